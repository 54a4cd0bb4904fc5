"""
The handful of snapshot scalars the hot path needs, and the single explicit unit
conversions applied before anything reaches the engine.

In the reference these live on the SWIFTCellGrid and travel as `unyt` quantities
(SOAP/core/swift_cells.py:224-287, SOAP/particle_selection/halo_properties.py:8-35);
here the engine works in one fixed system (comoving code lengths, code masses and
velocities) and every physical radius / density is converted exactly once by the
methods below.  Membership counts depend on these factors bit-for-bit, so tests pin
them against the oracle's.
"""

from __future__ import annotations

import dataclasses
import math


@dataclasses.dataclass
class SnapshotScalars:
    """Defaults: the reference's dummy snapshot, a FLAMINGO header at z=0.3
    (tests/dummy_halo_generator.py:40-126)."""

    a: float = 0.76923077
    boxsize: float = 100.0  # comoving code length
    critical_density: float = 17.58736923  # physical code density
    Omega_m: float = 0.304611
    Omega_k: float = 2.5212783e-09
    Omega_lambda: float = 0.693922
    U_L: float = 3.08567758e24
    U_M: float = 1.98841e43
    U_t: float = 3.08567758e19
    G_cgs: float = 6.6743e-08
    kpc_cm: float = 3.08567758e21
    comoving_softening: float = 0.0446
    max_physical_softening: float = 0.0114
    nu_density: float = 0.0  # physical code density of the neutrino background
    H: float = 79.60499176  # Hubble rate at this redshift, 1 / code time ("H [internal units]")
    Omega_g: float = 5.33243487e-05  # radiation density parameter (flow rates, SO_properties.py:3446-3451)

    @property
    def z(self):
        return 1.0 / self.a - 1.0

    @property
    def mean_density(self):
        return self.critical_density * self.Omega_m

    @property
    def kpc(self):
        """one physical kpc in physical code length"""
        return self.kpc_cm / self.U_L

    @property
    def G(self):
        """Newton's constant in physical code units (swift_units.py:77-81)"""
        return self.G_cgs * self.U_M * self.U_t**2 / self.U_L**3

    @property
    def softening(self):
        """physical code length, min(comoving*a, max physical) (swift_cells.py:235-247)"""
        return min(self.comoving_softening * self.a, self.max_physical_softening)

    @property
    def virBN98(self):
        """Bryan & Norman (1998) overdensity (swift_cells.py, dummy_halo_generator.py:462-470)"""
        a = self.a
        bnx = -(self.Omega_k / a**2 + self.Omega_lambda) / (
            self.Omega_k / a**2 + self.Omega_m / a**3 + self.Omega_lambda
        )
        return 18.0 * math.pi**2 + 82.0 * bnx - 39.0 * bnx**2

    # ---- the single explicit conversions -------------------------------------------
    def kpc_to_comoving(self, r_kpc: float) -> float:
        return r_kpc * self.kpc / self.a

    def mpc_to_comoving(self, r_mpc: float) -> float:
        return r_mpc * (1000.0 * self.kpc) / self.a

    def comoving_to_mpc(self, r: float) -> float:
        return r * self.a / (1000.0 * self.kpc)

    def density_to_comoving(self, rho_phys: float) -> float:
        return rho_phys * self.a**3

    def softening_comoving(self, ptype: int = 1) -> float:
        return self.softening / self.a

    @property
    def G_eff(self) -> float:
        return self.G / self.a

    @property
    def nu_density_comoving(self) -> float:
        return self.nu_density * self.a**3
