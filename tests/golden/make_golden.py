#!/usr/bin/env python
"""
Generate the golden vectors that pin oracle/halo_oracle.py to the reference's own code.

Runs ONLY in the build container (it imports the reference from /root/reference, which
never travels); the outputs (tests/golden/*.npz, small) are committed and are what the
`-m "not gpu"` tests check the oracle against, here and on the GPU box.

The reference is pure Python but needs `unyt` (not installed; no network).  We import its
modules under the unit-less stand-in in tests/golden/_shims/unyt (all quantities are bare
numbers in one unit system, `.to()` is the identity) and empty stubs for modules that are
imported for type hints / I-O only (h5py, mpi4py, virgo, astropy).  Functions exercised:

  SOAP/property_calculation/half_mass_radius.py   get_half_mass_radius
  SOAP/property_calculation/kinematic_properties.py  get_vmax, get_angular_momentum,
                                                     get_velocity_dispersion_matrix
  SOAP/particle_selection/SO_properties.py        find_SO_radius_and_mass,
                                                  SOParticleData (compute_basics,
                                                  compute_SO_radius_and_mass and the lazy
                                                  properties on the path)
  SOAP/particle_selection/aperture_properties.py  ApertureParticleData
  SOAP/particle_selection/subhalo_properties.py   SubhaloParticleData
  SOAP/particle_selection/projected_aperture_properties.py  ProjectedApertureParticleData,
                                                            SingleProjectionProjectedApertureParticleData
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF)

    class _Anything(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            sub = _Anything(f"{self.__name__}.{name}")
            setattr(self, name, sub)
            return sub

        def __call__(self, *a, **k):
            return self

    import importlib.abc
    import importlib.machinery

    roots = ("h5py", "mpi4py", "virgo", "astropy", "psutil")

    class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            if name.split(".")[0] in roots:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            m = _Anything(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _StubFinder())


class DS:
    """snapshot_datasets stand-in: get_dataset("PartType1/Masses", data) -> data["PartType1"]["Masses"]"""

    @staticmethod
    def get_dataset(name, data):
        pt, ds = name.split("/")
        return data[pt][ds]


def U(x, dtype=None):
    import unyt

    return unyt.unyt_array(np.array(x, dtype=dtype))


def random_halo(rng, n, boxsize=100.0, types=(1,), neg=False):
    """exponential radii about a random centre, first particle at r=0 (dummy_halo_generator.py:865-946 recipe)"""
    centre = rng.uniform(20, 80, 3)
    r = rng.exponential(1.0 / 60.0, n)
    r[0] = 0.0
    phi = 2 * np.pi * rng.random(n)
    ct = 2 * rng.random(n) - 1
    st = np.sqrt((1 - ct) * (1 + ct))
    pos = centre + np.stack([r * st * np.cos(phi), r * st * np.sin(phi), r * ct], axis=1)
    vel = rng.uniform(-500, 500, (n, 3)) + rng.uniform(-300, 300, 3)
    mass = rng.uniform(0.1, 0.5, n)
    tsel = rng.choice(len(types), n)
    tsel[0] = list(types).index(1) if 1 in types else 0
    grnr = np.where(rng.random(n) < 0.7, 1, np.where(rng.random(n) < 0.5, 2, -1)).astype(np.int32)
    grnr[0] = 1
    fof = np.where(rng.random(n) < 0.85, 1, 7).astype(np.int64)
    fof[0] = 1
    data = {}
    for k, t in enumerate(types):
        sel = tsel == k
        mname = "DynamicalMasses" if t == 5 else "Masses"
        data[f"PartType{t}"] = {
            "Coordinates": pos[sel], "Velocities": vel[sel], mname: mass[sel],
            "GroupNr_bound": grnr[sel], "FOFGroupIDs": fof[sel],
        }  # fmt: skip
    return centre, data


def main():
    _install_stubs()
    import unyt  # the stand-in

    from SOAP.property_calculation.half_mass_radius import get_half_mass_radius
    from SOAP.property_calculation import kinematic_properties as kp
    from SOAP.particle_selection import SO_properties as SO
    from SOAP.particle_selection import aperture_properties as AP
    from SOAP.particle_selection import subhalo_properties as SH
    from SOAP.particle_selection import projected_aperture_properties as PJ
    from SOAP.particle_selection.halo_properties import SearchRadiusTooSmallError

    G = float(unyt.Unit("newton_G"))
    out = {}

    # ---- half mass radius: the reference's own test recipe (tests/test_half_mass_radius.py) ----
    np.random.seed(203)
    hm_r, hm_m, hm_out, hm_n = [], [], [], []
    for i in range(300):
        npart = np.random.choice([1, 10, 100, 1000])
        radius = np.random.exponential(1.0, npart)
        mass = 1.0e9 * (1.0 + 0.2 * (np.random.random(npart) - 0.5))
        hm = get_half_mass_radius(U(radius), U(mass), U(mass).sum())
        hm_r.append(radius); hm_m.append(mass); hm_out.append(float(hm)); hm_n.append(npart)
    out["half_mass"] = dict(n=np.array(hm_n), radius=np.concatenate(hm_r), mass=np.concatenate(hm_m),
                            result=np.array(hm_out))

    # ---- SO radius finder: tests/test_SO_radius_calculation.py recipe (seed 62) -----------------
    np.random.seed(62)
    so_n, so_r, so_d, so_c, so_ref, so_res = [], [], [], [], [], []
    for i in range(50):
        npart = np.random.choice(np.array([10, 100, 1000]))
        Mpart = 1.0e9
        radius = np.random.exponential(1.0, npart)
        rmax = radius.max()
        ordered_radius = np.sort(radius)
        mass = Mpart * (1.0 + 0.2 * (np.random.random(npart) - 0.5))
        idx = np.unique(np.random.randint(0, len(mass), int(0.1 * npart)))
        mass[idx] *= -1.0
        cumulative_mass = np.cumsum(mass)
        ipos = np.argmax(ordered_radius > 0.0)
        ordered_radius = ordered_radius[ipos:]
        cumulative_mass = cumulative_mass[ipos:]
        density = cumulative_mass / (4.0 * np.pi / 3.0 * ordered_radius**3)
        reference_density = 200.0 * Mpart * npart / (4.0 * np.pi / 3.0 * rmax**3)
        try:
            r, m, v = SO.find_SO_radius_and_mass(U(ordered_radius), U(density), U(cumulative_mass), U(reference_density))
            res = [float(r), float(m), float(v)]
        except SearchRadiusTooSmallError:
            res = [-1.0, -1.0, -1.0]
        so_n.append(len(ordered_radius)); so_r.append(ordered_radius); so_d.append(density)
        so_c.append(cumulative_mass); so_ref.append(reference_density); so_res.append(res)
        # the same profile cut short / with a higher threshold: exercises the
        # SearchRadiusTooSmallError and the "all particles below threshold" branches
        for cut, fac in ((max(2, len(ordered_radius) // 4), 1.0), (len(ordered_radius), 1.0e4)):
            r2, d2, c2 = ordered_radius[:cut], density[:cut], cumulative_mass[:cut]
            rho2 = reference_density * fac
            try:
                r, m, v = SO.find_SO_radius_and_mass(U(r2), U(d2), U(c2), U(rho2))
                res = [float(r), float(m), float(v)]
            except SearchRadiusTooSmallError:
                res = [-1.0, -1.0, -1.0]
            so_n.append(len(r2)); so_r.append(r2); so_d.append(d2)
            so_c.append(c2); so_ref.append(rho2); so_res.append(res)
    out["so_radius"] = dict(n=np.array(so_n), radius=np.concatenate(so_r), density=np.concatenate(so_d),
                            cum=np.concatenate(so_c), rho=np.array(so_ref), result=np.array(so_res))

    # ---- kinematic helpers -------------------------------------------------------------------------
    rng = np.random.default_rng(7)
    kin = dict(n=[], mass=[], pos=[], vel=[], vmax=[], vmax1=[], L=[], vd=[])
    for i in range(60):
        n = int(rng.choice([1, 2, 5, 50, 500, 3000]))
        m = rng.uniform(0.1, 0.5, n)
        x = rng.normal(0, 0.1, (n, 3))
        if i % 3 == 0:
            x[0] = 0.0
        v = rng.normal(0, 200, (n, 3)) + 300
        r = np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2 + x[:, 2] ** 2)
        rv, vv = kp.get_vmax(U(m), U(r))
        rv1, vv1 = kp.get_vmax(U(m), U(r), nskip=1)
        frac = m / m.sum()
        vcom = (frac[:, None] * v).sum(axis=0)
        L = kp.get_angular_momentum(U(m), U(x), U(v), ref_velocity=U(vcom))
        vd = kp.get_velocity_dispersion_matrix(U(frac), U(v), U(vcom))
        kin["n"].append(n); kin["mass"].append(m); kin["pos"].append(x); kin["vel"].append(v)
        kin["vmax"].append([float(rv), float(vv)]); kin["vmax1"].append([float(rv1), float(vv1)])
        kin["L"].append(np.asarray(L, dtype=np.float64)); kin["vd"].append(np.asarray(vd, dtype=np.float64))
    out["kinematics"] = dict(n=np.array(kin["n"]), mass=np.concatenate(kin["mass"]), pos=np.concatenate(kin["pos"]),
                             vel=np.concatenate(kin["vel"]), vmax=np.array(kin["vmax"]), vmax1=np.array(kin["vmax1"]),
                             L=np.array(kin["L"]), vd=np.array(kin["vd"]), G=np.array(G))

    # ---- class level: particle-data classes with their lazy properties ---------------------------------
    rng = np.random.default_rng(4251)
    boxsize = 100.0
    soft = {f"PartType{t}": 0.0114 for t in (0, 1, 4, 5, 6)}
    rng_nu = np.random.default_rng(66)   # separate stream: the cases without neutrinos keep their numbers
    cases = []
    for ic in range(40):
        n = int(rng.choice([1, 3, 12, 100, 600, 2500]))
        hydro = ic % 2 == 1
        types = (0, 1, 4, 5) if hydro else (1,)
        centre, data = random_halo(rng, n, boxsize, types)
        # every other pair of cases also carries neutrino particles (PartType6: weighted masses, some negative)
        # and a mean neutrino density (SO_properties.py:380-406)
        nu_density = 0.0
        if ic % 4 >= 2:
            n_nu = int(rng_nu.integers(1, 300))
            r_nu = rng_nu.exponential(1.0 / 25.0, n_nu)
            ph, c2 = 2 * np.pi * rng_nu.random(n_nu), 2 * rng_nu.random(n_nu) - 1
            s2 = np.sqrt((1 - c2) * (1 + c2))
            data["PartType6"] = {
                "Coordinates": centre + np.stack([r_nu * s2 * np.cos(ph), r_nu * s2 * np.sin(ph), r_nu * c2], axis=1),
                "Masses": rng_nu.uniform(0.01, 0.05, n_nu), "Weights": rng_nu.normal(0.0, 0.5, n_nu),
            }  # fmt: skip
            nu_density = 0.4
        # H (1 / code time), Omega_g, Omega_m: the flow-rate terms (SO_properties.py:3446-3451); dummy snapshot values
        cosmology = {"nu_density": U(nu_density), "H": U(79.60499176), "Omega_g": 5.33243487e-05, "Omega_m": 0.304611}
        udata = {pt: {k: U(v) for k, v in d.items()} for pt, d in data.items()}
        input_halo = {"cofp": U(centre), "index": 1, "is_central": 1}
        types_present = [f"PartType{t}" for t in (0, 1, 4, 5) if f"PartType{t}" in data]
        case = dict(centre=centre, types=np.array(types), n=n, nu_density=np.array(nu_density))
        for pt, d in data.items():
            for k, v in d.items():
                case[f"{pt}/{k}"] = v
        # SO: density type and fixed physical radius
        allr = np.concatenate([np.sqrt(((d["Coordinates"] - centre) ** 2).sum(axis=1)) for pt, d in data.items() if pt != "PartType6"])
        mtot = sum(d.get("Masses", d.get("DynamicalMasses")).sum() for pt, d in data.items() if pt != "PartType6")
        rmax = allr.max() if n > 1 else 1.0
        rho_ref = 30.0 * mtot / (4.0 / 3.0 * np.pi * max(rmax, 1e-3) ** 3)
        for tag, rho, rphys, virial in (("SOd", rho_ref, 0.0, True), ("SOp", 0.0, 0.6 * rmax, False)):
            pp = SO.SOParticleData(input_halo, udata, types_present, None, None, DS, None, soft, virial,
                                   U(2 * rmax), cosmology, U(boxsize))
            try:
                exists = pp.compute_SO_radius_and_mass(U(rho) if rho > 0 else 0.0, U(rphys))
            except SearchRadiusTooSmallError:
                case[f"{tag}/status"] = np.array(1)
                continue
            except (AttributeError, IndexError):
                case[f"{tag}/status"] = np.array(2)  # reference undefined (SURVEY 8a pitfall 7)
                continue
            case[f"{tag}/status"] = np.array(0)
            case[f"{tag}/rho"] = np.array(rho); case[f"{tag}/rphys"] = np.array(rphys)
            case[f"{tag}/exists"] = np.array(bool(exists))
            if not exists:
                continue
            for name in ["r", "Mtot", "Mtotpart", "com", "vcom", "Vmax_soft", "R_vmax_soft", "spin_parameter",
                         "Mfrac_satellites", "Mfrac_external", "Mdm", "Ldm", "Ndm", "Ngas", "Nstar", "Nbh",
                         "Mgas", "Mstar", "Mbh_dynamical", "concentration_unsoft", "concentration_soft",
                         "com_gas", "com_star", "vcom_gas", "vcom_star", "Lgas", "Lstar", "Lbaryons",
                         "Nnu", "Mnu", "MnuNS", "DtoTgas", "DtoTstar",
                         "TotalInertiaTensor", "TotalInertiaTensorReduced", "TotalInertiaTensorNoniterative",
                         "TotalInertiaTensorReducedNoniterative", "DarkMatterInertiaTensor", "DarkMatterInertiaTensorReduced",
                         "DarkMatterInertiaTensorNoniterative", "DarkMatterInertiaTensorReducedNoniterative",
                         "GasInertiaTensor", "GasInertiaTensorReduced", "GasInertiaTensorNoniterative",
                         "GasInertiaTensorReducedNoniterative", "StellarInertiaTensor", "StellarInertiaTensorReduced",
                         "StellarInertiaTensorNoniterative", "StellarInertiaTensorReducedNoniterative"]:
                val = getattr(pp, name)
                case[f"{tag}/{name}"] = np.array(np.nan) if val is None else np.asarray(val, dtype=np.float64)
            # dark matter mass flow rates through the three shells (SO_properties.py:2816-2932); the outer shell
            # may reach beyond the search radius
            case[f"{tag}/search_radius"] = np.array(2 * rmax)
            try:
                val = pp.DarkMatterMassFlowRate
                case[f"{tag}/flow_status"] = np.array(0)
                case[f"{tag}/DarkMatterMassFlowRate"] = np.array(np.nan) if val is None else np.asarray(val, dtype=np.float64)
            except SearchRadiusTooSmallError:
                case[f"{tag}/flow_status"] = np.array(1)
            # the same sphere with a search radius just short of the outer shell's edge (1.05 R_SO)
            pp2 = SO.SOParticleData(input_halo, udata, types_present, None, None, DS, None, soft, virial,
                                    U(1.04 * float(pp.r)), cosmology, U(boxsize))
            pp2.compute_SO_radius_and_mass(U(rho) if rho > 0 else 0.0, U(rphys))
            try:
                val = pp2.DarkMatterMassFlowRate
                case[f"{tag}/flow_status_short"] = np.array(0)
            except SearchRadiusTooSmallError:
                case[f"{tag}/flow_status_short"] = np.array(1)
        # apertures
        for tag, incl in (("APi", True), ("APe", False)):
            rap = 0.5 * rmax
            pp = AP.ApertureParticleData(input_halo, udata, types_present, incl, U(rap), None, None, None, DS,
                                         soft, U(boxsize), cosmology)
            case[f"{tag}/radius"] = np.array(rap)
            for name in ["Mtot", "Mdm", "Mgas", "Mstar", "Mbh_dynamical", "Ndm", "Ngas", "Nstar", "Nbh", "com",
                         "vcom", "Ldm", "veldisp_matrix_dm", "HalfMassRadiusDM", "Vmax_soft", "R_vmax_soft",
                         "com_star", "vcom_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas",
                         "veldisp_matrix_star", "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
                         "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar"]:
                val = getattr(pp, name)
                case[f"{tag}/{name}"] = np.array(np.nan) if val is None else np.asarray(val, dtype=np.float64)
        # bound subhalo
        pp = SH.SubhaloParticleData(input_halo, udata, types_present, None, None, DS, soft, U(boxsize), cosmology)
        for name in ["Mtot", "Mdm", "Mgas", "Mstar", "Mbh_dynamical", "Ndm", "Ngas", "Nstar", "Nbh", "com", "vcom",
                     "Ldm", "veldisp_matrix_dm", "Vmax_unsoft", "R_vmax_unsoft", "Vmax_soft", "spin_parameter",
                     "HalfMassRadiusTot", "HalfMassRadiusDM", "EncloseRadius",
                     "com_star", "Lgas", "Lstar", "Lbaryons", "veldisp_matrix_gas", "veldisp_matrix_star",
                     "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
                     "kappa_corot_gas", "kappa_corot_star", "kappa_corot_baryons", "DtoTgas", "DtoTstar",
                     "TotalInertiaTensor", "TotalInertiaTensorReduced", "TotalInertiaTensorNoniterative",
                     "TotalInertiaTensorReducedNoniterative", "DarkMatterInertiaTensor", "DarkMatterInertiaTensorReduced",
                     "DarkMatterInertiaTensorNoniterative", "DarkMatterInertiaTensorReducedNoniterative",
                         "GasInertiaTensor", "GasInertiaTensorReduced", "GasInertiaTensorNoniterative",
                         "GasInertiaTensorReducedNoniterative", "StellarInertiaTensor", "StellarInertiaTensorReduced",
                         "StellarInertiaTensorNoniterative", "StellarInertiaTensorReducedNoniterative"]:
            val = getattr(pp, name)
            case[f"SH/{name}"] = np.array(np.nan) if val is None else np.asarray(val, dtype=np.float64)
        # projected
        rap = 0.5 * rmax
        pp = PJ.ProjectedApertureParticleData(input_halo, udata, types_present, U(rap), DS, U(boxsize))
        case["PJ/radius"] = np.array(rap)
        for proj in ("projx", "projy", "projz"):
            sp = PJ.SingleProjectionProjectedApertureParticleData(pp, proj)
            for name in ["Mtot", "Mdm", "Ndm", "Ngas", "Nstar", "Nbh", "com", "vcom", "HalfMassRadiusDM",
                         "proj_veldisp_dm", "com_star", "proj_veldisp_gas", "proj_veldisp_star",
                         "HalfMassRadiusGas", "HalfMassRadiusStar", "HalfMassRadiusBaryon",
                         *[f"Projected{c}InertiaTensor{s}" for c in ("Total", "Gas", "Stellar")
                           for s in ("", "Reduced", "Noniterative", "ReducedNoniterative")]]:
                val = getattr(sp, name)
                case[f"PJ/{proj}/{name}"] = np.array(np.nan) if val is None else np.asarray(val, dtype=np.float64)
        cases.append(case)
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat[f"case{i:02d}|{k}"] = v
    flat["ncases"] = np.array(len(cases))
    flat["G"] = np.array(G)
    flat["softening"] = np.array(0.0114)
    flat["boxsize"] = np.array(boxsize)
    flat["H"] = np.array(79.60499176); flat["Omega_g"] = np.array(5.33243487e-05); flat["Omega_m"] = np.array(0.304611)

    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **d)
    np.savez_compressed(os.path.join(HERE, "particle_data_classes.npz"), **flat)
    print("golden vectors written:", sorted(out) + ["particle_data_classes"])


if __name__ == "__main__":
    main()
