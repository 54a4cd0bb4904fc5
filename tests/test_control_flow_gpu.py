"""
The engine against vectors written by the REFERENCE's own scheduler and calculation classes
(tests/golden/control_flow.npz; generator tests/golden/make_control_flow_golden.py): every attempt of every halo -- loops
of the x1.2 radius search, verdict, radii handed back -- and, for finished halos, every value the reference stored.
"""
import dataclasses

import numpy as np
import pytest

from soap_amd import compute_halo_properties as chp
from soap_amd.core import halo_tasks
from soap_amd.core.category_filter import CategoryFilter
from soap_amd.units import SnapshotScalars
from tests import control_flow_cases as cases

pytestmark = pytest.mark.gpu


class _Parameters:
    def __init__(self, strict, inertia):
        self.strict, self.inertia = strict, inertia

    def get_property_filters(self, base_halo_type, names):
        return {n: ("basic" if (self.inertia or "InertiaTensor" not in n) else False) for n in names}

    def strict_halo_copy(self):
        return self.strict


def _scalars(g):
    s = g.scalars()
    # the unit-less stand-in of the generator: one physical kpc is 1e-3 of the length unit, G as given, a = 1
    sc = dataclasses.replace(SnapshotScalars(), a=s.get("a", 1.0), boxsize=s["boxsize"], U_L=1.0, U_M=1.0, U_t=1.0, kpc_cm=1.0e-3, G_cgs=s["G"],
                             comoving_softening=1.0e30)   # (the generator's softening is the maximal physical one at any a)
    assert sc.kpc == s["kpc"] and sc.G == s["G"] and sc.softening == s["softening"]
    assert sc.critical_density == s["critical_density"] and sc.mean_density == s["mean_density"] and sc.virBN98 == s["virBN98"]
    assert sc.H == s["H"] and sc.Omega_g == s["Omega_g"] and sc.Omega_m == s["Omega_m"]
    return sc


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("vname", list(cases.VARIANTS))
def test_engine_follows_the_reference_with_float32_velocities_and_masses(built_library, vname, resident):
    """The float32 family (tests/golden/control_flow_f32.npz, `make_control_flow_golden.py --f32`): Velocities and Masses go through the C
    ABI as float32 (`value_bytes` = 4), the engine mirrors the reference's cast of the cumulative mass of the SO profile back to float32
    (SO_properties.py:400-402).  Attempts, radius rounds, verdicts, radii handed back, particle counts of every selection and the SO
    radii as stored: exactly the reference's; the other values to the rounding of the reference's own float32 sums."""
    test_engine_follows_the_reference_scheduler_and_classes(built_library, vname, resident, False, f32=True)


@pytest.mark.parametrize("units", [False, True])
@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("vname", list(cases.VARIANTS))
def test_engine_follows_the_reference_scheduler_and_classes(built_library, vname, resident, units, f32=False):
    """units=True: the vectors made at a = 0.769 with unit-carrying quantities (tests/golden/_shims_units): the reference's
    own comoving <-> physical conversions and powers of a in every stored value."""
    g = cases.Golden(units=units, f32=f32)
    bname, dmo, strict, skip_incl = cases.VARIANTS[vname]
    data, halos = g.box(bname)
    sc = _scalars(g)
    cf = CategoryFilter(cases.FILTERS, dmo=dmo)
    so_vars, ap_vars, pj_vars = cases.variations(dmo, skip_incl)
    props = chp.build_halo_prop_list(sc, cf, so_vars, ap_vars, projected_vars=pj_vars, parameters=_Parameters(strict, not dmo))
    assert [p.name for p in props] == [str(n) for n in g.z[f"{vname}|calc_names"]]
    assert halo_tasks.target_density_of(props, sc) == pytest.approx(float(g.z[f"{vname}|target_density"]) * sc.a**3, rel=1e-15)

    engine = halo_tasks.make_engine(sc, device=0)
    halo_tasks.stage_chunk(engine, data)
    H = len(halos["index"])
    ha = dict({k: v.copy() for k, v in halos.items()}, done=np.zeros(H, dtype=np.int32), n_loop=np.zeros(H, dtype=np.float32),
              n_process=np.zeros(H, dtype=np.float32))
    results = []
    rtol_r = 1e-15 if units else 0.0   # (the stand-in rounds a conversion factor its own way: x * (1/a) against x / a)
    finished_in_call = {}
    n_calls = 0
    for call in range(10):
        todo = np.flatnonzero(ha["done"] == 0)
        if todo.size == 0:
            break
        before = {k: ha[k].copy() for k in ("search_radius", "read_radius")}
        n0 = len(results)
        out = halo_tasks.process_halos(engine, props, sc, ha, results, resident_radius=np.inf if resident else None)
        raw = out[4]
        n_calls += 1
        for j, i in enumerate(todo):
            att = g.attempts(vname, i)
            if resident:
                continue
            # one call = one attempt of the reference: same radii going in, same loops, same verdict, same radii coming out
            assert call < len(att), (vname, i, call)
            s_in, r_in, n_loop, done, s_out = att[call]
            assert before["search_radius"][i] == pytest.approx(s_in, rel=rtol_r) and before["read_radius"][i] == pytest.approx(r_in, rel=rtol_r), (vname, i, call)
            assert raw["stats"][j, 0] == n_loop and raw["stats"][j, 4] == n_loop and raw["stats"][j, 5] == 1, (vname, i, call, raw["stats"][j])
            assert (raw["status"][j] == 0) == bool(done), (vname, i, call)
            assert ha["search_radius"][i] == pytest.approx(s_out, rel=rtol_r), (vname, i, call, ha["search_radius"][i], s_out)
            if not done:
                assert ha["read_radius"][i] == pytest.approx(max(1.5 * r_in, s_out), rel=rtol_r)   # halo_tasks.py:393-402
        for r in results[n0:]:
            finished_in_call[int(r["InputHalos/HaloCatalogueIndex"][0])] = call
        if resident:
            # every retry happens inside the engine: ONE call finishes everything that can finish
            assert [bool(d) for d in ha["done"]] == [g.done(vname, i) for i in range(H)]
            break

    by_index = {int(r["InputHalos/HaloCatalogueIndex"][0]): r for r in results}
    n_values = n_exact = 0
    for i in range(H):
        att = g.attempts(vname, i)
        if not g.done(vname, i):
            assert ha["done"][i] == 0 and int(halos["index"][i]) not in by_index
            continue
        assert ha["done"][i] == 1
        res = by_index[int(halos["index"][i])]
        rec = g.record(vname, i)
        # the radii halo_arrays hold at the end, and the bookkeeping columns of --record-halo-timings
        assert ha["search_radius"][i] == pytest.approx(att[-1, 4], rel=rtol_r) and ha["read_radius"][i] == pytest.approx(att[-1, 1], rel=rtol_r), (vname, i)
        assert float(res["InputHalos/n_loop"][0]) == float(rec["InputHalos/n_loop"][0][0]) == att[:, 2].sum()
        assert float(res["InputHalos/n_process"][0]) == float(rec["InputHalos/n_process"][0][0]) == len(att)
        for full, (want, is_f32) in rec.items():
            if full in ("InputHalos/n_loop", "InputHalos/n_process"):
                continue
            assert full in res, f"{full}: stored by the reference, missing from the engine's halo_result"
            got = np.atleast_1d(np.asarray(res[full][0]))
            assert got.shape == want.shape, (full, got.shape, want.shape)
            if got.dtype.kind in "ui":
                np.testing.assert_array_equal(got.astype(np.int64), want.astype(np.int64), err_msg=f"{vname} halo {i} {full}")
                continue
            assert (got.dtype == np.float32) == is_f32, (full, got.dtype)
            want = want.astype(got.dtype)
            n_values += want.size
            n_exact += int(np.sum(got == want))
            # fp64 sums in another order than the reference's, then the same cast: the neighbouring float32 at most
            # (vector quantities: against the largest component; rotation fractions pass through float32 buffers
            # inside the reference, kinematic_properties.py:360-381)
            tol = 2.4e-7 if got.dtype == np.float32 else 1e-10
            scale = float(np.max(np.abs(want))) if want.size > 1 else 0.0
            if "KappaCorot" in full or "DiscToTotal" in full:
                scale = 1.0
            if "AngularMomentum" in full or "InertiaTensor" in full or "VelocityDispersionMatrix" in full:
                tol = 2e-6   # sums of signed terms (angular momenta of random velocities, off-diagonal elements)
            if f32:
                # the reference sums float32 masses, mass fractions and velocities in float32 (NumPy's pairwise order), the engine in
                # float64: values agree to the rounding of THOSE sums.  What decides selections is exact: the SO radius as stored ...
                if full.endswith("/SORadius"):
                    np.testing.assert_array_equal(got, want, err_msg=f"{vname} halo {i} {full}")
                tol = max(tol, 2e-5)
                if "DiscToTotal" in full:
                    tol = 1e-3   # (a member whose L_i changes sign with the rounding of the centre-of-mass velocity moves its whole mass)
                if "MassFlowRate" in full or "AngularMomentum" in full or "KappaCorot" in full:
                    tol = 2e-4
            err = np.abs(got.astype(float) - want.astype(float))
            assert np.all(err <= tol * np.maximum(np.abs(want.astype(float)), scale)), (vname, i, full, got, want)
    assert n_values > 2000 and n_exact >= (0.3 if f32 else 0.97) * n_values, (n_exact, n_values)
    engine.close()
