"""The oracle (oracle/c2ray_oracle.c) against vectors produced by the reference itself.

Everything here is bit-exact: the oracle restates the reference operation by operation and the
reference was built without FMA contraction (flang -O2, x86-64)."""
import numpy as np
import pytest

from conftest import rel_err, tap_case


def test_constants_match_reference_build(orc, gold):
    ref = gold("consts.npz")["consts"]
    mine = orc.constants()
    assert np.array_equal(mine, ref[: len(mine)])


def test_ini_rec_colion_factors(orc, gold):
    a = gold("funcvec.npz")["reccoef_T"].reshape(-1, 13)
    for row in a:
        assert np.array_equal(orc.ini_rec_colion_factors(row[0]), row[1:]), row[0]


@pytest.mark.parametrize("key,iso", [("photoion_iso", 1), ("photoion_heat", 0)])
def test_photoion_rates(orc, otables, gold, key, iso):
    a = gold("funcvec.npz")[key].reshape(-1, 30)
    for row in a:
        got = orc.photoion_rates(otables, row[:6], row[6], row[8], row[7], iso)
        assert np.array_equal(got, row[9:])


def test_doric(orc, gold):
    a = gold("funcvec.npz")["doric"].reshape(-1, 54)
    for row in a:
        got = orc.doric(row[0], row[1], row[2], row[20:35], row[4:7], row[35:39], row[7:19], row[3])
        assert np.array_equal(got, row[39:54])


def test_thermal(orc, otables, gold):
    a = gold("funcvec.npz")["thermal"].reshape(-1, 25)
    c = gold("consts.npz")["consts"]
    for row in a:
        te, ta = orc.thermal(otables, row[0], row[1], -1.0, row[2], row[3], row[6:21], row[4], row[5], c[42], c[43])
        assert te == row[21] and ta == row[22]


CASES = [("tap_N16_iso_1src.npz", 1), ("tap_N16_iso_1src.npz", 2), ("tap_N16_heat_3src.npz", 1),
         ("tap_N16_heat_3src.npz", 2), ("tap_N22_iso_2src.npz", 1)]


@pytest.mark.parametrize("fname,call", CASES)
def test_evolve3d_end_to_end(orc, otables, gold, fname, call):
    """Whole evolve3D calls: same iteration count, same non-converged counts per iteration, and
    every output array bit-identical to what the reference wrote."""
    i, o = tap_case(gold(fname), call)
    st = orc.Step.from_tap(i)
    s = orc.State(st, i["xh"], i["xhe"], i.get("temperature"))
    nit = orc.evolve3d(otables, st, s, i["dt"][0])
    assert nit == len(o["conv_flags"])
    assert s.conv_flags == list(o["conv_flags"])
    names = {"xh": "xh", "xhe": "xhe", "phih_grid": "phih", "phihe_grid": "phihe", "xh_av": "xh_av",
             "xhe_av": "xhe_av", "xh_intermed": "xh_intermed", "xhe_intermed": "xhe_intermed",
             "coldensh_out": "coldensh_out", "coldenshe_out": "coldenshe_out", "photon_loss_all": "photon_loss"}
    if not st.isothermal:
        names.update(temperature="temperature", phiheat="phiheat")
    for k, attr in names.items():
        assert np.array_equal(getattr(s, attr), o[k]), k
    assert s.c.sum_nbox == o["sum_nbox_all"][0]


def test_n22_far_layer_never_traced(orc, otables, gold):
    """(N/2-1) mod 10 == 0: the while-test of evolve_source.F90:136-139 stops after box 1 and the
    layer at offset -N/2 gets no column."""
    i, o = tap_case(gold("tap_N22_iso_2src.npz"), 1)
    col = o["coldensh_out"].reshape(22, 22, 22, order="F")
    src = i["srcpos"].reshape(-1, 3)[-1]  # columns are those of the source swept last
    far = (src - 1 - 11) % 22
    assert np.all(col[far[0], :, :] == 0.0) and np.all(col[:, far[1], :] == 0.0) and np.all(col[:, :, far[2]] == 0.0)
    assert np.count_nonzero(col) == 21 ** 3


def _pl_tables(orc, pkg, gold):
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        d = {k: t[k] for k in t.files}
    z = gold("rad_tables_pl_qpl.npz")
    d.update({k: z[k] for k in z.files})
    return orc.Tables(d)


def test_evolve3d_with_power_law_and_quasar_seds(orc, pkg, gold):
    """-DPL -DQUASARS build of the reference: three SEDs per source.  The hard photons keep a few dozen
    cells flickering, so the call runs into the 500-iteration cap: 501 outer iterations, all
    non-converged counts and every output array bit-identical."""
    T = _pl_tables(orc, pkg, gold)
    i, o = tap_case(gold("tap_N16_pl_heat_3src.npz"), 1)
    st = orc.Step.from_tap(i)
    s = orc.State(st, i["xh"], i["xhe"], i["temperature"])
    nit = orc.evolve3d(T, st, s, i["dt"][0])
    assert nit == 501 == len(o["conv_flags"])
    assert s.conv_flags == list(o["conv_flags"])
    for k, attr in {"xh": "xh", "xhe": "xhe", "temperature": "temperature", "phih_grid": "phih", "phihe_grid": "phihe",
                    "phiheat": "phiheat", "xh_av": "xh_av", "xhe_av": "xhe_av", "photon_loss_all": "photon_loss"}.items():
        assert np.array_equal(getattr(s, attr), o[k]), k


def test_evolve3d_with_lyman_limit_systems(orc, otables, gold):
    """use_LLS = .true. build of the reference (type 1: one LLS column per cell added to the incoming
    HI column, evolve_point.F90:177-180): two whole calls, bit-identical."""
    for call in (1, 2):
        i, o = tap_case(gold("tap_N16_lls_heat_2src.npz"), call)
        assert float(i["coldensh_LLS"][0]) > 1e16
        st = orc.Step.from_tap(i)
        s = orc.State(st, i["xh"], i["xhe"], i["temperature"])
        niter = orc.evolve3d(otables, st, s, float(i["dt"][0]))
        assert niter == len(o["conv_flags"]) and s.conv_flags == [int(x) for x in o["conv_flags"]]
        for k, a in (("xh", s.xh), ("xhe", s.xhe), ("temperature", s.temperature), ("phih_grid", s.phih),
                     ("phihe_grid", s.phihe), ("phiheat", s.phiheat), ("coldensh_out", s.coldensh_out),
                     ("coldenshe_out", s.coldenshe_out)):
            assert np.array_equal(a, o[k]), (call, k)


def _used_columns(lo, hi, kind):
    cols = []
    for b in range(int(lo), int(hi) + 1):
        cols += [b] if kind == "photo" else ([1] if b == 1 else ([2 * b - 2, 2 * b - 1] if b <= 27 else [3 * b - 30, 3 * b - 29, 3 * b - 28]))
    return [c - 1 for c in cols]


def test_table_construction_equals_reference(orc, pkg, gold):
    """spec_integration (radiation_tables.f90:172-422) restated: from the band set-up, Romberg weights and
    normalised SEDs dumped from the reference (tests/golden/sed_setup.npz), every entry of the black-body
    tables and of the power-law / quasar tables the reference built is reproduced bit for bit."""
    d = dict(gold("sed_setup.npz"))
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        bb = {k: t[k] for k in ("photo_thick", "photo_thin", "heat_thick", "heat_thin")}
    T = orc.build_tables(d, 0)
    for k in bb:
        assert np.array_equal(T[k], bb[k]), k
    sed = gold("rad_tables_pl_qpl.npz")
    for idx, pre in ((1, "pl_"), (2, "qpl_")):
        T = orc.build_tables(d, idx)
        lo, hi = sed[pre + "limits"]
        for kind, ncol in (("photo", 47), ("heat", 113)):
            used = _used_columns(lo, hi, kind)
            for tt in ("thick", "thin"):
                got = T[f"{kind}_{tt}"].reshape(ncol, 2001)[used]
                want = sed[f"{pre}{kind}_{tt}"].reshape(ncol, 2001)[used]
                assert np.array_equal(got, want), (pre, kind, tt)


@pytest.mark.parametrize("n,nsrc,iso", [(22, 3, True), (40, 4, False)])
def test_shell_order_parallel_sweep_equals_serial_sweep(orc, otables, pkg, n, nsrc, iso):
    """The ordering the HIP kernels rely on, proved on the CPU: sweeping a source in L-infinity shells with the
    cells of a shell in parallel (8 OpenMP threads) leaves the columns and the rate grids of the reference's
    serial sweep order bit for bit; only the photon loss (a sum over boundary cells) is added up in another
    order.  The global pass is cell-parallel trivially."""
    rng = np.random.default_rng(5)
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    ndens = hp.test_density(zred) * np.exp(rng.normal(0, 0.5, nc))
    x = 10.0 ** rng.uniform(-4, -0.3, nc)
    xh, xhe = np.concatenate([1 - x, x]), np.concatenate([1 - x, 0.8 * x, 0.2 * x])
    temp = None if iso else np.tile((1e4 * np.exp(rng.normal(0, 0.2, nc))).astype(np.float32), 3)
    srcpos = rng.integers(1, n + 1, size=(nsrc, 3)).astype(np.int32)
    srcpos[0] = (1, n, n // 2)
    flux = 10.0 ** rng.uniform(5, 7, nsrc)
    st = orc.Step((n, n, n), dr, vol, zred, hp.H0, hp.Omega0, iso, 1e4, 1.0, srcpos, flux, 1e48, ndens, hp.reccoef(1e4))
    a = orc.State(st, xh, xhe, temp)
    orc.begin_step(a)
    orc.pass_all_sources(otables, st, a)
    for threads in (1, 8):
        b = orc.State(st, xh, xhe, temp)
        orc.begin_step(b)
        orc.pass_all_sources_shells(otables, st, b, threads)
        assert np.array_equal(a.coldensh_out, b.coldensh_out) and np.array_equal(a.coldenshe_out, b.coldenshe_out)
        assert np.array_equal(a.phih, b.phih) and np.array_equal(a.phihe, b.phihe) and np.array_equal(a.phiheat, b.phiheat)
        assert a.c.sum_nbox == b.c.sum_nbox
        assert abs(a.photon_loss[0] - b.photon_loss[0]) <= 1e-13 * a.photon_loss[0]
    dt = 1.0e6 * hp.YEAR
    ca = orc.global_pass(otables, st, a, dt)
    cb = orc.global_pass_threads(otables, st, b, dt, 8)
    assert ca == cb and np.array_equal(a.xh_av, b.xh_av) and np.array_equal(a.xhe_intermed, b.xhe_intermed)
    if not iso:
        assert np.array_equal(a.temperature, b.temperature)


@pytest.mark.parametrize("fixture", ["n64_heat_1src.npz", "n128_iso_32src.npz"])
def test_oracle_against_the_larger_reference_fixtures(orc, otables, pkg, gold, fixture):
    """The oracle itself against the reference's larger runs (tests/golden/n64_heat_1src.npz: BASELINE configs[1], 83 outer
    iterations with heating; n128_iso_32src.npz: 32 faint sources at 128^3, 149 outer iterations with growing sub-boxes): same
    iteration history, same SHA-256 of every grid after every call -- the pins of the oracle at sizes beyond the 16^3 / 22^3 tap
    fixtures.  Minutes of CPU time: runs with C2R_ORACLE_LARGE=1 (DESIGN.md section 4 records the last run)."""
    import hashlib
    import os
    if not os.environ.get("C2R_ORACLE_LARGE"):
        pytest.skip("set C2R_ORACLE_LARGE=1 (minutes of CPU time)")
    z = gold(fixture)
    n = int(z["c1_mesh"][0])
    nc = n ** 3
    iso = bool(z["c1_isothermal"][0])
    nthreads = min(8, os.cpu_count() or 1)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    xh, xhe = np.repeat(z["c1_xh_uniform"], nc), np.repeat(z["c1_xhe_uniform"], nc)
    temp = None if iso else np.repeat(z["c1_temperature_uniform"].astype(np.float32), nc)
    for call in range(1, int(z["ncalls"]) + 1):
        g = lambda k: z[f"c{call}_{k}"]
        st = orc.Step((n, n, n), g("dr"), g("vol")[0], g("zred")[0], g("H0")[0], g("Omega0")[0], iso, g("temper_val")[0],
                      g("clumping")[0], g("srcpos"), g("NormFlux"), g("S_star")[0], np.full(nc, float(g("ndens_uniform"))), g("reccoef"))
        s = orc.State(st, xh, xhe, temp)
        orc.begin_step(s)
        dt = float(g("dt")[0])
        crit = min(int(np.float32(2.5e-4) * nc), st.c.nsrc)                        # evolve.F90:147
        it, conv, flags = 0, nc, []
        while not (conv < crit and it > 1) and it <= 500:                        # :163, :177
            it += 1
            orc.pass_all_sources_shells(otables, st, s, nthreads)
            conv = orc.global_pass_threads(otables, st, s, dt, nthreads)
            flags.append(conv)
        assert flags == [int(x) for x in g("conv_flags")], call
        if conv < crit:                                                           # :164-166: the final copy
            s.xh[:], s.xhe[:] = s.xh_intermed, s.xhe_intermed
            if temp is not None:
                s.temperature[2 * nc:] = s.temperature[:nc]
        got = {"xh": s.xh, "xhe": s.xhe, "phih_grid": s.phih, "phihe_grid": s.phihe, "xh_av": s.xh_av, "xhe_av": s.xhe_av}
        if not iso:
            got.update(temperature=s.temperature, phiheat=s.phiheat)
        for k, v in got.items():
            assert sha(v) == str(g("sha_" + k)), (call, k)
        assert int(s.c.sum_nbox) == int(g("sum_nbox")[0])
        xh, xhe = s.xh.copy(), s.xhe.copy()
        temp = None if iso else s.temperature.copy()
