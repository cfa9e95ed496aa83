"""The product's device functions (csrc/c2ray_device.hpp) compiled for the host, against the
reference's golden vectors and against the oracle.  On the host they use glibc's exp/log10/pow, the
same libm the reference build links, so the comparison is bit-exact; on the GPU the same source is
compiled by hipcc (tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import tap_case

dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def hh(harness, pkg):
    t = pkg.RadiationTables.load()
    keep = [t.fvec[k] for k in pkg.evolve.FVEC_ORDER]
    fv = (dp * 12)(*[_p(a) for a in keep])
    harness.hh_set_tables(_p(t.photo_thick), _p(t.photo_thin), _p(t.heat_thick), _p(t.heat_thin), _p(t.sigma_HI),
                          _p(t.sigma_HeI), _p(t.sigma_HeII), fv, C.c_int(t.bb_upper), _p(t.cool),
                          C.c_double(t.cool_mintemp), C.c_double(t.cool_dtemp))
    harness._keep = (t, keep)
    return harness


def test_constants(hh, gold):
    ref = gold("consts.npz")["consts"]
    buf = (C.c_double * 64)()
    n = hh.hh_constants(buf, 64)
    assert np.array_equal(np.array(buf[:n]), ref[:n])


def test_reccoef(hh, gold):
    a = gold("funcvec.npz")["reccoef_T"].reshape(-1, 13)
    out = np.empty(12)
    for row in a:
        hh.hh_reccoef(C.c_double(row[0]), _p(out))
        assert np.array_equal(out, row[1:])


@pytest.mark.parametrize("key,heat", [("photoion_iso", 0), ("photoion_heat", 1)])
def test_photoion(hh, gold, key, heat):
    a = gold("funcvec.npz")[key].reshape(-1, 30)
    out = np.empty(5)
    for row in a:
        cin = np.ascontiguousarray(row[:6])
        hh.hh_photoion(_p(cin), C.c_double(row[6]), C.c_double(row[8]), C.c_double(row[7]), C.c_int(heat), _p(out))
        ref = row[9:]
        # photo_cell_HI, HeI, HeII, heat, photo_out = members 0,1,2,18,20 of type photrates
        assert np.array_equal(out, ref[[0, 1, 2, 18, 20]])
        assert hh.hh_photo_out_only(_p(cin), C.c_double(row[8])) == ref[20]


def test_doric(hh, gold):
    a = gold("funcvec.npz")["doric"].reshape(-1, 54)
    for row in a:
        ion = np.ascontiguousarray(row[20:35]).copy()
        hh.hh_doric(C.c_double(row[0]), C.c_double(row[1]), _p(ion), _p(np.ascontiguousarray(row[4:7])),
                    _p(np.ascontiguousarray(row[35:39])), _p(np.ascontiguousarray(row[7:19])), C.c_double(row[3]))
        assert np.array_equal(ion, row[39:54])


def test_thermal(hh, gold):
    a = gold("funcvec.npz")["thermal"].reshape(-1, 25)
    c = gold("consts.npz")["consts"]
    for row in a:
        te, ta = C.c_double(row[1]), C.c_double(-1.0)
        hh.hh_thermal(C.c_double(row[0]), C.byref(te), C.byref(ta), C.c_double(row[2]), C.c_double(row[3]),
                      _p(np.ascontiguousarray(row[6:21])), C.c_double(row[4]), C.c_double(row[5]), C.c_double(c[42]),
                      C.c_double(c[43]))
        assert te.value == row[21] and ta.value == row[22]


def test_cinterp_equals_oracle_everywhere(hh, orc, gold):
    """short_characteristic + interp_column (what the sweep kernel runs) against the oracle's
    cinterp for EVERY offset of a 16^3 and a 22^3 box, on the reference's own column grids."""
    for fname in ["tap_N16_heat_3src.npz", "tap_N22_iso_2src.npz"]:
        i, o = tap_case(gold(fname), 1)
        mesh = np.ascontiguousarray(i["mesh"], dtype=np.int32)
        n = int(mesh[0])
        cH, cHe = np.ascontiguousarray(o["coldensh_out"]), np.ascontiguousarray(o["coldenshe_out"])
        src = np.ascontiguousarray(i["srcpos"].reshape(-1, 3)[-1], dtype=np.int32)
        ip = C.POINTER(C.c_int)
        out = np.empty(4)
        lo, hi = -(n // 2), n - n // 2 - 1
        for dk in range(lo, hi + 1):
            for dj in range(lo, hi + 1):
                for di in range(lo, hi + 1):
                    if di == 0 and dj == 0 and dk == 0:
                        continue
                    pos = np.array([src[0] + di, src[1] + dj, src[2] + dk], dtype=np.int32)
                    hh.hh_cinterp(mesh.ctypes.data_as(ip), _p(cH), _p(cHe), pos.ctypes.data_as(ip),
                                  src.ctypes.data_as(ip), _p(out))
                    a, b, c, d = (C.c_double(), C.c_double(), C.c_double(), C.c_double())
                    orc.lib().orc_cinterp(mesh.ctypes.data_as(ip), _p(cH), _p(cHe), pos.ctypes.data_as(ip),
                                          src.ctypes.data_as(ip), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
                    assert (out[0], out[1], out[2], out[3]) == (a.value, b.value, c.value, d.value), (di, dj, dk)


def test_multi_sed_photoion_equals_oracle(hh, orc, pkg, gold):
    """photoion_rates_multi / photo_out_multi (three SEDs: black body, power law, quasar-like) against the
    oracle's orc_photoion_rates3 on the reference's own tables, every on/off combination of the SEDs."""
    from test_oracle_golden import _pl_tables
    T = _pl_tables(orc, pkg, gold)
    z = gold("rad_tables_pl_qpl.npz")
    for idx, pre in ((1, "pl_"), (2, "qpl_")):
        a = [np.ascontiguousarray(z[pre + k]) for k in ("photo_thick", "photo_thin", "heat_thick", "heat_thin")]
        hh.hh_set_sed(idx, *[_p(x) for x in a], C.c_int(int(z[pre + "limits"][0])), C.c_int(int(z[pre + "limits"][1])))
    vec = gold("funcvec.npz")["photoion_heat"].reshape(-1, 30)[:120]
    rng = np.random.default_rng(5)
    out = np.empty(5)
    ref = orc.PhotRates()
    for n, row in enumerate(vec):
        nf = np.array([row[8], 10.0 ** rng.uniform(5, 7), 10.0 ** rng.uniform(5, 7)])
        nf *= np.array([(n >> 0) & 1, (n >> 1) & 1, (n >> 2) & 1], dtype=float)
        cin = np.ascontiguousarray(row[:6])
        for heat in (0, 1):
            hh.hh_photoion_multi(_p(cin), C.c_double(row[6]), _p(nf), C.c_double(row[7]), C.c_int(heat), _p(out))
            orc.lib().orc_photoion_rates3(C.byref(T.c), *[C.c_double(x) for x in cin], C.c_double(row[6]),
                                          (C.c_double * 3)(*nf), C.c_double(row[7]), C.c_int(1 - heat), C.byref(ref))
            r = ref.as_array()
            assert np.array_equal(out, r[[0, 1, 2, 18, 20]]), (n, heat)
            # ... and with cross sections and factors read band by band (BandDataByRow, the three-SED heating kernel)
            out_rows = np.empty(5)
            hh.hh_photoion_multi_rows(_p(cin), C.c_double(row[6]), _p(nf), C.c_double(row[7]), C.c_int(heat), _p(out_rows))
            assert np.array_equal(out_rows, out), (n, heat, "rows")
        hh.hh_photo_out_multi.restype = C.c_double
        assert hh.hh_photo_out_multi(_p(cin), _p(nf)) == r[20]


@pytest.mark.parametrize("src", [(1, 1, 1), (7, 3, 12), (128, 128, 128), (255, 1, 77), (512, 300, 2)])
def test_per_shell_geometry_of_the_fast_sweep_equals_the_general_functions(harness, src):
    """csrc/c2ray_shell.hpp (what k_sweep_shell_fast computes per cell from per-shell constants) against
    short_characteristic + shell_position + shell_decode, every cell of shells 2..40 and a few large shells: decode,
    weights and path bit for bit; corner positions equal wherever the weight is not exactly zero."""
    what = (C.c_int * 8)()
    assert harness.hh_check_shell_geometry(2, 40, *src, what) == 0, list(what)


@pytest.mark.parametrize("s", [127, 128, 255, 256, 639, 640])
def test_per_shell_geometry_of_large_shells(harness, s):
    """The same for single large shells, up to the last one the fast kernel serves (SHELL_FAST_MAX = 640): the magic
    divisions of the thread -> cell map, the corner formulas and Markstein's path division have no small-shell bias to
    hide behind (round-3 ADVICE: shells beyond 40 rested on the end-to-end GPU tests alone)."""
    what = (C.c_int * 8)()
    assert harness.hh_check_shell_geometry(s, s, 300, 17, 511, what) == 0, list(what)


def test_per_shell_constants_hold_for_every_shell_of_the_largest_mesh(harness):
    """(s - 1/2) / s * s == s - 1/2 in double (the corners on the edges of a face have weight exactly 0), and the magic
    divisions of the thread -> cell map, up to the largest shell a mesh can have (checked inside the harness for the
    shells it visits; here the arithmetic claim for all of them)."""
    s = np.arange(1, 4097, dtype=np.float64)
    assert np.array_equal(((s - 0.5) / s) * s, s - 0.5)
