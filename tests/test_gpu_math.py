"""Device arithmetic against the host libm (glibc, the one the reference links).  IEEE operations
(sqrt, division) and the restated exp/log10/pow of csrc/c2ray_math.hpp must be bit-identical to it;
the device's own libm (ocml) is measured beside them for the record (gpurun_out/gpu_math_vs_glibc.json,
DESIGN.md 'Conditioning')."""
import ctypes as C
import json
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _build(name, flags):
    so = ROOT / "tests" / name
    src = ROOT / "tests" / "gpu_math_probe.hip"
    hdrs = list((ROOT / "c2-ray3dm1d_helium_amd" / "csrc").glob("*.hpp"))
    if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in [src] + hdrs):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", *flags,
                        "-o", str(so), str(src)], check=True)
    return C.CDLL(str(so))


@pytest.fixture(scope="module")
def probe():
    return _build("_gpu_math_probe.so", ["-DC2R_PROBE_USE_PRODUCT_MATH"])


@pytest.fixture(scope="module")
def probe_ocml():
    return _build("_gpu_math_probe_ocml.so", [])


def run(probe, op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(x if y is None else y, dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    rc = probe.probe_math(op, x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
    assert rc == 0
    return out


def test_ieee_ops_bit_exact(probe):
    rng = np.random.default_rng(1)
    x = 10.0 ** rng.uniform(-300, 300, 2_000_000)
    y = 10.0 ** rng.uniform(-300, 300, 2_000_000)
    assert np.array_equal(run(probe, 3, x), np.sqrt(x))
    with np.errstate(over="ignore", under="ignore"):
        assert np.array_equal(run(probe, 4, x, y), x / y)


def test_reciprocal_and_volume_division_bit_exact(probe):
    """recip_nr (scale_int2/3's 1/x without operand scaling) and div_by_vol (Markstein's sequence with the
    quotient-magnitude guard) against the IEEE division, inside and outside their fast ranges."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    x = 10.0 ** rng.uniform(-150, 150, n)
    # mantissas of all ones and their neighbours: where the reciprocal's last Newton step is most delicate
    edge = np.ldexp(np.nextafter(2.0, 0.0), rng.integers(-400, 400, 4096))
    x[:4096] = edge
    x[4096:8192] = np.nextafter(edge[:4096], 0.0)
    x[8192:12288] = np.ldexp(1.0, rng.integers(-400, 400, 4096))
    assert np.array_equal(run(probe, 5, x), 1.0 / x)
    a = 10.0 ** rng.uniform(-320, 300, n) * rng.choice([-1.0, 1.0], n)
    a[:1000] = 0.0
    b = 10.0 ** rng.uniform(-30, 130, n)       # vol_ph ~ 1e60..1e75; below 1 the guard sends everything to the division
    b[1000:2000] = 10.0 ** rng.uniform(-300, 300, 1000)
    with np.errstate(over="ignore", under="ignore"):
        assert np.array_equal(run(probe, 6, a, b), a / b)
        assert np.array_equal(run(probe, 7, a, b), (a * 2.0 ** -30) / b)
        assert np.array_equal(run(probe, 8, a, b), np.zeros(n))


def test_sweep_weight_bit_exact(probe):
    """The column sweep's 1/max(0.6, N sigma) with the division's sequence minus operand scaling (c2ray_shell.hpp)
    against the IEEE operations."""
    rng = np.random.default_rng(12)
    n = 2_000_000
    cd = 10.0 ** rng.uniform(-5, 30, n)
    cd[:1000] = 0.0
    sig = np.full(n, 6.30e-18)
    sig[n // 2:] = 10.0 ** rng.uniform(-19, -16, n - n // 2)
    cd[1000:2000] = 0.6 / sig[1000:2000]                                           # the maximum's switch-over
    assert np.array_equal(run(probe, 13, cd, sig), 1.0 / np.maximum(0.6, cd * sig))


def test_pair_table_position_equals_single(probe):
    """tau_table_positions (both logs as one straight line, polynomial path on demand) == tau_table_position."""
    rng = np.random.default_rng(8)
    n = 1_000_000
    ta = 10.0 ** rng.uniform(-22, 5, n)
    tb = ta * (1.0 + 10.0 ** rng.uniform(-9, 1, n))
    ta[:1000] = 0.0
    ta[1000:2000] = rng.uniform(0.9, 1.1, 1000)
    one_a, one_b = run(probe, 11, ta), run(probe, 11, tb)
    assert np.array_equal(run(probe, 9, ta, tb), one_a)
    assert np.array_equal(run(probe, 10, ta, tb), one_b)
    # and the single version against glibc's log10 on the host
    lt = np.array([math.log10(max(1e-20, v)) for v in ta[:200_000]])
    od = np.minimum(2000.0, 1.0 + (lt + 20.0) / ((4.0 + 20.0) / 2000.0))
    ip = np.floor(od)
    assert np.array_equal(one_a[:200_000], ip + (od - ip) * 0.5)


def _cases(n):
    rng = np.random.default_rng(2)
    return [("exp", 0, -10.0 ** rng.uniform(-12, 3.2, n), None),
            ("exp_any", 0, rng.uniform(-800, 720, n), None),
            ("log10", 1, 10.0 ** rng.uniform(-20, 4, n), None),
            ("log10_wide", 1, 10.0 ** rng.uniform(-300, 300, n), None),
            ("pow", 2, 10.0 ** rng.uniform(-3, 3, n), rng.uniform(-2.5, 2.5, n)),
            ("pow_frac", 2, 10.0 ** rng.uniform(-20, 0, n), rng.uniform(0.1, 2.0, n))]


def _ref(op, x, y):
    with np.errstate(all="ignore"):
        if op == 0:
            return np.exp(x)  # numpy calls the C library's exp for float64 scalars loops -> checked below
        if op == 1:
            return np.log10(x)
        return np.power(x, y)


def _glibc(op, x, y):
    f = [math.exp, math.log10, math.pow][op]
    out = np.empty_like(x)
    for i in range(x.size):
        try:
            out[i] = f(x[i]) if op < 2 else f(x[i], y[i])
        except OverflowError:
            out[i] = np.inf
    return out


def test_restated_libm_bit_identical_on_device(probe, probe_ocml):
    """exp/log10/pow of csrc/c2ray_math.hpp on the GPU == glibc on the host, every input; ocml's
    mismatch rate is recorded beside it."""
    log = {}
    n = 400_000
    for name, op, x, y in _cases(n):
        ref = _glibc(op, x, y)
        got = run(probe, op, x, y)
        bad = ~((got == ref) | (np.isnan(got) & np.isnan(ref)))
        log[name + "_restated_mismatch_fraction"] = float(bad.mean())
        o = run(probe_ocml, op, x, y)
        log[name + "_ocml_mismatch_fraction"] = float((o != ref).mean())
        assert not bad.any(), (name, x[bad][:3], got[bad][:3], ref[bad][:3])
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "gpu_math_vs_glibc.json").write_text(json.dumps(log, indent=1))
    print(log)
