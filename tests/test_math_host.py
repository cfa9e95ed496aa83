"""csrc/c2ray_math.hpp (the restated glibc exp / log10 / pow) against the libm the reference links,
on the host: bit-identical on every input.  The same source runs on the GPU (tests/test_gpu_math.py)."""
import ctypes as C

import numpy as np
import pytest

dp = C.POINTER(C.c_double)


def run(fn, op, x, y=None):
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(x if y is None else y)
    out = np.empty_like(x)
    fn(op, x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def cases(n, seed=7):
    rng = np.random.default_rng(seed)
    return {
        "exp general": (0, rng.uniform(-800, 720, n), None),
        "exp doric (lambda*dt)": (0, -10.0 ** rng.uniform(-20, 3.2, n), None),
        "exp tiny/edge": (0, np.array([0.0, -0.0, 1e-300, -1e-300, 5e-17, -5e-17, -745.2, -745.0, -708.4, -708.39,
                                       -1023.9, -1024.0, -1e5, 709.7, 709.8, 1e4, -np.inf, np.inf]), None),
        "log10 wide": (1, 10.0 ** rng.uniform(-300, 300, n), None),
        "log10 tau": (1, 10.0 ** rng.uniform(-20, 4, n), None),
        "log10 near 1": (1, rng.uniform(0.9, 1.1, n), None),
        "log10 exact": (1, np.array([1.0, 10.0, 100.0, 1e-20, 0.5, 2.0, 0.9375, 1.064697265625]), None),
        "log10_pos wide": (3, 10.0 ** rng.uniform(-300, 300, n), None),
        "log10_pos tau": (3, 10.0 ** rng.uniform(-20, 4, n), None),
        "log10_pos near 1": (3, rng.uniform(0.9, 1.1, n), None),
        "log10_pos exact": (3, np.array([1.0, 10.0, 100.0, 1e-20, 0.5, 2.0, 0.9375, 1.064697265625, 1.0646972656249998,
                                         0.93749999999999989, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf]), None),
        "pow fits": (2, 10.0 ** rng.uniform(-4, 4, n), rng.uniform(-3, 3, n)),
        "pow x<1": (2, 10.0 ** rng.uniform(-20, 0, n), rng.uniform(0.1, 2.0, n)),
        "pow wide": (2, 10.0 ** rng.uniform(-100, 100, n), rng.uniform(-4, 4, n)),
        "pow edge": (2, np.array([1.0, 1.0, 1e-20, 1e-20, 2.0, 1e300, 1e-300, 0.5]),
                     np.array([0.4092, 1.7592, 0.2, 0.38, 0.5, 2.0, 2.0, -1.5])),
    }


@pytest.mark.parametrize("name", list(cases(1)))
def test_bit_identical_to_glibc(harness, name):
    op, x, y = cases(500_000)[name]
    with np.errstate(all="ignore"):
        a = run(harness.hh_math, op, x, y)
        b = run(harness.hh_libm, op, x, y)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), (name, x[~same][:4], a[~same][:4], b[~same][:4])


def test_division_through_reciprocal_is_correctly_rounded(harness):
    """div_recip (Markstein's fma correction, csrc/c2ray_device.hpp) == IEEE division, including the
    guarded corners: huge/tiny operands, zero, all-ones significands, subnormal quotients."""
    rng = np.random.default_rng(11)
    n = 10_000_000
    b = np.ldexp(1.0 + rng.random(n), rng.integers(-100, 100, n))
    a = np.ldexp(1.0 + rng.random(n), rng.integers(-200, 200, n)) * rng.choice([-1.0, 1.0], n)
    edge_a = np.array([0.0, 1e-310, 5e-324, 1e300, 1e-200, 3.0, 1.0, 2.0 ** 600, 2.0 ** -600])
    edge_b = np.array([3.0, 7.0, 1e10, 1e-10, 1e200, np.nextafter(2.0, 0.0), np.nextafter(2.0, 0.0), 3.0, 3.0])
    a = np.concatenate([a, edge_a])
    b = np.concatenate([b, edge_b])
    out = np.empty_like(a)
    harness.hh_div_recip(a.size, a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp))
    with np.errstate(all="ignore"):
        assert np.array_equal(out, a / b)


def test_log_table_with_the_power_of_two_folded_in(harness):
    """gm::log_table_path4 (the 256-entry table k_rates keeps in LDS) against gm::log_table_path and against glibc's
    log10, 4e6 arguments over the whole range of optical depths and around every table boundary."""
    import ctypes as C
    rng = np.random.default_rng(77)
    x = np.concatenate([10.0 ** rng.uniform(-20.0, 16.0, 2_000_000), rng.uniform(0.4, 2.2, 1_000_000),
                        np.ldexp(rng.uniform(0.5, 2.0, 1_000_000), rng.integers(-70, 60, 1_000_000))])
    # every boundary of the 256 cases, one ulp either side
    hx = (0x3FE00000 + (np.arange(257, dtype=np.uint64) << np.uint64(13))) << np.uint64(32)
    edges = np.concatenate([(hx - np.uint64(1)).view(np.float64), hx.view(np.float64)])
    x = np.ascontiguousarray(np.concatenate([x, edges[np.isfinite(edges) & (edges > 0)]]))
    assert harness.hh_check_log_table4(C.c_int(len(x)), x.ctypes.data_as(C.POINTER(C.c_double))) == 0
