import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
GOLD = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def otables(orc, pkg):
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        return orc.Tables({k: t[k] for k in t.files})


@pytest.fixture(scope="session")
def gold():
    def load(name):
        return np.load(GOLD / name)
    return load


@pytest.fixture(scope="session")
def harness():
    """The product's device functions compiled for the host (test-only build)."""
    import ctypes
    so = ROOT / "tests" / "_host_harness.so"
    src = ROOT / "tests" / "host_harness.cpp"
    hdrs = list((ROOT / "c2-ray3dm1d_helium_amd" / "csrc").glob("*.hpp"))
    if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in [src] + hdrs):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-std=c++17", "-o", str(so), str(src)],
                       check=True)
    lib = ctypes.CDLL(str(so))
    lib.hh_photo_out_only.restype = ctypes.c_double
    return lib


def tap_case(z, call):
    """(inputs dict, outputs dict) of one tapped evolve3D call of a golden file."""
    i = {k[len(f"c{call}_in_"):]: z[k] for k in z.files if k.startswith(f"c{call}_in_")}
    o = {k[len(f"c{call}_out_"):]: z[k] for k in z.files if k.startswith(f"c{call}_out_")}
    o["conv_flags"] = z[f"c{call}_conv_flags"]
    return i, o


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def collect_from_ranks(procs, q, nresults=1, timeout=300):
    """Results that worker processes put on `q`, without waiting out a long time-out in silence when a worker died:
    polls the queue, fails as soon as a worker has exited with an error, and ends every worker when time is up."""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < nresults:
        try:
            out.append(q.get(timeout=2))
            continue
        except queue.Empty:
            pass
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        if dead or time.time() - t0 > timeout:
            for p in procs:
                if p.is_alive():
                    p.kill()
            raise AssertionError(f"worker ranks failed: exit codes {[p.exitcode for p in procs]}" if dead else
                                 f"no result from the worker ranks within {timeout} s")
    for p in procs:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
            raise AssertionError("a worker rank did not exit")
        assert p.exitcode == 0, [p.exitcode for p in procs]
    return out[0] if nresults == 1 else out
