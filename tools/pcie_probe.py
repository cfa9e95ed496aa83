#!/usr/bin/env python3
"""Time what crosses PCIe per evolve3D call at the bench's size: the state handed over at the start (c2r_upload_state)
and read back at the end (c2r_download_state), on host arrays that stay allocated as the reference's module arrays do
(pageable; page-locking them with hipHostRegister was measured too: 11.7 against 12.0 ms each way, nothing to gain)."""
import ctypes as C, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, __graft_entry__ as ge
pkg = ge.load_package()
n = 256
mat, grid, src, cosmo = bench.config3_inputs(pkg, n, 8, heating=False, neutral=False)
e = pkg.HipEngine((n, n, n), 0)
e.set_tables(pkg.RadiationTables.load()); e.set_step(mat, grid, cosmo); e.set_sources(src)
e.upload_state(mat)
e.begin_step(); e.set_rates_to_zero(); e.pass_sources(1, 1); e.global_pass(1.0e7 * pkg.hostphys.YEAR)
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
xh, xhe = np.array(mat.xh, dtype=np.float64).ravel().copy(), np.array(mat.xhe, dtype=np.float64).ravel().copy()
out = {"bytes_each_way": int(xh.nbytes + xhe.nbytes)}
def timed(label):
    up, down = [], []
    for rep in range(4):
        e.synchronize(); t0 = time.perf_counter(); e._chk(e.lib.c2r_upload_state(e.h, dp(xh), dp(xhe), None)); e.synchronize(); up.append(1e3 * (time.perf_counter() - t0))
        e.synchronize(); t0 = time.perf_counter(); e._chk(e.lib.c2r_download_state(e.h, dp(xh), dp(xhe), None)); e.synchronize(); down.append(1e3 * (time.perf_counter() - t0))
    out[label] = {"upload_ms": [round(v, 2) for v in up], "download_ms": [round(v, 2) for v in down]}
timed("pageable")
print(json.dumps(out))
