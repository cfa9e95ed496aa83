#!/usr/bin/env python3
"""Time what crosses PCIe per evolve3D call at the bench's size: the state handed over at the start (c2r_set_state /
upload_state) and the results read back at the end (download_state), next to the iterations in between."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, __graft_entry__ as ge
pkg = ge.load_package()
n = 256
mat, grid, src, cosmo = bench.config3_inputs(pkg, n, 8, heating=False, neutral=False)
e = pkg.HipEngine((n, n, n), 0)
e.set_tables(pkg.RadiationTables.load()); e.set_step(mat, grid, cosmo); e.set_sources(src)
out = {}
for rep in range(3):
    e.synchronize(); t0 = time.perf_counter(); e.upload_state(mat); e.synchronize(); out.setdefault("upload_ms", []).append(1e3 * (time.perf_counter() - t0))
e.begin_step(); e.set_rates_to_zero(); e.pass_sources(1, 1); e.global_pass(1.0e7 * pkg.hostphys.YEAR)
for rep in range(3):
    e.synchronize(); t0 = time.perf_counter(); e.download_state(mat); e.synchronize(); out.setdefault("download_ms", []).append(1e3 * (time.perf_counter() - t0))
nb_up = sum(getattr(mat, k).nbytes for k in ("ndens", "xh", "xhe") if getattr(mat, k, None) is not None)
out["upload_bytes_min"] = nb_up
out["download_bytes"] = sum(getattr(mat, k).nbytes for k in ("xh", "xhe") if getattr(mat, k, None) is not None)
print(json.dumps(out))
