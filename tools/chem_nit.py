#!/usr/bin/env python3
"""Where the chemistry pass spends its iterations: per outer iteration of the bench workload
from its neutral start -- and of the following evolve3D calls --, the histogram of do_chemistry iterations per cell
(`nit`, evolve_point.F90:516-610), the same for the slowest lane of every wave (a wave lasts as long as its slowest
lane) and the kernel's time.  Needs the diagnostic build: tools/build_variant.sh nit "-DC2R_CHEM_NIT_HIST", run with
C2R_LIB_PATH=c2-ray3dm1d_helium_amd/libc2ray_hip_nit.so.

    C2R_LIB_PATH=$PWD/c2-ray3dm1d_helium_amd/libc2ray_hip_nit.so tools/chem_nit.py [--calls 2] [--max-iter 12]"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--max-iter", type=int, default=12)
    a = ap.parse_args()
    pkg = ge.load_package()
    lib = pkg._lib.load()
    n = a.mesh
    mat, grid, src, cosmo = bench.config3_inputs(pkg, n, 8, neutral=True)
    e = pkg.HipEngine((n, n, n), 0)
    e.set_tables(pkg.RadiationTables.load())
    e.set_batch(8)
    e.enable_timing(True)
    dt = 1.0e7 * pkg.hostphys.YEAR
    nc = n ** 3
    out = (ctypes.c_ulonglong * 136)()
    lib.c2r_debug_chem_nit(out, 1)
    calls = []
    for call in range(a.calls):
        e.set_step(mat, grid, cosmo)
        e.set_sources(src)
        e.upload_state(mat)
        e.begin_step()
        rows, niter, conv = [], 0, nc
        while not (conv < 8 and niter > 1) and niter < 500:
            niter += 1
            e.set_rates_to_zero()
            e.pass_sources(1, 1)
            conv = e.global_pass(dt)
            tm = e.timing()
            lib.c2r_debug_chem_nit(out, 1)
            h = np.array(out[:], dtype=np.int64)
            lane, wave, total = h[:64], h[64:128], int(h[128])
            if niter <= a.max_iter:
                nz = lambda v: {int(k): int(x) for k, x in enumerate(v) if x}
                waves = int(wave.sum())
                rows.append({"iter": niter, "nonconv": int(conv), "chem_ms": tm.chem_ms, "mean_nit": total / nc,
                             "mean_of_wave_max": float((wave * np.arange(64)).sum() / waves),
                             "lane_efficiency": total / (64.0 * float((wave * np.arange(64)).sum())),
                             "cells_by_nit": nz(lane), "waves_by_max_nit": nz(wave)})
        e.end_step()
        e.download_state(mat)
        calls.append({"call": call + 1, "iterations": niter, "per_iteration": rows})
    print(json.dumps({"workload": f"bench.py --neutral-start ({n}^3, 8 sources), {a.calls} evolve3D calls", "calls": calls}))
    e.close()


if __name__ == "__main__":
    main()
