#!/usr/bin/env python3
"""Time the product the north star describes: the reference's own Fortran driver (C2Ray.F90, set-up and output modules,
unmodified) linked with this library's drop-in modules (oracle/_ref/N256/C2Ray_3D_hip, built in the dev container by
`oracle/ref_build.sh 256`), on the benchmark's workload -- 256^3, uniform density, the bench's 8 seeded sources of
1e56 photons/s, isothermal, dt = 1e7 yr, from the reference's neutral test-problem start, all output streams off.

Run ON THE GPU BOX (through gpurun).  Per evolve3D call the drop-in writes "evolve3D loop: N iterations in S s" to
Timings.log (C2RAY_HIP_TIMING=1; the reference's own stamps have tenths of a second); next to it the script runs
`bench.py --neutral-start --warmup 0 --steps N` -- the Python host on the same iterations of the same first time step
-- and writes both to gpurun_out/dropin_timing.json (copy to profiles/rNN_dropin_timing.json).

    tools/time_dropin.py [--steps-per-slice 1] [--stepwise]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--stepwise", action="store_true", help="C2RAY_HIP_STEPWISE=1: the reference's call-by-call loop")
    ap.add_argument("--keep-state", type=int, default=0, help="C2RAY_HIP_KEEP_STATE: 1 keeps xh / xhe / ndens on the device between calls when "
                    "a sample of the host arrays allows it and leaves phihe_grid there; 2 leaves every rate grid there (no output stream 3)")
    ap.add_argument("--no-bench", action="store_true", help="skip the bench.py run on the same iterations")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "dropin_timing.json"))
    a = ap.parse_args()
    import refrun
    n = a.mesh
    exe = refrun.ref_binary(n, "hip")
    if not exe.exists():
        raise SystemExit(f"{exe} missing: run oracle/ref_build.sh {n} in the dev container")
    # the bench's sources (bench.py:config3_inputs)
    pos = np.random.default_rng(12345).integers(1, n + 1, size=(8, 3))
    run = refrun.REFDIR / f"run_dropin_timing_N{n}"
    if run.exists():
        import shutil
        shutil.rmtree(run)
    (run / "results").mkdir(parents=True)
    with open(run / "test_sources.dat", "w") as f:
        f.write("8\n")
        for p in pos:
            f.write(f"{p[0]} {p[1]} {p[2]} {1e56:.6e}\n")
    with open(run / "input", "w") as f:
        f.write("0 0 0 0 0\n1e4\ny\nn\nn\n1\n1\n1\n")   # no output streams, T0, isothermal, no restart, no midpoint, slice 1, 1 step, 1 output
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env["C2RAY_HIP_TIMING"] = "1"
    if a.stepwise:
        env["C2RAY_HIP_STEPWISE"] = "1"
    if a.keep_state:
        env["C2RAY_HIP_KEEP_STATE"] = str(a.keep_state)
    t0 = time.perf_counter()
    with open(run / "stdout.txt", "w") as so:
        subprocess.run([str(exe), "input"], cwd=run, env=env, stdout=so, stderr=subprocess.STDOUT, timeout=1500, check=True)
    wall = time.perf_counter() - t0
    calls = []
    for line in (run / "results" / "Timings.log").read_text(errors="replace").splitlines():
        m = re.search(r"evolve3D loop:\s*(\d+)\s*iterations in\s*([\d.]+)\s*s", line)
        if m:
            calls.append({"iterations": int(m.group(1)), "seconds": float(m.group(2)),
                          "ms_per_iteration": 1e3 * float(m.group(2)) / int(m.group(1))})
    parts = re.findall(r"evolve3D call: set-up\s*([\d.]+)\s*s, loop\s*([\d.]+)\s*s, results\s*([\d.]+)\s*s",
                       (run / "results" / "Timings.log").read_text(errors="replace"))
    for c, (a_, b_, c_) in zip(calls, parts):
        c.update(setup_s=float(a_), results_s=float(c_), call_s=float(a_) + float(b_) + float(c_),
                 ms_per_iteration_whole_call=1e3 * (float(a_) + float(b_) + float(c_)) / c["iterations"])
    # kernels of every iteration, call by call (an iteration count that starts again at 1 opens a new call)
    kern = re.findall(r"evolve3D kernels, iteration\s*(\d+): sweep\s*([\d.]+) ms, rates\s*([\d.]+) ms, chemistry\s*([\d.]+) ms",
                      (run / "results" / "Timings.log").read_text(errors="replace"))
    percall = []
    for it, sw, ra, ch in kern:
        if int(it) == 1:
            percall.append([])
        percall[-1].append((float(sw), float(ra), float(ch)))
    for c, rows in zip(calls, percall):
        c["kernel_ms_mean"] = {"sweep": sum(r[0] for r in rows) / len(rows), "rates": sum(r[1] for r in rows) / len(rows),
                               "chemistry": sum(r[2] for r in rows) / len(rows)}
        c["chemistry_ms_by_iteration"] = [r[2] for r in rows]
        c["sweep_ms_by_iteration"] = [r[0] for r in rows]
    if not calls:
        raise SystemExit("no 'evolve3D loop' line in Timings.log")
    sys.path.insert(0, str(ROOT))
    import bench
    out = {"source_sha16": bench.kernel_source_sha16(), "binary": str(exe.relative_to(ROOT)), "mode": "stepwise" if a.stepwise else "c2r_iteration", "keep_state": a.keep_state, "driver_wall_s": wall,
           "evolve3D_calls": calls, "iterations": calls[0]["iterations"], "ms_per_iteration": calls[0]["ms_per_iteration"]}
    log = (run / "results" / "C2Ray.log").read_text(errors="replace")
    out["calls_with_state_kept"] = log.count("xh, xhe, temperature_grid kept on the device")
    out["calls_with_ndens_kept_or_rescaled"] = log.count("ndens kept on the device") + log.count("ndens rescaled on the device")
    if a.no_bench:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(out, indent=1))
        print(json.dumps(out, indent=1))
        return
    # the Python host on the same iterations of the same first time step
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--neutral-start", "--warmup", "0", "--steps", str(calls[0]["iterations"]),
                        "--no-cpu-baseline", "--mesh", str(n)], capture_output=True, text=True, timeout=900)
    try:
        b = json.loads(r.stdout.strip().splitlines()[-1])
        out["bench_ms_per_step_same_box"] = b["ms_per_step"]
        out["bench_kernel_ms_per_step"] = b["kernel_ms_per_step"]
    except Exception as e:
        out["bench_error"] = f"{e}: {r.stderr[-400:]}"
    out["note"] = (f"reference driver + drop-in modules ({out['mode']}), {n}^3, the bench's 8 sources, neutral start, first evolve3D call: "
                   f"{calls[0]['iterations']} outer iterations; bench_ms_per_step_same_box = bench.py --neutral-start --warmup 0 over "
                   "the same iterations on the same box")
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
