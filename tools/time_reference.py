#!/usr/bin/env python3
"""The reference ITSELF (flang -O2 -fopenmp build of /root/reference, oracle/ref_build.sh N omp timer) on the
benchmark's own inputs: N^3 uniform density, the bench's 8 seeded sources of 1e56 photons/s, isothermal, neutral
start, dt = 1e7 yr, all output streams off -- SURVEY.md section 8d(1).  A whole evolve3D call at 256^3 is 55 outer
iterations of ~40 s; the clock of oracle/probe/pass_timer.c (linked around the reference's do_grid call) prints every
pass as it ends and stops the run after --iterations of them.  DEV CONTAINER only (the reference does not travel).

    tools/time_reference.py --mesh 256 --iterations 2 --threads 8 --out profiles/r04_reference_256.json
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))


def time_reference(mesh=256, iterations=2, threads=8, timeout=6 * 3600):
    """Run the timed reference binary for `iterations` outer iterations; returns the result dictionary, or None when the
    binary is not there (oracle/ref_build.sh N omp timer builds it in the dev container; it travels in oracle/_ref/)."""
    import refrun
    n = mesh
    exe = refrun.ref_binary(n, "timed", omp=True)
    if not exe.exists():
        return None
    pos = np.random.default_rng(12345).integers(1, n + 1, size=(8, 3))   # bench.py:config3_inputs
    run = refrun.REFDIR / f"run_reference_timing_N{n}"
    if run.exists():
        shutil.rmtree(run)
    (run / "results").mkdir(parents=True)
    with open(run / "test_sources.dat", "w") as f:
        f.write("8\n")
        for p in pos:
            f.write(f"{p[0]} {p[1]} {p[2]} {1e56:.6e}\n")
    (run / "input").write_text("0 0 0 0 0\n1e4\ny\nn\nn\n1\n1\n1\n")
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(threads)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env["C2R_REF_STOP_AFTER"] = str(iterations)
    t0 = time.perf_counter()
    # the clock's lines are passed on as they come (a long run must not look hung), and kept
    proc = subprocess.Popen([str(exe), "input"], cwd=run, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    lines = []

    def reader():
        for line in proc.stderr:
            lines.append(line)
            if "pass_timer" in line:
                sys.stderr.write(line)
                sys.stderr.flush()

    import threading
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    try:
        proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:      # a run that stopped printing is ended here, not waited for
        proc.kill()
        proc.wait()
    th.join(timeout=10)
    err = "".join(lines)
    wall = time.perf_counter() - t0
    shutil.rmtree(run, ignore_errors=True)
    its = [float(x) for x in re.findall(r"pass_timer: iteration \d+ took ([\d.]+) s", err)]
    passes = [float(x) for x in re.findall(r"pass_timer: pass \d+ \(niter \d+\) took ([\d.]+) s", err)]
    if not its:
        sys.stderr.write(err[-2000:])
        return None
    return {"binary": str(exe.relative_to(ROOT)), "what": f"the reference (flang -O2 -fopenmp), {n}^3 uniform density, the bench's 8 sources of 1e56 photons/s, "
            "isothermal 1e4 K, neutral start, dt = 1e7 yr, output streams off: its first outer iterations of evolve3D",
            "threads": threads, "host_cpus": os.cpu_count(),
            "how": "oracle/probe/pass_timer.c wrapped (ld --wrap) around the reference's do_grid call: s_per_iteration is entry-to-entry of "
                   "consecutive passes (pass_all_sources + global_pass + the loop's bookkeeping), s_per_pass the pass alone; the run is "
                   "ended at the entry of the next pass",
            "s_per_iteration": its, "s_per_pass": passes, "wall_s_including_setup": wall,
            "cell_updates_per_s_nominal": [n ** 3 * 8 / t for t in its],
            # the last four timed iterations: with enough of them (12+ at 256^3) every source's box has reached the mesh limit,
            # the state the benchmark's metric is defined on
            "s_per_iteration_at_mesh_limit": sum(its[-4:]) / len(its[-4:]),
            "cell_updates_per_s_at_mesh_limit": n ** 3 * 8 / (sum(its[-4:]) / len(its[-4:])),
            "note": "the first iterations of a neutral start trace small sub-boxes; the pre-ionised bench state (every box at the mesh limit) "
                    "is the expensive end -- see DESIGN.md section 5"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--threads", type=int, default=8, help="the reference's OpenMP sweep is at most 8-way (octants)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = time_reference(a.mesh, a.iterations, a.threads)
    if out is None:
        raise SystemExit(f"no timing: is oracle/_ref/N{a.mesh}_omp/C2Ray_3D_timed built (oracle/ref_build.sh {a.mesh} omp timer)?")
    txt = json.dumps(out, indent=1)
    if a.out:
        Path(a.out).write_text(txt)
    print(txt)


if __name__ == "__main__":
    main()
