#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: the first outer iterations of BASELINE configs[2] WITH HEATING at 256^3 -- bench.py's eight
sources of 1e56 photons/s, neutral start, thermal evolution on -- by the reference's SERIAL build: the non-converged count
after every global pass (the iteration history evolve3D's exit test reads, files_for_3D/evolve.F90:163,488).

A whole serial call at this size is 52 outer iterations of up to ten minutes; this script runs the first N with the clock of
oracle/probe/pass_timer.c linked in (oracle/ref_build.sh 256 timer: it also wraps evolve0D_global and prints the count of
every pass), and, for the record, the same with the OpenMP build.  The two DIFFER from the eighth iteration on
(1892540 / 1892544 cells): the reference's OpenMP build is not equivalent to its serial build in heating runs of this size
(in isothermal runs it is, bit for bit: make_golden_n64.py --check, tests/golden/n256_iso_8src.npz), so the fixture written by
`make_golden_n64.py 256 --bench-sources --omp 8` (heating) is NOT a golden vector; only its inputs are used here.

    python oracle/make_golden_first_iterations.py tests/golden/n256_heat_8src_first14.npz 9     (~25 min; the scalars evolve3D
    read -- c1_dt, c1_dr, c1_vol, c1_zred, ... -- are taken from the fixture given, which may be the one being rewritten)
"""
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(HERE))
import refrun  # noqa: E402


def run_timed(n, mode, iterations, threads, sources):
    exe = refrun.REFDIR / (f"N{n}_omp" if mode == "omp" else f"N{n}") / "C2Ray_3D_timed"
    run = refrun.REFDIR / f"run_first_iterations_{mode}"
    if run.exists():
        shutil.rmtree(run)
    (run / "results").mkdir(parents=True)
    with open(run / "test_sources.dat", "w") as f:
        f.write(f"{len(sources)}\n")
        for p in sources:
            f.write(f"{p[0]} {p[1]} {p[2]} {p[3]:.6e}\n")
    (run / "input").write_text("0 0 0 0 0\n1e4\nn\nn\nn\n1\n1\n1\n")   # no output streams, T0 = 1e4 K, NOT isothermal
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), C2R_REF_STOP_AFTER=str(iterations))
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe), "input"], cwd=run, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    shutil.rmtree(run, ignore_errors=True)
    return [int(c) for _, c in re.findall(r"iteration (\d+) took [\d.]+ s \(entry to entry\), non-converged cells (\d+)", r.stderr)]


def main():
    inputs = np.load(sys.argv[1])            # the c1_* scalars of a tapped run of the same problem (any build)
    niter = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    n = int(inputs["c1_mesh"][0])
    subprocess.run([str(HERE / "ref_build.sh"), str(n), "timer"], check=True)
    subprocess.run([str(HERE / "ref_build.sh"), str(n), "omp", "timer"], check=True)
    sys.path.insert(0, str(HERE))
    from make_golden_n64 import bench_sources
    src = bench_sources(n)
    inputs = {k: inputs[k] for k in inputs.files}
    inputs_files = list(inputs)
    out = {k: inputs[k] for k in inputs_files if k.startswith("c1_") and not k.startswith("c1_sha") and k not in
           ("c1_conv_flags", "c1_xHII_line", "c1_T_line", "c1_sum_nbox", "c1_photon_loss", "c1_reccoef_after")}
    out["conv_flags_serial"] = np.array(run_timed(n, "serial", niter, 1, src), dtype=np.int64)
    out["conv_flags_openmp"] = np.array(run_timed(n, "omp", niter, 8, src), dtype=np.int64)
    if "c1_conv_flags" in inputs:          # a fixture of make_golden_n64.py --omp: its own history, for the record
        out["conv_flags_openmp_whole_run"] = inputs["c1_conv_flags"][:niter]
    elif "conv_flags_openmp_whole_run" in inputs:
        out["conv_flags_openmp_whole_run"] = inputs["conv_flags_openmp_whole_run"]
    np.savez_compressed(ROOT / "tests" / "golden" / f"n{n}_heat_8src_first{niter}.npz", **out)
    print("serial", out["conv_flags_serial"], "openmp", out["conv_flags_openmp"])


if __name__ == "__main__":
    main()
