#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: BASELINE configs[1] -- 64^3 uniform density, one point source, with
heating -- run with the reference itself (flang build, tapped at evolve3D), reduced to a compact
fixture: the per-call scalars evolve3D read, the uniform initial state, and for every call the
iteration history, SHA-256 of every output array, and the ionised-fraction line through the source
(the reference's own Ifront diagnostic, files_for_3D/output.F90:192-244).

    python oracle/make_golden_n64.py      (dev container; ~2 minutes)
    python oracle/make_golden_n64.py 128 64,64,64,3e55 10,120,70,1e55     (a larger box, two sources: ~10 minutes)
    python oracle/make_golden_n64.py 256 --bench-sources --iso --omp 8    (BASELINE configs[2] exactly as bench.py
        runs it: 256^3, the eight seeded sources of 1e56 photons/s, isothermal, neutral start; the reference's
        OpenMP build -- four evolve3D calls, 55 + 9 + 8 + 8 outer iterations of up to 50 s on 8 threads: ~1 h)
    python oracle/make_golden_n64.py 64 --omp 8 --check    (no fixture written: the OpenMP build's hashes are
        compared with the committed fixture of the serial build -- the octants of files_for_3D/evolve_source.F90
        touch disjoint cells, so the two builds write the same bits)
"""
import argparse
import hashlib
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(HERE))
import refrun  # noqa: E402


def sha(a):
    h = hashlib.sha256()
    a = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    for o in range(0, a.size, 1 << 28):      # memory-mapped dumps of GBs: hashed piece by piece
        h.update(a[o:o + (1 << 28)].tobytes())
    return h.hexdigest()


def bench_sources(n):
    """bench.py:config3_inputs -- the eight sources of BASELINE configs[2]."""
    pos = np.random.default_rng(12345).integers(1, n + 1, size=(8, 3))
    return [(int(p[0]), int(p[1]), int(p[2]), 1e56) for p in pos]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mesh", nargs="?", type=int, default=64)
    ap.add_argument("sources", nargs="*", help="i,j,k,photons_per_s")
    ap.add_argument("--bench-sources", action="store_true")
    ap.add_argument("--iso", action="store_true", help="isothermal run (no temperature grid, no heating rates)")
    ap.add_argument("--omp", type=int, default=0, help="the reference's OpenMP build on this many threads")
    ap.add_argument("--pl", action="store_true", help="the -DPL -DQUASARS build: sources are i,j,k,S_BB,S_PL,S_QPL")
    ap.add_argument("--check", action="store_true", help="compare with the committed fixture instead of writing it")
    ap.add_argument("--keep", action="store_true", help="keep the run directory (tap dumps)")
    a = ap.parse_args()
    N = a.mesh
    SOURCES = [(32, 32, 32, 1e54)]
    if a.sources:
        SOURCES = [tuple(float(x) if i >= 3 else int(x) for i, x in enumerate(s.split(","))) for s in a.sources]
    if a.bench_sources:
        SOURCES = bench_sources(N)
    NAME = f"n{N}_{'iso' if a.iso else 'heat'}{'_pl' if a.pl else ''}_{len(SOURCES)}src"

    subprocess.run([str(HERE / "ref_build.sh"), str(N)] + (["omp"] if a.omp else []) + (["pl"] if a.pl else []), check=True)
    run = refrun.run_reference(N, SOURCES, isothermal=a.iso, steps_per_slice=1, name="golden_" + NAME,
                               omp=bool(a.omp), threads=max(1, a.omp), timeout=6 * 3600, pl=a.pl)
    conv = refrun.parse_log(run)
    out = {"ncalls": np.int32(len(conv))}
    grids = ["xh", "xhe", "phih_grid", "phihe_grid", "xh_av", "xhe_av"] + ([] if a.iso else ["temperature", "phiheat"])
    for call in range(1, len(conv) + 1):
        tin = refrun.read_records(run / "results" / f"tap_{call:04d}_in.bin", mmap=True)
        tout = refrun.read_records(run / "results" / f"tap_{call:04d}_out.bin", mmap=True)
        p = f"c{call}_"
        for k in ["mesh", "dt", "zred", "H0", "Omega0", "dr", "vol", "srcpos", "NormFlux", "S_star", "isothermal",
                  "temper_val", "clumping", "reccoef"] + (["NormFluxPL", "pl_S_star", "NormFluxQPL", "qpl_S_star"] if a.pl else []):
            out[p + k] = np.array(tin[k])
        assert np.all(tin["ndens"] == tin["ndens"][0])
        out[p + "ndens_uniform"] = tin["ndens"][0]
        if call == 1:
            for k in ["xh", "xhe"] + ([] if a.iso else ["temperature"]):
                comp = tin[k].reshape(-1, N ** 3)
                assert np.all(comp == comp[:, :1])
                out["c1_" + k + "_uniform"] = comp[:, 0].copy()
        out[p + "conv_flags"] = np.array(conv[call - 1], dtype=np.int32)
        for k in grids:
            out[p + "sha_" + k] = np.array(sha(tout[k]))
        xh1 = tout["xh"][N ** 3:].reshape(N, N, N, order="F")
        j0, k0 = SOURCES[0][1] - 1, SOURCES[0][2] - 1   # the line through the first source
        out[p + "xHII_line"] = xh1[:, j0, k0].copy()
        if not a.iso:
            out[p + "T_line"] = tout["temperature"][:N ** 3].reshape(N, N, N, order="F")[:, j0, k0].copy()
        out[p + "sum_nbox"] = np.array(tout["sum_nbox_all"])
        out[p + "photon_loss"] = np.array(tout["photon_loss_all"])
        out[p + "reccoef_after"] = np.array(tout["reccoef"])
        del tin, tout
    if a.omp:
        out["reference_build"] = np.array(f"flang -O2 -fopenmp -DMY_OPENMP, OMP_NUM_THREADS={a.omp}")
    fixture = ROOT / "tests" / "golden" / (NAME + ".npz")
    if a.check:
        old = np.load(fixture)
        bad = [k for k in old.files if k in out and not np.array_equal(old[k], out[k])]
        print("compared", len([k for k in old.files if k in out]), "entries with", fixture.name, "-> differing:", bad)
        if not a.keep:
            shutil.rmtree(run)
        sys.exit(1 if bad else 0)
    np.savez_compressed(fixture, **out)
    if not a.keep:
        shutil.rmtree(run)  # GBs of tap dumps: scratch
    print("calls", [len(c) for c in conv], "fixture", fixture.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
