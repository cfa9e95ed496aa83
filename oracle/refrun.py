"""TEST INFRASTRUCTURE ONLY (oracle/): run the flang-built reference binary.

Helpers to run `oracle/_ref/N<mesh>/C2Ray_3D_{test,tap}` (built from the sources
under /root/reference by oracle/ref_build.sh) in a scratch run directory and to
read back what it wrote.  Nothing here is imported by the product package.

Run-time inputs of the reference test target:
  stdin/file lines: 5 stream flags (files_for_3D/output.F90:90), T0
  (mat_ini_test.F90:126), isothermal y/n (:130), restart y/n (:142), midpoint
  y/n (:154), start slice (:157), steps per slice, outputs per slice
  (time_ini.F90:48-53); ./test_sources.dat (sourceprops_test.F90:112-166);
  ../tables/*.tab (cooling_h.f90:83-149) when not isothermal.
"""
from __future__ import annotations

import os
import re
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REFDIR = HERE / "_ref"

_KINDS = {1: np.int32, 2: np.float32, 3: np.float64}


def read_records(path, mmap=False) -> dict:
    """Read a tap dump (see oracle/probe/evolve_tap.f90 for the record layout).  mmap: arrays are read-only views of
    the memory-mapped file (dumps of a 256^3 run are GBs)."""
    out = {}
    data = np.memmap(path, dtype=np.uint8, mode="r") if mmap else Path(path).read_bytes()
    off = 0
    while off < len(data):
        name = bytes(data[off:off + 16]).decode().strip()
        kind, count = struct.unpack_from("<iq", bytes(data[off + 16:off + 28]), 0)
        off += 28
        dt = np.dtype(_KINDS[kind])
        a = np.frombuffer(data, dtype=dt, count=count, offset=off)
        out[name] = a if mmap else a.copy()
        off += count * dt.itemsize
    return out


def ref_binary(mesh: int, which: str = "tap", omp: bool = False, pl: bool = False, lls: bool = False, params: bool = False) -> Path:
    tag = f"N{mesh}" + ("_omp" if omp else "") + ("_pl" if pl else "") + ("_lls" if lls else "") + ("_params" if params else "")
    return REFDIR / tag / f"C2Ray_3D_{which}"


def run_reference(mesh: int, sources, *, T0=1e4, isothermal=True, steps_per_slice=1,
                  outputs_per_slice=1, which="tap", omp=False, name=None, threads=1,
                  timeout=3600, keep=True, pl=False, lls=False, streams="0 1 1 0 0", params=False):
    """Run the reference; returns the run directory (results in <run>/results).
    sources: (i, j, k, S_BB) or, for the -DPL -DQUASARS build (pl=True), (i, j, k, S_BB, S_PL, S_QPL)."""
    exe = ref_binary(mesh, which, omp, pl, lls, params)
    if not exe.exists():
        raise FileNotFoundError(f"{exe} missing: run oracle/ref_build.sh {mesh} [omp] [pl] [lls]")
    name = name or f"run_N{mesh}_{'iso' if isothermal else 'heat'}_{len(sources)}src"
    run = REFDIR / name
    if run.exists():
        shutil.rmtree(run)
    (run / "results").mkdir(parents=True)
    with open(run / "test_sources.dat", "w") as f:
        f.write(f"{len(sources)}\n")
        for src in sources:
            i, j, k = src[:3]
            f.write(f"{i} {j} {k} " + " ".join(f"{x:.6e}" for x in src[3:]) + "\n")
    with open(run / "input", "w") as f:
        f.write(streams + "\n")                          # output streams 1..5 (output.F90:90)
        f.write(f"{T0:g}\n")
        f.write("y\n" if isothermal else "n\n")
        f.write("n\nn\n1\n")
        f.write(f"{steps_per_slice}\n{outputs_per_slice}\n")
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(threads)
    llvm_lib = "/opt/rocm/lib/llvm/lib"
    env["LD_LIBRARY_PATH"] = llvm_lib + ":" + env.get("LD_LIBRARY_PATH", "")
    with open(run / "stdout.txt", "w") as so:
        subprocess.run([str(exe), "input"], cwd=run, env=env, stdout=so,
                       stderr=subprocess.STDOUT, timeout=timeout, check=True)
    return run


def parse_log(run) -> list[list[int]]:
    """Non-converged counts per outer iteration, one list per evolve3D call
    (files_for_3D/evolve.F90:488 'Number of non-converged points')."""
    calls, cur = [], None
    for line in (Path(run) / "results" / "C2Ray.log").read_text(errors="replace").splitlines():
        if line.startswith("Time, dt:") or "Time, dt:" in line:
            cur = []
            calls.append(cur)
        m = re.search(r"Number of non-converged points:\s+(\d+)", line)
        if m and cur is not None:
            cur.append(int(m.group(1)))
    return calls


def iteration_times(run) -> list[float]:
    """Wall-clock stamps 'Time after iteration' from Timings.log (evolve.F90:220)."""
    out = []
    for line in (Path(run) / "results" / "Timings.log").read_text(errors="replace").splitlines():
        m = re.search(r"Time after iteration\s+(\d+)\s*:\s*([\d.]+)", line)
        if m:
            out.append(float(m.group(2)))
    return out


def run_cubep3m(mesh: int, sources, *, seed=2024, T0=1e4, isothermal=False, steps_per_slice=2, which="test",
                name=None, zred=(9.0, 8.6), lls_scale=1.0, timeout=3600):
    """Run a binary of the cubep3m-material build (oracle/ref_build.sh N cubep3m: ../cubep3m.F90 +
    mat_ini_cubep3m.F90, type_of_clumping = 5, use_LLS with type_of_LLS = 2) on generated inputs in the formats
    of files_for_3D/mat_ini_cubep3m.F90:
      <root>/coarser_densities/halos_removed/<z>n_all.dat    density (:223-351): stream, 3 x int32 mesh header,
                                                             float32 N^3 in N-body grid units, seeded log-normal
      <root>/coarser_densities/halos_included/<z>n_all.dat   clumping input (:675-760), same format
      <root>/halos/<z>cross_section.bin                      Lyman-limit cross sections (:859-920), same format
      <root>/run/redshifts.dat, test_sources.dat, input      run directory (cwd)
    sources: (i, j, k, S_BB, S_PL, S_QPL) as for the -DPL -DQUASARS test build.  Returns <root>/run."""
    exe = REFDIR / f"N{mesh}_cubep3m" / f"C2Ray_3D_{which}"
    if not exe.exists():
        raise FileNotFoundError(f"{exe} missing: run oracle/ref_build.sh {mesh} cubep3m")
    root = REFDIR / (name or f"run_cubep3m_N{mesh}")
    if root.exists():
        shutil.rmtree(root)
    run = root / "run"
    (run / "results").mkdir(parents=True)
    (run / "sources").mkdir()
    d_rem = root / "coarser_densities" / "halos_removed"
    d_inc = root / "coarser_densities" / "halos_included"
    d_lls = root / "halos"
    for d in (d_rem, d_inc, d_lls):
        d.mkdir(parents=True)
    rng = np.random.default_rng(seed)
    n_box = 8000.0                                       # cubep3m.F90:43
    mean = (n_box / mesh) ** 3                           # N-body grid cells per coarse cell = mean density in "grid" units
    hdr = np.array([mesh, mesh, mesh], dtype=np.int32).tobytes()
    for z in zred[:-1]:
        zs = f"{z:6.3f}".strip()
        dens = (mean * np.exp(rng.normal(0.0, 0.8, mesh ** 3) - 0.32)).astype(np.float32)
        (d_rem / f"{zs}n_all.dat").write_bytes(hdr + dens.tobytes())
        clump = (dens * np.exp(rng.normal(0.0, 0.3, mesh ** 3))).astype(np.float32)
        (d_inc / f"{zs}n_all.dat").write_bytes(hdr + clump.tobytes())
        lls = (lls_scale * 10.0 ** rng.uniform(-0.5, 0.5, mesh ** 3)).astype(np.float32)
        (d_lls / f"{zs}cross_section.bin").write_bytes(hdr + lls.tobytes())
    (root / "tables").symlink_to(REFDIR / "tables")      # cooling curves, read from ../tables (cooling_h.f90:83-149)
    (d_inc / "clumping_fit.dat").write_text("9.0 0.0 1.0 0.0 0.0\n")
    (run / "redshifts.dat").write_text(f"{len(zred)}\n" + "".join(f"{z:.3f}\n" for z in zred))
    # sourceprops_test.F90:111 reads dir_src // "test_sources.dat", and dir_src is cubep3m.F90's "./sources/"
    with open(run / "sources" / "test_sources.dat", "w") as f:
        f.write(f"{len(sources)}\n")
        for src in sources:
            f.write(f"{src[0]} {src[1]} {src[2]} " + " ".join(f"{x:.6e}" for x in src[3:]) + "\n")
    with open(run / "input", "w") as f:
        f.write("0 1 1 0 0\n")                           # output streams (output.F90:90)
        f.write(f"{T0:g}\n")                             # mat_ini (mat_ini_cubep3m.F90:131-165)
        f.write("y\n" if isothermal else "n\n")
        f.write("n\nn\n1\n")
        f.write("clumping_fit.dat\n")
        f.write("redshifts.dat\n")                       # nbody_ini (cubep3m.F90:203)
        f.write(f"{steps_per_slice}\n1\n")               # time_ini
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:" + env.get("LD_LIBRARY_PATH", "")
    with open(run / "stdout.txt", "w") as so:
        subprocess.run([str(exe), "input"], cwd=run, env=env, stdout=so, stderr=subprocess.STDOUT, timeout=timeout, check=True)
    return run
