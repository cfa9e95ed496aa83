#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the evolve3D hot path on a synthetic 256^3 box
(BASELINE.json configs[2]: 256^3 uniform density, 8 sources, 1 MI355X).

A "step" is ONE outer iteration of evolve3D (files_for_3D/evolve.F90:185-217): set_rates_to_zero,
pass_all_sources (column sweep + rates for every source of this rank), the sum of the rate grids over
ranks when N > 1, and the global chemistry pass.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--workload config3|config4]

N > 1, three ways to get there -- the sum over ranks is an RCCL all-reduce INSIDE the library in all of them
(c2r_comm_init[_local] / c2r_pass_allreduce_chemistry, include/c2ray_hip.h):
  * launched by torch.distributed.run (WORLD_SIZE set), one rank per GPU: c2r_create + c2r_comm_init
    (ncclCommInitRank); torch.distributed (gloo) only carries the 128-byte RCCL id to the ranks and does the barrier /
    max-over-ranks of this harness;
  * `python bench.py --gpus N` with no launcher (WORLD_SIZE unset), the default: ONE process drives the N GPUs
    through c2r_create_multi + c2r_comm_init_local (ncclCommInitAll), one host thread per device inside the
    library -- how the reference's no_mpi build reaches a whole node; nothing of torch.distributed is involved;
  * `python bench.py --gpus N --launcher children`: this process, before it touches torch or HIP, starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and relays its result line.
`rccl_ranks` in the result line is what c2r_comm_nranks reports for an RCCL communicator (0 when the sum over ranks
was carried by anything else).

  --workload config3 (default; BASELINE configs[2]): every rank sweeps --sources sources of its own (weak
      scaling: per-GPU work fixed); value = mesh^3 x (sources x N) x K / max-over-ranks time.
  --workload config4 (BASELINE configs[3]): 1024 seeded sources over a log-normal 256^3 box, neutral start,
      dealt to the ranks as do_grid_static does (1+rank, NumSrc, N): strong scaling, the north star's 2/4/8 case.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

# SURVEY.md section 8(d), per cell.source of evolve0D (= column sweep + rates here), isothermal
EVOLVE0D_BYTES = 136.0                # one source at a time: 40 state + 48 columns + 48 rate read-modify-write
COLUMN_BYTES_PER_CELL_SOURCE = 88.0   # column sweep alone: 40 B state + 48 B columns
CHEM_BYTES_PER_CELL = 252.0           # SURVEY.md section 8(d), isothermal chemistry pass
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6               # MI355X_MICROARCH.md: 16 384 FP64 lanes x 2.4 GHz x 2 (vector; the matrix peak is the same)
PMC_SUMMARY = ROOT / "profiles" / "r05_bench_pmc_summary.json"
DROPIN_TIMING = ROOT / "profiles" / "r05_dropin_timing.json"
REFERENCE_AT_SIZE = ROOT / "profiles" / "r04_reference_256_gpubox.json"   # timed on a GPU box's host cores (tools/time_reference.py)


def kernel_source_sha16():
    """First 16 hex digits of the SHA-256 over the library's sources (csrc/ and the C ABI header): what ties a stored
    profile to the build it was taken from.  tools/pmc_summary.py and tools/time_dropin.py write the same figure into their
    JSONs; a stored number is only quoted in the result line when it comes from the sources that are being timed."""
    import hashlib
    h = hashlib.sha256()
    files = sorted((ROOT / "c2-ray3dm1d_helium_amd" / "csrc").glob("*")) + [ROOT / "include" / "c2ray_hip.h"]
    for f in files:
        if f.suffix in (".hip", ".hpp", ".inc", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def stored_profile(path):
    """(dict, same_build) of a stored JSON of profiles/, or (None, False)."""
    try:
        d = json.loads(path.read_text())
    except Exception:
        return None, False
    return d, d.get("source_sha16") == kernel_source_sha16()


def rates_bytes_per_launch(cells, nsrc, heating=False):
    """Compulsory HBM bytes of one k_rates launch over `nsrc` sources: the six columns of every cell.source,
    and per cell the state (40 B) and the rate grids written once (3 or 4 grids; the first launch of a pass that
    covers every cell -- this workload's only one -- starts its sums from zero instead of reading them)."""
    return cells * (nsrc * 48.0 + 40.0 + (32.0 if heating else 24.0))


def stored_counter(kernel, counter):
    """A per-launch counter of the committed rocprofv3 --pmc passes of this same command (tools/pmc_summary.py), or None
    -- also None when the profile was taken from other sources than the ones being timed.  Stored, not measured in this
    run: the keys that use it say so, and `stored_profiles` in the result line names file and source hash."""
    d, same = stored_profile(PMC_SUMMARY)
    if d is None or not same:
        return None
    try:
        v = d[kernel][counter]
        return v["per_launch"] if isinstance(v, dict) else v
    except Exception:
        return None


def config3_inputs(pkg, n=256, nsrc=8, seed=12345, first_source=0, heating=False, neutral=False):
    """Synthetic inputs of BASELINE configs[2]: uniform density of the
    reference's test problem at z = 9, isothermal 1e4 K, sources at seeded positions
    (numpy default_rng(12345), integers in [1, n]) of 1e56 photons/s each.  The gas starts highly
    ionised (x_HI ~ 1e-3) so that every source's sub-boxes run to the full box: swept cells ==
    mesh^3 per source, the regime the metric is defined on.  neutral=True: the reference's test-problem start
    (x_HI = 1 - 1e-20) instead -- small sub-boxes, chemistry in its expensive state."""
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    rng = np.random.default_rng(seed)
    allpos = rng.integers(1, n + 1, size=(first_source + nsrc, 3)).astype(np.int32)
    srcpos = allpos[first_source:]
    ndens = np.full(nc, hp.test_density(zred))
    if neutral:
        eps = 1.0e-20
        xh = np.concatenate([np.full(nc, 1.0 - eps), np.full(nc, eps)])
        xhe = np.concatenate([np.full(nc, 1.0 - 2 * eps), np.full(nc, eps), np.full(nc, eps)])
    else:
        x0 = 1.0e-3 * (1.0 + 0.5 * np.sin(np.arange(nc, dtype=np.float64) * 1.0e-3))  # neutral fraction
        xh = np.concatenate([x0, 1.0 - x0])
        xhe = np.concatenate([x0, 1.0 - x0 - 0.1, np.full(nc, 0.1)])
    temp = np.full(3 * nc, 1.0e4, dtype=np.float32) if heating else None
    mat = pkg.Material(ndens, xh, xhe, temp, not heating, 1.0e4, 1.0, hp.reccoef(1.0e4))
    grid = pkg.GridProps((n, n, n), dr, vol)
    src = pkg.SourceProps(srcpos, np.full(nsrc, 1.0e56 / 1.0e48), 1.0e48)
    cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
    return mat, grid, src, cosmo


def config4_inputs(pkg, n=256, nsrc=1024, seed=2024, heating=False):
    """BASELINE configs[3]: log-normal density (sigma_ln = 1, seed 2024, mean of the test
    problem at z = 9), 1024 seeded sources with log-uniform luminosities 1e52..1e54 photons/s, neutral start."""
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    rng = np.random.default_rng(seed)
    ln = rng.normal(0.0, 1.0, nc)
    ndens = hp.test_density(zred) * np.exp(ln - 0.5)          # mean-preserving log-normal
    srcpos = rng.integers(1, n + 1, size=(nsrc, 3)).astype(np.int32)
    flux = 10.0 ** rng.uniform(52.0, 54.0, nsrc) / 1.0e48
    eps = 1.0e-20
    xh = np.concatenate([np.full(nc, 1.0 - eps), np.full(nc, eps)])
    xhe = np.concatenate([np.full(nc, 1.0 - 2 * eps), np.full(nc, eps), np.full(nc, eps)])
    temp = np.full(3 * nc, 1.0e4, dtype=np.float32) if heating else None
    mat = pkg.Material(ndens, xh, xhe, temp, not heating, 1.0e4, 1.0, hp.reccoef(1.0e4))
    grid = pkg.GridProps((n, n, n), dr, vol)
    src = pkg.SourceProps(srcpos, flux, 1.0e48)
    cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
    return mat, grid, src, cosmo


def cpu_baseline(pkg, mesh=256, nsrc=8):
    """The oracle (C port of the reference's path, bit-identical to it on the golden fixtures) on ONE STEP OF THIS
    VERY WORKLOAD -- 256^3, 8 sources, the same seeded inputs --, on the host cores of one GPU's share: the sweep in
    L-infinity shell order with the cells of a shell over OpenMP threads (columns and rates equal the serial
    sweep's bit for bit, tests/test_oracle_golden.py), the global pass cell-parallel.  Reported, not a target."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    orc.build()
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    mat, grid, src, cosmo = config3_inputs(pkg, mesh, nsrc)
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        T = orc.Tables({k: t[k] for k in t.files})
    st = orc.Step(grid.mesh, grid.dr, grid.vol, cosmo.zred, cosmo.H0, cosmo.Omega0, 1, 1.0e4, 1.0, src.srcpos,
                  src.NormFlux, src.S_star, mat.ndens, mat.reccoef)
    s = orc.State(st, mat.xh, mat.xhe)
    orc.begin_step(s)
    t0 = time.perf_counter()
    orc.pass_all_sources_shells(T, st, s, threads)
    orc.global_pass_threads(T, st, s, 1.0e7 * pkg.hostphys.YEAR, threads)
    dt = time.perf_counter() - t0
    return {"value": mesh ** 3 * nsrc / dt, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"one step of this workload ({mesh}^3 box, {nsrc} sources, same seeded inputs): shell-parallel sweep + "
                      f"cell-parallel chemistry of the oracle on {threads} OpenMP threads, {dt:.1f} s wall"}


def cpu_baseline_reference(mesh=64):
    """The reference ITSELF (flang -O2 build of /root/reference made by oracle/ref_build.sh in the dev
    container; the binary travels in oracle/_ref/) on its own test problem: mesh^3, one 1e54 source,
    isothermal, four time slices of three steps.  Wall time of the evolve3D iterations from the reference's own
    Timings.log stamps (evolve.F90:150,220).  Returns None when the binary is not there."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import re
    import shutil
    import refrun
    omp = refrun.ref_binary(mesh, "test", omp=True).exists()
    if not omp and not refrun.ref_binary(mesh, "test").exists():
        return None
    # the reference's OpenMP path is at most 8-way (6 axes / 12 planes / 8 octants, evolve_source.F90:158-189)
    threads = min(8, os.cpu_count() or 1) if omp else 1
    run = refrun.run_reference(mesh, [(mesh // 2, mesh // 2, mesh // 2, 1e54)], isothermal=True, steps_per_slice=3,
                               which="test", omp=omp, threads=threads, name="bench_reference_run")
    text = (run / "results" / "Timings.log").read_text(errors="replace")
    total, niter, t0 = 0.0, 0, None
    for line in text.splitlines():
        m = re.search(r"Time before starting iteration:\s*([\d.]+)", line)
        if m:
            t0 = float(m.group(1))
        m = re.search(r"Time after iteration\s+(\d+)\s*:\s*([\d.]+)", line)
        if m and t0 is not None:
            total += float(m.group(2)) - t0
            t0 = float(m.group(2))
            niter += 1
    shutil.rmtree(run, ignore_errors=True)
    if niter == 0 or total <= 0:
        return None
    return {"value": mesh ** 3 * niter / total, "unit": "cell-updates/s", "cores": threads, "kind": "reference",
            "sample": f"reference binary (flang -O2{', OpenMP' if omp else ', serial'}), its own test problem: {mesh}^3 box, 1 source, "
                      f"isothermal, {niter} outer iterations of evolve3D in {total:.1f} s (nominal mesh^3 x sources per iteration, as the metric)"}


def cpu_baseline_reference_at_size(mesh=256, iterations=2):
    """The reference ITSELF on the benchmark's own inputs at the benchmark's own size, on THIS box's host cores: the timed
    build (oracle/ref_build.sh 256 omp timer -> oracle/_ref/N256_omp/C2Ray_3D_timed, made in the dev container, travels
    in oracle/_ref/) run for its first `iterations` outer iterations from the neutral start -- a bounded sample (~20 s);
    the sub-boxes are still small then, so the figure flatters the reference (profiles/r04_reference_256.json has the 16
    iterations it takes to reach the mesh limit: 41-50 s each).  None when the binary is not there."""
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import time_reference
        r = time_reference.time_reference(mesh, iterations, threads=min(8, os.cpu_count() or 1), timeout=600)
    except Exception as ex:  # noqa: BLE001 -- a baseline that cannot run is reported as absent, never fatal
        sys.stderr.write(f"bench.py: reference at size not timed: {ex}\n")
        return None
    if r is None:
        return None
    t = sum(r["s_per_iteration"])
    return {"value": mesh ** 3 * 8 * len(r["s_per_iteration"]) / t, "unit": "cell-updates/s", "cores": r["threads"], "kind": "reference",
            "s_per_iteration": r["s_per_iteration"],
            "sample": f"reference binary (flang -O2 -fopenmp), THIS workload's inputs ({mesh}^3, the same 8 sources), its first "
                      f"{len(r['s_per_iteration'])} outer iterations from the neutral start in {t:.1f} s on this box's host cores "
                      "(sub-boxes still small: an upper bound of its rate; at the mesh limit see cpu_baseline_reference_at_size)"}


def launch_children(a):
    """`python bench.py --gpus N --launcher children`: start the N ranks as a CHILD torch.distributed.run and relay the one
    result line.  Runs before this process has imported torch or touched HIP (a process that has may not exec or be
    replaced; a child is always fine)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [x for x in sys.argv[1:] if x != "children" and x != "--launcher" and not x.startswith("--launcher=")]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if r.returncode != 0 or not lines:
        raise SystemExit(r.returncode or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["config3", "config4"], default="config3")
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--sources", type=int, default=None, help="config3: sources per GPU (default 8); config4: total (default 1024)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--heating", action="store_true",
                    help="non-isothermal variant (heating tables, thermal evolution); not the headline config")
    ap.add_argument("--neutral-start", action="store_true",
                    help="config3 from the reference's neutral test-problem start (small sub-boxes, chemistry in its expensive state)")
    ap.add_argument("--one-process", action="store_true",
                    help="one process drives all --gpus devices (c2r_create_multi + c2r_comm_init_local); the default for "
                         "--gpus N > 1 when no launcher set WORLD_SIZE; with --gpus 1 a one-device RCCL communicator")
    ap.add_argument("--launcher", choices=["none", "children"], default="none",
                    help="children: start torch.distributed.run with --gpus ranks as a child process and relay its result")
    a = ap.parse_args()

    # dmabuf IPC for RCCL's peer-to-peer transport on this driver (must be set before the HIP runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1        # torch.distributed.run (or another launcher) made us a rank
    if not launched and a.launcher == "children" and not a.one_process:   # also with --gpus 1 (rehearsal)
        return launch_children(a)
    one_process = a.one_process or (a.gpus > 1 and not launched)

    # stdout carries ONE line, the result: whatever libraries print there while they start up (gloo, RCCL's version
    # banner) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    pkg = ge.load_package()
    if one_process:
        world, rank, local = a.gpus, 0, 0
        have = pkg._lib.load().c2r_device_count()
        # C2R_BENCH_SHARE_DEVICE=1: all "devices" are GPU 0 (rehearsal on a one-GPU box: RCCL refuses duplicate devices,
        # the library then sums its replicas itself and rccl_ranks says 0)
        share = bool(os.environ.get("C2R_BENCH_SHARE_DEVICE"))
        if have < a.gpus and not share:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} HIP device(s) visible")
        devices = [0] * a.gpus if share else list(range(a.gpus))
    else:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if a.gpus != world:
            raise SystemExit(f"bench.py --gpus {a.gpus} inside a launch of WORLD_SIZE={world}")
        # C2R_BENCH_SHARE_DEVICE=1 under a launcher: every rank on GPU 0 -- a rehearsal of this very launch shape on a one-GPU
        # box, possible only with a stand-in for RCCL that accepts it (C2R_RCCL_LIBRARY; the line then says STAND-IN)
        if os.environ.get("C2R_BENCH_SHARE_DEVICE"):
            local = 0
        devices = local
    torch.cuda.set_device(local)
    dist = None
    force_comm = bool(os.environ.get("C2R_BENCH_FORCE_COMM"))   # rehearses the N > 1 code path (RCCL, one rank) on one GPU
    if not one_process and (world > 1 or force_comm):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # plumbing only: the id of the library's RCCL communicator, the barrier and the max over ranks below
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n = a.mesh
    cfg4 = a.workload == "config4"
    if cfg4:
        total_src = a.sources or 1024
        per_gpu = len(range(rank, total_src, world))
        mat, grid, src, cosmo = config4_inputs(pkg, n, total_src, heating=a.heating)
        batch = a.batch or 256
    else:
        per_gpu = a.sources or 8
        total_src = per_gpu * world
        # every rank holds the full source list; rank r sweeps r+1, r+1+world, ... (master_slave.F90:85)
        mat, grid, src, cosmo = config3_inputs(pkg, n, total_src, heating=a.heating, neutral=a.neutral_start)
        batch = a.batch or 8
    if cfg4 and n >= 256:
        # the column scratch of a many-source share (a block per source in flight, up to 0.8 GB each at 256^3) is allocated when
        # the first step begins, in front of the warm-up, instead of piece by piece inside the timed steps: a cold device
        # allocation costs ~24 ms per GB and stalls every launch of the process
        os.environ.setdefault("C2R_ARENA_RESERVE_GB", "110")
    tables = pkg.RadiationTables.load()
    e = pkg.HipEngine((n, n, n), devices)
    e.set_tables(tables)
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    e.set_batch(batch)
    e.enable_timing(True)
    comm = None
    transport = None
    if one_process:
        e.comm_init_local()       # ncclCommInitAll over this process's devices; failure is fatal: there is no other transport here
        comm = pkg.parallel.LocalComm(e)
        transport = (f"one process, {a.gpus} device(s): RCCL all-reduce of the rate grids inside the library (c2r_create_multi + "
                     "c2r_comm_init_local = ncclCommInitAll, slab-wise, overlapped)") if e.rccl_ranks() else \
                    f"one process, {a.gpus} replicas SHARING ONE DEVICE (rehearsal): summed by the library itself, no RCCL"
    elif dist is not None:
        try:
            fail = os.environ.get("C2R_BENCH_FAIL_LIB_COMM")     # rehearses the fall-back below: "1" every rank, "rank0" rank 0 only
            comm = pkg.parallel.RcclComm(e, dist, fail_on_ranks=() if not fail else ((0,) if fail == "rank0" else range(world)))
            transport = "RCCL all-reduce of the rate grids inside the library (c2r_comm_init = ncclCommInitRank, slab-wise, overlapped)"
        except Exception as ex:  # noqa: BLE001 -- whatever the library reports: no RCCL to load, version, init error
            # The library's communicator has never met a second GPU.  If it cannot be set up --
            # RcclComm makes that decision collectively: every rank is here, or none -- the bench still measures the path,
            # with the same slab-wise overlapped sum carried by torch.distributed's RCCL instead, and SAYS SO in its result
            # line (rccl_ranks 0): a diagnostic, never the result.
            sys.stderr.write(f"bench.py: rank {rank}: the library's RCCL communicator failed ({ex}); "
                             "falling back to torch.distributed (backend nccl = RCCL) for the sum over ranks\n")
            e.use_torch_rates_buffer(f"cuda:{local}")   # the reduction buffer as a tensor torch.distributed can sum in place
            comm = pkg.parallel.TorchComm(dist.new_group(backend="nccl"))
            transport = f"FALL-BACK: torch.distributed nccl (= RCCL) all-reduce of the rate grids, slab-wise, overlapped; the library's own communicator failed: {ex}"
    rccl_ranks = e.rccl_ranks()
    rccl_library = None
    if rccl_ranks:
        # what carried the sums: the loader's librccl -- or whatever C2R_RCCL_LIBRARY names (the one-device stand-in of
        # tests/fake_rccl.hip rehearses the N-rank control flow on a one-GPU box); only the former is an RCCL result
        rccl_library = pkg.HipEngine.comm_library()
        if not os.path.basename(rccl_library).startswith("librccl"):
            transport = f"STAND-IN for RCCL ({rccl_library}, C2R_RCCL_LIBRARY) -- NOT an RCCL result: {transport}"
            rccl_ranks = 0
    dt = 1.0e7 * pkg.hostphys.YEAR
    e.begin_step()

    def step():
        e.set_rates_to_zero()
        if comm is not None:
            # pass, sum over ranks (RCCL inside the library) and global pass overlapped slab by slab
            return comm.pass_allreduce_chemistry(e, dt)
        e.pass_sources(1, 1)
        return e.global_pass(dt)

    def barrier():
        e.synchronize()           # every device of the context
        for d in (sorted(set(devices)) if one_process else [local]):
            torch.cuda.synchronize(d)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    sweep_ms = rates_ms = chem_ms = 0.0
    swept = 0
    rates_launches = 0
    ndev_here = a.gpus if one_process else 1
    other = [{"swept": 0, "sweep_ms": 0.0, "rates_ms": 0.0, "chem_ms": 0.0} for _ in range(ndev_here - 1)]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        tm = e.timing()
        sweep_ms += tm.sweep_ms
        rates_ms += tm.rates_ms
        chem_ms += tm.chem_ms
        swept += tm.cells_swept
        rates_launches += tm.rates_launches
        for i, o in enumerate(other):     # the other devices of a one-process run
            t = e.timing(i + 1)
            o["swept"] += t.cells_swept
            o["sweep_ms"] += t.sweep_ms
            o["rates_ms"] += t.rates_ms
            o["chem_ms"] += t.chem_ms
    barrier()
    elapsed = time.perf_counter() - t0
    mine = {"elapsed": elapsed, "swept": swept, "sweep_ms": sweep_ms, "rates_ms": rates_ms, "chem_ms": chem_ms}
    if dist is not None and world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
    else:
        allr = [mine] + [{"elapsed": elapsed, **o} for o in other]
    elapsed = max(r["elapsed"] for r in allr)
    swept_total = sum(r["swept"] for r in allr)

    if rank == 0:
        units = n ** 3 * total_src * a.steps
        coverage = swept_total / units
        heating = a.heating
        nl = max(1, rates_launches)
        rates_per_launch_ms = rates_ms / nl
        # cell.sources of one k_rates launch on this rank
        cs_per_launch = swept / nl
        src_per_launch = cs_per_launch / n ** 3
        rates_bytes = rates_bytes_per_launch(n ** 3, src_per_launch, heating) if not cfg4 else cs_per_launch * 48.0
        achieved = rates_bytes / (rates_per_launch_ms * 1e-3) / 1e9
        headline = (not cfg4) and n == 256 and per_gpu == 8 and batch == 8 and not heating and not a.neutral_start
        n_instr = stored_counter("k_rates", "SQ_INSTS_VALU") if headline else None
        n_flop = stored_counter("k_rates", "fp64_flop_per_launch_upper") if headline else None
        out = {
            "metric": "cell-updates/sec (grid_cells x sources x iters / wall) on 256^3 box; % HBM roofline",
            "value": units / elapsed, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "strong" if cfg4 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # ranks of the RCCL communicator INSIDE the library that summed the rate grids (c2r_comm_nranks; 0: one rank
            # without communicator, a fall-back transport, or replicas sharing a device)
            "rccl_ranks": rccl_ranks,
            "rccl_library": rccl_library,
            "config": {"workload": (f"BASELINE configs[3]: {n}^3 log-normal density, {total_src} sources dealt over {world} GPU(s) "
                                    f"({per_gpu} on rank 0), neutral start, {'heating' if heating else 'isothermal 1e4 K'}, one evolve3D "
                                    f"outer iteration per step (the state evolves from step to step)") if cfg4 else
                                   (f"{'BASELINE configs[2]' if headline or (n == 256 and not heating and not a.neutral_start) else 'variant of BASELINE configs[2]'}: "
                                    f"{n}^3 uniform density, {per_gpu} sources per GPU ({total_src} total), "
                                    f"{'heating + thermal evolution' if heating else 'isothermal 1e4 K'}, "
                                    f"{'neutral start, ' if a.neutral_start else ''}one evolve3D outer iteration per step"),
                       "mesh": n, "sources_per_gpu": per_gpu, "batch": batch, "coverage": coverage,
                       "swept_cell_updates_per_s": swept_total / elapsed,
                       "parallelism": f"sources over {world} GPU(s), replicated chemistry; sum over ranks: {transport or 'none (one rank)'}"},
            # the dominant kernel on ITS OWN compulsory bytes: six columns per cell.source, state and rate grids per cell
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": stored_counter("k_rates", "hbm_bytes_per_launch_corrected") if headline else None,
                         "kernel": "k_rates",
                         "note": "compulsory bytes of one k_rates launch (48 B of columns per cell.source + 64 B of state and "
                                 "rate grids per cell) / mean launch time (HIP events on the library's stream); traffic: stored "
                                 "rocprofv3 --pmc FETCH_SIZE (doubled, gfx950) + WRITE_SIZE of the same command (profiles/). The "
                                 "kernel is bound by FP64 instruction issue, see roofline_valu_issue"},
            # SURVEY 8(d)'s figure for the whole of evolve0D (the reference's sweep = column sweep + rates here)
            "roofline_evolve0d": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                  "achieved": EVOLVE0D_BYTES * swept / ((sweep_ms + rates_ms) * 1e-3) / 1e9,
                                  "note": "136 B per cell.source (SURVEY 8d) over column sweep + rates time"},
            # the bound that actually holds for k_rates: VALU issue rate (stored SQ_INSTS_VALU of the same command over
            # the live launch time, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction)
            "roofline_valu_issue": None if n_instr is None else {
                "bound": "valu-issue", "unit": "wave-instructions/s", "achieved": n_instr / (rates_per_launch_ms * 1e-3),
                "peak": 1024 * 2.4e9 / 4.0, "frac": n_instr / (rates_per_launch_ms * 1e-3) / (1024 * 2.4e9 / 4.0),
                # the same priced with what was measured on this part instead of the nominal figures: 2.31 GHz under this
                # kernel's load (tools/clock_probe.sh) and 4.3 cycles per FP64 wave-instruction (tools/micro/valu_rates.hip)
                "frac_at_measured_clock_and_cost": n_instr / (rates_per_launch_ms * 1e-3) / (1024 * 2.31e9 / 4.3),
                "kernel": "k_rates", "instructions_per_launch": n_instr, "instructions_per_cell_source": n_instr * 64.0 / max(1.0, cs_per_launch),
                "instructions": "stored counter (profiles/, same sources as this run), live time; per cell.source: wave-instructions x 64 lanes / (cells x sources)"},
            # "report both HBM % and FP64 %".  FLOPs of one k_rates launch from the stored
            # SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 counters of this same command (wave-instructions x 64 lanes, an fma
            # counted twice: an upper bound, lanes masked off in divergent branches included) over the live launch time
            "roofline_fp64": None if n_flop is None else {
                "bound": "fp64-vector", "unit": "TFLOP/s", "achieved": n_flop / (rates_per_launch_ms * 1e-3) / 1e12,
                "peak": FP64_PEAK_TFLOPS, "frac": n_flop / (rates_per_launch_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "kernel": "k_rates", "flop_per_launch": n_flop, "flop_per_cell_source": n_flop / max(1.0, cs_per_launch),
                "note": "stored counters (profiles/), live time; two thirds of the kernel's vector instructions are FP64 "
                        "arithmetic, the rest integer / conversion / compare work of the bit-exact log10 and the table "
                        "look-ups, which is why the issue-rate figure (roofline_valu_issue) is the one that binds"},
            # with a communicator the global pass runs slab by slab behind the sum over ranks, while the rates of
            # later slabs are still being computed: its figure is then the span from the first slab's launch to
            # the end of the last, most of which overlaps the rates
            "kernel_ms_per_step": {"column_sweep": sweep_ms / a.steps, "rates": rates_ms / a.steps,
                                   ("chemistry" if comm is None else "chemistry_span_overlapping_rates"): chem_ms / a.steps},
            "roofline_column_sweep": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                      "achieved": COLUMN_BYTES_PER_CELL_SOURCE * swept / (sweep_ms * 1e-3) / 1e9,
                                      "note": "88 B per cell.source; all shell launches of a step incl. boundary-loss work"},
            "roofline_chemistry": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                   "achieved": CHEM_BYTES_PER_CELL * n ** 3 * a.steps / (chem_ms * 1e-3) / 1e9,
                                   "state": ("CONVERGED STEADY STATE: every step of this run starts from the state the previous one left, so "
                                             "after the warm-up nearly every cell is done after one do_chemistry iteration; where the pass is "
                                             "expensive -- the first iterations of a time step -- is in chemistry_ms_in_a_time_step")
                                   if not a.neutral_start else "from the neutral start"},
        }
        if comm is not None:
            del out["roofline_chemistry"]   # a span that overlaps the rates kernel is not a kernel time
        for k in ("roofline_evolve0d", "roofline_column_sweep", "roofline_chemistry"):
            if k in out:
                out[k]["frac"] = out[k]["achieved"] / HBM_PEAK_GBS
        out["config"]["launch"] = ("one process, c2r_create_multi" if one_process else
                                   "one process per GPU (torch.distributed.run)" if world > 1 else "one process, one GPU")
        if world > 1:
            # per-rank step time (imbalance) and what the kernels of a step do not account for (exposed sum over ranks,
            # host time); in a one-process run every device shares the one wall clock
            out["per_rank_ms_per_step"] = [1e3 * r["elapsed"] / a.steps for r in allr]
            out["per_rank_kernel_ms_per_step"] = [(r["sweep_ms"] + r["rates_ms"]) / a.steps for r in allr]   # pass only (see above)
        if headline:
            # what is quoted from profiles/ rather than measured in this run, and whether it was taken from these very sources
            sha = kernel_source_sha16()
            out["stored_profiles"] = {"running_source_sha16": sha}
            for key, path in (("pmc", PMC_SUMMARY), ("dropin", DROPIN_TIMING), ("reference_at_size", REFERENCE_AT_SIZE)):
                d, same = stored_profile(path)
                if d is not None:
                    entry = {"file": str(path.relative_to(ROOT)), "source_sha16": d.get("source_sha16")}
                    if key == "reference_at_size":
                        entry["depends_on_sources"] = False    # a run of the REFERENCE: nothing of this library is in it
                    else:
                        entry["same_sources_as_this_run"] = same
                    out["stored_profiles"][key] = entry
            # the product the north star describes -- the reference's Fortran driver with this library's modules linked
            # in (oracle/ref_build.sh 256 hip) -- on this very workload, by the drop-in's own clock; measured by
            # tools/time_dropin.py on a GPU box and stored (the driver's bench box has no reference sources to build the
            # binary from); quoted only when taken from the sources being timed
            dj, same = stored_profile(DROPIN_TIMING)
            if dj is not None and same:
                out["dropin_ms_per_iteration"] = {k: dj[k] for k in ("ms_per_iteration", "iterations", "bench_ms_per_step_same_box", "note") if k in dj}
                calls = dj.get("evolve3D_calls", [])
                out["dropin_ms_per_iteration"]["evolve3D_calls"] = [{k: c[k] for k in ("iterations", "ms_per_iteration", "setup_s", "results_s", "call_s", "kernel_ms_mean") if k in c} for c in calls]
                # the chemistry pass where it is expensive (stored, same sources): by outer iteration of the neutral-start
                # call (first ten) and of the time step after it
                if calls and "chemistry_ms_by_iteration" in calls[0]:
                    first = calls[0]["chemistry_ms_by_iteration"][:10]
                    out["chemistry_ms_in_a_time_step"] = {
                        "neutral_start_first_10_iterations": first, "neutral_start_first_10_mean": sum(first) / len(first),
                        "second_time_step_by_iteration": calls[1]["chemistry_ms_by_iteration"] if len(calls) > 1 else None,
                        "source": str(DROPIN_TIMING.relative_to(ROOT))}
            # the reference's own OpenMP build on THIS workload's inputs at THIS size to the mesh limit (sixteen
            # outer iterations, four minutes: too long for every bench run), timed once on a GPU box's host cores
            # (tools/time_reference.py) and stored; it does not depend on this library's sources.  The first iterations of the
            # same run ARE timed live below (cpu_baseline_reference_at_size_first_iterations)
            rj, _ = stored_profile(REFERENCE_AT_SIZE)
            if rj is not None:
                out["cpu_baseline_reference_at_size"] = {
                    "stored": str(REFERENCE_AT_SIZE.relative_to(ROOT)), "kind": "reference", "unit": "cell-updates/s", "cores": rj.get("threads"),
                    "value": rj.get("cell_updates_per_s_at_mesh_limit"), "s_per_iteration": rj.get("s_per_iteration"),
                    "sample": rj.get("what"), "how": rj.get("how"),
                    "where": rj.get("where", "dev container host cores"), "s_per_iteration_at_mesh_limit": rj.get("s_per_iteration_at_mesh_limit")}
        if not a.no_cpu_baseline and world == 1 and not cfg4:
            out["cpu_baseline"] = cpu_baseline(pkg)
            ref = cpu_baseline_reference()
            if ref is not None:
                out["cpu_baseline_reference"] = ref
            if headline:
                ref = cpu_baseline_reference_at_size(n)
                if ref is not None:
                    out["cpu_baseline_reference_at_size_first_iterations"] = ref
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    e.close()


if __name__ == "__main__":
    main()
