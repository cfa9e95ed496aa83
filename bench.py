#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the evolve3D hot path on a synthetic 256^3 box
(BASELINE.json configs[2]: 256^3 uniform density, 8 sources, 1 MI355X).

A "step" is ONE outer iteration of evolve3D (files_for_3D/evolve.F90:185-217): set_rates_to_zero,
pass_all_sources (column sweep + rates for every source of this rank), the sum of the rate grids over
ranks when N > 1, and the global chemistry pass.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; every rank sweeps --sources sources
(weak scaling: per-GPU work fixed), the rate grids are all-reduced over RCCL, chemistry is replicated
as in the reference.  value = mesh^3 x (sources x N) x K / max-over-ranks time.
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

SWEEP_BYTES_PER_CELL_SOURCE = 136.0   # SURVEY.md section 8(d), isothermal sweep (evolve0D)
COLUMN_BYTES_PER_CELL_SOURCE = 88.0   # column sweep alone: 40 B state + 48 B columns
CHEM_BYTES_PER_CELL = 252.0           # SURVEY.md section 8(d), isothermal chemistry pass
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_SUMMARY = ROOT / "profiles" / "r01_bench_pmc_summary.json"


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this same command (tools/pmc_summary.py; FETCH_SIZE doubled as the guide's gfx950
    correction prescribes), or None."""
    try:
        return json.loads(PMC_SUMMARY.read_text())[kernel]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def measured_counter(kernel, counter):
    try:
        return json.loads(PMC_SUMMARY.read_text())[kernel][counter]["per_launch"]
    except Exception:
        return None


def config3_inputs(pkg, n=256, nsrc=8, seed=12345, first_source=0, heating=False):
    """Synthetic inputs of BASELINE configs[2]: uniform density of the
    reference's test problem at z = 9, isothermal 1e4 K, sources at seeded positions
    (numpy default_rng(12345), integers in [1, n]) of 1e56 photons/s each.  The gas starts highly
    ionised (x_HI ~ 1e-3) so that every source's sub-boxes run to the full box: swept cells ==
    mesh^3 per source, the regime the metric is defined on."""
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    rng = np.random.default_rng(seed)
    allpos = rng.integers(1, n + 1, size=(first_source + nsrc, 3)).astype(np.int32)
    srcpos = allpos[first_source:]
    ndens = np.full(nc, hp.test_density(zred))
    x0 = 1.0e-3 * (1.0 + 0.5 * np.sin(np.arange(nc, dtype=np.float64) * 1.0e-3))  # neutral fraction
    xh = np.concatenate([x0, 1.0 - x0])
    xhe = np.concatenate([x0, 1.0 - x0 - 0.1, np.full(nc, 0.1)])
    temp = np.full(3 * nc, 1.0e4, dtype=np.float32) if heating else None
    mat = pkg.Material(ndens, xh, xhe, temp, not heating, 1.0e4, 1.0, hp.reccoef(1.0e4))
    grid = pkg.GridProps((n, n, n), dr, vol)
    src = pkg.SourceProps(srcpos, np.full(nsrc, 1.0e56 / 1.0e48), 1.0e48)
    cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
    return mat, grid, src, cosmo


def cpu_baseline(pkg, mesh=160, nsrc=8):
    """The oracle (C port of the reference's path, bit-identical to it) on a bounded sample of the same
    workload, on ALL host cores: one outer iteration on a mesh^3 box with nsrc sources, the sweep in
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
            "sample": f"{mesh}^3 box, {nsrc} sources, 1 outer iteration (shell-parallel sweep + chemistry) on {threads} "
                      f"OpenMP threads, {dt:.1f} s wall"}


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
            "sample": f"reference binary (flang -O2{', OpenMP' if omp else ', serial'}), {mesh}^3 box, 1 source, isothermal, {niter} outer "
                      f"iterations of evolve3D in {total:.1f} s (nominal mesh^3 x sources per iteration, as the metric)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--sources", type=int, default=8, help="sources per GPU")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--heating", action="store_true",
                    help="non-isothermal variant (heating tables, thermal evolution); not the headline config")
    a = ap.parse_args()

    import torch
    pkg = ge.load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    comm = None
    if world > 1 or os.environ.get("C2R_BENCH_FORCE_COMM"):   # the env knob rehearses the N > 1 code path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        comm = pkg.parallel.TorchComm()

    n = a.mesh
    total_src = a.sources * world
    # every rank holds the full source list; rank r sweeps r+1, r+1+world, ... (master_slave.F90:85)
    mat, grid, src, cosmo = config3_inputs(pkg, n, total_src, heating=a.heating)
    tables = pkg.RadiationTables.load()
    e = pkg.HipEngine((n, n, n), local)
    e.set_tables(tables)
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    e.set_batch(a.batch)
    e.enable_timing(True)
    if comm is not None:
        e.use_torch_rates_buffer(f"cuda:{local}")
    dt = 1.0e7 * pkg.hostphys.YEAR
    e.begin_step()

    def step():
        e.set_rates_to_zero()
        if comm is not None:
            # pass, sum over ranks and global pass overlapped slab by slab (parallel.py)
            return comm.pass_allreduce_chemistry(e, dt)
        e.pass_sources(1, 1)
        return e.global_pass(dt)

    def barrier():
        e.synchronize()
        torch.cuda.synchronize()
        if comm is not None:
            comm.dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    sweep_ms = rates_ms = chem_ms = 0.0
    swept = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        tm = e.timing()
        sweep_ms += tm.sweep_ms
        rates_ms += tm.rates_ms
        chem_ms += tm.chem_ms
        swept += tm.cells_swept
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        comm.dist.all_reduce(tt, op=comm.dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        sw = torch.tensor([float(swept)], dtype=torch.float64, device=f"cuda:{local}")
        comm.dist.all_reduce(sw)
        swept_total = int(sw.item())
    else:
        swept_total = swept

    if rank == 0:
        units = n ** 3 * total_src * a.steps
        coverage = swept_total / units
        rates_per_launch_ms = rates_ms / max(1, a.steps * ((a.sources + a.batch - 1) // a.batch))
        units_per_launch = n ** 3 * min(a.batch, a.sources)
        achieved = SWEEP_BYTES_PER_CELL_SOURCE * units_per_launch / (rates_per_launch_ms * 1e-3) / 1e9
        out = {
            "metric": "cell-updates/sec (grid_cells x sources x iters / wall) on 256^3 box; % HBM roofline",
            "value": units / elapsed, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'BASELINE configs[2]' if n == 256 and not a.heating else 'variant of BASELINE configs[2]'}: {n}^3 uniform density, {a.sources} sources per GPU "
                                   f"({total_src} total), {'heating + thermal evolution' if a.heating else 'isothermal 1e4 K'}, "
                                   f"one evolve3D outer iteration per step",
                       "mesh": n, "sources_per_gpu": a.sources, "batch": a.batch, "coverage": coverage,
                       "parallelism": f"sources over {world} GPU(s), all-reduce of rate grids, replicated chemistry"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("k_rates") if (n == 256 and a.sources == 8 and a.batch == 8) else None,
                         "kernel": "k_rates",
                         "note": "136 B per cell.source (SURVEY 8d) x cells x sources of one launch / mean launch "
                                 "time (HIP events on the library stream); the kernel is FP64-ALU bound (VALU busy "
                                 "~85 %, profiles/), not HBM bound"},
            # the bound that actually holds for k_rates (SURVEY 8d asks for HBM % and FP64 %): VALU issue rate.
            # wave-instructions per launch from the committed PMC pass (SQ_INSTS_VALU) / live launch time, against
            # 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
            "roofline_valu_issue": (lambda n_instr: None if n_instr is None or not (n == 256 and a.sources == 8 and a.batch == 8 and not a.heating) else {
                "bound": "valu-issue", "unit": "wave-instructions/s", "achieved": n_instr / (rates_per_launch_ms * 1e-3),
                "peak": 1024 * 2.4e9 / 4.0, "frac": n_instr / (rates_per_launch_ms * 1e-3) / (1024 * 2.4e9 / 4.0),
                "kernel": "k_rates", "instructions_per_launch": n_instr})(measured_counter("k_rates", "SQ_INSTS_VALU")),
            "kernel_ms_per_step": {"column_sweep": sweep_ms / a.steps, "rates": rates_ms / a.steps,
                                   "chemistry": chem_ms / a.steps},
            # the two kernels the north star names, priced the same way (HIP-event time of their launches)
            "roofline_column_sweep": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                      "achieved": COLUMN_BYTES_PER_CELL_SOURCE * n ** 3 * a.sources * a.steps
                                      / (sweep_ms * 1e-3) / 1e9,
                                      "note": "88 B per cell.source; all shell launches of a step incl. boundary-loss work"},
            "roofline_chemistry": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                   "achieved": CHEM_BYTES_PER_CELL * n ** 3 * a.steps / (chem_ms * 1e-3) / 1e9},
        }
        for k in ("roofline_column_sweep", "roofline_chemistry"):
            out[k]["frac"] = out[k]["achieved"] / HBM_PEAK_GBS
        if not a.no_cpu_baseline and world == 1:
            port = cpu_baseline(pkg)
            ref = cpu_baseline_reference()
            out["cpu_baseline"] = ref if ref is not None else port
            out["cpu_baseline_port"] = port
        print(json.dumps(out))
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
