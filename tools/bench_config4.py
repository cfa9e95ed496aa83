#!/usr/bin/env python3
"""Production-like workload (BASELINE configs[3]): 256^3 log-normal density
(sigma_ln = 1, seed 2024), 1024 seeded sources with log-uniform luminosities 1e52..1e54 photons/s, gas
neutral at the start, one whole evolve3D call (all outer iterations to convergence).

On one GPU this plays rank `--rank` of `--ranks`: it sweeps sources rank+1, rank+1+ranks, ... only (the other
ranks' rate contributions are missing, so the ionisation history is that of a 1/ranks-luminosity run --
fine for timing the per-GPU work of a sharded run).  With --ranks 1 it is the complete problem.

--config5 switches to BASELINE configs[4]'s shape: 512^3, 10^4 sources, heating on, and the three SEDs of the
-DPL -DQUASARS build (every third source gets a power-law component, every fifth a quasar-like one), with
the photo-ionisation and heating tables integrated on the device (c2r_build_tables) from
tests/golden/sed_setup.npz.

Not the headline bench (bench.py); prints one JSON line with per-phase times.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def correlated_unit_field(white, n, index):
    """The white Gaussian field `white` (n^3, unit variance) filtered to a power spectrum P(k) ~ k^-index and
    scaled back to unit variance: densities that vary smoothly from cell to cell, like an N-body snapshot."""
    f = np.fft.rfftn(white.reshape(n, n, n))
    k1 = np.fft.fftfreq(n) * n
    kz = np.fft.rfftfreq(n) * n
    k2 = k1[:, None, None] ** 2 + k1[None, :, None] ** 2 + kz[None, None, :] ** 2
    k2[0, 0, 0] = 1.0
    f *= k2 ** (-0.25 * index)
    f[0, 0, 0] = 0.0
    out = np.fft.irfftn(f, s=(n, n, n)).reshape(-1)
    return out / out.std()


def config4_inputs(pkg, n=256, nsrc=1024, seed=2024, heating=False, corr_index=0.0):
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    rng = np.random.default_rng(seed)
    ln = rng.normal(0.0, 1.0, nc)
    if corr_index > 0.0:
        ln = correlated_unit_field(ln, n, corr_index)
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


def lane_census(pkg):
    """Counters of the -DC2R_RATES_COUNT build (c2ray_device.hpp count_lanes), read and reset."""
    import ctypes
    lib = pkg._lib.load()
    out = (ctypes.c_ulonglong * 24)()
    if lib.c2r_debug_rates_counters(out, 1) != 0:
        raise RuntimeError("c2r_debug_rates_counters failed")
    bands, lanes, skipped, srcs, inbox, srcskip = (int(out[k]) for k in range(6))
    return {"band_bodies_per_wave": bands, "band_lane_fill": lanes / (64.0 * bands) if bands else None, "bands_skipped_by_wave": skipped,
            "div_doubt_waves": int(out[6]), "div_doubt_lanes": int(out[7]), "div_no_doubt_waves": int(out[8]),
            "div_redo_waves": int(out[9]), "div_redo_lanes": int(out[10]), "log_near1_waves": int(out[12]), "log_near1_lanes": int(out[13]),
            "log_table_only_waves": int(out[14]),
            "log_near1_both_args_waves": int(out[15]), "log_near1_both_args_lanes": int(out[16]), "log_near1_two_blocks_waves": int(out[18]),
            "thin_band_waves": int(out[21]), "thin_band_lanes": int(out[22]),
            "source_bodies_per_wave": srcs, "source_lane_fill": inbox / (64.0 * srcs) if srcs else None, "sources_skipped_by_wave": srcskip}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--sources", type=int, default=1024)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--max-iter", type=int, default=40)
    ap.add_argument("--dt-years", type=float, default=1.0e7)
    ap.add_argument("--heating", action="store_true")
    ap.add_argument("--config5", action="store_true")
    ap.add_argument("--pl", action="store_true", help="the SEDs, heating and device-built tables of --config5 at --mesh / --sources")
    ap.add_argument("--all-seds", action="store_true", help="with --pl / --config5: EVERY source has a power-law and a quasar-like component")
    ap.add_argument("--correlated", type=float, default=0.0, metavar="INDEX",
                    help="log-density with power spectrum k^-INDEX (e.g. 2.5) instead of white noise: neighbouring cells alike")
    ap.add_argument("--headline", action="store_true", help="bench.py's workload instead (256^3, 8 bright sources, pre-ionised gas): for --lane-census")
    ap.add_argument("--all-ranks", action="store_true",
                    help="play ALL --ranks ranks one after another on this GPU in every iteration (their rates add up in the one set of "
                         "rate grids, then one global pass): the full problem's ionisation history, and every rank's share of "
                         "each pass timed separately -- what decides the scaling of the sharded run")
    ap.add_argument("--calls", type=int, default=1, help="with --headline --neutral: evolve3D calls (time steps) in a row, each to convergence")
    ap.add_argument("--neutral", action="store_true", help="with --headline: the reference's neutral start instead of the pre-ionised bench state")
    ap.add_argument("--lane-census", action="store_true",
                    help="library built with -DC2R_RATES_COUNT: per iteration, how well k_rates' band loop fills its lanes")
    a = ap.parse_args()
    pkg = ge.load_package()
    if a.config5:
        a.mesh, a.sources = 512, 10000
    if a.config5 or a.pl:
        a.config5 = a.heating = True
    n = a.mesh
    if a.headline:
        import bench
        a.sources, a.ranks, a.rank = 8, 1, 0
        mat, grid, src, cosmo = bench.config3_inputs(pkg, n, 8, heating=a.heating, neutral=a.neutral)
    else:
        mat, grid, src, cosmo = config4_inputs(pkg, n, a.sources, heating=a.heating, corr_index=a.correlated)
    tables = pkg.RadiationTables.load()
    if a.config5:
        gold = ROOT / "tests" / "golden"
        tables.add_sed_file(gold / "rad_tables_pl_qpl.npz")   # band ranges (the tables themselves are rebuilt)
        with np.load(gold / "sed_setup.npz") as z:
            tables.setup = {k: z[k] for k in z.files}
        tables.build_on_device = True
        idx = np.arange(a.sources)
        src.NormFluxPL = np.where((idx % 3 == 0) | a.all_seds, 0.3 * src.NormFlux, 0.0)
        src.NormFluxQPL = np.where((idx % 5 == 0) | a.all_seds, 0.5 * src.NormFlux, 0.0)
    e = pkg.HipEngine((n, n, n), 0)
    t_tab = time.perf_counter()
    e.set_tables(tables)
    t_tab = time.perf_counter() - t_tab
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    e.set_batch(a.batch)
    e.enable_timing(True)
    dt = a.dt_years * pkg.hostphys.YEAR
    mine = len(range(a.rank, a.sources, a.ranks))
    t_begin = time.perf_counter()
    e.begin_step()            # with C2R_ARENA_RESERVE_GB: the column scratch is allocated here, in front of the iterations
    e.synchronize()
    t_begin = time.perf_counter() - t_begin
    conv_criterion = min(int(2.5e-4 * n ** 3), a.sources)
    hist = []
    t_all = time.perf_counter()
    niter, conv = 0, n ** 3
    calls_left = a.calls
    call_no = 1
    while True:
        if (conv < conv_criterion and niter > 1) or niter >= a.max_iter:
            calls_left -= 1
            if calls_left <= 0:
                break
            # the next time step: end this one, hand the state over as evolve3D's caller would, begin anew
            e.end_step()
            e.download_state(mat)
            e.set_step(mat, grid, cosmo)
            e.set_sources(src)
            e.upload_state(mat)
            e.begin_step()
            niter, conv = 0, n ** 3
            call_no += 1
        niter += 1
        t0 = time.perf_counter()
        e.set_rates_to_zero()
        shares = []
        if a.all_ranks:
            for r in range(a.ranks):
                ta = time.perf_counter()
                e.pass_sources(1 + r, a.ranks)
                tmr = e.timing()
                shares.append({"rank": r, "pass_ms": 1e3 * (time.perf_counter() - ta), "sweep_kernel_ms": tmr.sweep_ms,
                               "rates_kernel_ms": tmr.rates_ms, "cells_swept": int(tmr.cells_swept)})
        else:
            e.pass_sources(1 + a.rank, a.ranks)
        t1 = time.perf_counter()
        conv = e.global_pass(dt)
        t2 = time.perf_counter()
        tm = e.timing()
        census = lane_census(pkg) if a.lane_census else None
        hist.append({"call": call_no, "iter": niter, "pass_ms": 1e3 * (t1 - t0), "chem_ms": 1e3 * (t2 - t1), "sweep_kernel_ms": tm.sweep_ms,
                     "rates_kernel_ms": tm.rates_ms, "cells_swept": int(tm.cells_swept), "sweep_launches": tm.sweep_launches,
                     "rates_launches": tm.rates_launches, "nonconv": int(conv), "sum_nbox": int(e.get_loss()[1])})
        # a line per iteration on stderr: a 512^3 pass takes most of a minute, and a run that prints nothing for minutes is taken
        # for hung by the GPU pool's watchdog
        sys.stderr.write(f"bench_config4: iteration {niter}: pass {hist[-1]['pass_ms'] / 1e3:.1f} s, chemistry {hist[-1]['chem_ms'] / 1e3:.2f} s, "
                         f"{hist[-1]['nonconv']} cells not converged\n")
        sys.stderr.flush()
        if census:
            hist[-1]["lane_census"] = census
        if shares:
            p = [x["pass_ms"] for x in shares]
            hist[-1]["rank_shares"] = shares
            hist[-1]["share_max_ms"], hist[-1]["share_mean_ms"] = max(p), sum(p) / len(p)
            hist[-1]["share_max_over_mean"] = max(p) / (sum(p) / len(p))
    wall = time.perf_counter() - t_all
    swept = sum(h["cells_swept"] for h in hist)
    out = {"tables_s": t_tab,
           "workload": f"configs[{4 if a.config5 else 3}]-like{' (heating, BB + PL + QPL SEDs, device-built tables)' if a.config5 else ''}: {n}^3 log-normal density, {a.sources} sources, rank {a.rank} of {a.ranks} "
                       f"({mine} sources on this GPU), neutral start, dt = {a.dt_years:g} yr, batch {a.batch}",
           "niter": niter, "wall_s": wall, "begin_step_s": t_begin, "arena_reserve_gb": __import__("os").environ.get("C2R_ARENA_RESERVE_GB"),
           "arena": e.arena_stats(), "swept_cell_updates_per_s": swept / wall,
           "nominal_cell_updates_per_s": n ** 3 * mine * niter / wall,
           "mean_subboxes_per_source": float(np.mean([h["sum_nbox"] for h in hist])) / max(1, mine),
           "iterations": hist}
    if a.all_ranks:
        # what an N-GPU step would take: the slowest share, the sum over the ranks (403 MB of rate grids at 256^3,
        # isothermal; ring all-reduce over xGMI, 2 (N-1)/N x bytes at the per-link rate of MI355X_MICROARCH.md --
        # the slab pipeline hides part of it behind the pass and the chemistry) and the replicated chemistry.
        # A PREDICTION from one GPU's timings, not a measurement.
        bytes_sum = (3 if not a.heating else 4) * n ** 3 * 8.0
        link_gbs = 153.0 * 0.8
        allreduce_ms = 1e3 * 2.0 * (a.ranks - 1) / a.ranks * bytes_sum / (link_gbs * 1e9 * min(7, a.ranks - 1)) if a.ranks > 1 else 0.0
        pred = []
        for h in hist:
            one_gpu = sum(x["pass_ms"] for x in h["rank_shares"]) + h["chem_ms"]
            step = h["share_max_ms"] + allreduce_ms + h["chem_ms"]
            pred.append({"iter": h["iter"], "one_gpu_ms": one_gpu, "predicted_step_ms": step, "predicted_speedup": one_gpu / step})
        out["prediction"] = {"label": "PREDICTED from one GPU's per-share timings, not measured on N GPUs",
                             "allreduce_ms_exposed_upper": allreduce_ms, "per_iteration": pred,
                             "share_max_over_mean": [h["share_max_over_mean"] for h in hist]}
    print(json.dumps(out))
    e.close()


if __name__ == "__main__":
    main()
