"""TEST INFRASTRUCTURE ONLY: the product's RCCL path (c2r_comm_kind == 1) with 2..8 ranks on ONE device, its sums carried by
the stand-in of tests/fake_rccl.hip.  Run by tests/test_gpu_rccl_standin.py in a process of its own -- the library binds
its RCCL once per process -- with C2R_RCCL_LIBRARY and C2R_COMM_SHARED_DEVICE_RCCL set:

    python tests/rccl_standin_worker.py OUTDIR

writes OUTDIR/<scenario>.npz (rate grids, iteration state, non-converged counts of rank 0 -- and of every rank where
ranks are contexts of their own) and OUTDIR/summary.json (what library carried the sums, its counters, the failure
scenarios' messages and times).  Nothing here is compared with anything: the parent test holds the oracle."""
import ctypes as C
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def case_heating16(pkg):
    """tests/golden/tap_N16_heat_3src.npz, second call: 16^3, three sources, heating -- every launch covers the mesh"""
    from test_host_logic import _inputs
    gold = lambda n: np.load(ROOT / "tests" / "golden" / n)
    i, _, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, "tap_N16_heat_3src.npz", 2)
    return mesh, mat, grid, src, cosmo, float(i["dt"][0])


def case_tiles64(pkg, iso):
    """the mesh of test_early_stopping_subboxes_vs_oracle: 64^3, five sources in gas of very different opacity -- some stop
    after one sub-box, the rates launches work from tile lists"""
    n, nsrc = 64, 5
    rng = np.random.default_rng(11)
    hp = pkg.hostphys
    zred = 9.0
    dr, vol = hp.test_grid(n, zred)
    nc = n ** 3
    ndens = hp.test_density(zred) * np.exp(rng.normal(0.0, 0.5, nc)) * 3.0
    xn = np.minimum(0.99, 10.0 ** (-2.6 + 2.4 * (np.arange(nc) % n) / n + rng.uniform(-0.2, 0.2, nc)))
    xh = np.concatenate([xn, 1.0 - xn])
    xhe = np.concatenate([xn, 0.8 * (1.0 - xn), 0.2 * (1.0 - xn)])
    temp = None if iso else np.tile((1e4 * np.exp(rng.normal(0, 0.2, nc))).astype(np.float32), 3)
    srcpos = rng.integers(1, n + 1, size=(nsrc, 3)).astype(np.int32)
    srcpos[:, 0] = [3, 16, 30, 45, 60]
    flux = np.array([3e3, 1e5, 3e6, 1e2, 1e8])
    mat = pkg.Material(ndens, xh, xhe, temp, iso, 1.0e4, 1.0, hp.reccoef(1.0e4))
    grid = pkg.GridProps((n, n, n), dr, vol)
    src = pkg.SourceProps(srcpos, flux, 1.0e48)
    cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
    return (n, n, n), mat, grid, src, cosmo, 1.0e5 * hp.YEAR


def engine(pkg, tables, case, devices, upload=True):
    mesh, mat, grid, src, cosmo, _ = case
    e = pkg.HipEngine(mesh, devices)
    e.set_tables(tables)
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    if upload:
        e.upload_state(mat)
    return e


def iterate(e, dt, niter, fused, first=1, stride=1, nslab=3):
    e.begin_step()
    conv = []
    for _ in range(niter):
        e.set_rates_to_zero()
        if fused:
            conv.append(e.pass_allreduce_chemistry(dt, first, stride, nslab))
        else:
            e.pass_sources(first, stride)
            e.allreduce_rates()
            conv.append(e.global_pass(dt))
    out = {**e.download_rates(), **e.download_iter_state()}
    out["conv"] = np.array(conv)
    return out


def run_multi(pkg, tables, case, n, fused, niter=3):
    """ONE process, ONE thread issuing every collective: c2r_create_multi([0] * n) + c2r_comm_init_local (ncclCommInitAll)"""
    e = engine(pkg, tables, case, [0] * n)
    e.comm_init_local()
    assert e.num_devices() == n and e.rccl_ranks() == n, (e.num_devices(), e.rccl_ranks())
    out = iterate(e, case[5], niter, fused)
    e.close()
    return out


def run_threads(pkg, tables, case, n, fused, devs_per_rank=1, niter=3, broken=False):
    """The launch shape of torch.distributed.run / MPI inside one process: a context per rank (c2r_create, or c2r_create_multi
    with `devs_per_rank` devices), c2r_comm_init (ncclCommInitRank) with one id, every rank driven by a host thread of its own.
    broken: a failure scenario (C2R_FAULT_INJECT is set): afterwards every context is asked for one more iteration."""
    uid = pkg.HipEngine.comm_unique_id()
    nproc = n // devs_per_rank
    res, err, took = [None] * nproc, [None] * nproc, [0.0] * nproc
    eng = [None] * nproc

    def body(p):
        t0 = time.time()
        try:
            e = engine(pkg, tables, case, 0 if devs_per_rank == 1 else [0] * devs_per_rank, upload=True)
            eng[p] = e
            e.comm_init(p * devs_per_rank, n, uid)
            assert e.rccl_ranks() == n
            # process p of nproc: its device i takes sources 1 + p + i * nproc, step nproc * devs_per_rank (c2r_pass_sources)
            res[p] = iterate(e, case[5], niter, fused, first=1 + p, stride=nproc)
        except Exception as ex:  # noqa: BLE001 -- reported to the parent test
            err[p] = f"{type(ex).__name__}: {ex}"
        took[p] = time.time() - t0

    th = [threading.Thread(target=body, args=(p,)) for p in range(nproc)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    again = [None] * nproc
    if broken:
        # a context whose communicator was aborted refuses further collective work, at once
        for p in range(nproc):
            t0 = time.time()
            try:
                eng[p].set_rates_to_zero()
                eng[p].pass_allreduce_chemistry(case[5], 1 + p, nproc, 3)
                again[p] = "no error"
            except Exception as ex:  # noqa: BLE001
                again[p] = f"{type(ex).__name__}: {ex} [{time.time() - t0:.2f} s]"
    t0 = time.time()
    for e in eng:
        if e is not None:
            e.close()
    return res, err, took, again, time.time() - t0


def save(outdir, name, d):
    np.savez(outdir / (name + ".npz"), **{k: np.asarray(v) for k, v in d.items()})


def main():
    outdir = Path(sys.argv[1])
    outdir.mkdir(parents=True, exist_ok=True)
    assert os.environ.get("C2R_RCCL_LIBRARY"), "run by tests/test_gpu_rccl_standin.py"
    import __graft_entry__ as ge
    pkg = ge.load_package()
    tables = pkg.RadiationTables.load()
    summary = {"library": pkg.HipEngine.comm_library()}
    fake = C.CDLL(os.environ["C2R_RCCL_LIBRARY"])
    heat16 = case_heating16(pkg)

    for n in (2, 4, 8):                      # 3 sources: with 4 and 8 ranks most ranks sweep nothing and still owe every sum
        for fused in (True, False):
            save(outdir, f"multi_heat16_N{n}_{'fused' if fused else 'plain'}", run_multi(pkg, tables, heat16, n, fused))
    for n in (2, 4):
        for fused in (True, False):
            res, err, _, _, _ = run_threads(pkg, tables, heat16, n, fused)
            assert not any(err), err
            for p in range(n):
                save(outdir, f"threads_heat16_N{n}_{'fused' if fused else 'plain'}_rank{p}", res[p])
    res, err, _, _, _ = run_threads(pkg, tables, heat16, 4, True, devs_per_rank=2)   # two "processes" of two devices each
    assert not any(err), err
    for p in range(2):
        save(outdir, f"composed_heat16_2x2_fused_rank{p}", res[p])

    # the whole convergence loop inside the library (c2r_evolve3d), four communicators of one process
    e = engine(pkg, tables, heat16, [0] * 4)
    e.comm_init_local()
    niter, flags = e.evolve3d(heat16[5])
    m = pkg.Material(ndens=heat16[1].ndens, xh=heat16[1].xh.copy(), xhe=heat16[1].xhe.copy(), temperature_grid=heat16[1].temperature_grid.copy())
    e.download_state(m)
    out = {**e.download_rates(), **e.download_iter_state(), "conv": np.array(flags), "niter": niter, "xh": m.xh, "xhe": m.xhe,
           "temperature": m.temperature_grid}
    e.close()
    save(outdir, "multi_heat16_N4_evolve3d", out)

    tiles_iso = case_tiles64(pkg, True)
    save(outdir, "multi_tiles64iso_N3_fused", run_multi(pkg, tables, tiles_iso, 3, True, niter=2))
    tiles_heat = case_tiles64(pkg, False)
    res, err, _, _, _ = run_threads(pkg, tables, tiles_heat, 2, True, niter=2)
    assert not any(err), err
    save(outdir, "threads_tiles64heat_N2_fused_rank0", res[0])

    stats = (C.c_longlong * 4)()
    fake.fake_rccl_stats(stats)
    summary["stats_after_parity_scenarios"] = list(stats)

    # ---- fail together ------------------------------------------------------------------------------------------
    # (a) rank 1 returns from its second pass with an error (after its sweeps, before its share of the sums): it aborts its
    # communicator, its peer's collective ends with an error
    os.environ["C2R_FAULT_INJECT"] = "1:2"
    _, err, took, again, t_close = run_threads(pkg, tables, heat16, 2, True, broken=True)
    summary["fail_abort"] = dict(errors=err, seconds=took, again=again, close_seconds=t_close)
    # (a') the same inside a multi-device context: one thread issues the sums for both devices
    t0 = time.time()
    try:
        run_multi(pkg, tables, heat16, 2, True)
        msg = "no error"
    except Exception as ex:  # noqa: BLE001
        msg = f"{type(ex).__name__}: {ex}"
    summary["fail_multi"] = dict(error=msg, seconds=time.time() - t0)
    # (b) the same, but the peer's collective does not notice (what a real RCCL kernel does when a peer is gone): its stream
    # waits; the library's watchdog ends the wait after C2R_COMM_TIMEOUT_S and aborts
    os.environ["FAKE_RCCL_HANG"] = "1"
    os.environ["C2R_COMM_TIMEOUT_S"] = "4"
    _, err, took, again, t_close = run_threads(pkg, tables, heat16, 2, True, broken=True)
    summary["fail_watchdog"] = dict(errors=err, seconds=took, again=again, close_seconds=t_close)
    del os.environ["FAKE_RCCL_HANG"], os.environ["C2R_COMM_TIMEOUT_S"], os.environ["C2R_FAULT_INJECT"]
    # ... and the library still works afterwards (new contexts, new communicators)
    save(outdir, "multi_heat16_N2_fused_after_failures", run_multi(pkg, tables, heat16, 2, True))
    fake.fake_rccl_stats(stats)
    summary["stats_at_end"] = list(stats)
    (outdir / "summary.json").write_text(json.dumps(summary, indent=1))
    print("rccl_standin_worker: done", flush=True)


if __name__ == "__main__":
    main()
