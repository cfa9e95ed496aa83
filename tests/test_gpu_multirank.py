"""The N > 1 path with the REAL HIP engine: two ranks (both on cuda:0 of the one-GPU box, gloo as
transport because RCCL refuses two ranks on one device) share the sources, all-reduce the
device-resident rate buffer through torch.distributed and replicate the chemistry -- the code path
bench.py --gpus N runs over RCCL.  The two-rank result is compared with the single-rank one."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, tap_case

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, pipelined=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import __graft_entry__ as ge
    from test_host_logic import _inputs
    pkg = ge.load_package()
    gold = lambda n: np.load(ROOT / "tests" / "golden" / n)
    i, o, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, "tap_N16_heat_3src.npz", 2)
    torch.cuda.set_device(0)
    comm = pkg.parallel.TorchComm()
    if not pipelined:   # whole-buffer all-reduce after the pass, the path CPU engines take
        def plain(e, nslab=8):
            e.pass_sources(1 + comm.rank, comm.size)
            comm.allreduce_rates(e)
        comm.pass_and_allreduce = plain
        comm.pass_allreduce_chemistry = None
    ev = pkg.Evolve(mesh, pkg.RadiationTables.load(), device=0, comm=comm)
    n = ev.evolve3D(0.0, float(i["dt"][0]), 0, mat, grid, src, cosmo)
    if rank == 0:
        q.put(dict(niter=n, xh=mat.xh, temp=mat.temperature_grid, nbox=ev.sum_nbox_all, loss=ev.photon_loss_all[0],
                   phih=ev.rates["phih_grid"]))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(pipelined, port, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, pipelined)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_ranks_on_one_gpu(pkg, gold):
    _, o = tap_case(gold("tap_N16_heat_3src.npz"), 2)
    port = 29600 + (os.getpid() % 2000)
    res = _run_ranks(True, port)
    # the slab-pipelined iteration (rates of slab s+1 computed while slab s is reduced, chemistry of slab s as
    # soon as its sum is complete) changes no bit against pass -> whole-buffer all-reduce -> global pass:
    # two ranks, a + b == b + a
    plain = _run_ranks(False, port + 1)
    assert res["niter"] == plain["niter"] and res["nbox"] == plain["nbox"] and res["loss"] == plain["loss"]
    for k in ("xh", "temp", "phih"):
        assert np.array_equal(res[k], plain[k]), k
    n = 16 ** 3
    assert abs(res["niter"] - len(o["conv_flags"])) <= 2
    assert res["nbox"] == int(o["sum_nbox_all"][0])
    assert abs(res["loss"] / o["photon_loss_all"][0] - 1) < 1e-9
    assert abs(res["xh"][n:].mean() / o["xh"][n:].mean() - 1) < 1e-3
    assert np.max(np.abs(res["xh"] - o["xh"])) < 0.05
    assert np.all(np.isfinite(res["phih"])) and np.all(res["phih"] >= 0)


def test_four_ranks_one_of_them_without_sources(pkg, gold):
    """Three sources over four ranks on the one GPU: rank 3 sweeps nothing, but still owes its slab events, its
    (zero) share of every all-reduce and the replicated global pass.  With more than two ranks the order in
    which a ring all-reduce adds the contributions depends on how the buffer is chunked, so the slab-wise sums
    and the whole-buffer sum (and the serial reference) agree to rounding per pass, not bit for bit."""
    _, o = tap_case(gold("tap_N16_heat_3src.npz"), 2)
    port = 31700 + (os.getpid() % 2000)
    res = _run_ranks(True, port, 4)
    plain = _run_ranks(False, port + 1, 4)
    n = 16 ** 3
    for r in (res, plain):
        assert abs(r["niter"] - len(o["conv_flags"])) <= 2
        assert r["nbox"] == int(o["sum_nbox_all"][0])
        assert abs(r["loss"] / o["photon_loss_all"][0] - 1) < 1e-9
        assert abs(r["xh"][n:].mean() / o["xh"][n:].mean() - 1) < 1e-3
        assert np.all(np.isfinite(r["phih"])) and np.all(r["phih"] >= 0)
    assert abs(res["loss"] / plain["loss"] - 1) < 1e-12
    assert abs(res["xh"][n:].mean() / plain["xh"][n:].mean() - 1) < 1e-4
