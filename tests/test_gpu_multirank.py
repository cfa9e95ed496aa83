"""The N > 1 path with the REAL HIP engine: two ranks (both on cuda:0 of the one-GPU box, gloo as
transport because RCCL refuses two ranks on one device) share the sources, all-reduce the
device-resident rate buffer through torch.distributed and replicate the chemistry -- the code path
bench.py --gpus N runs over RCCL.  The two-rank result is compared with the single-rank one."""
import json
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, collect_from_ranks, tap_case

pytestmark = pytest.mark.gpu
OUT = ROOT / "gpurun_out"


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
    return collect_from_ranks(procs, q)


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
    # ... and against the ORACLE run with this very association of the sum over sources: rank 0 adds up sources 1
    # and 3, rank 1 has source 2, the sum over the ranks is (s1 + s3) + s2 -- not the serial (s1 + s2) + s3 of the
    # reference's one-rank run, which is why the comparison with the reference's own numbers below is to rounding.
    # Same iteration count, every grid bit for bit.
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    from oracle_engine import OracleEngine
    from test_host_logic import _inputs
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        T = orc.Tables({k: t[k] for k in t.files})

    class TwoRanksInOne:
        rank, size = 0, 2   # size 2: the step-wise loop of the Python host

        def pass_and_allreduce(self, e, nslab=8):
            parts = []
            for r in range(2):
                e.set_rates_to_zero()
                e.pass_sources(1 + r, 2)
                parts.append((e.s.phih.copy(), e.s.phihe.copy(), e.s.phiheat.copy(), e.loss, e.nbox))
            a, b = parts
            e.s.phih[:], e.s.phihe[:], e.s.phiheat[:] = a[0] + b[0], a[1] + b[1], a[2] + b[2]
            e.loss, e.nbox = a[3] + b[3], a[4] + b[4]

    i, _, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, "tap_N16_heat_3src.npz", 2)
    ev = pkg.Evolve(mesh, pkg.RadiationTables.load(), engine=OracleEngine(mesh, T), comm=TwoRanksInOne())
    n_orc = ev.evolve3D(0.0, float(i["dt"][0]), 0, mat, grid, src, cosmo)
    assert res["niter"] == n_orc
    assert np.array_equal(res["xh"], mat.xh)
    assert np.array_equal(res["temp"], mat.temperature_grid)
    assert np.array_equal(res["phih"], ev.rates["phih_grid"])
    assert res["nbox"] == ev.sum_nbox_all
    assert abs(res["loss"] / ev.photon_loss_all[0] - 1) < 1e-13   # a source's loss is summed in block order on the device
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


# ---------------------------------------------------------------------------------------------------------
# The sum over ranks BEHIND the C ABI (c2r_comm_*, c2r_allreduce_rates, c2r_pass_allreduce_chemistry): no
# torch.distributed anywhere near the data.

def _engine(pkg, gold, devices, case="tap_N16_heat_3src.npz", call=2):
    from test_host_logic import _inputs
    i, o, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, case, call)
    e = pkg.HipEngine(mesh, devices)
    e.set_tables(pkg.RadiationTables.load())
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    return e, float(i["dt"][0]), src


def _iterate(e, dt, niter, fused, first=1, stride=1):
    e.begin_step()
    conv = []
    for _ in range(niter):
        e.set_rates_to_zero()
        if fused:
            conv.append(e.pass_allreduce_chemistry(dt, first, stride, 3))
        else:
            e.pass_sources(first, stride)
            e.allreduce_rates()
            conv.append(e.global_pass(dt))
    out = {**e.download_rates(), **e.download_iter_state()}
    out["conv"] = conv
    return out


def test_rccl_single_rank_through_the_cabi(pkg, gold):
    """c2r_comm_unique_id + c2r_comm_init (ncclCommInitRank, one rank) + the slab-pipelined iteration with its
    ncclAllReduce calls: the sum over one rank changes nothing, so every grid equals the plain pass + global pass."""
    e0, dt, _ = _engine(pkg, gold, 0)
    ref = _iterate(e0, dt, 3, fused=False)          # no communicator: allreduce_rates is a no-op
    e0.close()
    e1, dt, _ = _engine(pkg, gold, 0)
    e1.comm_init(0, 1, pkg.HipEngine.comm_unique_id())
    assert e1.comm_size() == 1
    got = _iterate(e1, dt, 3, fused=True)
    e1.close()
    e2, dt, _ = _engine(pkg, gold, 0)                # (a fresh engine: the passes have moved the temperatures)
    e2.comm_init(0, 1, pkg.HipEngine.comm_unique_id())
    got2 = _iterate(e2, dt, 3, fused=False)          # whole-buffer ncclAllReduce
    e2.close()
    for k, v in ref.items():
        assert np.array_equal(np.asarray(v), np.asarray(got[k])), k
        assert np.array_equal(np.asarray(v), np.asarray(got2[k])), k


@pytest.mark.parametrize("nrep", [2, 4])
def test_replicas_of_one_process_share_the_sources(pkg, gold, nrep):
    """c2r_create_multi with the same device nrep times (the rehearsal mode of the one-process multi-GPU path:
    RCCL refuses duplicate devices, so the replicas are summed in the process, in rank order): every replica
    sweeps its share of the sources on its own host thread, the buffers are summed, the chemistry is replicated.
    The sum must equal, bit for bit, the rank-ordered sum of single-engine passes over the same shares, and the
    fused iteration must equal pass -> allreduce -> global pass."""
    em, dt, src = _engine(pkg, gold, [0] * nrep)
    em.comm_init_local()
    assert em.num_devices() == nrep and em.comm_size() == nrep
    assert em.rccl_ranks() == 0          # replicas that share a device are summed by the library itself (c2r_comm_kind == 2)
    em.enable_timing(True)
    em.begin_step()
    em.set_rates_to_zero()
    em.pass_sources(1, 1)
    # every replica's own pass is timed (c2r_get_timing_device): 3 sources over nrep replicas
    swept = [em.timing(i).cells_swept for i in range(nrep)]
    assert sum(1 for x in swept if x > 0) == min(3, nrep) and em.timing().cells_swept == swept[0]
    with pytest.raises(pkg.C2RayHipError):
        em.timing(nrep)
    em.allreduce_rates()
    got = em.download_rates()
    # the grids chosen by a mask instead of by null pointers (c2r_download_rates_sel, for Fortran hosts)
    import ctypes as C
    n = em.ncell
    a, b, h = np.full(n, -1.0), np.full(2 * n, -1.0), np.full(n, -1.0)
    loss, nbox = np.empty(47), C.c_int(0)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    assert em.lib.c2r_download_rates_sel(em.h, 3, dp(a), dp(b), dp(h), dp(loss), C.byref(nbox)) == 0
    assert np.array_equal(a, got["phih_grid"]) and np.array_equal(b, got["phihe_grid"]) and np.all(h == -1.0)
    assert np.array_equal(loss, got["photon_loss"]) and nbox.value == got["sum_nbox"]
    # the same shares on single engines, summed in rank order on the host
    acc = None
    for r in range(nrep):
        e, _, _ = _engine(pkg, gold, 0)
        e.begin_step()
        e.set_rates_to_zero()
        e.pass_sources(1 + r, nrep)
        d = e.download_rates()
        e.close()
        if acc is None:
            acc = d
        else:
            for k in ("phih_grid", "phihe_grid", "phiheat", "photon_loss"):
                acc[k] = acc[k] + d[k]
            acc["sum_nbox"] += d["sum_nbox"]
    for k in ("phih_grid", "phihe_grid", "phiheat", "photon_loss"):
        assert np.array_equal(got[k], acc[k]), k
    assert got["sum_nbox"] == acc["sum_nbox"]
    em.close()
    ea, dt, _ = _engine(pkg, gold, [0] * nrep)
    ea.comm_init_local()
    plain = _iterate(ea, dt, 3, fused=False)
    ea.close()
    eb, dt, _ = _engine(pkg, gold, [0] * nrep)
    eb.comm_init_local()
    fused = _iterate(eb, dt, 3, fused=True)
    eb.close()
    for k, v in plain.items():
        assert np.array_equal(np.asarray(v), np.asarray(fused[k])), (k, "non-converged counts, plain / slab-wise:", plain["conv"], fused["conv"])


# ---------------------------------------------------------------------------------------------------------
# More than one REAL device: RCCL (ncclCommInitAll / ncclCommInitRank + ncclAllReduce) inside the library.
# Skipped on the one-GPU box; they run wherever `pytest -m gpu` meets two devices.

def _device_count(pkg):
    return int(pkg._lib.load().c2r_device_count())


def _oracle_two_ranks(pkg, gold, niter, case="tap_N16_heat_3src.npz", call=2):
    """`niter` outer iterations of the oracle with the sources dealt over two ranks as do_grid_static does and the
    rate grids summed as a two-rank all-reduce does: (s1 + s3) + s2."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    from oracle_engine import OracleEngine
    from test_host_logic import _inputs
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        T = orc.Tables({k: t[k] for k in t.files})
    i, _, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, case, call)
    e = OracleEngine(mesh, T)
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    e.begin_step()
    conv = []
    for _ in range(niter):
        parts = []
        for r in range(2):
            e.set_rates_to_zero()
            e.pass_sources(1 + r, 2)
            parts.append((e.s.phih.copy(), e.s.phihe.copy(), e.s.phiheat.copy(), e.loss, e.nbox))
        a, b = parts
        e.s.phih[:], e.s.phihe[:], e.s.phiheat[:] = a[0] + b[0], a[1] + b[1], a[2] + b[2]
        e.loss, e.nbox = a[3] + b[3], a[4] + b[4]
        conv.append(e.global_pass(float(i["dt"][0])))
    out = {**e.download_rates(), **e.download_iter_state()}
    out["conv"] = conv
    return out


def _same_as_oracle(got, ref):
    assert got["conv"] == ref["conv"]
    for k in ("phih_grid", "phihe_grid", "phiheat", "xh_av", "xhe_av", "xh_intermed", "xhe_intermed"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["sum_nbox"] == ref["sum_nbox"]
    assert abs(got["photon_loss"][0] / ref["photon_loss"][0] - 1) < 1e-13   # block-ordered sum on the device


def test_two_devices_rccl_vs_oracle(pkg, gold):
    """ONE process, two GPUs: c2r_create_multi([0, 1]) + c2r_comm_init_local (ncclCommInitAll); three fused outer
    iterations (pass, slab-wise ncclAllReduce, slab-wise global pass) against the oracle run with the same
    association of the sum, bit for bit; then pass -> whole-buffer all-reduce -> global pass, the same bits again
    (two ranks: a + b == b + a whatever the ring does)."""
    if _device_count(pkg) < 2:
        pytest.skip("needs two HIP devices")
    ref = _oracle_two_ranks(pkg, gold, 3)
    for fused in (True, False):
        e, dt, _ = _engine(pkg, gold, [0, 1])
        e.comm_init_local()
        assert e.num_devices() == 2 and e.rccl_ranks() == 2
        got = _iterate(e, dt, 3, fused=fused)
        e.close()
        _same_as_oracle(got, ref)


def _rccl_rank(rank, world, port, q, fused):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    gold = lambda n: np.load(ROOT / "tests" / "golden" / n)
    # one process per GPU: device = rank (C2R_TEST_ONE_DEVICE: every rank on device 0, for the stand-in librccl)
    e, dt, _ = _engine(pkg, gold, 0 if os.environ.get("C2R_TEST_ONE_DEVICE") else rank)
    comm = pkg.parallel.RcclComm(e, dist)                # c2r_comm_unique_id on rank 0, gloo carries it, c2r_comm_init
    assert e.rccl_ranks() == world
    got = _iterate(e, dt, 3, fused=fused, first=1 + rank, stride=world)
    if rank == 0:
        q.put(got)
    dist.barrier()
    e.close()
    dist.destroy_process_group()


def test_two_processes_rccl_vs_oracle(pkg, gold):
    """One process per GPU, the launch shape of torch.distributed.run / MPI: c2r_create(rank) + c2r_comm_init
    (ncclCommInitRank) with the id carried by gloo; same comparison as above."""
    if _device_count(pkg) < 2:
        pytest.skip("needs two HIP devices")
    ref = _oracle_two_ranks(pkg, gold, 3)
    ctx = mp.get_context("spawn")
    for n, fused in enumerate((True, False)):
        q = ctx.Queue()
        port = 33100 + (os.getpid() % 2000) + n
        procs = [ctx.Process(target=_rccl_rank, args=(r, 2, port, q, fused)) for r in range(2)]
        for p in procs:
            p.start()
        got = collect_from_ranks(procs, q)
        _same_as_oracle(got, ref)


@pytest.mark.parametrize("world", [2, 3])
def test_processes_with_standin_rccl_vs_oracle(pkg, gold, monkeypatch, world):
    """THE LAUNCH SHAPE OF THE DRIVER -- one process per rank, c2r_create + c2r_comm_init (ncclCommInitRank) through
    parallel.RcclComm with the id carried by gloo -- executed on the one device: the sums go through the multi-process mode of the
    stand-in for librccl (tests/fake_rccl.hip, FAKE_RCCL_MULTIPROCESS=1: a clique in POSIX shared memory, the ranks' buffers
    mapped into the summing process with hipIpcOpenMemHandle, rank-ordered sum).  Three fused iterations and three plain ones
    against the oracle with the same dealing and association, bit for bit -- what test_two_processes_rccl_vs_oracle will do with
    the real RCCL once it meets two devices."""
    from test_gpu_rccl_standin import build_fake, oracle_ranks, same, static_shares
    sys.path.insert(0, str(ROOT / "tests"))
    import rccl_standin_worker as w
    monkeypatch.setenv("C2R_RCCL_LIBRARY", str(build_fake()))
    monkeypatch.setenv("FAKE_RCCL_MULTIPROCESS", "1")
    monkeypatch.setenv("C2R_TEST_ONE_DEVICE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ref = oracle_ranks(pkg, w.case_heating16(pkg), static_shares(3, world), 3)
    ctx = mp.get_context("spawn")
    for n, fused in enumerate((True, False)):
        q = ctx.Queue()
        port = 35100 + (os.getpid() % 2000) + 10 * world + n
        procs = [ctx.Process(target=_rccl_rank, args=(r, world, port, q, fused)) for r in range(world)]
        for p in procs:
            p.start()
        got = collect_from_ranks(procs, q)
        same(got, ref, (world, fused))


def test_bench_under_torchrun_with_standin_ranks(pkg):
    """bench.py exactly as the driver launches it for N > 1 -- python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    -- on the one device (C2R_BENCH_SHARE_DEVICE=1), the library's communicator made by parallel.RcclComm over gloo, its sums
    carried by the multi-process stand-in: the rank glue of bench.py (id broadcast, barrier, max over ranks, gathered per-rank
    timings, ONE result line from rank 0) runs to the end, and the line says what carried the sum."""
    import subprocess
    from test_gpu_rccl_standin import build_fake
    env = dict(os.environ, C2R_RCCL_LIBRARY=str(build_fake()), FAKE_RCCL_MULTIPROCESS="1", C2R_BENCH_SHARE_DEVICE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 36100 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--mesh", "64", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    (OUT / "bench_torchrun_standin.log").write_text(r.stdout[-5000:] + "\n--- stderr ---\n" + r.stderr[-10000:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["sources_per_gpu"] == 8
    assert d["rccl_ranks"] == 0 and d["rccl_library"].endswith("_fake_rccl.so") and "STAND-IN" in d["config"]["parallelism"]
    assert "c2r_comm_init = ncclCommInitRank" in d["config"]["parallelism"] and "FALL-BACK" not in d["config"]["parallelism"]
    assert len(d["per_rank_ms_per_step"]) == 2 and d["config"]["coverage"] > 0.99
    assert d["value"] > 0 and abs(d["value"] - 64 ** 3 * 16 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_all_devices_of_the_node_rccl(pkg, gold):
    """Every device of the box in one process (3 sources over up to 8 devices: most sweep nothing and still owe their
    share of every sum).  With more than two ranks the ring's association is RCCL's own: agreement with the serial
    oracle to rounding per pass, and the slab-wise sums against the whole-buffer sum likewise."""
    nd = min(8, _device_count(pkg))
    if nd < 3:
        pytest.skip("needs three or more HIP devices")
    _, o = tap_case(gold("tap_N16_heat_3src.npz"), 2)
    # one outer iteration: its rate grids depend on the initial state only, so the two ways of summing differ by the
    # association of at most three non-zero terms
    res = []
    for fused in (True, False):
        e, dt, _ = _engine(pkg, gold, list(range(nd)))
        e.comm_init_local()
        assert e.rccl_ranks() == nd
        res.append(_iterate(e, dt, 1, fused=fused))
        e.close()
    a, b = res
    assert a["sum_nbox"] == b["sum_nbox"] == int(o["sum_nbox_all"][0])
    for k in ("phih_grid", "phihe_grid", "phiheat"):
        scale = np.maximum(np.abs(b[k]), 1e-300)
        assert np.max(np.abs(a[k] - b[k]) / scale) < 1e-12, k
    # three iterations: the chemistry amplifies last-bit differences of the sums (DESIGN.md "Conditioning"), so the two
    # orders agree like two builds of the reference agree, not to rounding
    res = []
    for fused in (True, False):
        e, dt, _ = _engine(pkg, gold, list(range(nd)))
        e.comm_init_local()
        res.append(_iterate(e, dt, 3, fused=fused))
        e.close()
    a, b = res
    n = 16 ** 3
    for r in (a, b):
        assert np.all(np.isfinite(r["xh_intermed"])) and np.all(np.isfinite(r["phih_grid"])) and np.all(r["phih_grid"] >= 0)
    assert abs(a["xh_intermed"][n:].mean() / b["xh_intermed"][n:].mean() - 1) < 1e-4
    assert np.max(np.abs(a["xh_intermed"] - b["xh_intermed"])) < 0.05


def _same_device_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import time
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    gold = lambda n: np.load(ROOT / "tests" / "golden" / n)
    e, dt, _ = _engine(pkg, gold, 0)                     # BOTH ranks on device 0: RCCL refuses that
    t0 = time.time()
    try:
        pkg.parallel.RcclComm(e, dist)
        verdict = "ok"
    except RuntimeError as ex:
        verdict = "raised: " + str(ex)[:200]
    q.put((rank, verdict, time.time() - t0, e.rccl_ranks()))
    dist.barrier()
    e.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_fail_together(pkg):
    """Real RCCL, the failure it is known to give on a one-GPU box: two ranks whose contexts sit on the same device.
    ncclCommInitRank refuses duplicate devices; what matters is HOW the refusal arrives -- on both ranks, as the same
    RuntimeError of parallel.RcclComm, within the time-out, and with no communicator left behind (the state bench.py's
    fall-back starts from).  A hang here would be a hang of the first multi-GPU launch."""
    from conftest import collect_from_ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34100 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_same_device_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (v, t, n) for r, v, t, n in collect_from_ranks(procs, q, nresults=2, timeout=240)}
    assert got[0][0].startswith("raised") and got[1][0].startswith("raised"), got
    assert got[0][2] == 0 and got[1][2] == 0, got
    (OUT / "rccl_two_ranks_one_device.json").write_text(json.dumps({str(r): {"verdict": v, "seconds": t} for r, (v, t, _) in got.items()}, indent=1))
