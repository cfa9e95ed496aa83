"""The library's RCCL path (c2r_comm_kind == 1: csrc/c2ray_comm.inc) executed with 2, 4 and 8 ranks on the ONE device of
the GPU box.  The real RCCL refuses two ranks on one device, so the sums are carried by a test-only stand-in
(tests/fake_rccl.hip -> tests/_fake_rccl.so, loaded through C2R_RCCL_LIBRARY) that adds the ranks' buffers in RANK ORDER on
the device, honouring stream order like the real collective.  Everything above the ten ncclXxx entry points is the
product's own code: grouped all-reduces for several communicators, per-slab events on several contexts, the tail sum, one
host thread per device, communicators made by ncclCommInitAll (one process, c2r_create_multi) and by ncclCommInitRank (a
context per rank, the launch shape of torch.distributed.run and MPI), and both composed.  Every grid is compared BIT FOR BIT
with the oracle run with the same dealing of the sources (do_grid_static, master_slave.F90:85) and the same association of
the sum over the ranks (mpi_accumulate_grid_quantities, files_for_3D/evolve.F90:505-548).

What this does NOT execute is RCCL itself: its transports, its rings, its bootstrap between processes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
FAKE_SRC = ROOT / "tests" / "fake_rccl.hip"
FAKE_SO = ROOT / "tests" / "_fake_rccl.so"


def build_fake():
    if not FAKE_SO.exists() or FAKE_SO.stat().st_mtime < FAKE_SRC.stat().st_mtime:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-pthread",
                        "-o", str(FAKE_SO), str(FAKE_SRC)], check=True)
    return FAKE_SO


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    """One worker process runs every scenario (the library binds its RCCL once per process); the tests below compare."""
    out = tmp_path_factory.mktemp("rccl_standin")
    env = dict(os.environ, C2R_RCCL_LIBRARY=str(build_fake()), C2R_COMM_SHARED_DEVICE_RCCL="1", FAKE_RCCL_TIMEOUT_S="60")
    for k in ("C2R_FAULT_INJECT", "FAKE_RCCL_HANG", "C2R_COMM_TIMEOUT_S"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_standin_worker.py"), str(out)], env=env, capture_output=True,
                       text=True, timeout=600)
    OUT.mkdir(exist_ok=True)
    (OUT / "rccl_standin_worker.log").write_text(r.stdout[-20000:] + "\n--- stderr ---\n" + r.stderr[-20000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    summary = json.loads((out / "summary.json").read_text())
    (OUT / "rccl_standin_summary.json").write_text(json.dumps(summary, indent=1))
    return out, summary


def oracle_ranks(pkg, case, shares, niter):
    """`niter` outer iterations of the oracle; rank r sweeps the sources shares[r] (in that order, from zeroed grids), the rate
    grids are added in rank order -- ((r0 + r1) + r2) + ... -- and the global pass is applied to the sums."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    from oracle_engine import OracleEngine
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        T = orc.Tables({k: t[k] for k in t.files})
    mesh, mat, grid, src, cosmo, dt = case
    e = OracleEngine(mesh, T)
    e.set_step(mat, grid, cosmo)
    e.set_sources(src)
    e.upload_state(mat)
    e.begin_step()
    conv = []
    for _ in range(niter):
        acc = None
        for mine in shares:
            e.set_rates_to_zero()
            for ns in mine:
                nbox, loss = orc.do_source_accumulate(e.T, e.st, e.s, ns)
                e.loss += loss
                e.nbox += nbox
            part = [e.s.phih.copy(), e.s.phihe.copy(), e.s.phiheat.copy(), e.loss, e.nbox]
            acc = part if acc is None else [a + b for a, b in zip(acc, part)]
        e.s.phih[:], e.s.phihe[:], e.s.phiheat[:] = acc[0], acc[1], acc[2]
        e.loss, e.nbox = acc[3], acc[4]
        conv.append(e.global_pass(dt))
    out = {**e.download_rates(), **e.download_iter_state()}
    out["conv"] = conv
    return out


def static_shares(nsrc, nranks):
    """do_grid_static: rank r takes sources 1 + r, 1 + r + nranks, ..."""
    return [list(range(1 + r, nsrc + 1, nranks)) for r in range(nranks)]


def same(got, ref, what, heating=True):
    assert [int(x) for x in got["conv"]] == [int(x) for x in ref["conv"]], what
    keys = ["phih_grid", "phihe_grid", "xh_av", "xhe_av", "xh_intermed", "xhe_intermed"] + (["phiheat"] if heating else [])
    for k in keys:
        assert np.array_equal(got[k], ref[k]), (what, k)
    assert int(got["sum_nbox"]) == int(ref["sum_nbox"]), what
    assert abs(got["photon_loss"][0] / ref["photon_loss"][0] - 1) < 1e-13, what   # a source's loss is a block-ordered sum on the device


@pytest.fixture(scope="module")
def cases(pkg):
    sys.path.insert(0, str(ROOT / "tests"))
    import rccl_standin_worker as w
    return w, {"heat16": w.case_heating16(pkg), "tiles64iso": w.case_tiles64(pkg, True), "tiles64heat": w.case_tiles64(pkg, False)}


def test_the_standin_carried_the_sums(standin):
    _, summary = standin
    assert summary["library"].endswith("_fake_rccl.so"), summary["library"]
    cliques, calls, launched, largest = summary["stats_after_parity_scenarios"]
    assert largest == 8 and cliques >= 13 and launched > 100 and calls > 2 * launched, summary


@pytest.mark.parametrize("n", [2, 4, 8])
def test_one_process_n_communicators_vs_oracle(pkg, standin, cases, n):
    """c2r_create_multi([0] * n) + c2r_comm_init_local -> ncclCommInitAll: ONE host thread queues every slab's grouped sum for
    all n communicators behind each context's slab event, the slab's chemistry behind the sum, and the tail sum; three fused
    iterations and three plain ones (pass, whole-buffer sum, global pass).  Three sources: with 4 and 8 ranks most ranks
    sweep nothing and still owe their share of every sum."""
    out, _ = standin
    _, cs = cases
    ref = oracle_ranks(pkg, cs["heat16"], static_shares(3, n), 3)
    for mode in ("fused", "plain"):
        same(np.load(out / f"multi_heat16_N{n}_{mode}.npz"), ref, (n, mode))


@pytest.mark.parametrize("n", [2, 4])
def test_a_context_per_rank_vs_oracle(pkg, standin, cases, n):
    """c2r_create + c2r_comm_init(rank, n, id) -> ncclCommInitRank, every rank on a host thread of its own: each rank's thread
    queues its own sums (pass_slabs_one with own_sums) -- what a process per GPU does.  Every rank ends with the same grids."""
    out, _ = standin
    _, cs = cases
    ref = oracle_ranks(pkg, cs["heat16"], static_shares(3, n), 3)
    for mode in ("fused", "plain"):
        for r in range(n):
            same(np.load(out / f"threads_heat16_N{n}_{mode}_rank{r}.npz"), ref, (n, mode, r))


def test_two_processes_of_two_devices_vs_oracle(pkg, standin, cases):
    """c2r_comm_init on multi-device contexts: "process" p of 2 drives ranks 2p and 2p + 1; its device i takes the sources
    1 + p + 2 i, step 4 (c2r_pass_sources' dealing) -- source 1 -> rank 0, source 3 -> rank 1, source 2 -> rank 2."""
    out, _ = standin
    _, cs = cases
    ref = oracle_ranks(pkg, cs["heat16"], [[1], [3], [2], []], 3)
    for p in range(2):
        same(np.load(out / f"composed_heat16_2x2_fused_rank{p}.npz"), ref, p)


def test_whole_evolve3d_in_the_library_vs_oracle(pkg, standin, cases):
    """c2r_evolve3d on a four-communicator context: the library's own convergence loop (evolve.F90:147-217) around the slab-wise
    iteration, to convergence.  Same number of outer iterations as the oracle's loop with four ranks' association, same
    non-converged count after every one of them, and the state c2r_end_step leaves (xh = xh_intermed, ...) bit for bit."""
    out, _ = standin
    _, cs = cases
    got = np.load(out / "multi_heat16_N4_evolve3d.npz")
    niter = int(got["niter"])
    assert niter > 3
    ref = oracle_ranks(pkg, cs["heat16"], static_shares(3, 4), niter)
    same(got, ref, "evolve3d")
    crit = min(int(np.float32(2.5e-4) * 16 ** 3), 3)                  # evolve.F90:147
    conv = [int(x) for x in ref["conv"]]
    assert conv[-1] < crit and all(not (c < crit and i + 1 > 1) for i, c in enumerate(conv[:-1])), conv   # ... and no earlier exit
    assert np.array_equal(got["xh"], ref["xh_intermed"]) and np.array_equal(got["xhe"], ref["xhe_intermed"])


def test_tile_list_launches_vs_oracle(pkg, standin, cases):
    """64^3, five sources of which some stop after their first sub-box: rates launches from tile lists (which clear the grids
    first instead of writing them), isothermal over three communicators of one process and with heating over two ranks with
    a context each."""
    out, _ = standin
    _, cs = cases
    same(np.load(out / "multi_tiles64iso_N3_fused.npz"), oracle_ranks(pkg, cs["tiles64iso"], static_shares(5, 3), 2), "iso", heating=False)
    same(np.load(out / "threads_tiles64heat_N2_fused_rank0.npz"), oracle_ranks(pkg, cs["tiles64heat"], static_shares(5, 2), 2), "heat")


def test_ranks_fail_together(pkg, standin, cases):
    """Rank 1 returns from its second pass with an error (C2R_FAULT_INJECT: after its sweeps, before its share of the sums).
    (a) It aborts its communicator; rank 0, waiting in a collective, gets an error from it; both contexts refuse further
    collective work at once, and closing them does not wait for anything.  (a') The same inside one multi-device context.
    (b) Rank 0's collective does not notice (FAKE_RCCL_HANG: its stream just waits, like an RCCL kernel whose peer is gone):
    the library's watchdog (C2R_COMM_TIMEOUT_S = 4 s here) ends the wait with an error and aborts.  Afterwards new contexts
    and communicators work as before."""
    out, summary = standin
    _, cs = cases
    a = summary["fail_abort"]
    assert "fault injected into pass 2 of rank 1" in a["errors"][1] and "aborted" in a["errors"][1], a
    assert a["errors"][0] and "a peer aborted its communicator" in a["errors"][0] and "aborted" in a["errors"][0], a
    assert max(a["seconds"]) < 30 and a["close_seconds"] < 10, a
    for msg in a["again"]:
        assert "was aborted after an earlier error" in msg, a
    m = summary["fail_multi"]
    assert "fault injected into pass 2 of rank 1" in m["error"] and m["seconds"] < 30, m
    w = summary["fail_watchdog"]
    assert "fault injected" in w["errors"][1], w
    assert "waited" in w["errors"][0] and "C2R_COMM_TIMEOUT_S" in w["errors"][0], w
    assert 3.5 < w["seconds"][0] < 40 and w["close_seconds"] < 10, w
    same(np.load(out / "multi_heat16_N2_fused_after_failures.npz"), oracle_ranks(pkg, cs["heat16"], static_shares(3, 2), 3), "after")
