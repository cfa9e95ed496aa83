"""Host logic above the C ABI, on CPU: the evolve3D iteration loop of the Python host, the
partition of sources over ranks and the single all-reduce of the rate buffer (gloo, world_size 2),
and the small host-side set-up helpers."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, collect_from_ranks, tap_case

sys.path.insert(0, str(ROOT / "tests"))


def test_reccoef_equals_reference_module_values(pkg, gold):
    """hostphys.reccoef(T0) == the twelve module-global coefficients mat_ini leaves behind
    (mat_ini_test.F90:168), as dumped from the reference."""
    i, _ = tap_case(gold("tap_N16_iso_1src.npz"), 1)
    assert np.array_equal(pkg.hostphys.reccoef(float(i["temper_val"][0])), i["reccoef"])
    a = gold("funcvec.npz")["reccoef_T"].reshape(-1, 13)
    for row in a:
        assert np.array_equal(pkg.hostphys.reccoef(row[0]), row[1:])


def test_test_problem_density_and_constants(pkg, gold):
    c = gold("consts.npz")["consts"]
    hp = pkg.hostphys
    assert hp.H0 == c[42] and hp.Omega0 == c[43]
    i, _ = tap_case(gold("tap_N16_iso_1src.npz"), 1)
    # density at the redshift of the first step (set at z=9, then diluted by cosmo_evol)
    assert abs(hp.test_density(float(i["zred"][0])) / i["ndens"][0] - 1) < 1e-12


def _inputs(pkg, gold, fname, call):
    i, o = tap_case(gold(fname), call)
    iso = bool(i["isothermal"][0])
    mesh = tuple(int(m) for m in i["mesh"])
    mat = pkg.Material(ndens=i["ndens"], xh=i["xh"].copy(), xhe=i["xhe"].copy(),
                       temperature_grid=None if iso else i["temperature"].copy(), isothermal=iso,
                       temper_val=float(i["temper_val"][0]), clumping=float(i["clumping"][0]), reccoef=i["reccoef"])
    grid = pkg.GridProps(mesh, tuple(i["dr"]), float(i["vol"][0]))
    src = pkg.SourceProps(i["srcpos"].reshape(-1, 3), i["NormFlux"], float(i["S_star"][0]))
    cosmo = pkg.Cosmology(float(i["zred"][0]), float(i["H0"][0]), float(i["Omega0"][0]))
    return i, o, mesh, mat, grid, src, cosmo


def test_python_host_loop_equals_reference(pkg, orc, otables, gold):
    """Evolve.evolve3D with the step-wise loop of the Python host (as used for N > 1) reproduces a
    whole reference evolve3D call when the engine is the oracle."""
    from oracle_engine import OracleEngine

    class OneRank:
        rank, size = 0, 2  # size 2 forces the step-wise path; no partner: reduce is the identity

        def pass_and_allreduce(self, e, nslab=8):
            e.pass_sources(1, 1)  # every source on this single "rank"
            e.rates_buffer()
            e.rates_reduced()

    i, o, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, "tap_N16_heat_3src.npz", 1)
    comm = OneRank()
    ev = pkg.Evolve(mesh, pkg.RadiationTables.load(), engine=OracleEngine(mesh, otables), comm=comm)
    n = ev.evolve3D(0.0, float(i["dt"][0]), 0, mat, grid, src, cosmo)
    assert n == len(o["conv_flags"]) and ev.conv_flags == [int(x) for x in o["conv_flags"]]
    assert np.array_equal(mat.xh, o["xh"]) and np.array_equal(mat.xhe, o["xhe"])
    assert np.array_equal(mat.temperature_grid, o["temperature"])


def _worker(rank, world, port, fname, call, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tests"))
    import __graft_entry__ as ge
    import oracle as orc
    from oracle_engine import OracleEngine
    pkg = ge.load_package()
    with np.load(pkg.evolve.DEFAULT_TABLES) as t:
        T = orc.Tables({k: t[k] for k in t.files})
    gold = lambda n: np.load(ROOT / "tests" / "golden" / n)
    i, o, mesh, mat, grid, src, cosmo = _inputs(pkg, gold, fname, call)
    ev = pkg.Evolve(mesh, pkg.RadiationTables.load(), engine=OracleEngine(mesh, T), comm=pkg.parallel.TorchComm())
    n = ev.evolve3D(0.0, float(i["dt"][0]), 0, mat, grid, src, cosmo)
    if rank == 0:
        q.put(dict(niter=n, conv=ev.conv_flags, xh=mat.xh, xhe=mat.xhe, nbox=ev.sum_nbox_all,
                   loss=ev.photon_loss_all[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_sources_sharded(pkg, gold):
    """world_size 2: rank r sweeps sources r+1, r+3, ...; one all-reduce per outer iteration; chemistry
    replicated.  3 sources -> ranks hold 2 and 1.  The sum over ranks associates the additions
    differently from the serial source loop ((s1+s3)+s2 vs (s1+s2)+s3), so results agree with the
    reference to rounding (and the iteration amplifies that, hence the looser cell-wise bound)."""
    fname, call = "tap_N16_heat_3src.npz", 2
    _, o = tap_case(gold(fname), call)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fname, call, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = collect_from_ranks(procs, q, timeout=600)
    assert abs(res["niter"] - len(o["conv_flags"])) <= 2
    assert res["nbox"] == int(o["sum_nbox_all"][0])
    assert abs(res["loss"] / o["photon_loss_all"][0] - 1) < 1e-9
    n = 16 ** 3
    assert abs(res["xh"][n:].mean() / o["xh"][n:].mean() - 1) < 1e-3
    assert np.max(np.abs(res["xh"] - o["xh"])) < 0.05


def test_mpi_lines_of_the_fortran_shim_compile(tmp_path):
    """The image has no MPI Fortran compiler, so the `#ifdef MPI` lines of the drop-in modules (MPI_BCAST of the RCCL
    id, of a restart dump and of the dump decision, MPI_ABORT) are never part of a build here.  Compile them all the
    same: -DMPI with the handful of MPI datatype constants they name given as cpp macros, against the module files of
    the reference's no-MPI build (whose my_mpi exports rank, npr, MPI_COMM_NEW like the MPI one); MPI_BCAST / MPI_ABORT
    stay implicit-interface externals, as with mpif.h.  Compile only -- nothing is linked or run."""
    import shutil
    import subprocess
    from conftest import ROOT
    fc = "/opt/rocm/lib/llvm/bin/flang"
    mods = ROOT / "oracle" / "_ref" / "N16"
    if not Path(fc).exists() or not (mods / "material.mod").exists():
        pytest.skip("flang or the reference's module files (oracle/ref_build.sh 16) not present")
    src = ROOT / "c2-ray3dm1d_helium_amd" / "fortran"
    defs = ["-DMPI", "-DMPI_INTEGER=1", "-DMPI_DOUBLE_PRECISION=2", "-DMPI_REAL=3", "-DMPI_CHARACTER=4", "-DMPI_LOGICAL=5"]
    for f in ("c2ray_hip_binding.f90", "evolve_data.F90", "evolve_source.F90", "evolve_point.F90", "evolve.F90"):
        r = subprocess.run([fc, "-cpp", "-O0", "-DGFORT", "-w", *defs, f"-I{mods}", "-c", str(src / f), "-o", str(tmp_path / (f + ".o"))],
                           cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, f + "\n" + r.stderr[-2000:]
    # and the lines are really there (an #ifdef that hides nothing proves nothing)
    text = (src / "evolve.F90").read_text() + (src / "evolve_data.F90").read_text()
    assert text.count("MPI_BCAST") >= 10 and "MPI_ABORT" in text


# ---------------------------------------------------------------------------------------------------------
# parallel.RcclComm sets its communicator up COLLECTIVELY: every rank leaves with one, or every rank raises.

class _FakeRcclEngine:
    """The four calls RcclComm makes, with failures injected per rank (no GPU, no RCCL)."""
    mode = "ok"
    rank = 0
    log: list = []

    @staticmethod
    def comm_available():
        return "no librccl here" if _FakeRcclEngine.mode == "unavailable_on_1" and _FakeRcclEngine.rank == 1 else None

    @staticmethod
    def comm_unique_id():
        if _FakeRcclEngine.mode == "id_fails":
            raise RuntimeError("ncclGetUniqueId failed (injected)")
        return b"x" * 128

    def comm_size(self):
        return 2 if _FakeRcclEngine.mode == "has_comm_on_1" and _FakeRcclEngine.rank == 1 else 1

    def comm_init(self, rank, size, uid):
        assert uid == b"x" * 128
        if _FakeRcclEngine.mode == "init_fails_on_1" and rank == 1:
            raise RuntimeError("ncclCommInitRank failed (injected)")
        if _FakeRcclEngine.mode == "init_hangs_on_0" and rank == 0:
            import time
            time.sleep(3600)          # a collective whose peer never arrives
        self.log.append("init")

    def comm_destroy(self):
        self.log.append("destroy")


def _comm_setup_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    _FakeRcclEngine.mode, _FakeRcclEngine.rank, _FakeRcclEngine.log = mode, rank, []
    eng = _FakeRcclEngine()
    try:
        pkg.parallel.RcclComm(eng, dist, fail_on_ranks=(0,) if mode == "forced_on_0" else ())
        q.put((rank, "ok", list(eng.log)))
    except RuntimeError as ex:
        q.put((rank, "raised: " + str(ex), list(eng.log)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "forced_on_0", "unavailable_on_1", "has_comm_on_1", "id_fails", "init_fails_on_1"])
def test_rccl_comm_setup_is_collective(mode):
    """Round-3 ADVICE: a failure on one rank only (rank 0 cannot obtain the id; one rank cannot load RCCL or fails in
    c2r_comm_init) left the other ranks waiting inside a collective.  Now both ranks return within the time-out,
    with the same verdict, and a rank whose own c2r_comm_init succeeded gives its communicator back."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_setup_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (verdict, log) for r, verdict, log in collect_from_ranks(procs, q, nresults=2, timeout=120)}
    if mode == "ok":
        assert got[0] == ("ok", ["init"]) and got[1] == ("ok", ["init"])
    else:
        assert got[0][0].startswith("raised") and got[1][0].startswith("raised"), got
        if mode == "init_fails_on_1":
            assert got[0][1] == ["init", "destroy"] and got[1][1] == []
        else:
            assert got[0][1] == [] and got[1][1] == []


def test_rccl_comm_init_watchdog_ends_the_process(monkeypatch):
    """Round-4 ADVICE: a rank stuck INSIDE the collective c2r_comm_init (its peer failed in ncclCommInitRank, or never came)
    cannot reach RcclComm's third agreement.  Its watchdog says so and exits the process with status 3 after
    C2R_COMM_INIT_TIMEOUT_S, which is what makes a launcher end the job; here the test plays the launcher."""
    import time
    monkeypatch.setenv("C2R_COMM_INIT_TIMEOUT_S", "3")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_setup_worker, args=(r, 2, port, "init_hangs_on_0", q)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    procs[0].join(timeout=90)
    try:
        assert procs[0].exitcode == 3, procs[0].exitcode
        assert time.time() - t0 < 80
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
            p.join(timeout=30)
