"""The drop-in proof: the reference's OWN, unmodified Fortran driver (C2Ray.F90), set-up modules and
output routines, linked once with the reference's evolve chain (oracle/_ref/N16/C2Ray_3D_test) and
once with the product's Fortran modules + libc2ray_hip.so (oracle/_ref/N16/C2Ray_3D_hip, see
oracle/ref_build.sh and INTEGRATION.md).  Both binaries are run on the same inputs on the GPU box
and every output file they write must be byte-identical.

The binaries are built in the dev container (they contain compiled reference code, so they are
git-ignored) and travel to the GPU box with the snapshot; without them the test is skipped."""
import filecmp
import json
import sys
from pathlib import Path

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(ROOT / "oracle"))


def _reference_snapshot(golden_name, run_reference):
    """What the all-reference binary produced for a case: from the fixture tests/golden/<golden_name> (made by
    oracle/make_golden_dropin.py from a run of that binary in the dev container -- the two cases whose reference
    run spends a minute or more integrating the PL / QPL tables on the host), or, with C2R_DROPIN_RUN_REFERENCE=1
    or without a fixture, from running it now (`run_reference()` returns the run directory)."""
    import json
    import os
    sys.path.insert(0, str(ROOT / "oracle"))
    import make_golden_dropin
    path = ROOT / "tests" / "golden" / golden_name if golden_name else None
    if path is not None and path.exists() and not os.environ.get("C2R_DROPIN_RUN_REFERENCE"):
        return json.loads(path.read_text())
    return json.loads(json.dumps(make_golden_dropin.snapshot(run_reference())))   # same types as the fixture


@pytest.mark.parametrize("iso,pl,sources", [
    (False, "lls", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),   # use_LLS = .true. build
    (False, False, [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # C2Ray_3D_hip_dogrid: the reference's own master_slave.F90 (unmodified) deals out the sources and calls
    # do_source of the product's module evolve_source -- the do_source call surface, one source per call
    (False, "dogrid", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # C2Ray_3D_hip_bycell: the global pass through the per-cell interface of the product's module evolve_point
    # (evolve0D_global for every cell, the reference's own loop)
    (False, "bycell", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),
    # C2Ray_3D_hip_bypoint: the reference's own evolve_source.F90 and master_slave.F90 (both unmodified) on top of the
    # per-cell interface of the product's module evolve_point: evolve0D(dt,rtpos,ns,niter) for every cell of every
    # sub-box, in the order of the reference's evolve2D sweeps (c2r_evolve0d, one launch per cell)
    (False, "bypoint", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),
    # C2RAY_HIP_BUILD_TABLES=1: the shim has the photo-ionisation / heating tables integrated on the device
    # (c2r_build_tables) instead of uploading rad_ini's host arrays
    (False, "devtables", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    (True, False, [(8, 8, 8, 1e55)]),
    # C2RAY_HIP_STEPWISE=1: every outer iteration call by call in the reference's order (pass, sum, minima, global pass,
    # means, state sums, total rates) instead of one c2r_iteration per iteration, which is what all other cases run
    (False, "stepwise", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # C2RAY_HIP_FORCE_COMM=1: the shim creates a one-rank RCCL communicator (c2r_comm_unique_id, c2r_comm_init)
    # and every iteration's c2r_allreduce_rates is a real ncclAllReduce
    (False, "comm1", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # C2RAY_HIP_NGPU=2 on one GPU (C2RAY_HIP_SAME_DEVICE=1): c2r_create_multi, one host thread per replica, each
    # sweeping one of the two sources, the in-process sum over the replicas, replicated chemistry.  Two sources
    # over two ranks add up as (0 + s1) + s2 either way: still byte-identical files
    (False, "ngpu2", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),
    # ... and the same with the sums carried by the library's RCCL path (comm_kind 1: ncclCommInitAll, grouped ncclAllReduce
    # per slab, c2r_iteration) through the one-device stand-in for librccl of tests/fake_rccl.hip (C2R_RCCL_LIBRARY,
    # C2R_COMM_SHARED_DEVICE_RCCL=1): the Fortran driver on "two GPUs" whose sum is not the in-process one
    (False, "ngpu2_standin", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),
    # C2RAY_HIP_KEEP_STATE=1 (opt-in): from the second evolve3D call on xh, xhe, temperature_grid stay on the device when a
    # sample of the host arrays still shows what the previous call downloaded, ndens is divided by cosmo_evol's zfactor**3 on
    # the device when that is all that happened to it (two time steps per slice: the step inside a slice), sent otherwise
    # (the first step of a slice: two cosmo_evol calls lie in between), and phihe_grid is not brought to the host
    (False, "keepstate", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # ... =2, for runs without output stream 3 (the only reader of phih_grid / phiheat on the host): no rate grid comes back
    (True, "keepstate2", [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]),
    # a host with ANOTHER c2ray_parameters.f90 (subboxsize = 4, max_subbox = 9, convergence_fraction = 1.0e-3; oracle/ref_build.sh
    # 16 params) and the library built for the same values (libc2ray_hip_params.so): evolve_ini's comparison of the compiled
    # constants passes, and the files are the reference's
    (False, "params", [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)]),
    # the -DPL -DQUASARS build (the flags of the reference's production targets, Makefile:185-186,207-208)
    (False, True, [(8, 8, 8, 1e55, 3e54, 0.0), (2, 15, 4, 0.0, 2e54, 4e54), (16, 1, 9, 2e54, 0.0, 1e54)]),
])
def test_reference_driver_with_hip_evolve_writes_identical_files(iso, pl, sources, monkeypatch):
    import refrun
    if pl == "devtables":
        monkeypatch.setenv("C2RAY_HIP_BUILD_TABLES", "1")
    devtables = pl == "devtables"
    if pl == "comm1":
        monkeypatch.setenv("C2RAY_HIP_FORCE_COMM", "1")
    if pl in ("ngpu2", "ngpu2_standin"):
        monkeypatch.setenv("C2RAY_HIP_NGPU", "2")
        monkeypatch.setenv("C2RAY_HIP_SAME_DEVICE", "1")
    standin = pl == "ngpu2_standin"
    if standin:
        fake = ROOT / "tests" / "_fake_rccl.so"
        if not fake.exists():
            pytest.skip("tests/_fake_rccl.so is built by __graft_entry__.build() / tests/test_gpu_rccl_standin.py")
        monkeypatch.setenv("C2R_RCCL_LIBRARY", str(fake))
        monkeypatch.setenv("C2R_COMM_SHARED_DEVICE_RCCL", "1")
    comm1, ngpu2 = pl == "comm1", pl in ("ngpu2", "ngpu2_standin")
    bycell = pl == "bycell"
    bypoint = pl == "bypoint"
    if bypoint:
        monkeypatch.setenv("C2RAY_HIP_POINT_INTERFACE", "1")
    stepwise = pl == "stepwise"
    if stepwise:
        monkeypatch.setenv("C2RAY_HIP_STEPWISE", "1")
    keep = {"keepstate": 1, "keepstate2": 2}.get(pl, 0)
    if keep:
        monkeypatch.setenv("C2RAY_HIP_KEEP_STATE", str(keep))
    steps = 2 if keep else 1
    streams = "0 1 0 0 0" if keep == 2 else "0 1 1 0 0"
    pvar = pl == "params"
    lls, dogrid, pl = pl == "lls", pl == "dogrid", pl is True
    which_hip = "hip_dogrid" if dogrid else ("hip_bycell" if bycell else ("hip_bypoint" if bypoint else "hip"))
    ref, hip = refrun.ref_binary(16, "test", pl=pl, lls=lls, params=pvar), refrun.ref_binary(16, which_hip, pl=pl, lls=lls, params=pvar)
    if not ref.exists() or not hip.exists():
        pytest.skip("oracle/_ref binaries not present (built only where /root/reference exists)")
    tag = ("iso" if iso else "heat") + ("_pl" if pl else "") + ("_lls" if lls else "") + ("_dogrid" if dogrid else "") + ("_devtables" if devtables else "") + ("_comm1" if comm1 else "") + ("_ngpu2" if ngpu2 else "") + ("_standin" if standin else "") + ("_bycell" if bycell else "") + ("_stepwise" if stepwise else "") + ("_bypoint" if bypoint else "") + (f"_keepstate{keep}" if keep else "") + ("_params" if pvar else "")
    import make_golden_dropin
    if pl:
        assert sources == make_golden_dropin.PL_SOURCES and not iso     # what the fixture was made from
    s1 = _reference_snapshot("dropin_ref_heat_pl.json" if pl else None,
                             lambda: refrun.run_reference(16, sources, isothermal=iso, steps_per_slice=steps, which="test",
                                                          name=f"dropin_ref_{tag}", pl=pl, lls=lls, streams=streams, params=pvar))
    r2 = refrun.run_reference(16, sources, isothermal=iso, steps_per_slice=steps, which=which_hip, name=f"dropin_hip_{tag}", pl=pl, lls=lls,
                              streams=streams, params=pvar)
    s2 = json.loads(json.dumps(make_golden_dropin.snapshot(r2)))
    assert len(s1["sha256"]) >= (9 if keep == 2 else 15), sorted(s1["sha256"])
    # every output file byte for byte (SHA-256 of the reference's file against the drop-in's)
    assert sorted(s1["sha256"]) == sorted(s2["sha256"])
    for f in s1["sha256"]:
        assert s1["sha256"][f] == s2["sha256"][f], f
    # same iteration history in the log
    assert s1["log_calls"] == s2["log_calls"]
    log2 = (r2 / "results" / "C2Ray.log").read_text(errors="replace")
    if keep:
        # 8 calls: state kept from the second on; ndens rescaled on the device inside a slice (calls 2, 4, 6, 8), sent at
        # the start of a slice, where cosmo_evol ran twice since the call before
        assert log2.count("xh, xhe, temperature_grid kept on the device") == 7, log2.count("kept on the device")
        assert log2.count("ndens rescaled on the device") == 4
    if devtables:
        assert "tables built on the device for SED  0" in log2
    if comm1:
        assert "RCCL communicator of one rank" in log2
    if ngpu2:
        assert "devices per rank:   2" in log2
        if standin:
            assert "the sum over the devices is an ncclAllReduce of" in log2 and "_fake_rccl.so" in log2
        else:
            assert "their sum is made by the library itself" in log2
    if stepwise or dogrid or bycell or bypoint:
        assert "outer iterations call by call" in log2
    else:
        assert "one library call per outer iteration" in log2
    # the reference's "min xh_av" / "min xhe_av" lines (evolve.F90:463-466), from a device minimum: same numbers
    assert len(s1["mins"]) > 0 and s1["mins"] == s2["mins"]
    # photon statistics (written from host arrays the HIP path filled): compare the numbers
    assert s1["photoncounts2"] == s2["photoncounts2"]
    # PhotonCounts.out (unit 90, report_photonstatistics): one line per global pass plus one per
    # evolve3D call, in both runs; the HIP shim fills it from grid sums reduced on the device (a
    # different summation order: equal to the printed 4 digits, up to a last-digit flip)
    import numpy as np

    def numbers(text):
        rows = []
        for line in text.splitlines():
            try:
                v = [float(x) for x in line.split()]
            except ValueError:
                continue
            if len(v) == 9:
                rows.append(v)
        return np.array(rows)

    pa = numbers(s1["photoncounts"])
    pb = numbers(s2["photoncounts"])
    calls = s1["log_calls"]
    assert pa.shape == pb.shape and pa.shape[0] == sum(len(c) + 1 for c in calls)
    # columns: total_ion, totalsrc, recomions, photon_loss, totrec, totcollisions, 3 ratios.  total_ion
    # is a difference of two large sums (before - after), so compare it relative to the source term
    # (the reference itself prints Infinity in ratio columns of steps without lost photons: same in both)
    assert np.array_equal(np.isfinite(pa), np.isfinite(pb))
    fin = np.isfinite(pa)
    pa, pb = np.where(fin, pa, 0.0), np.where(fin, pb, 0.0)
    scale = np.maximum(np.abs(pa), np.abs(pa[:, 1:2]) * 1e-3)
    assert np.all(np.abs(pb - pa) <= 2.5e-3 * scale), np.max(np.abs(pb - pa) / scale)


def test_random_source_lists_through_both_binaries():
    """Seeded random source lists (1-6 sources anywhere on the 16^3 mesh, 1e53-1e55 photons/s, isothermal or
    heating) through the reference binary and the reference driver + HIP evolve3D: byte-identical output files.
    C2R_DROPIN_FUZZ_CASES widens it."""
    import os
    import numpy as np
    import refrun
    ref, hip = refrun.ref_binary(16, "test"), refrun.ref_binary(16, "hip")
    if not ref.exists() or not hip.exists():
        pytest.skip("oracle/_ref binaries not present (built only where /root/reference exists)")
    rng = np.random.default_rng(int(os.environ.get("C2R_FUZZ_SEED", "20261004")))
    for ic in range(int(os.environ.get("C2R_DROPIN_FUZZ_CASES", "2"))):
        nsrc = int(rng.integers(1, 7))
        sources = [(int(rng.integers(1, 17)), int(rng.integers(1, 17)), int(rng.integers(1, 17)), float(10.0 ** rng.uniform(53, 55)))
                   for _ in range(nsrc)]
        iso = bool(rng.random() < 0.5)
        r1 = refrun.run_reference(16, sources, isothermal=iso, steps_per_slice=1, which="test", name=f"dropin_fz_ref_{ic}")
        r2 = refrun.run_reference(16, sources, isothermal=iso, steps_per_slice=1, which="hip", name=f"dropin_fz_hip_{ic}")
        files = sorted(p.name for p in (r1 / "results").glob("*.bin"))
        assert len(files) >= 15, files
        for f in files:
            assert filecmp.cmp(r1 / "results" / f, r2 / "results" / f, shallow=False), (ic, sources, iso, f)
        assert refrun.parse_log(r1) == refrun.parse_log(r2), (ic, sources, iso)


def test_cubep3m_material_through_the_dropin():
    """The material side of the production target C2Ray_3D_cubep3m_kyl_periodic (files_for_3D/Makefile:207-210)
    through the drop-in: the reference's own cubep3m.F90 + mat_ini_cubep3m.F90 read a generated snapshot in the
    cubep3m formats -- `<z>n_all.dat` density (stream, 3 x int32 header + float32 N^3, mat_ini_cubep3m.F90:223-351),
    the halos-included grid behind type_of_clumping = 5 and the cross-section grid behind use_LLS / type_of_LLS = 2
    (oracle/refrun.py:run_cubep3m writes them, seeded) -- with -DPL -DQUASARS sources and heating; once linked with
    the reference's evolve chain, once with the product's modules + libc2ray_hip.so.  Two time steps of 501 outer
    iterations each (a handful of cells in the position-dependent clumping field never settle, so both runs go to
    the iteration cap): every output file byte-identical, the same non-converged count after every iteration.
    (The target's own source module, sourceprops_cubep3m.F90, has unbalanced ENDIFs at :307 and :397 in this
    snapshot of the reference and compiles with no compiler, so the sources come through sourceprops_test.F90.)"""
    import refrun
    ref = refrun.REFDIR / "N16_cubep3m" / "C2Ray_3D_test"
    hip = refrun.REFDIR / "N16_cubep3m" / "C2Ray_3D_hip"
    if not ref.exists() or not hip.exists():
        pytest.skip("oracle/_ref/N16_cubep3m binaries not present (built only where /root/reference exists)")
    import json
    import make_golden_dropin
    sources = make_golden_dropin.CUBEP3M_SOURCES
    s1 = _reference_snapshot("dropin_ref_cubep3m.json", lambda: refrun.run_cubep3m(16, sources, which="test", name="dropin_cubep3m_ref"))
    r2 = refrun.run_cubep3m(16, sources, which="hip", name="dropin_cubep3m_hip")
    s2 = json.loads(json.dumps(make_golden_dropin.snapshot(r2)))
    assert len(s1["sha256"]) >= 10, sorted(s1["sha256"])
    assert sorted(s1["sha256"]) == sorted(s2["sha256"])
    for f in s1["sha256"]:
        assert s1["sha256"][f] == s2["sha256"][f], f
    a, b = s1["log_calls"], s2["log_calls"]
    assert a == b and len(a) == 2 and len(a[0]) > 100
    # both runs read the generated snapshot through the reference's own ingest
    assert len(s1["log_markers"]) == 4 and s1["log_markers"] == s2["log_markers"]
