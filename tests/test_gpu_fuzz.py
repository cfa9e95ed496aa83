"""Randomised GPU-vs-oracle parity: meshes of odd shapes and sizes, sources anywhere (edges, corners, the same
cell twice), any batch size, isothermal or heating, with or without the extra SEDs, Lyman-limit systems and
clumping grids -- one pass and one global pass each, every array compared bit for bit.  The case list is seeded;
C2R_FUZZ_CASES / C2R_FUZZ_SEED widen or move it."""
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _tables(pkg, orc, multi):
    t = pkg.RadiationTables.load()
    with np.load(pkg.evolve.DEFAULT_TABLES) as z:
        d = {k: z[k] for k in z.files}
    if multi:
        t.add_sed_file(GOLD / "rad_tables_pl_qpl.npz")
        with np.load(GOLD / "rad_tables_pl_qpl.npz") as z:
            d.update({k: z[k] for k in z.files})
    return t, orc.Tables(d)


def _case(rng):
    mesh = tuple(int(x) for x in rng.choice([2, 3, 5, 8, 9, 12, 16, 21, 24, 30, 33, 40], size=3))
    if rng.random() < 0.4:
        mesh = (mesh[0],) * 3
    return dict(mesh=mesh, nsrc=int(rng.integers(1, 7)), iso=bool(rng.random() < 0.5), multi=bool(rng.random() < 0.3),
                lls=int(rng.integers(0, 3)), clump=bool(rng.random() < 0.3), batch=int(rng.integers(1, 9)),
                opacity=float(rng.uniform(-4.5, -0.5)), seed=int(rng.integers(1 << 30)))


def _progress(text):
    """A long run keeps telling the GPU box that it is alive (a line per case in gpurun_out/fuzz_progress.log)."""
    try:
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        out.mkdir(exist_ok=True)
        with open(out / "fuzz_progress.log", "a") as f:
            f.write(text + "\n")
    except OSError:
        pass


def test_random_cases_vs_oracle(pkg, orc):
    ncases = int(os.environ.get("C2R_FUZZ_CASES", "40"))
    master = np.random.default_rng(int(os.environ.get("C2R_FUZZ_SEED", "20261004")))
    hp = pkg.hostphys
    cache = {}
    # found by this test: a mesh only two cells deep never opens a sub-box (evolve_source.F90:136-139), and such
    # a source must then contribute nothing at all
    fixed = [dict(mesh=(24, 2, 2), nsrc=3, iso=True, multi=False, lls=1, clump=False, batch=2, opacity=-3.8, seed=393658753),
             dict(mesh=(5, 9, 2), nsrc=2, iso=False, multi=True, lls=0, clump=True, batch=1, opacity=-2.0, seed=17)]
    for ic in range(ncases + len(fixed)):
        cs = fixed[ic] if ic < len(fixed) else _case(master)
        _progress(f"case {ic}: {cs}")
        rng = np.random.default_rng(cs["seed"])
        n1, n2, n3 = cs["mesh"]
        nc = n1 * n2 * n3
        zred = 9.0
        dr, vol = hp.test_grid(max(cs["mesh"]), zred)
        ndens = hp.test_density(zred) * np.exp(rng.normal(0.0, 0.6, nc))
        xn = np.minimum(0.999, 10.0 ** (cs["opacity"] + rng.uniform(-0.5, 0.5, nc)))   # neutral fraction
        xh = np.concatenate([xn, 1.0 - xn])
        xhe = np.concatenate([xn, 0.8 * (1.0 - xn), 0.2 * (1.0 - xn)])
        temp = None if cs["iso"] else np.tile((1e4 * np.exp(rng.normal(0, 0.3, nc))).astype(np.float32), 3)
        srcpos = np.stack([rng.integers(1, n + 1, size=cs["nsrc"]) for n in cs["mesh"]], axis=1).astype(np.int32)
        if cs["nsrc"] > 1 and rng.random() < 0.3:
            srcpos[1] = srcpos[0]                       # two sources in one cell
        if rng.random() < 0.3:
            srcpos[0] = (1, n2, 1)                      # a corner
        flux = 10.0 ** rng.uniform(4.0, 7.5, cs["nsrc"])
        pl = qpl = None
        if cs["multi"]:
            pl = np.where(rng.random(cs["nsrc"]) < 0.6, 10.0 ** rng.uniform(4.0, 7.0, cs["nsrc"]), 0.0)
            qpl = np.where(rng.random(cs["nsrc"]) < 0.6, 10.0 ** rng.uniform(4.0, 7.0, cs["nsrc"]), 0.0)
            flux = np.where(rng.random(cs["nsrc"]) < 0.8, flux, 0.0)
        lls_grid = clump = None
        coldensh_lls = None
        if cs["lls"] == 1:
            coldensh_lls = float(10.0 ** rng.uniform(15, 17))
        elif cs["lls"] == 2:
            lls_grid = (10.0 ** rng.uniform(15.0, 17.5, nc)).astype(np.float32)
        if cs["clump"]:
            clump = (1.0 + 5.0 * rng.random(nc)).astype(np.float32)
        key = cs["multi"]
        if key not in cache:
            cache[key] = _tables(pkg, orc, cs["multi"])
        t, T = cache[key]
        mat = pkg.Material(ndens, xh, xhe, temp, cs["iso"], 1.0e4, 1.7, hp.reccoef(1.0e4), clumping_grid=clump,
                           use_LLS=cs["lls"] > 0, coldensh_LLS=coldensh_lls or 0.0, LLS_grid=lls_grid)
        grid = pkg.GridProps(cs["mesh"], dr, vol)
        src = pkg.SourceProps(srcpos, flux, 1.0e48, NormFluxPL=pl, pl_S_star=1.5e48, NormFluxQPL=qpl, qpl_S_star=0.7e48)
        cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
        e = pkg.HipEngine(cs["mesh"], 0)
        e.set_tables(t)
        e.set_step(mat, grid, cosmo)
        e.set_sources(src)
        e.upload_state(mat)
        e.set_batch(cs["batch"])
        e.begin_step()
        e.set_rates_to_zero()
        if rng.random() < 0.4:      # the slab-wise hand-over of the same pass
            for sl in range(e.pass_sources_begin(1, 1, int(rng.integers(1, 9)))):
                e.pass_wait_slab(sl)
            e.pass_sources_end()
        else:
            e.pass_sources(1, 1)
        rates = e.download_rates()
        cols = e.download_columns()
        st = orc.Step(cs["mesh"], dr, vol, zred, hp.H0, hp.Omega0, cs["iso"], 1.0e4, 1.7, srcpos, flux, 1.0e48, ndens,
                      hp.reccoef(1.0e4), normflux_pl=pl, normflux_qpl=qpl, pl_s_star=1.5e48, qpl_s_star=0.7e48,
                      coldensh_lls=coldensh_lls, lls_grid=lls_grid, clumping_grid=clump)
        s = orc.State(st, xh, xhe, temp)
        orc.begin_step(s)
        orc.pass_all_sources(T, st, s)
        tag = (ic, cs)
        assert rates["sum_nbox"] == int(s.c.sum_nbox), tag
        assert np.array_equal(cols["coldensh_out"], s.coldensh_out), tag
        assert np.array_equal(cols["coldenshe_out"], s.coldenshe_out), tag
        assert np.array_equal(rates["phih_grid"], s.phih) and np.array_equal(rates["phihe_grid"], s.phihe), tag
        if not cs["iso"]:
            assert np.array_equal(rates["phiheat"], s.phiheat), tag
        if s.photon_loss[0] > 0:
            assert rel_err(rates["photon_loss"][0], s.photon_loss[0]) <= 1e-12, tag
        dt = float(10.0 ** rng.uniform(5, 7.3)) * hp.YEAR
        if rng.random() < 0.4:      # the global pass in pieces
            cut = int(rng.integers(0, nc + 1))
            e.global_pass_cells(dt, 0, cut)
            e.global_pass_cells(dt, cut, nc - cut)
            conv = e.global_pass_finish()
        else:
            conv = e.global_pass(dt)
        assert conv == orc.global_pass(T, st, s, dt), tag
        it = e.download_iter_state()
        for k in ("xh_av", "xhe_av", "xh_intermed", "xhe_intermed"):
            assert np.array_equal(it[k], getattr(s, k)), (k, tag)
        if not cs["iso"]:
            m2 = pkg.Material(None, np.empty(2 * nc), np.empty(3 * nc), np.empty(3 * nc, dtype=np.float32), False)
            e.download_state(m2)
            assert np.array_equal(m2.temperature_grid, s.temperature), tag
        e.close()


def test_random_whole_evolve3d_vs_oracle(pkg, orc):
    """Whole evolve3D calls (all outer iterations, end-of-step copies, the heating pass's tiers from the second
    iteration on) on random small boxes: iteration count, per-iteration non-converged counts and the final
    state bit for bit against the oracle."""
    ncases = int(os.environ.get("C2R_FUZZ_EVOLVE_CASES", "8"))
    master = np.random.default_rng(int(os.environ.get("C2R_FUZZ_SEED", "20261004")) + 1)
    hp = pkg.hostphys
    t, T = _tables(pkg, orc, False)
    for ic in range(ncases):
        rng = np.random.default_rng(int(master.integers(1 << 30)))
        sizes = [60, 64, 72] if os.environ.get("C2R_FUZZ_BIG") else [8, 11, 14, 16, 20]   # big: the sampled heating pass
        mesh = tuple(int(x) for x in rng.choice(sizes, size=3))
        iso = bool(rng.random() < 0.4)
        nsrc = int(rng.integers(1, 4))
        nc = mesh[0] * mesh[1] * mesh[2]
        zred = 9.0
        dr, vol = hp.test_grid(max(mesh), zred)
        ndens = hp.test_density(zred) * np.exp(rng.normal(0.0, 0.5, nc)) * float(10.0 ** rng.uniform(0, 1.5))
        xn = np.minimum(0.999, 10.0 ** (rng.uniform(-3, -0.3) + rng.uniform(-0.3, 0.3, nc)))
        xh = np.concatenate([xn, 1.0 - xn])
        xhe = np.concatenate([xn, 0.8 * (1.0 - xn), 0.2 * (1.0 - xn)])
        temp = None if iso else np.tile((1e4 * np.exp(rng.normal(0, 0.3, nc))).astype(np.float32), 3)
        srcpos = np.stack([rng.integers(1, n + 1, size=nsrc) for n in mesh], axis=1).astype(np.int32)
        flux = 10.0 ** rng.uniform(5.0, 7.0, nsrc)
        dt = float(10.0 ** rng.uniform(5.5, 7.0)) * hp.YEAR
        mat = pkg.Material(ndens, xh.copy(), xhe.copy(), None if temp is None else temp.copy(), iso, 1.0e4, 1.0, hp.reccoef(1.0e4))
        grid = pkg.GridProps(mesh, dr, vol)
        src = pkg.SourceProps(srcpos, flux, 1.0e48)
        cosmo = pkg.Cosmology(zred, hp.H0, hp.Omega0)
        ev = pkg.Evolve(mesh, t, device=0)
        niter = ev.evolve3D(0.0, dt, 0, mat, grid, src, cosmo)
        st = orc.Step(mesh, dr, vol, zred, hp.H0, hp.Omega0, iso, 1.0e4, 1.0, srcpos, flux, 1.0e48, ndens, hp.reccoef(1.0e4))
        s = orc.State(st, xh, xhe, temp)
        tag = (ic, mesh, iso, nsrc)
        _progress(f"evolve3D case {tag}: {niter} iterations on the device, the oracle next")
        assert orc.evolve3d(T, st, s, dt) == niter, tag
        assert ev.conv_flags == s.conv_flags, tag
        assert np.array_equal(mat.xh, s.xh) and np.array_equal(mat.xhe, s.xhe), tag
        if not iso:
            assert np.array_equal(mat.temperature_grid, s.temperature), tag
        ev.engine.close()
