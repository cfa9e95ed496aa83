"""Host-side mirror of the reference's `evolve` / `evolve_data` modules for a Python host.

The reference's drop-in boundary is a set of Fortran modules with fixed public names
(SURVEY.md section 8b); fortran/ holds those modules for a Fortran host.  This file offers the
same call surface to Python (tests, bench.py): the same names, argument meaning and state
ownership -- the *host* owns ndens, xh, xhe, temperature_grid, srcpos, NormFlux; `Evolve` owns the
work arrays (phih_grid, xh_av, ...) which live on the GPU and are downloaded on request.

    material / grid / sourceprops state  ->  Evolve.evolve3D(time, dt, restart)
                                             (files_for_3D/evolve.F90:78-229)

All numerical work happens in libc2ray_hip.so (HIP, gfx950).  There is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from . import _lib
from ._lib import C2RayHipError, NFREQ

PKG = Path(__file__).resolve().parent
DEFAULT_TABLES = PKG / "data" / "rad_tables_bb5e4.npz"

FVEC_ORDER = ["f1ion_HI", "f1ion_HeI", "f1ion_HeII", "f2ion_HI", "f2ion_HeI", "f2ion_HeII",
              "f1heat_HI", "f1heat_HeI", "f1heat_HeII", "f2heat_HI", "f2heat_HeI", "f2heat_HeII"]

convergence_fraction = float(np.float32(2.5e-4))  # c2ray_parameters.f90:26 (REAL(4) literal)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class RadiationTables:
    """What rad_ini (radiation_tables.f90:141-168) and setup_cool (cooling_h.f90:76-171) leave
    behind: inputs of the hot path, built once on the host."""
    photo_thick: np.ndarray
    photo_thin: np.ndarray
    heat_thick: np.ndarray | None
    heat_thin: np.ndarray | None
    sigma_HI: np.ndarray
    sigma_HeI: np.ndarray
    sigma_HeII: np.ndarray
    fvec: dict
    bb_upper: int
    cool: np.ndarray | None = None
    cool_mintemp: float = 1.0
    cool_dtemp: float = 0.01
    # -DPL / -DQUASARS builds: sed[1] (power law) / sed[2] (quasar-like) =
    # dict(photo_thick, photo_thin, heat_thick, heat_thin, lower, upper)
    sed: dict = field(default_factory=dict)
    # What spec_integration starts from (band set-up, Romberg weights, normalised SEDs; keys as in
    # tests/golden/sed_setup.npz).  With build_on_device the tables above are ignored and the engine
    # builds them on the GPU (c2r_build_tables) -- bit-identical to the host-built ones.
    setup: dict | None = None
    build_on_device: bool = False

    def add_sed_file(self, path):
        """pl_* / qpl_* tables and band limits as dumped from a -DPL -DQUASARS reference build."""
        with np.load(path) as z:
            for idx, pre in ((1, "pl_"), (2, "qpl_")):
                if pre + "photo_thick" in z.files:
                    self.sed[idx] = dict(photo_thick=_f64(z[pre + "photo_thick"]), photo_thin=_f64(z[pre + "photo_thin"]),
                                         heat_thick=_f64(z[pre + "heat_thick"]) if pre + "heat_thick" in z.files else None,
                                         heat_thin=_f64(z[pre + "heat_thin"]) if pre + "heat_thin" in z.files else None,
                                         lower=int(z[pre + "limits"][0]), upper=int(z[pre + "limits"][1]))
        return self

    @classmethod
    def load(cls, path=DEFAULT_TABLES):
        with np.load(path) as z:
            g = lambda k: _f64(z[k]) if k in z.files else None
            return cls(g("photo_thick"), g("photo_thin"), g("heat_thick"), g("heat_thin"), g("sigma_HI"),
                       g("sigma_HeI"), g("sigma_HeII"), {k: g(k) for k in FVEC_ORDER if k in z.files},
                       int(z["bb_upper"]), g("cool"),
                       float(z["cool_mintemp"]) if "cool_mintemp" in z.files else 1.0,
                       float(z["cool_dtemp"]) if "cool_dtemp" in z.files else 0.01)


@dataclass
class Material:
    """module material (files_for_3D/mat_ini_test.F90:27-36): host-owned state arrays, Fortran
    layout flattened (i fastest, component slowest)."""
    ndens: np.ndarray            # (N^3,) float64
    xh: np.ndarray               # (2*N^3,) float64, components 0:1
    xhe: np.ndarray              # (3*N^3,) float64, components 0:2
    temperature_grid: np.ndarray | None = None  # (3*N^3,) float32, slots 0:2; None when isothermal
    isothermal: bool = True
    temper_val: float = 1.0e4
    clumping: float = 1.0
    reccoef: np.ndarray = field(default_factory=lambda: np.zeros(12))  # cgsconstants.f90:106-133
    # type_of_clumping = 5: REAL(4) clumping_grid(mesh) read per cell (mat_ini_cubep3m.F90:635-646)
    clumping_grid: np.ndarray | None = None
    # use_LLS (c2ray_parameters.f90:72-78): coldensh_LLS of type 1, or the REAL(4) LLS_grid of type 2
    use_LLS: bool = False
    coldensh_LLS: float = 0.0
    LLS_grid: np.ndarray | None = None


@dataclass
class GridProps:
    """module grid (files_for_3D/grid.F90): cell sizes and volume (proper cm, cm^3)."""
    mesh: tuple
    dr: tuple
    vol: float


@dataclass
class SourceProps:
    """module sourceprops (files_for_3D/sourceprops_test.F90:38-40)."""
    srcpos: np.ndarray           # (NumSrc, 3) int32, 1-based mesh coordinates
    NormFlux: np.ndarray         # (NumSrc,) photons/s divided by S_star
    S_star: float = 1.0e48
    NormFluxPL: np.ndarray | None = None    # -DPL builds
    pl_S_star: float = 1.0e48
    NormFluxQPL: np.ndarray | None = None   # -DQUASARS builds
    qpl_S_star: float = 1.0e48

    @property
    def NumSrc(self):
        return int(len(self.NormFlux))


@dataclass
class Cosmology:
    zred: float = 0.0
    H0: float = 0.0
    Omega0: float = 0.0


class HipEngine:
    """Thin object wrapper over the C ABI.  `device`: one GPU, or a list of GPUs driven by this process
    (c2r_create_multi: every state-setting call goes to all of them, results are read from the first)."""

    def __init__(self, mesh, device=0):
        self.lib = _lib.load()
        self.mesh = tuple(int(m) for m in mesh)
        self.ncell = int(np.prod(self.mesh))
        h = C.c_void_p()
        m = (C.c_int * 3)(*self.mesh)
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            rc = self.lib.c2r_create_multi(C.byref(h), len(device), devs, m)
        else:
            rc = self.lib.c2r_create(C.byref(h), int(device), m)
        if rc != 0:
            raise C2RayHipError(self.lib.c2r_create_error().decode())
        self.h = h
        self._ext_rates = None

    # -- several GPUs: the sum over ranks behind the C ABI (RCCL) ----------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId, for ONE rank to obtain and the launcher to pass to the others."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        if lib.c2r_comm_unique_id(buf) != 0:
            raise C2RayHipError(lib.c2r_create_error().decode())
        return buf.raw

    @staticmethod
    def comm_available():
        """None when RCCL can be loaded and used by this library, else the reason (a string)."""
        lib = _lib.load()
        return None if lib.c2r_comm_available() == 0 else lib.c2r_create_error().decode()

    @staticmethod
    def comm_library():
        """Path of the library that carries the sums over ranks (c2r_comm_library); raises when none can be loaded."""
        lib = _lib.load()
        buf = C.create_string_buffer(1024)
        if lib.c2r_comm_library(buf, 1024) != 0:
            raise C2RayHipError(lib.c2r_create_error().decode())
        return buf.value.decode()

    def comm_destroy(self):
        self._chk(self.lib.c2r_comm_destroy(self.h))

    def comm_init(self, first_rank, nranks, unique_id):
        assert len(unique_id) == 128
        self._chk(self.lib.c2r_comm_init(self.h, int(first_rank), int(nranks), unique_id))

    def comm_init_local(self):
        self._chk(self.lib.c2r_comm_init_local(self.h))

    def comm_size(self):
        return int(self.lib.c2r_comm_nranks(self.h))

    def rccl_ranks(self):
        """Ranks of the RCCL communicator that sums the rate grids (0: no communicator, or the in-process sum of
        replicas that share a device)."""
        return self.comm_size() if int(self.lib.c2r_comm_kind(self.h)) == 1 else 0

    def num_devices(self):
        return int(self.lib.c2r_num_devices(self.h))

    def allreduce_rates(self):
        self._chk(self.lib.c2r_allreduce_rates(self.h))

    def pass_allreduce_chemistry(self, dt, first=1, stride=1, nslab=4):
        """pass_all_sources + sum over ranks + global pass of one outer iteration, overlapped slab by slab;
        returns the non-converged count."""
        cf = C.c_int(0)
        self._chk(self.lib.c2r_pass_allreduce_chemistry(self.h, int(first), int(stride), int(nslab), float(dt), C.byref(cf)))
        return cf.value

    def iteration(self, dt, first=1, stride=1, nslab=4):
        """c2r_iteration: one outer iteration (pass, sum over ranks, global pass) and every grid reduction the reference's
        loop reports after it, with one synchronisation; returns the report as a dict of numpy arrays / ints."""
        rep = _lib.IterationReport()
        self._chk(self.lib.c2r_iteration(self.h, int(first), int(stride), int(nslab), float(dt), C.byref(rep)))
        out = {"conv_flag": rep.conv_flag, "sum_nbox": rep.sum_nbox}
        for k in ("photon_loss", "means_intermed", "sums_intermed", "total_rates", "minima_av", "reccoef"):
            out[k] = np.array(getattr(rep, k)[:])
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.c2r_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise C2RayHipError(self.lib.c2r_last_error(self.h).decode())

    # -- inputs ------------------------------------------------------------------------------
    def set_tables(self, t: RadiationTables):
        fv = None
        if t.fvec and len(t.fvec) == 12:
            arr = (C.POINTER(C.c_double) * 12)(*[_dp(t.fvec[k]) for k in FVEC_ORDER])
            fv = arr
        on_dev = bool(t.build_on_device)
        if on_dev and t.setup is None:
            raise C2RayHipError("build_on_device needs RadiationTables.setup")
        heat = t.heat_thick is not None or (on_dev and fv is not None)
        self._chk(self.lib.c2r_set_tables(self.h, None if on_dev else _dp(t.photo_thick), None if on_dev else _dp(t.photo_thin),
                                          None if on_dev else _dp(t.heat_thick), None if on_dev else _dp(t.heat_thin),
                                          _dp(t.sigma_HI), _dp(t.sigma_HeI), _dp(t.sigma_HeII), fv, int(t.bb_upper)))
        if on_dev:
            self.build_tables(t.setup, 0, heat)
        if t.cool is not None:
            self._chk(self.lib.c2r_set_cooling(self.h, _dp(t.cool), float(t.cool_mintemp), float(t.cool_dtemp)))
        for idx, d in t.sed.items():
            if on_dev:
                self._chk(self.lib.c2r_set_sed_tables(self.h, int(idx), None, None, None, None, d["lower"], d["upper"]))
                self.build_tables(t.setup, int(idx), heat)
            else:
                self._chk(self.lib.c2r_set_sed_tables(self.h, int(idx), _dp(d["photo_thick"]), _dp(d["photo_thin"]),
                                                      _dp(d["heat_thick"]), _dp(d["heat_thin"]), d["lower"], d["upper"]))

    def build_tables(self, d, sed=0, heat=True):
        """spec_integration on the device (c2r_build_tables) from a set-up dictionary."""
        from ._lib import SedSetup
        b = np.arange(1, 48)
        xi = _f64(np.where(b <= 1, d["pl_index_HI"], np.where(b <= 27, d["pl_index_HeI"], d["pl_index_HeII"])))
        keep = [_f64(d[k]) for k in ("freq_min", "delta_freq", "tau", "romw9")] + [xi]
        s = SedSetup()
        s.nfreq, s.sed = 512, int(sed)
        s.freq_min, s.delta_freq, s.tau, s.romw, s.xsec_index = (_dp(a) for a in keep)
        s.R_star2, s.h_over_kT, s.two_pi_over_c_square = (float(x) for x in d["sed_setup"])
        c = d["consts"]
        s.pi, s.hplanck = float(c[0]), float(c[5])
        s.ion_freq_HI, s.ion_freq_HeI, s.ion_freq_HeII = float(c[22]), float(c[23]), float(c[24])
        if sed:
            s.pl_scaling, s.pl_index = (float(x) for x in d["pl_setup" if sed == 1 else "qpl_setup"])
        self._chk(self.lib.c2r_build_tables(self.h, C.byref(s), int(bool(heat))))

    def download_tables(self, sed=0, heat=True):
        nt = 2001
        out = {"photo_thick": np.empty(47 * nt), "photo_thin": np.empty(47 * nt)}
        if heat:
            out.update(heat_thick=np.empty(113 * nt), heat_thin=np.empty(113 * nt))
        self._chk(self.lib.c2r_download_tables(self.h, int(sed), _dp(out["photo_thick"]), _dp(out["photo_thin"]),
                                               _dp(out.get("heat_thick")), _dp(out.get("heat_thin"))))
        return out

    def set_step(self, mat: Material, grid: GridProps, cosmo: Cosmology):
        nd = _f64(mat.ndens).reshape(-1)
        assert nd.size == self.ncell, "ndens has the wrong size"
        dr = (C.c_double * 3)(*[float(x) for x in grid.dr])
        rc = _f64(mat.reccoef).reshape(-1)
        assert rc.size == 12
        self.isothermal = bool(mat.isothermal)
        self._chk(self.lib.c2r_set_step(self.h, _dp(nd), dr, float(grid.vol), float(mat.clumping), float(cosmo.zred),
                                        float(cosmo.H0), float(cosmo.Omega0), int(bool(mat.isothermal)),
                                        float(mat.temper_val), _dp(rc)))
        lls = None if mat.LLS_grid is None else np.ascontiguousarray(mat.LLS_grid, dtype=np.float32).reshape(-1)
        cg = None if mat.clumping_grid is None else np.ascontiguousarray(mat.clumping_grid, dtype=np.float32).reshape(-1)
        assert lls is None or lls.size == self.ncell
        assert cg is None or cg.size == self.ncell
        fp = C.POINTER(C.c_float)
        self._chk(self.lib.c2r_set_lls(self.h, int(bool(mat.use_LLS)), float(mat.coldensh_LLS),
                                       None if lls is None else lls.ctypes.data_as(fp)))
        self._chk(self.lib.c2r_set_clumping_grid(self.h, None if cg is None else cg.ctypes.data_as(fp)))

    def set_step_scalars(self, mat: Material, grid: GridProps, cosmo: Cosmology):
        """c2r_set_step without the density: what is on the device stays (see scale_ndens)."""
        dr = (C.c_double * 3)(*[float(x) for x in grid.dr])
        rc = _f64(mat.reccoef).reshape(-1)
        self.isothermal = bool(mat.isothermal)
        self._chk(self.lib.c2r_set_step_scalars(self.h, dr, float(grid.vol), float(mat.clumping), float(cosmo.zred), float(cosmo.H0),
                                                float(cosmo.Omega0), int(bool(mat.isothermal)), float(mat.temper_val), _dp(rc)))

    def arena_stats(self):
        """c2r_arena_stats as a dict (column scratch: segments allocated, of them inside a pass, ...)."""
        out = (C.c_longlong * 6)()
        self._chk(self.lib.c2r_arena_stats(self.h, out))
        return dict(zip(("segments", "segments_in_pass", "doubles_allocated", "block_moves", "batch_restarts", "doubles_held"), out))

    def scale_ndens(self, divisor):
        """cosmo_evol's ndens = ndens / zfactor3 on the device copy (cosmology.f90:193)."""
        self._chk(self.lib.c2r_scale_ndens(self.h, float(divisor)))

    def set_sources(self, src: SourceProps):
        pos = np.ascontiguousarray(src.srcpos, dtype=np.int32).reshape(-1)
        nf = _f64(src.NormFlux).reshape(-1)
        assert pos.size == 3 * nf.size
        self._chk(self.lib.c2r_set_sources(self.h, int(nf.size), pos.ctypes.data_as(C.POINTER(C.c_int)), _dp(nf),
                                           float(src.S_star)))
        self.nsrc = int(nf.size)
        for idx, flux, star in ((1, src.NormFluxPL, src.pl_S_star), (2, src.NormFluxQPL, src.qpl_S_star)):
            if flux is not None:
                f = _f64(flux).reshape(-1)
                assert f.size == nf.size
                self._chk(self.lib.c2r_set_sources_sed(self.h, idx, _dp(f), float(star)))

    def upload_state(self, mat: Material):
        xh, xhe = _f64(mat.xh).reshape(-1), _f64(mat.xhe).reshape(-1)
        assert xh.size == 2 * self.ncell and xhe.size == 3 * self.ncell
        tp = None
        if mat.temperature_grid is not None:
            t = np.ascontiguousarray(mat.temperature_grid, dtype=np.float32).reshape(-1)
            assert t.size == 3 * self.ncell
            tp = t.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.lib.c2r_upload_state(self.h, _dp(xh), _dp(xhe), tp))

    def download_state(self, mat: Material):
        xh = np.empty(2 * self.ncell)
        xhe = np.empty(3 * self.ncell)
        t = None if mat.temperature_grid is None else np.empty(3 * self.ncell, dtype=np.float32)
        tp = None if t is None else t.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.lib.c2r_download_state(self.h, _dp(xh), _dp(xhe), tp))
        mat.xh, mat.xhe = xh, xhe
        if t is not None:
            mat.temperature_grid = t

    # -- the pieces of evolve3D --------------------------------------------------------------
    def begin_step(self):
        self._chk(self.lib.c2r_begin_step(self.h))

    def set_rates_to_zero(self):
        self._chk(self.lib.c2r_set_rates_to_zero(self.h))

    def pass_sources(self, first=1, stride=1):
        self._chk(self.lib.c2r_pass_sources(self.h, int(first), int(stride)))

    def pass_sources_begin(self, first=1, stride=1, nslab=8):
        """Queue the pass and return at once; returns the number of slabs to wait for."""
        self._chk(self.lib.c2r_pass_sources_begin(self.h, int(first), int(stride), int(nslab)))
        return int(self.lib.c2r_pass_slab_count(self.h))

    def pass_wait_slab(self, slab):
        """Block until the rate grids of this slab are final; (first_cell, ncells) of every component."""
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.lib.c2r_pass_wait_slab(self.h, int(slab), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def pass_sources_end(self):
        self._chk(self.lib.c2r_pass_sources_end(self.h))

    def do_source(self, ns):
        self._chk(self.lib.c2r_do_source(self.h, int(ns)))

    def global_pass(self, dt):
        cf = C.c_int(0)
        self._chk(self.lib.c2r_global_pass(self.h, float(dt), C.byref(cf)))
        return cf.value

    def global_pass_cells(self, dt, first_cell, ncells, after_event=None):
        """Queue the chemistry of a range of cells, after a hipEvent_t handle (int) of another stream."""
        self._chk(self.lib.c2r_global_pass_cells(self.h, float(dt), int(first_cell), int(ncells),
                                                 C.c_void_p(after_event) if after_event else None))

    def global_pass_finish(self):
        conv = C.c_int(0)
        self._chk(self.lib.c2r_global_pass_finish(self.h, C.byref(conv)))
        return int(conv.value)

    def end_step(self):
        self._chk(self.lib.c2r_end_step(self.h))

    def evolve3d(self, dt):
        n = C.c_int(0)
        flags = (C.c_int * 512)()
        self._chk(self.lib.c2r_evolve3d(self.h, float(dt), C.byref(n), flags, 512))
        return n.value, list(flags[: n.value])

    def synchronize(self):
        self._chk(self.lib.c2r_synchronize(self.h))

    def set_batch(self, n):
        self._chk(self.lib.c2r_set_batch(self.h, int(n)))

    def enable_timing(self, on=True):
        self._chk(self.lib.c2r_enable_timing(self.h, int(on)))

    def timing(self, idev=None):
        t = _lib.Timing()
        if idev is None:
            self._chk(self.lib.c2r_get_timing(self.h, C.byref(t)))
        else:
            self._chk(self.lib.c2r_get_timing_device(self.h, int(idev), C.byref(t)))
        return t

    # -- outputs -----------------------------------------------------------------------------
    def download_rates(self):
        n = self.ncell
        phih, phihe, phiheat = np.empty(n), np.empty(2 * n), np.empty(n)
        loss = np.empty(NFREQ)
        nbox = C.c_int(0)
        self._chk(self.lib.c2r_download_rates(self.h, _dp(phih), _dp(phihe), _dp(phiheat), _dp(loss), C.byref(nbox)))
        return dict(phih_grid=phih, phihe_grid=phihe, phiheat=phiheat, photon_loss=loss, sum_nbox=nbox.value)

    def get_loss(self):
        """(photon_loss(1:47), sum_nbox) of this rank's last pass, without touching the grids."""
        loss = np.empty(47)
        nbox = C.c_int(0)
        self._chk(self.lib.c2r_get_loss(self.h, _dp(loss), C.byref(nbox)))
        return loss, int(nbox.value)

    def download_iter_state(self):
        n = self.ncell
        a, b, c, d = np.empty(2 * n), np.empty(3 * n), np.empty(2 * n), np.empty(3 * n)
        self._chk(self.lib.c2r_download_iter_state(self.h, _dp(a), _dp(b), _dp(c), _dp(d)))
        return dict(xh_av=a, xhe_av=b, xh_intermed=c, xhe_intermed=d)

    def upload_rates(self, phih=None, phihe=None, phiheat=None):
        a = [None if x is None else _f64(x).reshape(-1) for x in (phih, phihe, phiheat)]
        self._chk(self.lib.c2r_upload_rates(self.h, *[_dp(x) for x in a]))

    def upload_iter_state(self, xh_av=None, xhe_av=None, xh_intermed=None, xhe_intermed=None):
        a = [None if x is None else _f64(x).reshape(-1) for x in (xh_av, xhe_av, xh_intermed, xhe_intermed)]
        self._chk(self.lib.c2r_upload_iter_state(self.h, *[_dp(x) for x in a]))

    def download_columns(self):
        n = self.ncell
        a, b = np.empty(n), np.empty(2 * n)
        self._chk(self.lib.c2r_download_columns(self.h, _dp(a), _dp(b)))
        return dict(coldensh_out=a, coldenshe_out=b)

    # -- photon statistics (photonstatistics.f90) on the device -----------------------------------
    def state_sums(self, which=0):
        out = np.empty(5)
        self._chk(self.lib.c2r_state_sums(self.h, int(which), _dp(out)))
        return out

    def total_rates(self, dt, reccoef):
        out = np.empty(3)
        rc = _f64(reccoef).reshape(-1)
        self._chk(self.lib.c2r_total_rates(self.h, float(dt), _dp(rc), _dp(out)))
        return out

    def get_reccoef(self):
        out = np.empty(12)
        self._chk(self.lib.c2r_get_reccoef(self.h, _dp(out)))
        return out

    # -- reduction buffer ----------------------------------------------------------------------
    def rates_count(self):
        return int(self.lib.c2r_rates_count(self.h))

    def rates_buffer(self):
        if self._ext_rates is None:
            raise C2RayHipError("no shared reduction buffer: call use_torch_rates_buffer(device) first")
        return self._ext_rates

    def rates_reduced(self):
        return None

    def use_torch_rates_buffer(self, device):
        """Allocate the reduction buffer as a torch tensor so torch.distributed (RCCL) can
        all-reduce it in place; returns the tensor."""
        import torch
        t = torch.zeros(self.rates_count(), dtype=torch.float64, device=device)
        self._chk(self.lib.c2r_set_rates_buffer(self.h, C.c_void_p(t.data_ptr()), t.numel()))
        self._ext_rates = t
        return t


class Evolve:
    """module evolve + evolve_data: evolve_ini() == construction, evolve3D(time, dt, restart).

    comm: optional object with .rank, .size and .allreduce_sum_(buffer) (see parallel.py) -- the
    MPI analogue of the reference: sources are dealt round-robin over the ranks
    (do_grid_static, master_slave.F90:74-96: `do ns1 = 1+rank, NumSrc, npr`), the rate grids are
    summed over ranks (mpi_accumulate_grid_quantities, evolve.F90:505-548) and the global
    chemistry pass is replicated on every rank (evolve.F90:477-484).
    """

    def __init__(self, mesh, tables: RadiationTables | None = None, device=0, engine=None, comm=None):
        self.mesh = tuple(int(m) for m in mesh)
        self.engine = engine if engine is not None else HipEngine(self.mesh, device)
        self.tables = tables if tables is not None else RadiationTables.load()
        self.engine.set_tables(self.tables)
        self.comm = comm
        if comm is not None and comm.size > 1 and isinstance(self.engine, HipEngine):
            self.engine.use_torch_rates_buffer(f"cuda:{device}")
        self.niter = 0
        self.conv_flags: list[int] = []
        self.sum_nbox_all = 0
        self.photon_loss_all = np.zeros(NFREQ)
        self.iteration_dump: dict | None = None

    # subroutine evolve3D (time,dt,restart) -- evolve.F90:78
    def evolve3D(self, time, dt, restart, material: Material, grid: GridProps, sources: SourceProps,
                 cosmology: Cosmology | None = None, dump: dict | None = None, dump_at=None):
        """evolve3D(time,dt,restart) (evolve.F90:78-229).

        restart != 0 continues from an iteration dump (start_from_dump + global_pass, evolve.F90:138-140):
        `dump` is the dictionary an earlier call left in `self.iteration_dump`.  `dump_at` (an iterable of
        iteration numbers) stands for the reference's wall-clock trigger (a dump every 15 minutes, :196-210):
        after pass_all_sources of those iterations the dump content is taken off the device
        (write_iteration_dump, :233-275: niter, photon_loss_all, phih_grid, phihe_grid, [phiheat], xh_av, xhe_av,
        xh_intermed, xhe_intermed)."""
        cosmology = cosmology or Cosmology()
        e = self.engine
        e.set_step(material, grid, cosmology)
        e.set_sources(sources)
        if restart != 0 and dump is not None and dump.get("temperature") is not None and not material.isothermal:
            # not part of the reference's dump file: the temperature slots as the interrupted call left them, so
            # that a non-isothermal continuation is bit-identical (the reference restarts from the last output)
            material.temperature_grid = np.array(dump["temperature"], dtype=np.float32)
        e.upload_state(material)
        single = self.comm is None or self.comm.size == 1
        if restart == 0 and not dump_at and single:
            self.niter, self.conv_flags = e.evolve3d(dt)
        else:
            if restart != 0 and dump is None:
                raise C2RayHipError("evolve3D: restart /= 0 needs the iteration dump to start from")
            self._evolve3d_stepwise(dt, sources.NumSrc, dump if restart != 0 else None, set(dump_at or ()))
        e.download_state(material)
        r = e.download_rates()
        self.sum_nbox_all = r["sum_nbox"]
        self.photon_loss_all = r["photon_loss"]
        return self.niter

    def _evolve3d_stepwise(self, dt, numsrc, dump, dump_at):
        e = self.engine
        comm = self.comm if self.comm is not None and self.comm.size > 1 else None
        ncell = int(np.prod(self.mesh))
        self.conv_flags = []
        if dump is None:
            e.begin_step()
            niter, conv_flag = 0, ncell
        else:  # start_from_dump, then global_pass
            e.upload_rates(dump["phih_grid"], dump["phihe_grid"], dump.get("phiheat"))
            e.upload_iter_state(dump["xh_av"], dump["xhe_av"], dump["xh_intermed"], dump["xhe_intermed"])
            niter = int(dump["niter"])
            conv_flag = e.global_pass(dt)
            self.conv_flags.append(conv_flag)
        conv_criterion = min(int(convergence_fraction * self.mesh[0] * self.mesh[1] * self.mesh[2]), numsrc)
        while True:
            if conv_flag < conv_criterion and niter > 1:
                e.end_step()
                break
            elif niter > 500:
                break
            niter += 1
            e.set_rates_to_zero()
            fused = comm is not None and numsrc > 0 and niter not in dump_at and callable(getattr(comm, "pass_allreduce_chemistry", None))
            if fused:   # pass, sum over ranks and global pass overlapped slab by slab (parallel.py)
                conv_flag = comm.pass_allreduce_chemistry(e, dt)
                self.conv_flags.append(conv_flag)
                continue
            if numsrc > 0:
                if comm is not None:
                    comm.pass_and_allreduce(e)
                else:
                    e.pass_sources(1, 1)
                if niter in dump_at:
                    self.iteration_dump = {"niter": niter, **e.download_rates(), **e.download_iter_state()}
                    if not e.isothermal:
                        tmp = Material(ndens=None, xh=np.empty(2 * ncell), xhe=np.empty(3 * ncell),
                                       temperature_grid=np.empty(3 * ncell, dtype=np.float32), isothermal=False)
                        e.download_state(tmp)
                        self.iteration_dump["temperature"] = tmp.temperature_grid
            conv_flag = e.global_pass(dt)
            self.conv_flags.append(conv_flag)
        self.niter = niter

    # use evolve_data, only: phih_grid, phiheat (output.F90:26)
    @property
    def rates(self):
        return self.engine.download_rates()

    @property
    def iter_state(self):
        return self.engine.download_iter_state()
