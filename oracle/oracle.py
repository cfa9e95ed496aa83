"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/liboracle.so (the plain-C
restatement of the reference's hot path, oracle/c2ray_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
NFREQ, NHEAT, NTAU, NCOOL = 47, 113, 2000, 801

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


class PhotRates(C.Structure):
    _names = ["photo_cell_HI", "photo_cell_HeI", "photo_cell_HeII", "heat_cell_HI", "heat_cell_HeI",
              "heat_cell_HeII", "photo_in_HI", "photo_in_HeI", "photo_in_HeII", "heat_in_HI",
              "heat_in_HeI", "heat_in_HeII", "photo_out_HI", "photo_out_HeI", "photo_out_HeII",
              "heat_out_HI", "heat_out_HeI", "heat_out_HeII", "heat", "photo_in", "photo_out"]
    _fields_ = [(n, C.c_double) for n in _names]

    def as_array(self):
        return np.array([getattr(self, n) for n in self._names])


class IonStates(C.Structure):
    _fields_ = [("h", C.c_double * 2), ("he", C.c_double * 3), ("h_av", C.c_double * 2),
                ("he_av", C.c_double * 3), ("h_old", C.c_double * 2), ("he_old", C.c_double * 3)]

    @classmethod
    def from_array(cls, a):
        s = cls()
        C.memmove(C.byref(s), np.ascontiguousarray(a, dtype=np.float64).ctypes.data, 15 * 8)
        return s

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float64).copy()


class RecCoef(C.Structure):
    _names = ["arech0", "brech0", "areche0", "breche0", "oreche0", "areche1", "breche1", "treche1",
              "colli_HI", "colli_HeI", "colli_HeII", "v"]
    _fields_ = [(n, C.c_double) for n in _names]

    @classmethod
    def from_array(cls, a):
        s = cls()
        for n, x in zip(cls._names, a):
            setattr(s, n, float(x))
        return s

    def as_array(self):
        return np.array([getattr(self, n) for n in self._names])


class CTables(C.Structure):
    _fields_ = [("photo_thick", _dp), ("photo_thin", _dp), ("heat_thick", _dp), ("heat_thin", _dp),
                ("sigma_HI", _dp), ("sigma_HeI", _dp), ("sigma_HeII", _dp),
                ("f1ion_HI", _dp), ("f1ion_HeI", _dp), ("f1ion_HeII", _dp),
                ("f2ion_HI", _dp), ("f2ion_HeI", _dp), ("f2ion_HeII", _dp),
                ("f1heat_HI", _dp), ("f1heat_HeI", _dp), ("f1heat_HeII", _dp),
                ("f2heat_HI", _dp), ("f2heat_HeI", _dp), ("f2heat_HeII", _dp),
                ("bb_upper", C.c_int),
                ("pl_photo_thick", _dp), ("pl_photo_thin", _dp), ("pl_heat_thick", _dp), ("pl_heat_thin", _dp),
                ("qpl_photo_thick", _dp), ("qpl_photo_thin", _dp), ("qpl_heat_thick", _dp), ("qpl_heat_thin", _dp),
                ("pl_lower", C.c_int), ("pl_upper", C.c_int), ("qpl_lower", C.c_int), ("qpl_upper", C.c_int),
                ("cool", _dp), ("cool_mintemp", C.c_double), ("cool_dtemp", C.c_double)]


class CStep(C.Structure):
    _fields_ = [("mesh", C.c_int * 3), ("dr", C.c_double * 3), ("vol", C.c_double),
                ("zred", C.c_double), ("H0", C.c_double), ("Omega0", C.c_double),
                ("isothermal", C.c_int), ("temper_val", C.c_double), ("clumping", C.c_float),
                ("nsrc", C.c_int), ("srcpos", _ip), ("normflux", _dp), ("s_star", C.c_double),
                ("normflux_pl", _dp), ("normflux_qpl", _dp), ("pl_s_star", C.c_double), ("qpl_s_star", C.c_double),
                ("ndens", _dp), ("rc", RecCoef),
                ("use_lls", C.c_int), ("coldensh_lls", C.c_double), ("lls_grid", _fp), ("clumping_grid", _fp)]


class CState(C.Structure):
    _fields_ = [("xh", _dp), ("xhe", _dp), ("temperature", _fp),
                ("phih", _dp), ("phihe", _dp), ("phiheat", _dp),
                ("xh_av", _dp), ("xhe_av", _dp), ("xh_intermed", _dp), ("xhe_intermed", _dp),
                ("coldensh_out", _dp), ("coldenshe_out", _dp),
                ("photon_loss", C.c_double * NFREQ), ("sum_nbox", C.c_int), ("niter", C.c_int),
                ("conv_flags", C.c_int * 512)]


_lib = None


def build(force=False):
    so = HERE / "liboracle.so"
    src = HERE / "c2ray_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "liboracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _lib
    if _lib is None:
        # ORC_LIB_PATH: a build of the oracle with other compile-time parameters (oracle/Makefile: liboracle_params.so),
        # the checker of a product library built with the same ones (tests/test_gpu_parity.py)
        import os
        alt = os.environ.get("ORC_LIB_PATH")
        _lib = C.CDLL(alt if alt else str(build()))
        _lib.orc_coolin.restype = C.c_double
        _lib.orc_electrondens.restype = C.c_double
    return _lib


def _p(a, t=_dp):
    return None if a is None else a.ctypes.data_as(t)


SED_TABLE_KEYS = ["pl_photo_thick", "pl_photo_thin", "pl_heat_thick", "pl_heat_thin",
                  "qpl_photo_thick", "qpl_photo_thin", "qpl_heat_thick", "qpl_heat_thin"]
TABLE_KEYS = ["photo_thick", "photo_thin", "heat_thick", "heat_thin", "sigma_HI", "sigma_HeI", "sigma_HeII",
              "f1ion_HI", "f1ion_HeI", "f1ion_HeII", "f2ion_HI", "f2ion_HeI", "f2ion_HeII",
              "f1heat_HI", "f1heat_HeI", "f1heat_HeII", "f2heat_HI", "f2heat_HeI", "f2heat_HeII"]


class Tables:
    """Radiation + cooling tables as rad_ini / setup_cool leave them (a dict of numpy arrays)."""

    def __init__(self, d):
        self.a = {k: np.ascontiguousarray(d[k], dtype=np.float64) for k in TABLE_KEYS + SED_TABLE_KEYS if k in d}
        self.bb_upper = int(d["bb_upper"])
        self.cool = np.ascontiguousarray(d["cool"], dtype=np.float64) if "cool" in d else None
        self.cool_mintemp = float(d["cool_mintemp"]) if "cool_mintemp" in d else 1.0
        self.cool_dtemp = float(d["cool_dtemp"]) if "cool_dtemp" in d else 0.01
        c = CTables()
        for k in TABLE_KEYS:
            setattr(c, k, _p(self.a.get(k)))
        c.bb_upper = self.bb_upper
        for k in SED_TABLE_KEYS:
            setattr(c, k, _p(self.a.get(k)))
        if "pl_limits" in d:
            c.pl_lower, c.pl_upper = int(d["pl_limits"][0]), int(d["pl_limits"][1])
        if "qpl_limits" in d:
            c.qpl_lower, c.qpl_upper = int(d["qpl_limits"][0]), int(d["qpl_limits"][1])
        c.cool = _p(self.cool)
        c.cool_mintemp = self.cool_mintemp
        c.cool_dtemp = self.cool_dtemp
        self.c = c

    @classmethod
    def load(cls, path):
        with np.load(path) as z:
            return cls({k: z[k] for k in z.files})


class Step:
    """Host-side inputs of one evolve3D call (SURVEY.md section 8b 'Host data read')."""

    def __init__(self, mesh, dr, vol, zred, H0, Omega0, isothermal, temper_val, clumping, srcpos,
                 normflux, s_star, ndens, reccoef, normflux_pl=None, normflux_qpl=None, pl_s_star=1.0,
                 qpl_s_star=1.0, coldensh_lls=None, lls_grid=None, clumping_grid=None):
        self.srcpos = np.ascontiguousarray(srcpos, dtype=np.int32).reshape(-1)
        self.normflux = np.ascontiguousarray(normflux, dtype=np.float64).reshape(-1)
        self.ndens = np.ascontiguousarray(ndens, dtype=np.float64).reshape(-1)
        c = CStep()
        c.mesh[:] = [int(m) for m in mesh]
        c.dr[:] = [float(x) for x in dr]
        c.vol = float(vol)
        c.zred, c.H0, c.Omega0 = float(zred), float(H0), float(Omega0)
        c.isothermal = int(isothermal)
        c.temper_val = float(temper_val)
        c.clumping = float(clumping)
        c.nsrc = self.normflux.size
        c.srcpos = _p(self.srcpos, _ip)
        c.normflux = _p(self.normflux)
        c.s_star = float(s_star)
        self.normflux_pl = None if normflux_pl is None else np.ascontiguousarray(normflux_pl, dtype=np.float64).reshape(-1)
        self.normflux_qpl = None if normflux_qpl is None else np.ascontiguousarray(normflux_qpl, dtype=np.float64).reshape(-1)
        c.normflux_pl, c.normflux_qpl = _p(self.normflux_pl), _p(self.normflux_qpl)
        c.pl_s_star, c.qpl_s_star = float(pl_s_star), float(qpl_s_star)
        c.ndens = _p(self.ndens)
        c.rc = RecCoef.from_array(reccoef)
        self.lls_grid = None if lls_grid is None else np.ascontiguousarray(lls_grid, dtype=np.float32).reshape(-1)
        self.clumping_grid = None if clumping_grid is None else np.ascontiguousarray(clumping_grid, dtype=np.float32).reshape(-1)
        c.use_lls = int(coldensh_lls is not None or lls_grid is not None)
        c.coldensh_lls = 0.0 if coldensh_lls is None else float(coldensh_lls)
        c.lls_grid, c.clumping_grid = _p(self.lls_grid, _fp), _p(self.clumping_grid, _fp)
        self.c = c
        self.ncell = int(np.prod(mesh))
        self.isothermal = bool(isothermal)

    @classmethod
    def from_tap(cls, t):
        return cls(t["mesh"], t["dr"], t["vol"][0], t["zred"][0], t["H0"][0], t["Omega0"][0],
                   t["isothermal"][0], t["temper_val"][0], t["clumping"][0], t["srcpos"], t["NormFlux"],
                   t["S_star"][0], t["ndens"], t["reccoef"], t.get("NormFluxPL"), t.get("NormFluxQPL"),
                   t["pl_S_star"][0] if "pl_S_star" in t else 1.0, t["qpl_S_star"][0] if "qpl_S_star" in t else 1.0,
                   coldensh_lls=t["coldensh_LLS"][0] if "coldensh_LLS" in t else None)


class State:
    def __init__(self, step: Step, xh, xhe, temperature=None):
        n = step.ncell
        self.xh = np.array(xh, dtype=np.float64).reshape(-1).copy()
        self.xhe = np.array(xhe, dtype=np.float64).reshape(-1).copy()
        self.temperature = None if temperature is None else np.array(temperature, dtype=np.float32).reshape(-1).copy()
        z = lambda m: np.zeros(m * n, dtype=np.float64)
        self.phih, self.phihe, self.phiheat = z(1), z(2), z(1)
        self.xh_av, self.xhe_av, self.xh_intermed, self.xhe_intermed = z(2), z(3), z(2), z(3)
        self.coldensh_out, self.coldenshe_out = z(1), z(2)
        c = CState()
        for k in ["xh", "xhe", "phih", "phihe", "phiheat", "xh_av", "xhe_av", "xh_intermed", "xhe_intermed",
                  "coldensh_out", "coldenshe_out"]:
            setattr(c, k, _p(getattr(self, k)))
        c.temperature = _p(self.temperature, _fp)
        self.c = c

    @property
    def photon_loss(self):
        return np.array(self.c.photon_loss[:])

    @property
    def conv_flags(self):
        return list(self.c.conv_flags[: self.c.niter])


def evolve3d(tables: Tables, step: Step, state: State, dt, max_iter=0):
    return lib().orc_evolve3d(C.byref(tables.c), C.byref(step.c), C.byref(state.c), C.c_double(dt),
                              C.c_int(max_iter))


def begin_step(state: State):
    """evolve.F90:131-134: *_av = *_intermed = current state."""
    state.xh_av[:] = state.xh
    state.xh_intermed[:] = state.xh
    state.xhe_av[:] = state.xhe
    state.xhe_intermed[:] = state.xhe


def pass_all_sources(tables, step, state):
    lib().orc_pass_all_sources(C.byref(tables.c), C.byref(step.c), C.byref(state.c))


def pass_all_sources_shells(tables, step, state, nthreads=1):
    """pass_all_sources in L-infinity shell order, the cells of a shell over `nthreads` OpenMP threads."""
    l = lib()
    l.orc_pass_all_sources_shells.restype = None
    l.orc_pass_all_sources_shells.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    l.orc_pass_all_sources_shells(C.byref(tables.c), C.byref(step.c), C.byref(state.c), int(nthreads))


def global_pass(tables, step, state, dt):
    return lib().orc_global_pass(C.byref(tables.c), C.byref(step.c), C.byref(state.c), C.c_double(dt))


def global_pass_threads(tables, step, state, dt, nthreads=1):
    l = lib()
    l.orc_global_pass_threads.restype = C.c_int
    l.orc_global_pass_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int]
    return int(l.orc_global_pass_threads(C.byref(tables.c), C.byref(step.c), C.byref(state.c), float(dt), int(nthreads)))


def do_source(tables, step, state, ns):
    loss = C.c_double(0)
    nbox = lib().orc_do_source(C.byref(tables.c), C.byref(step.c), C.byref(state.c), C.c_int(ns), C.byref(loss))
    return nbox, loss.value


def do_source_accumulate(tables, step, state, ns):
    """do_source + the bookkeeping of evolve_source.F90:233-236; the caller sums loss and nbox."""
    return do_source(tables, step, state, ns)


def ini_rec_colion_factors(T):
    rc = RecCoef()
    lib().orc_ini_rec_colion_factors(C.c_double(T), C.byref(rc))
    return rc.as_array()


def photoion_rates(tables, cin, vol, normflux, i_state, isothermal):
    out = PhotRates()
    lib().orc_photoion_rates(C.byref(tables.c), *[C.c_double(x) for x in cin], C.c_double(vol),
                             C.c_double(normflux), C.c_double(i_state), C.c_int(int(isothermal)), C.byref(out))
    return out.as_array()


def prepare_doric_factors(NH, NHe):
    o = [C.c_double() for _ in range(4)]
    a = (C.c_double * 2)(*NHe)
    lib().orc_prepare_doric_factors(C.c_double(NH), a, *[C.byref(x) for x in o])
    return [x.value for x in o]


def doric(dt, de, nd, ion15, phi3, fr4, rc12, clumping):
    ion = IonStates.from_array(ion15)
    phi = PhotRates()
    phi.photo_cell_HI, phi.photo_cell_HeI, phi.photo_cell_HeII = [float(x) for x in phi3]
    rc = RecCoef.from_array(rc12)
    lib().orc_doric(C.c_double(dt), C.c_double(de), C.c_double(nd), C.byref(ion), C.byref(phi),
                    *[C.c_double(x) for x in fr4], C.byref(rc), C.c_float(clumping))
    return ion.as_array()


def thermal(tables, dt, tend, tavg, de, nd, ion15, heat, zred, H0, Omega0):
    ion = IonStates.from_array(ion15)
    phi = PhotRates()
    phi.heat = float(heat)
    te, ta = C.c_double(tend), C.c_double(tavg)
    lib().orc_thermal(C.byref(tables.c), C.c_double(dt), C.byref(te), C.byref(ta), C.c_double(de),
                      C.c_double(nd), C.byref(ion), C.byref(phi), C.c_double(zred), C.c_double(H0),
                      C.c_double(Omega0))
    return te.value, ta.value


def constants():
    buf = (C.c_double * 64)()
    n = lib().orc_constants(buf, 64)
    return np.array(buf[:n])


class CSedSetup(C.Structure):
    _fields_ = [("nfreq", C.c_int), ("sed", C.c_int), ("freq_min", _dp), ("delta_freq", _dp), ("xsec_index", _dp),
                ("tau", _dp), ("romw", _dp), ("R_star2", C.c_double), ("h_over_kT", C.c_double),
                ("two_pi_over_c_square", C.c_double), ("hplanck", C.c_double), ("pi", C.c_double),
                ("ion_freq_HI", C.c_double), ("ion_freq_HeI", C.c_double), ("ion_freq_HeII", C.c_double),
                ("pl_scaling", C.c_double), ("pl_index", C.c_double)]


def xsec_index(d):
    """The index spec_integration passes per band: HI's for band 1, HeI's for 2..27, HeII's for 28..47
    (radiation_tables.f90:278,315,349)."""
    b = np.arange(1, 48)
    return np.where(b <= 1, d["pl_index_HI"], np.where(b <= 27, d["pl_index_HeI"], d["pl_index_HeII"])).astype(np.float64)


def build_tables(d, sed=0, heat=True):
    """spec_integration for one SED from a set-up dictionary (keys as dumped by the tap: freq_min,
    delta_freq, pl_index_*, tau, romw9, sed_setup, consts, [pl_setup | qpl_setup])."""
    lib_ = lib()
    lib_.orc_build_tables.restype = None
    lib_.orc_build_tables.argtypes = [C.POINTER(CSedSetup), C.c_int, _dp, _dp, _dp, _dp]
    keep = [np.ascontiguousarray(d[k], dtype=np.float64) for k in ("freq_min", "delta_freq", "tau", "romw9")]
    keep.append(np.ascontiguousarray(xsec_index(d)))
    s = CSedSetup()
    s.nfreq, s.sed = 512, int(sed)
    s.freq_min, s.delta_freq, s.tau, s.romw, s.xsec_index = (_p(a) for a in keep)
    s.R_star2, s.h_over_kT, s.two_pi_over_c_square = (float(x) for x in d["sed_setup"])
    c = d["consts"]
    s.pi, s.hplanck = float(c[0]), float(c[5])
    s.ion_freq_HI, s.ion_freq_HeI, s.ion_freq_HeII = float(c[22]), float(c[23]), float(c[24])
    if sed:
        s.pl_scaling, s.pl_index = (float(x) for x in d["pl_setup" if sed == 1 else "qpl_setup"])
    nt = NTAU + 1
    out = [np.zeros(47 * nt), np.zeros(47 * nt), np.zeros(113 * nt), np.zeros(113 * nt)]
    lib_.orc_build_tables(C.byref(s), int(bool(heat)), *[_p(a) for a in out])
    return dict(zip(("photo_thick", "photo_thin", "heat_thick", "heat_thin"), out))
