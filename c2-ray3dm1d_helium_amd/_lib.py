"""ctypes binding of include/c2ray_hip.h (the C ABI a Fortran host binds with iso_c_binding).

There is no CPU path: if the shared library is missing or no HIP device is present the calls
raise; nothing here falls back to another implementation.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _build

NFREQ, NHEAT, NTAU, NCOOL = 47, 113, 2000, 801

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


class C2RayHipError(RuntimeError):
    pass


class Timing(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("rates_ms", C.c_double), ("chem_ms", C.c_double),
                ("sweep_launches", C.c_int), ("rates_launches", C.c_int), ("chem_launches", C.c_int),
                ("cells_swept", C.c_longlong)]


class IterationReport(C.Structure):
    """c2r_iteration_report (include/c2ray_hip.h)."""
    _fields_ = [("conv_flag", C.c_int), ("sum_nbox", C.c_int), ("photon_loss", C.c_double * NFREQ),
                ("means_intermed", C.c_double * 5), ("sums_intermed", C.c_double * 5), ("total_rates", C.c_double * 3),
                ("minima_av", C.c_double * 2), ("reccoef", C.c_double * 12)]


class SedSetup(C.Structure):
    """struct c2r_sed_setup (include/c2ray_hip.h)."""
    _fields_ = [("nfreq", C.c_int), ("sed", C.c_int), ("freq_min", _dp), ("delta_freq", _dp), ("xsec_index", _dp),
                ("tau", _dp), ("romw", _dp), ("R_star2", C.c_double), ("h_over_kT", C.c_double),
                ("two_pi_over_c_square", C.c_double), ("hplanck", C.c_double), ("pi", C.c_double),
                ("ion_freq_HI", C.c_double), ("ion_freq_HeI", C.c_double), ("ion_freq_HeII", C.c_double),
                ("pl_scaling", C.c_double), ("pl_index", C.c_double)]


# every symbol include/c2ray_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "c2r_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _ip]),
    "c2r_destroy": (None, [C.c_void_p]),
    "c2r_last_error": (C.c_char_p, [C.c_void_p]),
    "c2r_create_error": (C.c_char_p, []),
    "c2r_set_tables": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(_dp), C.c_int]),
    "c2r_set_cooling": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double]),
    "c2r_set_step": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double, C.c_float, C.c_double, C.c_double, C.c_double,
                               C.c_int, C.c_double, _dp]),
    "c2r_arena_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "c2r_set_step_scalars": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_float, C.c_double, C.c_double, C.c_double,
                                       C.c_int, C.c_double, _dp]),
    "c2r_scale_ndens": (C.c_int, [C.c_void_p, C.c_double]),
    "c2r_set_sources": (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, C.c_double]),
    "c2r_set_sed_tables": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int]),
    "c2r_set_sources_sed": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_double]),
    "c2r_build_tables": (C.c_int, [C.c_void_p, C.POINTER(SedSetup), C.c_int]),
    "c2r_download_tables": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]),
    "c2r_set_lls": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _fp]),
    "c2r_set_clumping_grid": (C.c_int, [C.c_void_p, _fp]),
    "c2r_upload_state": (C.c_int, [C.c_void_p, _dp, _dp, _fp]),
    "c2r_download_state": (C.c_int, [C.c_void_p, _dp, _dp, _fp]),
    "c2r_evolve3d": (C.c_int, [C.c_void_p, C.c_double, _ip, _ip, C.c_int]),
    "c2r_begin_step": (C.c_int, [C.c_void_p]),
    "c2r_set_rates_to_zero": (C.c_int, [C.c_void_p]),
    "c2r_pass_sources": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "c2r_pass_sources_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "c2r_pass_slab_count": (C.c_int, [C.c_void_p]),
    "c2r_pass_wait_slab": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "c2r_pass_sources_end": (C.c_int, [C.c_void_p]),
    "c2r_do_source": (C.c_int, [C.c_void_p, C.c_int]),
    "c2r_global_pass": (C.c_int, [C.c_void_p, C.c_double, _ip]),
    "c2r_global_pass_cells": (C.c_int, [C.c_void_p, C.c_double, C.c_size_t, C.c_size_t, C.c_void_p]),
    "c2r_global_pass_finish": (C.c_int, [C.c_void_p, _ip]),
    "c2r_end_step": (C.c_int, [C.c_void_p]),
    "c2r_download_rates": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _ip]),
    "c2r_download_rates_sel": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "c2r_get_loss": (C.c_int, [C.c_void_p, _dp, _ip]),
    "c2r_download_iter_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "c2r_download_columns": (C.c_int, [C.c_void_p, _dp, _dp]),
    "c2r_upload_rates": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "c2r_upload_iter_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "c2r_state_sums": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "c2r_fraction_means": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "c2r_total_rates": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp]),
    "c2r_get_reccoef": (C.c_int, [C.c_void_p, _dp]),
    "c2r_rates_count": (C.c_size_t, [C.c_void_p]),
    "c2r_rates_device_ptr": (C.c_void_p, [C.c_void_p]),
    "c2r_set_rates_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "c2r_synchronize": (C.c_int, [C.c_void_p]),
    "c2r_set_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "c2r_evolve0d_global": (C.c_int, [C.c_void_p, C.c_double, _ip, _ip]),
    "c2r_evolve0d": (C.c_int, [C.c_void_p, _ip, C.c_int, C.c_int, C.c_int, _dp]),
    "c2r_fraction_minima": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "c2r_get_constants": (C.c_int, [_dp, C.c_int]),
    "c2r_device_count": (C.c_int, []),
    "c2r_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _ip, _ip]),
    "c2r_num_devices": (C.c_int, [C.c_void_p]),
    "c2r_comm_available": (C.c_int, []),
    "c2r_comm_unique_id": (C.c_int, [C.c_char_p]),
    "c2r_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "c2r_comm_init_local": (C.c_int, [C.c_void_p]),
    "c2r_comm_destroy": (C.c_int, [C.c_void_p]),
    "c2r_comm_rank": (C.c_int, [C.c_void_p]),
    "c2r_comm_nranks": (C.c_int, [C.c_void_p]),
    "c2r_comm_kind": (C.c_int, [C.c_void_p]),
    "c2r_comm_library": (C.c_int, [C.c_char_p, C.c_int]),
    "c2r_allreduce_rates": (C.c_int, [C.c_void_p]),
    "c2r_pass_allreduce_chemistry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, _ip]),
    "c2r_iteration": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(IterationReport)]),
    "c2r_get_timing": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "c2r_get_timing_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Timing)]),
    "c2r_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
}

_lib = None


def library_path() -> Path:
    return _build.LIB


def load():
    """Load libc2ray_hip.so (building it first if the sources are newer). Raises if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    # C2R_LIB_PATH: another build of this library (an A/B on one GPU box: tools/ab.sh); never a different product
    import os
    alt = os.environ.get("C2R_LIB_PATH")
    path = Path(alt) if alt else _build.build()
    if not path.exists():
        raise C2RayHipError(f"{path} is missing and could not be built: the HIP extension is required")
    lib = C.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
