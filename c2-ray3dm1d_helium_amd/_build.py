"""Build the HIP shared library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libc2ray_hip.so"
SOURCES = [CSRC / "c2ray_hip.hip"]
HEADERS = [CSRC / "c2ray_device.hpp", CSRC / "c2ray_shell.hpp", CSRC / "c2ray_math.hpp", CSRC / "c2ray_math_tables.hpp", CSRC / "c2ray_comm.inc",
           PKG.parent / "include" / "c2ray_hip.h"]
# -ffp-contract=off: the reference's flang -O2 x86-64 build performs no FMA contraction; fusing
# a*b+c on the GPU changes results in the last bit, which the outer iteration amplifies.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-pthread", "-ldl"]


# the compile-time physics parameters (csrc/c2ray_device.hpp: C2R_PARAM_<NAME>), by the names of the reference's modules
PARAM_NAMES = ("subboxsize", "max_subbox", "epsilon", "convergence_fraction", "minimum_fractional_change",
               "minimum_fraction_of_atoms", "minitemp", "relative_denergy", "abu_he", "abu_c")


def param_flags(params) -> list:
    """-D flags for a dictionary (or the text "name=value,name=value" of C2R_PARAMS) of c2ray_parameters.f90 /
    abundances.f90 values: Fortran literals without kind suffix ("2.5e-4", 10)."""
    if not params:
        return []
    if isinstance(params, str):
        params = dict(item.split("=", 1) for item in params.replace(";", ",").split(",") if item.strip())
    flags = []
    for k, v in params.items():
        k = k.strip().lower()
        if k not in PARAM_NAMES:
            raise ValueError(f"unknown parameter {k!r}: one of {', '.join(PARAM_NAMES)}")
        v = str(v).strip().lower().replace("d", "e")        # a Fortran 1.0d-20 is a C 1.0e-20
        if k in ("subboxsize", "max_subbox"):
            v = str(int(v))
        elif "." not in v and "e" not in v:
            v += ".0"                                         # the f suffix needs a floating literal
        flags.append(f"-DC2R_PARAM_{k.upper()}={v}")
    return flags


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, verbose: bool = False, params=None, out: Path | None = None) -> Path:
    """params (or the environment variable C2R_PARAMS): the library for a host whose c2ray_parameters.f90 / abundances.f90
    differ from the reference's defaults (param_flags); out: another file name for such a build (C2R_LIB_PATH selects it)."""
    params = params if params is not None else os.environ.get("C2R_PARAMS")
    if out is not None:
        cmd = [hipcc(), *HIPCC_FLAGS, *param_flags(params), *os.environ.get("C2R_EXTRA_HIPCC_FLAGS", "").split(), "-o", str(out),
               *map(str, SOURCES)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return Path(out)
    if params:
        force = True       # whatever is there was built for other values (c2r_get_constants says which)
    if not force and not needs_build():
        return LIB
    # One builder at a time (N ranks of a multi-GPU launch import the package together), and the library
    # appears atomically: compile to a private name, then rename.
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():   # another process built it while we waited
                return LIB
            # C2R_EXTRA_HIPCC_FLAGS: extra compiler flags for experiments
            extra = os.environ.get("C2R_EXTRA_HIPCC_FLAGS", "").split() + param_flags(params)
            tmp = LIB.with_name(f"{LIB.name}.tmp{os.getpid()}")
            cc = hipcc()
            # build-time requirement besides hipcc itself: the RCCL header <rccl/rccl.h> (csrc/c2ray_comm.inc binds librccl
            # through its declarations; the library is loaded at run time, on first use of a communicator).  Whether the
            # compiler finds it is the compiler's business (ROCM_PATH, CPATH, versioned installs, wrappers): c2ray_comm.inc
            # carries an #error with __has_include whose text says what is missing, and that text is what is raised below.
            cmd = [cc, *HIPCC_FLAGS, *extra, "-o", str(tmp), *map(str, SOURCES)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                tmp.unlink(missing_ok=True)
                hint = ""
                if "rccl/rccl.h" in r.stderr:
                    hint = ("\n(the RCCL development header is needed to BUILD libc2ray_hip even for single-GPU use: install ROCm's "
                            "rccl-dev files or point the compiler at them, e.g. C2R_EXTRA_HIPCC_FLAGS=-I/path/to/rocm/include)")
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr + hint)
            os.replace(tmp, LIB)
            if verbose:
                print(" ".join(cmd))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB
