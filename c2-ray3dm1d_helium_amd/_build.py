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


def build(force: bool = False, verbose: bool = False) -> Path:
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
            extra = os.environ.get("C2R_EXTRA_HIPCC_FLAGS", "").split()
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
