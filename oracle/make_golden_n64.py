#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: BASELINE configs[1] -- 64^3 uniform density, one point source, with
heating -- run with the reference itself (flang build, tapped at evolve3D), reduced to a compact
fixture: the per-call scalars evolve3D read, the uniform initial state, and for every call the
iteration history, SHA-256 of every output array, and the ionised-fraction line through the source
(the reference's own Ifront diagnostic, files_for_3D/output.F90:192-244).

    python oracle/make_golden_n64.py      (dev container; ~2 minutes)
    python oracle/make_golden_n64.py 128 64,64,64,3e55 10,120,70,1e55     (a larger box, two sources: ~10 minutes)
"""
import hashlib
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(HERE))
import refrun  # noqa: E402

N = 64
SOURCES = [(32, 32, 32, 1e54)]
if len(sys.argv) > 2:
    N = int(sys.argv[1])
    SOURCES = [tuple(float(x) if i == 3 else int(x) for i, x in enumerate(a.split(","))) for a in sys.argv[2:]]
NAME = f"n{N}_heat_{len(SOURCES)}src"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    subprocess.run([str(HERE / "ref_build.sh"), str(N)], check=True)
    run = refrun.run_reference(N, SOURCES, isothermal=False, steps_per_slice=1, name="golden_" + NAME)
    conv = refrun.parse_log(run)
    out = {"ncalls": np.int32(len(conv))}
    for call in range(1, len(conv) + 1):
        tin = refrun.read_records(run / "results" / f"tap_{call:04d}_in.bin")
        tout = refrun.read_records(run / "results" / f"tap_{call:04d}_out.bin")
        p = f"c{call}_"
        for k in ["mesh", "dt", "zred", "H0", "Omega0", "dr", "vol", "srcpos", "NormFlux", "S_star", "isothermal",
                  "temper_val", "clumping", "reccoef"]:
            out[p + k] = tin[k]
        assert np.all(tin["ndens"] == tin["ndens"][0])
        out[p + "ndens_uniform"] = tin["ndens"][0]
        if call == 1:
            for k in ["xh", "xhe", "temperature"]:
                comp = tin[k].reshape(-1, N ** 3)
                assert np.all(comp == comp[:, :1])
                out["c1_" + k + "_uniform"] = comp[:, 0].copy()
        out[p + "conv_flags"] = np.array(conv[call - 1], dtype=np.int32)
        for k in ["xh", "xhe", "temperature", "phih_grid", "phihe_grid", "phiheat", "xh_av", "xhe_av"]:
            out[p + "sha_" + k] = np.array(sha(tout[k]))
        xh1 = tout["xh"][N ** 3:].reshape(N, N, N, order="F")
        j0, k0 = SOURCES[0][1] - 1, SOURCES[0][2] - 1   # the line through the first source
        out[p + "xHII_line"] = xh1[:, j0, k0].copy()
        out[p + "T_line"] = tout["temperature"][:N ** 3].reshape(N, N, N, order="F")[:, j0, k0].copy()
        out[p + "sum_nbox"] = tout["sum_nbox_all"]
        out[p + "reccoef_after"] = tout["reccoef"]
    np.savez_compressed(ROOT / "tests" / "golden" / (NAME + ".npz"), **out)
    shutil.rmtree(run)  # 300 MB of tap dumps: scratch
    print("calls", [len(c) for c in conv], "fixture", (ROOT / "tests/golden" / (NAME + ".npz")).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
