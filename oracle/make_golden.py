#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: regenerate tests/golden/ and the package's table data
by RUNNING THE REFERENCE ITSELF (flang build of /root/reference, see
oracle/ref_build.sh; the evolve3D boundary is tapped by oracle/probe/evolve_tap.f90).

Run in the dev container (needs /root/reference and flang):
    python oracle/make_golden.py

Outputs (all data, no reference source text):
  c2-ray3dm1d_helium_amd/data/rad_tables_bb5e4.npz
                                     radiation tables + per-band vectors as rad_ini leaves them
                                     (radiation_tables.f90:141-168), cooling curves as setup_cool
                                     leaves them (cooling_h.f90:76-171) -- inputs of the hot path
  tests/golden/consts.npz            module constants as evaluated by the reference build
  tests/golden/funcvec.npz           input/output vectors of ini_rec_colion_factors,
                                     photoion_rates, doric, thermal
  tests/golden/tap_<case>.npz        every array evolve3D read / wrote, per call, + the
                                     per-iteration non-converged counts from C2Ray.log
"""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(HERE))
import refrun  # noqa: E402

GOLD = ROOT / "tests" / "golden"
PKGDATA = ROOT / "c2-ray3dm1d_helium_amd" / "data"
REF_TABLES = Path("/root/reference/tables")

COOL_FILES = ["H0-cool", "H1-cool-B", "He0-cool_new", "He1-cool_new_nocollion", "He2-cool"]

# name, mesh, sources (i,j,k,photons/s), isothermal, calls kept
CASES = [
    ("N16_iso_1src", 16, [(8, 8, 8, 1e55)], True, [1, 2]),
    ("N16_heat_3src", 16, [(8, 8, 8, 1e55), (2, 15, 4, 3e54), (16, 1, 9, 2e54)], False, [1, 2]),
    # (N/2-1) mod 10 == 0: the far "left" layer is never traced (evolve_source.F90:136-139)
    ("N22_iso_2src", 22, [(11, 11, 11, 3e55), (3, 20, 7, 1e55)], True, [1]),
]


def cooling_tables():
    """setup_cool (cooling_h.f90:76-171): 801 rows 'log10T log10Lambda', stored as 10**value."""
    cols, temp = [], None
    for n in COOL_FILES:
        a = np.loadtxt(REF_TABLES / f"{n}.tab", skiprows=1)
        assert a.shape == (801, 2)
        temp = a[:, 0] if temp is None else temp
        cols.append(np.array([10.0 ** float(x) for x in a[:, 1]]))
    return np.concatenate(cols), float(temp[0]), float(temp[1]) - float(temp[0])


SED_SETUP_KEYS = ["freq_min", "delta_freq", "pl_index_HI", "pl_index_HeI", "pl_index_HeII", "tau", "romw9", "sed_setup",
                  "pl_setup", "qpl_setup", "consts", "pl_limits", "qpl_limits"]
KEEP_OUT = ["xh", "xhe", "temperature", "phih_grid", "phihe_grid", "phiheat", "xh_av", "xhe_av", "photon_loss_all",
            "sum_nbox_all", "reccoef", "coldensh_out", "coldenshe_out"]


def lls_case():
    """use_LLS = .true. build (type_of_LLS = 1: the same Lyman-limit-system column in every cell,
    evolve_point.F90:177-180), heating on, two sources; calls 1 and 2."""
    subprocess.run([str(HERE / "ref_build.sh"), "16", "lls"], check=True)
    srcs = [(8, 8, 8, 1e55), (2, 15, 4, 3e54)]
    run = refrun.run_reference(16, srcs, isothermal=False, steps_per_slice=1, lls=True, name="golden_N16_lls_heat_2src")
    conv = refrun.parse_log(run)
    out = {"conv_flags_per_call": np.array([len(c) for c in conv], dtype=np.int32)}
    for call in (1, 2):
        tin = refrun.read_records(run / "results" / f"tap_{call:04d}_in.bin")
        tout = refrun.read_records(run / "results" / f"tap_{call:04d}_out.bin")
        for k, v in tin.items():
            out[f"c{call}_in_{k}"] = v
        for k in KEEP_OUT:
            out[f"c{call}_out_{k}"] = tout[k]
        out[f"c{call}_conv_flags"] = np.array(conv[call - 1], dtype=np.int32)
    np.savez_compressed(GOLD / "tap_N16_lls_heat_2src.npz", **out)
    print("wrote tap_N16_lls_heat_2src.npz", [len(c) for c in conv])
    shutil.rmtree(run)


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    if sys.argv[1:] == ["lls"]:
        return lls_case()
    lls_case()
    for mesh in sorted({c[1] for c in CASES}):
        subprocess.run([str(HERE / "ref_build.sh"), str(mesh)], check=True)

    first = True
    for name, mesh, sources, iso, keep in CASES:
        run = refrun.run_reference(mesh, sources, isothermal=iso, steps_per_slice=1, name="golden_" + name)
        res = run / "results"
        conv = refrun.parse_log(run)
        out = {"conv_flags_per_call": np.array([len(c) for c in conv], dtype=np.int32)}
        for call in keep:
            tin = refrun.read_records(res / f"tap_{call:04d}_in.bin")
            tout = refrun.read_records(res / f"tap_{call:04d}_out.bin")
            for k, v in tin.items():
                out[f"c{call}_in_{k}"] = v
            for k, v in tout.items():
                out[f"c{call}_out_{k}"] = v
            out[f"c{call}_conv_flags"] = np.array(conv[call - 1], dtype=np.int32)
        np.savez_compressed(GOLD / f"tap_{name}.npz", **out)
        print("wrote", f"tap_{name}.npz", [len(c) for c in conv])

        if first:
            first = False
            tb = refrun.read_records(res / "tables.bin")
            cool, mint, dtemp = cooling_tables()
            d = {k: tb[k] for k in tb if k not in ("consts", "ints", "coolin_probe")}
            d["bb_upper"] = np.int32(tb["ints"][3])
            d["cool"] = cool
            d["cool_mintemp"] = np.float64(mint)
            d["cool_dtemp"] = np.float64(dtemp)
            np.savez_compressed(PKGDATA / "rad_tables_bb5e4.npz", **d)
            np.savez_compressed(GOLD / "consts.npz", consts=tb["consts"], ints=tb["ints"])
        if not iso:
            fv = refrun.read_records(res / "funcvec.bin")
            np.savez_compressed(GOLD / "funcvec.npz", **fv)
            tb = refrun.read_records(res / "tables.bin")
            np.savez_compressed(GOLD / "coolin_probe.npz", coolin_probe=tb["coolin_probe"])
    # -DPL -DQUASARS build: three sources with black-body, power-law and quasar-like components;
    # the hard photons keep ~60 cells flickering, so every evolve3D call runs into the 500-iteration
    # cap (evolve.F90:177-181) -- 501 outer iterations to reproduce
    subprocess.run([str(HERE / "ref_build.sh"), "16", "pl"], check=True)
    srcs = [(8, 8, 8, 1e55, 3e54, 0.0), (2, 15, 4, 0.0, 2e54, 4e54), (16, 1, 9, 2e54, 0.0, 1e54)]
    run = refrun.run_reference(16, srcs, isothermal=False, steps_per_slice=1, pl=True, name="golden_N16_pl_heat_3src")
    conv = refrun.parse_log(run)
    tin = refrun.read_records(run / "results" / "tap_0001_in.bin")
    tout = refrun.read_records(run / "results" / "tap_0001_out.bin")
    out = {"conv_flags_per_call": np.array([len(c) for c in conv], dtype=np.int32)}
    for k, v in tin.items():
        out["c1_in_" + k] = v
    for k in ["xh", "xhe", "temperature", "phih_grid", "phihe_grid", "phiheat", "xh_av", "xhe_av", "photon_loss_all",
              "sum_nbox_all", "reccoef"]:
        out["c1_out_" + k] = tout[k]
    out["c1_conv_flags"] = np.array(conv[0], dtype=np.int32)
    np.savez_compressed(GOLD / "tap_N16_pl_heat_3src.npz", **out)
    tb = refrun.read_records(run / "results" / "tables.bin")
    sed = {}
    for pre in ("pl_", "qpl_"):
        lo, hi = tb[pre + "limits"]
        sed[pre + "limits"] = tb[pre + "limits"]
        for kind, ncol in (("photo", 47), ("heat", 113)):
            for tt in ("thick", "thin"):
                a = tb[f"{pre}{kind}_{tt}"].reshape(ncol, 2001).copy()
                # columns outside the SED's band range are never read: zero them so the file stays small
                used = np.zeros(ncol, dtype=bool)
                for b in range(lo, hi + 1):
                    cols = [b] if kind == "photo" else ([1] if b == 1 else ([2 * b - 2, 2 * b - 1] if b <= 27 else
                                                                            [3 * b - 30, 3 * b - 29, 3 * b - 28]))
                    for c in cols:
                        used[c - 1] = True
                a[~used] = 0.0
                sed[f"{pre}{kind}_{tt}"] = a.reshape(-1)
    np.savez_compressed(GOLD / "rad_tables_pl_qpl.npz", **sed)
    # what spec_integration starts from (band set-up, Romberg weights, normalised SEDs): the inputs of the
    # table builders (orc_build_tables, c2r_build_tables)
    np.savez_compressed(GOLD / "sed_setup.npz", **{k: tb[k] for k in SED_SETUP_KEYS})
    for d in refrun.REFDIR.glob("golden_*"):
        shutil.rmtree(d)  # run directories are scratch
    for p in sorted(list(GOLD.glob("*.npz")) + list(PKGDATA.glob("*.npz"))):
        print(f"{p.name:32s} {p.stat().st_size/1024:8.1f} KiB")


if __name__ == "__main__":
    main()
