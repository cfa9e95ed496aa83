#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/).  Runs the all-reference binaries of the two slow drop-in cases of
tests/test_gpu_dropin.py here, in the dev container (where /root/reference was compiled by oracle/ref_build.sh), and
stores what the tests compare -- SHA-256 of every output file, the iteration history of C2Ray.log, the "min xh_av"
lines, the photon-count files -- as small JSON fixtures under tests/golden/.  The GPU tests then run only the
drop-in binary (reference driver + product modules + libc2ray_hip.so) and compare with the fixture; setting
C2R_DROPIN_RUN_REFERENCE=1 makes them run the reference binary again instead.

    python3 oracle/make_golden_dropin.py
"""
import hashlib
import json
import re
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import refrun  # noqa: E402

GOLD = HERE.parent / "tests" / "golden"

# the same inputs as tests/test_gpu_dropin.py
PL_SOURCES = [(8, 8, 8, 1e55, 3e54, 0.0), (2, 15, 4, 0.0, 2e54, 4e54), (16, 1, 9, 2e54, 0.0, 1e54)]
CUBEP3M_SOURCES = [(8, 8, 8, 3e59, 1e59, 0.0), (2, 15, 4, 0.0, 5e58, 2e59), (16, 1, 9, 2e59, 0.0, 5e58)]


def snapshot(run) -> dict:
    """What the drop-in tests compare, from a finished run directory."""
    run = Path(run)
    res = run / "results"
    log = (res / "C2Ray.log").read_text(errors="replace")
    out = {
        "sha256": {p.name: hashlib.sha256(p.read_bytes()).hexdigest() for p in sorted(res.glob("*.bin"))},
        "log_calls": refrun.parse_log(run),
        "mins": re.findall(r"min xhe?_av:\s+(\S+)", log),
        "photoncounts2": (res / "PhotonCounts2.out").read_text().split() if (res / "PhotonCounts2.out").exists() else None,
        "photoncounts": (res / "PhotonCounts.out").read_text() if (res / "PhotonCounts.out").exists() else None,
        "log_markers": [m for m in ("clumping input from ../coarser_densities/halos_included/9.000n_all.dat",
                                    "density input from ../coarser_densities/halos_removed/9.000n_all.dat",
                                    "(type  5 )", "(type  2 )") if m in log],
    }
    return out


def main():
    r = refrun.run_reference(16, PL_SOURCES, isothermal=False, steps_per_slice=1, which="test", name="golden_dropin_ref_heat_pl", pl=True)
    (GOLD / "dropin_ref_heat_pl.json").write_text(json.dumps(snapshot(r), indent=1))
    r = refrun.run_cubep3m(16, CUBEP3M_SOURCES, which="test", name="golden_dropin_cubep3m_ref")
    (GOLD / "dropin_ref_cubep3m.json").write_text(json.dumps(snapshot(r), indent=1))
    import shutil
    for name in ("golden_dropin_ref_heat_pl", "golden_dropin_cubep3m_ref"):   # run directories: not needed any more
        shutil.rmtree(refrun.REFDIR / name, ignore_errors=True)
    print("written:", GOLD / "dropin_ref_heat_pl.json", GOLD / "dropin_ref_cubep3m.json")


if __name__ == "__main__":
    main()
