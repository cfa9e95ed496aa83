#!/usr/bin/env python3
"""Registers, scratch (private segment) and LDS of every kernel in the built library, from the gfx950 code object's
metadata notes.   tools/kernel_resources.py [filter-regex] [--lib path]"""
import re
import subprocess
import struct
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def device_code_object(lib: Path) -> bytes:
    blob = lib.read_bytes()
    at = blob.index(b"__CLANG_OFFLOAD_BUNDLE__")
    n, = struct.unpack_from("<Q", blob, at + 24)
    pos = at + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24: pos + 24 + tl].decode()
        pos += 24 + tl
        if "gfx950" in triple:
            return blob[at + off: at + off + size]
    raise SystemExit("no gfx950 code object in " + str(lib))


def resources(lib: Path):
    with tempfile.TemporaryDirectory() as d:
        co = Path(d) / "device.co"
        co.write_bytes(device_code_object(lib))
        notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    out = []
    for t in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", t) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                        lds=g("group_segment_fixed_size"), spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count")))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = ROOT / "c2-ray3dm1d_helium_amd" / "libc2ray_hip.so"
    if "--lib" in sys.argv:
        lib = Path(sys.argv[sys.argv.index("--lib") + 1])
        args = [a for a in args if a != str(lib)]
    pat = re.compile(args[0]) if args else None
    rows = resources(lib)
    names = subprocess.run([FILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':70s} vgpr agpr sgpr scratch   lds spillV spillS")
    for r, nm in zip(rows, names):
        nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
        nm = re.sub(r"\(.*", "", nm)
        if pat and not pat.search(nm):
            continue
        print(f"{nm[:70]:70s} {r['vgpr']:>4} {r['agpr']:>4} {r['sgpr']:>4} {r['scratch']:>7} {r['lds']:>5} {r['spill_v']:>6} {r['spill_s']:>6}")
