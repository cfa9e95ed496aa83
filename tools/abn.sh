#!/bin/bash
# Several builds of the library on ONE GPU box, round-robin, twice:
#   tools/abn.sh <out-dir> <variant> [<variant> ...] -- <command that prints one JSON line>
# <variant> = the <name> of tools/build_variant.sh, or "default" for the product build.
out=$1; shift
libs=()
while [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
mkdir -p "$out"
for round in 1 2; do
  for v in "${libs[@]}"; do
    lib=c2-ray3dm1d_helium_amd/libc2ray_hip_$v.so; [ "$v" = default ] && lib=c2-ray3dm1d_helium_amd/libc2ray_hip.so
    C2R_LIB_PATH=$PWD/$lib timeout -k 10 900 "$@" > "$out/${v}_${round}.json" 2> "$out/${v}_${round}.err" || { echo "$v $round failed"; tail -3 "$out/${v}_${round}.err"; }
    python3 - "$out/${v}_${round}.json" "$v/$round" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as ex:
    print(sys.argv[2], "no result", ex); sys.exit(0)
if "iterations" in d:
    it = d["iterations"]
    print(f"{sys.argv[2]:24s} rates ms/iter:", " ".join(f"{h['rates_kernel_ms']:.1f}" for h in it), "| sweep:", " ".join(f"{h['sweep_kernel_ms']:.1f}" for h in it[1:]),
          "| chem:", " ".join(f"{h['chem_ms']:.1f}" for h in it))
else:
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:.2f}", {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
PY
  done
done
