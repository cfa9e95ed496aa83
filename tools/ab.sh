#!/bin/bash
# A/B of two builds of the library on ONE GPU box (boxes differ by several per cent: only same-box numbers compare).
#   tools/ab.sh <out-dir> <lib A> <lib B> -- <command that prints one JSON line>   (A, B, A, B)
out=$1; A=$2; B=$3; shift 4
mkdir -p "$out"
for round in 1 2; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    C2R_LIB_PATH=$PWD/$lib timeout -k 10 900 "$@" > "$out/${v}${round}.json" 2> "$out/${v}${round}.err" || { echo "$v$round failed"; tail -3 "$out/${v}${round}.err"; }
    python3 - "$out/${v}${round}.json" "$v$round $lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as ex:
    print(sys.argv[2], "no result", ex); sys.exit(0)
if "iterations" in d:
    it = d["iterations"]
    print(sys.argv[2], "rates ms/iter:", " ".join(f"{h['rates_kernel_ms']:.1f}" for h in it), "| sweep:", " ".join(f"{h['sweep_kernel_ms']:.1f}" for h in it),
          "| chem:", " ".join(f"{h['chem_ms']:.1f}" for h in it))
else:
    print(sys.argv[2], f"ms/step {d['ms_per_step']:.2f}", d["kernel_ms_per_step"])
PY
  done
done
