#!/bin/bash
# chemistry time per iteration of the drop-in's evolve3D calls, for values of an environment switch, on ONE box
#   tools/chem_ab.sh <out-dir> VAR v1 v2 ...
out=$1; var=$2; shift 2
mkdir -p "$out"
for round in 1 2; do
  for v in "$@"; do
    env $var=$v python tools/time_dropin.py --out "$out/${var}_${v}_$round.json" > "$out/${var}_${v}_$round.log" 2>&1
    python3 - "$out/${var}_${v}_$round.json" "$var=$v/$round" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for c in j["evolve3D_calls"]:
    k = c.get("kernel_ms_mean", {})
    print(f"{sys.argv[2]:22s} its {c['iterations']:3d} loop {c['ms_per_iteration']:.2f} ms/it  chem mean {k.get('chemistry', 0):.2f}  by iteration:",
          " ".join(f"{x:.2f}" for x in c.get("chemistry_ms_by_iteration", [])[:9]))
PY
  done
done
