#!/usr/bin/env bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of the kernels of a short bench run with the library as built
# (rocprofv3 --pmc pass with --kernel-trace only), printed per kernel.   tools/rates_fetch.sh [label]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fetch_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$ctr" -o bench -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$ctr.log" 2>&1 || { tail -5 "$OUT/$ctr.log"; exit 1; }
done
python3 - "$OUT" "${1:-x}" <<'PY'
import collections, csv, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{sys.argv[1]}/{ctr}/bench_counter_collection.csv")):
        if r["Counter_Name"] == ctr:
            m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if m:
                e = tot[m.group(1)][ctr]
                e[0] += float(r["Counter_Value"]); e[1] += 1
for k, v in tot.items():
    print(sys.argv[2], k, "  ".join(f"{c} {x[0] / x[1]:.0f} x{x[1]}" for c, x in v.items()))
PY
