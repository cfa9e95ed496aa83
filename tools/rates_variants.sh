#!/usr/bin/env bash
# Run ON THE GPU BOX: rebuild the library with each set of extra hipcc flags given as arguments (one quoted
# string per variant, "" = default build) and print the kernel times of a short bench run.
#   tools/rates_variants.sh "" "-DC2R_RATES_WAVES_ISO=4"
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
  C2R_EXTRA_HIPCC_FLAGS="$flags" python3 -c "
import __graft_entry__ as ge
ge.load_package().build(force=True)" || exit 1
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA_ARGS:-} > gpurun_out/variant.log 2>&1 || { tail -5 gpurun_out/variant.log; exit 1; }
  python3 - "$flags" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/variant.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]!r:60s} ms/step {d['ms_per_step']:.2f}  " + "  ".join(f"{k} {v:.2f}" for k, v in d["kernel_ms_per_step"].items()))
PY
done
# leave the default build behind
python3 -c "
import __graft_entry__ as ge
ge.load_package().build(force=True)"
