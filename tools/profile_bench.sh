#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics and PMC passes of the default bench.py
# command, written under gpurun_out/prof_bench/.  Counters go in their own passes with --kernel-trace only
# (MI355X_MICROARCH.md).  Afterwards, in the dev container:
#   cp gpurun_out/prof_bench/stats/*kernel_stats.csv profiles/rNN_bench_kernel_stats.csv
#   cp gpurun_out/prof_bench/pmc_summary.json        profiles/rNN_bench_pmc_summary.json
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${PROF_NAME:-prof_bench}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline ${BENCH_EXTRA_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
i=0
# one small counter set per pass (FETCH_SIZE and WRITE_SIZE together already exceed what one pass can collect)
# (the FP64 instruction mix -- SURVEY 8d's "FP64 %" -- needs two passes: four SQ counters do not fit one)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  echo "pmc pass $i: $set"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary.json" "$OUT"/pmc? > "$OUT/pmc_summary.txt"
python3 "$ROOT/tools/shell_timeline.py" "$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)" --out "$OUT/sweep_timeline.txt" > /dev/null || true
head -8 "$OUT/kernel_stats.csv" | cut -c1-60,200-
cat "$OUT/pmc_summary.txt"
