#!/usr/bin/env bash
# Run ON THE GPU BOX: PMC passes of the heating, three-SED rates kernel on the 256^3 faint-source workload
# (tools/bench_config4.py --pl --batch 128 --max-iter 2), one small counter set per pass, summarised into
# gpurun_out/prof_three_seds/summary.json (copy to profiles/rNN_rates_three_seds_pmc.json).
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_three_seds
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--pl --mesh 256 --sources 1024 --rank 0 --batch 128 --max-iter 2"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM SQ_WAVES" "FETCH_SIZE"; do
  i=$((i+1))
  echo "pmc pass $i: $set"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- python3 "$ROOT/tools/bench_config4.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/tools/bench_config4.py" $ARGS > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ratesILb1ELb1E" in r["Kernel_Name"] or "k_rates<true, true>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
res = {k: agg[k] / n[k] for k in agg}
dur = []
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ratesILb1ELb1E" in r["Kernel_Name"] or "k_rates<true, true>" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
sys.path.insert(0, out + "/../..")
try:
    import bench; sha = bench.kernel_source_sha16()
except Exception: sha = None
j = {"source_sha16": sha,
     "command": "rocprofv3 --kernel-trace --pmc <one set per pass> -- python3 tools/bench_config4.py --pl --mesh 256 --sources 1024 --rank 0 --batch 128 --max-iter 2",
     "k_rates<heat, three SEDs> per launch": res, "launch_ms_kernel_trace": dur}
if dur and "SQ_INSTS_VALU" in res:
    ms = sum(dur) / len(dur)
    j["issue_fraction_at_4.3_cycles_2.31GHz"] = res["SQ_INSTS_VALU"] * 4.3 / (1024 * 2.31e9) / (ms * 1e-3)
    j["issue_fraction_at_4_cycles_2.4GHz"] = res["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) / (ms * 1e-3)
if "SQ_WAIT_ANY" in res and "SQ_WAVE_CYCLES" in res:
    j["wait_any_over_wave_cycles"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
json.dump(j, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(j, indent=1))
PY
