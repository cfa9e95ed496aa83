#!/bin/bash
# Run ON THE GPU BOX at the end of a round, once the sources are final (bench.py quotes stored profiles only for the
# sources they were taken from): the GPU test-suite, the rocprofv3 statistics and PMC passes of the bench command, the
# drop-in timing, the numbers DESIGN.md section 5 quotes and the launch modes of bench.py.  Afterwards, in the dev container:
#   R=r04; cp gpurun_out/prof_bench_end/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
#   cp gpurun_out/prof_bench_end/pmc_summary.json profiles/${R}_bench_pmc_summary.json
#   cp gpurun_out/prof_bench_end/sweep_timeline.txt profiles/${R}_sweep_timeline.txt
#   cp gpurun_out/dropin_timing_end.json profiles/${R}_dropin_timing.json
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
echo "== GPU tests"; python -u -m pytest tests -m gpu -x -q --timeout=900 -p no:cacheprovider 2>&1 | tee gpurun_out/gputest_end.log | tail -4
echo "== profile"; PROF_NAME=prof_bench_end tools/profile_bench.sh > gpurun_out/prof_bench_end.log 2>&1; tail -3 gpurun_out/prof_bench_end.log
echo "== drop-in"; python tools/time_dropin.py --out gpurun_out/dropin_timing_end.json > gpurun_out/dropin_end.log 2>&1; tail -2 gpurun_out/dropin_end.log
echo "== numbers"; bash tools/final_numbers.sh 2>&1 | tail -8
echo "== modes"; tools/bench_modes.sh gpurun_out/bench_modes_end 2>&1 | grep -v "^$" | cut -c1-230
echo "== the driver's own command (with the CPU baselines)"; t0=$(date +%s); python bench.py > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err; echo "$(( $(date +%s) - t0 )) s wall"; python -c "
import json; j=json.load(open('gpurun_out/bench_driver_like.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], {k: (v.get('value'), v.get('cores'), v.get('kind')) for k, v in j.items() if k.startswith('cpu_baseline')})"
echo "== eight replicas on one device"; C2R_BENCH_SHARE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/share8.err | cut -c1-400
