#!/usr/bin/env bash
# Run ON THE GPU BOX: sample the engine clock and the power while bench.py runs (why do boxes differ by 8 % in k_rates?)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 900 --warmup 5 --no-cpu-baseline > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
pid=$!
sleep 14
for i in $(seq 1 10); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $pid
python - <<PY
import json
d=json.load(open('gpurun_out/clk_bench.json'))
print('bench', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
