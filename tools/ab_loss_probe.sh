set -e
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.load(open('gpurun_out/ab_$name.json'))
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
}
run nosplit C2R_SWEEP_SPLIT_FROM=0
run split_hi C2R_SWEEP_B_PRIO=0
run split_normal C2R_SWEEP_B_PRIO=1
run split_lo C2R_SWEEP_B_PRIO=2
run split_normal_gpuq8 C2R_SWEEP_B_PRIO=1 GPU_MAX_HW_QUEUES=8
run split_hi_gpuq8 C2R_SWEEP_B_PRIO=0 GPU_MAX_HW_QUEUES=8
