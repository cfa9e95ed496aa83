set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or early_stopping or batching or benchmark_size or config3 or scratch" > gpurun_out/ab_tests.log 2>&1 || true
tail -3 gpurun_out/ab_tests.log
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.load(open('gpurun_out/ab_$name.json'))
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
}
run new A=1
run legacy C2R_FINAL_LOSS_KERNEL=1
run new2 A=1
run legacy2 C2R_FINAL_LOSS_KERNEL=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_ab.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/shell_timeline.py $(find gpurun_out/prof_ab/stats -name "*kernel_trace.csv") --out gpurun_out/prof_ab_timeline.txt > /dev/null
