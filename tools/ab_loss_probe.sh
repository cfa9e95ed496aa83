set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "heat or fuzz or config4 or n64 or n128" > gpurun_out/ab_tests.log 2>&1 || true
tail -3 gpurun_out/ab_tests.log
run() { dir=$1; name=$2; shift; shift; (cd $dir && env "$@" python bench.py --no-cpu-baseline $ARGS > $GRAFT_REPO_ROOT/gpurun_out/ab_$name.json 2> $GRAFT_REPO_ROOT/gpurun_out/ab_$name.err); python - <<PY
import json
d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/ab_$name.json'))
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'cov', round(d['config']['coverage'],3))
PY
}
ARGS="--steps 30 --warmup 5 --heating"
run . heat_now A=1
run .r02_tree heat_r02 A=1
run . heat_now2 A=1
ARGS="--steps 40 --warmup 5"
run . iso_now A=1
run .r02_tree iso_r02 A=1
C2R_CHEM_LOG=1 python tools/bench_config4.py --all-ranks --heating --batch 128 --max-iter 7 > gpurun_out/c4_heat3.json 2> gpurun_out/c4_heat3.err; python - <<PY
import json
d=json.load(open("gpurun_out/c4_heat3.json"))
print("c4 heating chem", [round(h["chem_ms"],1) for h in d["iterations"]])
PY
