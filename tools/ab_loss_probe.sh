set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "fuzz or benchmark_size or config4 or full_evolve3d or one_iteration or n64" > gpurun_out/ab_tests.log 2>&1 || true
tail -3 gpurun_out/ab_tests.log
run() { name=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.load(open('gpurun_out/ab_$name.json'))
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
}
run ytab_a A=1
run ytab_b A=1
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
os.environ['C2R_EXTRA_HIPCC_FLAGS'] = '-DC2R_NO_LOG10_YTAB'
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.build(force=True)
PY
run noytab_a A=1
run noytab_b A=1
