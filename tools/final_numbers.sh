# Run ON THE GPU BOX: the numbers DESIGN.md section 5 quotes, all from one box
set -e
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/fin_$name.json 2> gpurun_out/fin_$name.err; python - <<PY
import json
d=json.load(open('gpurun_out/fin_$name.json'))
print('$name', round(d['ms_per_step'],3), '%.3e' % d['value'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'cov', round(d['config']['coverage'],3), 'sweep frac', round(d['roofline_column_sweep']['frac'],4))
PY
}
run default --steps 100 --warmup 5
run default2 --steps 100 --warmup 5
C2R_BENCH_FORCE_COMM=1 run comm1 --steps 50 --warmup 5
run neutral --steps 40 --warmup 5 --neutral-start
run heating --steps 40 --warmup 5 --heating
run config4 --workload config4 --steps 6 --warmup 1
