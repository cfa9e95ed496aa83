set -e
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.load(open('gpurun_out/ab_$name.json'))
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
}
for i in 1 2 3; do
run chem$i A=1
run start$i C2R_PACK_AT_PASS_START=1
run generic$i C2R_SWEEP_GENERIC=1
done
