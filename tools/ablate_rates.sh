set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_outer or full_size" > gpurun_out/pytest_stateT.log 2>&1; tail -2 gpurun_out/pytest_stateT.log
for v in BASE C2R_ABL_LOG C2R_ABL_DIVVOL C2R_ABL_TABLE C2R_ABL_SCALE; do
  if [ $v = BASE ]; then export C2R_EXTRA_HIPCC_FLAGS=""; else export C2R_EXTRA_HIPCC_FLAGS="-D$v"; fi
  touch c2-ray3dm1d_helium_amd/csrc/c2ray_hip.hip
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$v.log").read().strip().splitlines()[-1])
print("$v", round(d["ms_per_step"],2), d["kernel_ms_per_step"])
PY
done
export C2R_EXTRA_HIPCC_FLAGS=""; touch c2-ray3dm1d_helium_amd/csrc/c2ray_hip.hip; python -c "import __graft_entry__ as g; g.load_package().build()"
