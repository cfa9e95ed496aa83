#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): the library's HOST code -- the sub-box loop, the column-scratch arena, the tile /
# source lists, the tiers of the heating pass, the communicator set-up -- under AddressSanitizer and
# UndefinedBehaviorSanitizer, driven by the real GPU tests.  Only the host half of csrc/c2ray_hip.hip is instrumented
# (-fno-gpu-sanitize: device-side ASan needs xnack, which this pool does not offer); kernels run as always.
# The run-time libraries are gcc's (libasan.so.6, libubsan.so.1; same __asan_* / __ubsan_* interface as clang's
# instrumentation asks for): ROCm's own ASan runtime intercepts hsa_amd_memory_pool_allocate for device-side ASan and
# fails every device allocation on a pool without xnack.
# The instrumented library replaces libc2ray_hip.so in the box's scratch copy of the repository only.
#   tools/asan_host_run.sh [pytest arguments ...]        default: the fuzz cases and the parity tests up to 64^3
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
ASAN_RT=/usr/lib/x86_64-linux-gnu/libasan.so.6
UBSAN_RT=/usr/lib/x86_64-linux-gnu/libubsan.so.1
[ -f "$ASAN_RT" ] && [ -f "$UBSAN_RT" ] || { echo "gcc's sanitizer run-time libraries are not in this image"; exit 1; }
export C2R_EXTRA_HIPCC_FLAGS="-g -fsanitize=address,undefined -fno-sanitize=vptr,function -fno-gpu-sanitize -fno-omit-frame-pointer"
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
print(ge.load_package().build(force=True))
PY
export LD_PRELOAD="$ASAN_RT $UBSAN_RT"
mkdir -p "$ROOT/gpurun_out"
rm -f "$ROOT"/gpurun_out/asan_report*
# leaks: CPython and the HIP runtime keep what they allocate
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:log_path=$ROOT/gpurun_out/asan_report"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:log_path=$ROOT/gpurun_out/ubsan_report"
ARGS=("$@")
[ ${#ARGS[@]} -gt 0 ] || ARGS=(tests/test_gpu_fuzz.py tests/test_gpu_parity.py -k "not 256 and not 512 and not benchmark_size and not config3 and not n128")
set +e
python3 -X faulthandler -m pytest -m gpu -x -q "${ARGS[@]}"
rc=$?
for f in "$ROOT"/gpurun_out/asan_report* "$ROOT"/gpurun_out/ubsan_report*; do
  [ -f "$f" ] && { echo "== $f"; head -60 "$f" | cut -c1-220; }
done
exit $rc
