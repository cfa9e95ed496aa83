#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): the library's HOST code -- the sub-box loop, the column-scratch arena, the tile /
# source lists, the tiers of the heating pass, the communicator set-up -- under AddressSanitizer and
# UndefinedBehaviorSanitizer, driven by the real GPU tests.  Only the host half of csrc/c2ray_hip.hip is instrumented
# (-fno-gpu-sanitize: device-side ASan needs xnack, which this pool does not offer); kernels run as always.
# The instrumented library replaces libc2ray_hip.so in the box's scratch copy of the repository only.
#   tools/asan_host_run.sh [pytest arguments ...]        default: the fuzz cases and the parity tests up to 64^3
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
[ -n "$RT" ] || { echo "no ASan runtime in this image"; exit 1; }
export C2R_EXTRA_HIPCC_FLAGS="-g -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer"
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
print(ge.load_package().build(force=True))
PY
export LD_PRELOAD="$RT"
export LD_LIBRARY_PATH="$(dirname "$RT"):${LD_LIBRARY_PATH:-}"
# leaks: CPython and the HIP runtime keep what they allocate; shadow gap: ROCr maps memory where ASan wants none
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
ARGS=("$@")
[ ${#ARGS[@]} -gt 0 ] || ARGS=(tests/test_gpu_fuzz.py tests/test_gpu_parity.py -k "not 256 and not 512 and not benchmark_size and not config3 and not n128")
python3 -m pytest -m gpu -x -q "${ARGS[@]}"
