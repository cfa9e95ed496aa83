#!/bin/bash
# The launch modes of bench.py on a one-GPU box, same box, one after another:
# default, one-process with a one-device RCCL communicator, one rank through torch.distributed + c2r_comm_init,
# two replicas sharing the device (rehearsal), and --gpus 2 without a second device (must fail loudly).
out=${1:-gpurun_out/bench_modes}
mkdir -p "$out"
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 400 "$@" > "$out/$name.json" 2> "$out/$name.err"; echo "   rc=$? $(cut -c1-260 "$out/$name.json")"; }
run default python bench.py --steps 20 --warmup 3 --no-cpu-baseline
run one_process python bench.py --gpus 1 --one-process --steps 20 --warmup 3 --no-cpu-baseline
C2R_BENCH_FORCE_COMM=1 run force_comm python bench.py --steps 20 --warmup 3 --no-cpu-baseline
C2R_BENCH_SHARE_DEVICE=1 run share2 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline
run two_without_second_device python bench.py --gpus 2 --steps 2 --no-cpu-baseline
run children1 python bench.py --gpus 1 --launcher children --steps 5 --warmup 2 --no-cpu-baseline
tail -2 "$out/two_without_second_device.err"
