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
# the 2-, 4- and 8-rank control flow of the RCCL path (c2r_comm_kind 1) on this one device, sums carried by the stand-in of
# tests/fake_rccl.hip: a rehearsal of `bench.py --gpus N` before a node exists -- the line says STAND-IN and rccl_ranks 0
for n in 2 4 8; do
  C2R_BENCH_SHARE_DEVICE=1 C2R_COMM_SHARED_DEVICE_RCCL=1 C2R_RCCL_LIBRARY=$PWD/tests/_fake_rccl.so run standin$n python bench.py --gpus $n --steps 4 --warmup 1 --no-cpu-baseline
done
# ... and the driver's own launch shape, torch.distributed.run with one process per rank, on this one device: the multi-process
# mode of the stand-in (FAKE_RCCL_MULTIPROCESS=1; C2R_BENCH_SHARE_DEVICE=1 puts every rank on GPU 0)
for n in 2 4; do
  C2R_BENCH_SHARE_DEVICE=1 FAKE_RCCL_MULTIPROCESS=1 C2R_RCCL_LIBRARY=$PWD/tests/_fake_rccl.so run torchrun${n}_standin python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --steps 4 --warmup 1 --no-cpu-baseline
done
run two_without_second_device python bench.py --gpus 2 --steps 2 --no-cpu-baseline
run children1 python bench.py --gpus 1 --launcher children --steps 5 --warmup 2 --no-cpu-baseline
tail -2 "$out/two_without_second_device.err"
