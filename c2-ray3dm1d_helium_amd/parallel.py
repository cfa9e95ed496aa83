"""Sources over ranks, one all-reduce of the rate grids per outer iteration.

The reference's only distributed strategy: every rank holds the full grid,
rank r sweeps sources r+1, r+1+npr, ... (master_slave.F90:85), then
mpi_accumulate_grid_quantities (evolve.F90:505-548) sums four grids, photon_loss(47) and sum_nbox
with six MPI_ALLREDUCE calls.  Here that is ONE fp64 SUM all-reduce over the contiguous buffer
[phih | phihe(0) | phihe(1) | phiheat | photon_loss(1:47) | sum_nbox] through torch.distributed
(backend "nccl" == RCCL over xGMI on the GPUs; "gloo" in the CPU tests).
"""
from __future__ import annotations


class TorchComm:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def allreduce_rates(self, engine):
        """Sum the engine's reduction buffer over all ranks, in place."""
        buf = engine.rates_buffer()          # torch tensor aliasing the engine's buffer
        engine.synchronize()                 # the sweep wrote it on the engine's own stream
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        if buf.is_cuda:
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
        engine.rates_reduced()


class SingleComm:
    rank, size = 0, 1

    def allreduce_rates(self, engine):
        return None
