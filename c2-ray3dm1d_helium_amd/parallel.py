"""Sources over ranks with the sum carried by torch.distributed: the TEST TRANSPORT of the host loop.

The product's sum over ranks lives inside the library (csrc/c2ray_comm.inc: c2r_comm_init, c2r_allreduce_rates,
c2r_pass_allreduce_chemistry -- one RCCL all-reduce of the contiguous buffer, or slab-wise and overlapped with the
pass and the chemistry); bench.py and the Fortran drop-in use that.  This module keeps the same host loop
(rank r sweeps sources r+1, r+1+npr, ... as master_slave.F90:85, then one fp64 SUM over
[phih | phihe(0) | phihe(1) | phiheat | photon_loss(1:47) | sum_nbox] as mpi_accumulate_grid_quantities,
evolve.F90:505-548, then the replicated global pass) over a torch.distributed process group, so that the
distribution logic can be exercised where RCCL cannot run: world_size 2 over gloo on the CPU with the oracle as
engine (tests/test_host_logic.py), and several ranks sharing one GPU over gloo (tests/test_gpu_multirank.py;
RCCL refuses ranks that share a device).  It is not a second production path.
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


    def pass_and_allreduce(self, engine, nslab=None):
        """pass_all_sources + mpi_accumulate_grid_quantities (evolve.F90:385-431, :505-548) for this rank's
        sources, the sum over ranks overlapped with the pass where the engine can hand over slabs."""
        if not hasattr(engine, "pass_sources_begin"):
            engine.pass_sources(1 + self.rank, self.size)
            return self.allreduce_rates(engine)
        if nslab is None:
            import os
            nslab = int(os.environ.get("C2R_ALLREDUCE_SLABS", "4"))
        buf = engine.rates_buffer()
        nc = (buf.numel() - 48) // 4
        ncomp = 3 if getattr(engine, "isothermal", False) else 4   # phiheat stays zero in isothermal runs
        works = []
        n = engine.pass_sources_begin(1 + self.rank, self.size, nslab)
        for s in range(n):
            c0, cnt = engine.pass_wait_slab(s)
            for comp in range(ncomp):
                works.append(self.dist.all_reduce(buf[comp * nc + c0: comp * nc + c0 + cnt], op=self.dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))
        engine.pass_sources_end()
        works.append(self.dist.all_reduce(buf[4 * nc:], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if buf.is_cuda:
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
        engine.rates_reduced()


    def pass_allreduce_chemistry(self, engine, dt, nslab=None):
        """One outer iteration's pass_all_sources + mpi_accumulate_grid_quantities + global_pass with all three
        overlapped: slab s is all-reduced while the rates of slab s+1 are computed, and its chemistry runs as soon
        as its sum is complete, i.e. while later slabs are still on the wire.  Returns the non-converged count."""
        if not (hasattr(engine, "pass_sources_begin") and hasattr(engine, "global_pass_cells")):
            self.pass_and_allreduce(engine, nslab)
            return engine.global_pass(dt)
        import os
        import torch
        if nslab is None:
            nslab = int(os.environ.get("C2R_ALLREDUCE_SLABS", "4"))
        buf = engine.rates_buffer()
        nc = (buf.numel() - 48) // 4
        ncomp = 3 if getattr(engine, "isothermal", False) else 4
        n = engine.pass_sources_begin(1 + self.rank, self.size, nslab)
        slabs = []
        for s in range(n):
            c0, cnt = engine.pass_wait_slab(s)
            works = [self.dist.all_reduce(buf[comp * nc + c0: comp * nc + c0 + cnt], op=self.dist.ReduceOp.SUM,
                                          group=self.group, async_op=True) for comp in range(ncomp)]
            slabs.append((c0, cnt, works))
        engine.pass_sources_end()
        tail = self.dist.all_reduce(buf[4 * nc:], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        events = []
        for c0, cnt, works in slabs:
            for w in works:
                w.wait()                       # torch's current stream now waits for this slab's sum
            handle = None
            if buf.is_cuda:
                ev = torch.cuda.Event()
                ev.record()                    # ... and the library's stream waits for that point
                events.append(ev)
                handle = ev.cuda_event
            engine.global_pass_cells(dt, c0, cnt, handle)
        tail.wait()
        if buf.is_cuda:
            torch.cuda.current_stream(buf.device).synchronize()
        engine.rates_reduced()
        return engine.global_pass_finish()


class RcclComm:
    """One process per GPU, the sum over ranks inside the library (c2r_comm_init + ncclAllReduce): what
    bench.py and a torch.distributed.run launch use.  torch.distributed only carries the 128-byte RCCL id from
    rank 0 to the others (any backend) and the barrier / max-over-ranks of the timing harness; nothing of it is
    in the data path."""

    def __init__(self, engine, dist=None, fail_on_ranks=()):
        """Collective over `dist`: either every rank leaves with a communicator or every rank raises the same
        error (and holds none) -- a rank that failed alone would leave the others waiting inside the collective
        ncclCommInitRank, or inside the broadcast of the id.  Three agreements, each over torch.distributed:
        (1) every rank can load RCCL (c2r_comm_available) AND meets c2r_comm_init's local preconditions (its context has
        no communicator yet) -- everything a rank can know on its own is settled before the collective call, (2) rank 0
        obtained the id (the error text travels in its place), (3) every rank's c2r_comm_init succeeded.
        What (3) cannot cover: a rank that fails INSIDE ncclCommInitRank (a device error) while its peers are already
        blocked in the same collective -- they never reach the agreement.  For that window c2r_comm_init runs under a
        watchdog: a rank whose call has not returned after C2R_COMM_INIT_TIMEOUT_S seconds (default 600; 0: no limit)
        says so on stderr and EXITS the process with status 3, so that the launcher (torch.distributed.run, mpirun) ends
        the job instead of leaving it hung.  fail_on_ranks: ranks that pretend step (1) failed (rehearsal of the
        fall-back in bench.py and tests/test_host_logic.py)."""
        self.engine = engine
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.size = dist.get_world_size() if dist is not None else 1
        many = dist is not None and self.size > 1

        def agree(err):
            """None when no rank reports an error, else the first rank's text (the same on every rank)."""
            if not many:
                return err
            errs = [None] * self.size
            dist.all_gather_object(errs, err)
            return next((f"rank {r}: {e}" for r, e in enumerate(errs) if e is not None), None)

        err = "forced failure (rehearsal)" if self.rank in tuple(fail_on_ranks) else type(engine).comm_available()
        if err is None and engine.comm_size() > 1:
            err = "the context already has a communicator"
        err = agree(err)
        if err is not None:
            raise RuntimeError(f"RCCL is not usable on every rank: {err}")
        box = [None]
        if self.rank == 0:
            try:
                box[0] = ("id", type(engine).comm_unique_id())
            except Exception as ex:  # noqa: BLE001 -- travels to the other ranks instead of the id
                box[0] = ("error", str(ex))
        if many:
            dist.broadcast_object_list(box, src=0)
        if box[0][0] != "id":
            raise RuntimeError(f"rank 0 could not obtain the RCCL id: {box[0][1]}")
        err = self._init_with_watchdog(engine, box[0][1])
        all_err = agree(err)
        if all_err is not None:
            if err is None:
                engine.comm_destroy()
            raise RuntimeError(f"c2r_comm_init failed: {all_err}")

    def _init_with_watchdog(self, engine, unique_id):
        """c2r_comm_init; returns the error text or None.  Does not return at all when the call outlives the time-out."""
        import os
        import sys
        import threading
        limit = float(os.environ.get("C2R_COMM_INIT_TIMEOUT_S", "600"))
        out = {}

        def call():
            try:
                engine.comm_init(self.rank, self.size, unique_id)
                out["err"] = None
            except Exception as ex:  # noqa: BLE001
                out["err"] = str(ex)

        if limit <= 0 or self.size == 1:
            call()
            return out["err"]
        th = threading.Thread(target=call, daemon=True)
        th.start()
        th.join(limit)
        if th.is_alive():
            sys.stderr.write(f"c2ray_hip: rank {self.rank} of {self.size}: c2r_comm_init (ncclCommInitRank) has not returned after "
                             f"{limit:.0f} s -- a peer rank is missing or failed inside the collective; exiting so that the launcher "
                             "ends the job (C2R_COMM_INIT_TIMEOUT_S)\n")
            sys.stderr.flush()
            os._exit(3)
        return out["err"]

    def allreduce_rates(self, engine=None):
        self.engine.allreduce_rates()

    def pass_and_allreduce(self, engine=None, nslab=None):
        self.engine.pass_sources(1 + self.rank, self.size)
        self.engine.allreduce_rates()

    def pass_allreduce_chemistry(self, engine, dt, nslab=None):
        import os
        if nslab is None:
            nslab = int(os.environ.get("C2R_ALLREDUCE_SLABS", "4"))
        return self.engine.pass_allreduce_chemistry(dt, 1 + self.rank, self.size, nslab)


class LocalComm:
    """One process, the devices of a c2r_create_multi context with c2r_comm_init_local's communicators
    (ncclCommInitAll): the reference's no_mpi build on a whole node.  The library deals the sources over its devices
    (device i: 1 + i, 1 + i + ndev, ...), sums over them and replicates the chemistry; this host sees one rank."""
    rank = 0

    def __init__(self, engine):
        self.engine = engine
        self.size = 1        # ranks of THIS host loop: the split over the devices is the library's

    def allreduce_rates(self, engine=None):
        self.engine.allreduce_rates()

    def pass_and_allreduce(self, engine=None, nslab=None):
        self.engine.pass_sources(1, 1)
        self.engine.allreduce_rates()

    def pass_allreduce_chemistry(self, engine, dt, nslab=None):
        import os
        if nslab is None:
            nslab = int(os.environ.get("C2R_ALLREDUCE_SLABS", "4"))
        return self.engine.pass_allreduce_chemistry(dt, 1, 1, nslab)


class SingleComm:
    rank, size = 0, 1

    def allreduce_rates(self, engine):
        return None

    def pass_and_allreduce(self, engine, nslab=8):
        engine.pass_sources(1, 1)
