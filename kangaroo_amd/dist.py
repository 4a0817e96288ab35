"""Multi-GPU plumbing: independent herds, NO collective on the data path.

The jump path shards embarrassingly (disjoint herds per GPU, Kangaroo.cpp:1041-1047); what the GPUs
share is the HOST distinguished-point table.  Launched as one process per GPU (torchrun), rank 0 therefore drives
every device through kng_solver (a host thread per GPU, one table) and the other ranks only take part in the
rendezvous, the barriers around the timed region and the max-reduce of the elapsed time (timed_on_rank0).

The control plane is CPU-side: a `gloo` process group.  Nothing of it may touch a GPU -- an RCCL barrier parks a
spinning kernel on the device of every waiting rank, and those are the very devices rank 0's walk kernels are being
timed on (one 512-thread workgroup per CU, no slack for a co-resident spinner).  There is no GPU collective to run:
herds never exchange anything (set KNG_DIST_BACKEND=nccl only to prove that point).

Failure protocol: a rank that fails says so through all_ok() (an all-reduce every rank reaches), so that EVERY rank leaves
with a non-zero status instead of one rank dying while the others sit in a barrier until the launcher's timeout.
"""
from __future__ import annotations

import os
import time
from typing import Callable


class RankFailure(RuntimeError):
    """some rank of the job failed; raised on every rank"""


class Ranks:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = max(int(os.environ.get("WORLD_SIZE", "1")), 1)
        self.backend = backend or os.environ.get("KNG_DIST_BACKEND", "gloo")
        self.dist = None
        self._torch = None
        if self.world > 1:
            import datetime

            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=self.backend, timeout=datetime.timedelta(seconds=int(os.environ.get("KNG_DIST_TIMEOUT", "1800"))))
            self.dist = dist
            self._torch = torch

    # -- what every rank derives from its rank: device and herd seed ------------------------------
    @property
    def device(self) -> int:
        return self.local_rank

    def herd_seed(self, base: int) -> int:
        """Distinct, reproducible herd per rank (herds must be disjoint random walks)."""
        return (base + 0x9E3779B97F4A7C15 * (self.rank + 1)) & ((1 << 64) - 1)

    def total_kangaroos(self, per_gpu: int) -> int:
        """totalRW of Kangaroo.cpp:946-959: every GPU contributes its own herd."""
        return per_gpu * self.world

    # -- synchronisation -----------------------------------------------------------------------------
    def _tensor(self, value, dtype):
        dev = "cuda" if self.backend == "nccl" else "cpu"
        return self._torch.tensor([value], dtype=dtype, device=dev)

    def sync(self) -> None:
        """barrier.  The engines synchronise their own streams (kng_wait / kngs_wait); nothing here touches a GPU
        unless the nccl backend was asked for."""
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                self._torch.cuda.synchronize()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self._tensor(value, self._torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok: bool) -> bool:
        """True when every rank passed True.  Every rank must call it at the same point."""
        if self.dist is None:
            return bool(ok)
        t = self._tensor(1 if ok else 0, self._torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def close(self) -> None:
        if self.dist is not None:
            try:
                self.dist.barrier()
            finally:
                self.dist.destroy_process_group()
                self.dist = None

    def abort(self) -> None:
        """leave without the closing barrier (some rank failed and said so through all_ok)"""
        if self.dist is not None:
            try:
                self.dist.destroy_process_group()
            except Exception:
                pass
            self.dist = None


def timed_steps(ranks: Ranks, step: Callable[[int], None], finish: Callable[[], None], steps: int) -> float:
    """Time exactly `steps` calls of step(i) (+ finish()) between two barrier+sync pairs and return
    the MAX over ranks of the elapsed seconds (the bench contract)."""
    ranks.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    finish()
    ranks.sync()
    return ranks.max_over_ranks(time.perf_counter() - t0)


def timed_on_rank0(ranks: Ranks, job: Callable[[], object] | None) -> float:
    """Time job() on rank 0 between two barriers; every rank returns the same elapsed seconds (max over ranks: the
    idle ranks measure the same interval through the barriers).  If the job raises, every rank raises RankFailure
    after the closing barrier -- nobody is left waiting."""
    ranks.sync()
    t0 = time.perf_counter()
    err = None
    if ranks.rank == 0 and job is not None:
        try:
            job()
        except BaseException as e:  # noqa: BLE001 -- reported to every rank below
            err = e
    ranks.sync()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    if not ranks.all_ok(err is None):
        raise RankFailure(f"rank 0's job failed: {err!r}" if err is not None else "rank 0's job failed") from err
    return elapsed


def whole_job_rate(ranks: Ranks, units_per_rank_per_step: int, steps: int, elapsed: float) -> float:
    """Units per second of the whole job: every rank processed the same number of units (weak scaling)."""
    return ranks.world * units_per_rank_per_step * steps / elapsed
