"""Multi-GPU plumbing: independent herds, NO collective on the data path.

The jump path shards embarrassingly (disjoint herds per GPU, Kangaroo.cpp:1041-1047); what the GPUs
share is the HOST distinguished-point table.  Launched as one process per GPU (torchrun), rank 0 therefore drives
every device through kng_solver (a host thread per GPU, one table) and the other ranks only take part in the
rendezvous, the barriers around the timed region and the max-reduce of the elapsed time (timed_on_rank0).
backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
import time
from typing import Callable


class Ranks:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = max(int(os.environ.get("WORLD_SIZE", "1")), 1)
        self.backend = backend
        self.dist = None
        self._torch = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or "nccl"
            self.backend = backend
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=backend)
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
    def sync(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
        if self.backend != "gloo":
            try:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except ImportError:
                pass

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([value], dtype=self._torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
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
    """Time job() on rank 0 between two barrier+sync pairs; every rank returns the same elapsed seconds (max over
    ranks: the idle ranks measure the same interval through the barriers)."""
    ranks.sync()
    t0 = time.perf_counter()
    if ranks.rank == 0 and job is not None:
        job()
    ranks.sync()
    return ranks.max_over_ranks(time.perf_counter() - t0)


def whole_job_rate(ranks: Ranks, units_per_rank_per_step: int, steps: int, elapsed: float) -> float:
    """Units per second of the whole job: every rank processed the same number of units (weak scaling)."""
    return ranks.world * units_per_rank_per_step * steps / elapsed
