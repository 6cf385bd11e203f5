"""Response (d) sharding across the GPUs of one node.

Inside a sweep every response column is independent given the per-covariate terms
(src/coreLoop.cpp:34-50), so rank r owns responses [k_lo, k_hi) and X is replicated.
The only exchange is ONE all-reduce (sum, fp64) per CAVI iteration of
  [ rowSums(gam_vb) (p)  |  shard-local scalar sums ]
-- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  Every rank gets
bit-identical sums, so the replicated hyper-parameter updates stay in lock step.

The reference has no counterpart (its only parallelism is parallel::mclapply over
independent sub-problems, R/locus.R:601).
"""
import torch


class Comm:
    """Single-process stand-in: world of one, all-reduce is the identity."""
    rank = 0
    world = 1

    def shard(self, d):
        return 0, d

    def all_reduce_sum_(self, t):
        return t

    def barrier(self):
        pass


class DistComm(Comm):
    """torch.distributed-backed communicator (backend nccl == RCCL on ROCm, or gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def shard(self, d):
        return (d * self.rank) // self.world, (d * (self.rank + 1)) // self.world

    def all_reduce_sum_(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        self._dist.barrier(group=self.group)


def shard_bounds(d, rank, world):
    return (d * rank) // world, (d * (rank + 1)) // world
