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
    solo = True      # nothing to exchange: the drivers skip the staging buffer and the collectives

    def shard(self, d):
        return 0, d

    def all_reduce_sum_(self, t):
        return t

    def barrier(self):
        pass


class DistComm(Comm):
    """torch.distributed-backed communicator (backend nccl == RCCL on ROCm, or gloo)."""

    def __init__(self, group=None, force_collectives=False):
        """`force_collectives`: a group of ONE rank still goes through the packed all-reduce and
        the all-gather (a one-GPU box can then execute the RCCL branches the N > 1 runs use)."""
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.solo = self.world == 1 and not force_collectives
        self._sync_first = dist.get_backend(group) == "gloo"

    def shard(self, d):
        return (d * self.rank) // self.world, (d * (self.rank + 1)) // self.world

    def all_reduce_sum_(self, t):
        if t.is_cuda and self._sync_first:
            # gloo stages device tensors through the host on streams of its own.  Two ranks that
            # share one GPU (the one-GPU rehearsal of the N > 1 path, tests) stalled there with a
            # long sweep still queued: drain this rank's stream first.  RCCL enqueues the
            # collective on the stream and needs none of this.
            import torch
            torch.cuda.current_stream(t.device).synchronize()
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        self._dist.barrier(group=self.group)

    def all_gather_columns(self, local, d):
        """rows x d_r column shards of every rank -> the rows x d matrix (Fortran order) on every
        rank.  Shards may differ in width by one column (shard()), so each rank's piece travels
        as a [d_r][rows] block padded to the widest shard; the pad is dropped on arrival."""
        import numpy as np
        dist = self._dist
        rows = local.shape[0]
        bounds = [shard_bounds(d, r, self.world) for r in range(self.world)]
        wmax = max(hi - lo for lo, hi in bounds)
        lo, hi = bounds[self.rank]
        assert local.shape[1] == hi - lo
        mine = torch.zeros((wmax, rows), dtype=torch.float64)
        mine[: hi - lo] = torch.from_numpy(np.ascontiguousarray(local.T))
        on_gpu = dist.get_backend(self.group) == "nccl"
        if on_gpu:
            mine = mine.cuda()
        out = torch.empty((self.world * wmax, rows), dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        out = out.cpu().numpy().reshape(self.world, wmax, rows)
        full = np.empty((rows, d), order="F")
        for r, (a, b) in enumerate(bounds):
            full[:, a:b] = out[r, : b - a].T
        return full


def shard_bounds(d, rank, world):
    return (d * rank) // world, (d * (rank + 1)) // world
