"""Sharded (multi-GPU) fit: host-side plumbing over the C ABI's all-reduce hook.

SURVEY.md section 8e: data points are independent rows, so the fit shards by points.
Every rank bins and assembles ITS points; the nearest-node histogram, then the normal
equations (half-stencil + right-hand side), then each refinement residual are
sum-all-reduced (RCCL over xGMI = torch.distributed backend "nccl" on ROCm).  The Cholesky of
the normal equations: with a hook that accepts any device pointer (make_allreduce's does) the
nested-dissection factorisation is distributed by subtrees -- a rank eliminates its own, their
Schur complements are summed through the same hook, the top of the tree and all memory are
replicated (route (a)); band-path grids and two-argument hooks: the
factorisation is replicated.  The route whose MEMORY is partitioned is the one-process multi-GPU
plan (capi.MultiPlan).  Evaluation shards the queries and needs no collective.

The reduction buffers live in ONE torch tensor (`comm`) that is handed to the plan at
creation; the HIP library calls back with a pointer into it and this module all-reduces
the corresponding view on the caller's stream.  One process per GPU.
"""
from __future__ import annotations

import numpy as np

from . import capi


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block partition of `n_total` points: -> (first, count) of `rank`."""
    base, rem = divmod(int(n_total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def make_allreduce(comm, dist, group=None):
    """-> fn(offset, count[, view]) that sum-all-reduces comm[offset:offset+count] (or the given device view) in place."""
    def _fn(offset: int, count: int, view=None):
        if view is None:
            if offset < 0 or offset + count > comm.numel():
                raise ValueError(f"all-reduce window [{offset}, {offset + count}) outside the buffer")
            view = comm[offset:offset + count]
        if comm.is_cuda and dist.get_backend(group) != "nccl":
            # A host backend (gloo: the CPU tests and the one-GPU rehearsal of the multi-rank path) on a device buffer:
            # stage through host memory with explicit synchronisation.  gloo's own device-tensor path only orders its
            # copies by stream events, and two ranks sharing ONE GPU were seen to sum a buffer that the fit's kernels
            # had not finished writing (round 3: garbage normal equations in `bench.py --gpus 2` on one device).
            import torch
            torch.cuda.current_stream().synchronize()
            host = view.cpu()
            dist.all_reduce(host, group=group)
            view.copy_(host)
            torch.cuda.current_stream().synchronize()
            return
        dist.all_reduce(view, group=group)       # RCCL: enqueued behind the current (= the library's) stream
    return _fn


class ShardedFit:
    """A fit plan whose reductions go through torch.distributed (any backend)."""

    def __init__(self, ndim, nodes, xmin, xmax, xtrap, max_ndata, device, dist=None, group=None):
        import torch
        self.dist = dist
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        nodes_a = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
        self.comm_len = int(capi.lib().splpak_plan_comm_len(ndim, capi._p(nodes_a, capi._ip)))
        self.comm = torch.zeros(self.comm_len, dtype=torch.float64, device=device)
        self.plan = capi.Plan(ndim, nodes, xmin, xmax, xtrap, max_ndata, comm=self.comm)
        if self.world > 1:
            self.plan.set_allreduce(make_allreduce(self.comm, dist, group), self.rank, self.world)

    def fit(self, xdata, ydata, wdata, coef, stream=0):
        return self.plan.fit(xdata, ydata, wdata, coef, stream)

    def close(self):
        self.plan.close()
