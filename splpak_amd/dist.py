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


def native_rccl_comm(dist, device, group=None):
    """ncclComm_t (int) of the LIBRARY's own RCCL hook for the ranks of `group`: rank 0 draws the ncclUniqueId
    (splpak_rccl_unique_id), torch.distributed only carries its 128 bytes to the other ranks, every rank then calls
    splpak_rccl_comm_create on its device.  The fit's collectives do not pass through Python after that."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_dev = dist.get_backend(group) == "nccl"
    idt = torch.zeros(128, dtype=torch.uint8, device=device if on_dev else "cpu")
    err = None
    if rank == 0:
        try:
            idt.copy_(torch.frombuffer(bytearray(capi.rccl_unique_id()), dtype=torch.uint8))
        except capi.SplpakError as exc:      # the others wait in the broadcast: they get a zero id and give up with rank 0
            err = exc
    dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    raw = bytes(idt.cpu().numpy().tobytes())
    if err is not None or not any(raw):
        raise capi.SplpakError(f"no ncclUniqueId from rank 0{': ' + str(err) if err else ''}")
    with torch.cuda.device(device):
        return capi.rccl_comm_create(raw, rank, world)


class ShardedFit:
    """A fit plan whose reductions go over RCCL: through the library's NATIVE hook (splpak_plan_set_rccl -- ncclAllReduce
    enqueued on the fit's stream by the library itself, no Python and no host synchronisation in the loop) when the process
    group's backend is nccl (= RCCL on ROCm), through a torch.distributed callback otherwise (gloo: CPU tests, one-GPU
    rehearsals) or when `native=False` / the native communicator cannot be made.  `collective` says which."""

    def __init__(self, ndim, nodes, xmin, xmax, xtrap, max_ndata, device, dist=None, group=None, native=None):
        import os
        import torch
        self.dist = dist
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        nodes_a = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
        self.comm_len = int(capi.lib().splpak_plan_comm_len(ndim, capi._p(nodes_a, capi._ip)))
        self.comm = torch.zeros(self.comm_len, dtype=torch.float64, device=device)
        self.plan = capi.Plan(ndim, nodes, xmin, xmax, xtrap, max_ndata, comm=self.comm)
        self.collective = None
        self.nccl_comm = None
        self.native_error = None
        if self.world > 1:
            if native is None:
                native = dist.get_backend(group) == "nccl" and os.environ.get("SPLPAK_NATIVE_RCCL", "1") != "0"
            if native:
                # every rank must take the same route: the outcome of the attempt is made collective
                try:
                    self.nccl_comm = native_rccl_comm(dist, device, group)
                except Exception as exc:      # noqa: BLE001 -- whatever it is, the torch callback still works
                    self.native_error = f"{type(exc).__name__}: {exc}"
                ok = torch.tensor([1.0 if self.nccl_comm else 0.0], dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                if float(ok.item()) == 0.0 and self.nccl_comm:
                    capi.rccl_comm_destroy(self.nccl_comm)
                    self.nccl_comm = None
            if self.nccl_comm:
                self.plan.set_rccl(self.nccl_comm, self.rank, self.world)
                self.collective = "rccl-native"
            else:
                self.plan.set_allreduce(make_allreduce(self.comm, dist, group), self.rank, self.world)
                self.collective = "torch.distributed:" + str(dist.get_backend(group))

    def fit(self, xdata, ydata, wdata, coef, stream=0):
        return self.plan.fit(xdata, ydata, wdata, coef, stream)

    def close(self):
        self.plan.close()
        if self.nccl_comm:
            capi.rccl_comm_destroy(self.nccl_comm)
            self.nccl_comm = None
