"""Data-parallel training: one process per GPU, gradients summed over RCCL (``torch.distributed`` backend
"nccl" on ROCm) and averaged inside the Adam kernel.

The reference has no distributed code; the exchange step is this build's.  Every rank holds a replica
(weights, Adam moments), computes the mean-reduced loss on its shard of the minibatch and contributes its
flat gradient buffer to ONE sum all-reduce per optimizer step; dividing by the world size inside the Adam
launch (``grad_scale``) makes N ranks at batch B identical to one rank at batch N*B for the mean-type
losses used here (BatchNorm statistics stay per-rank unless ``sync_batchnorm`` is added).

Gradient messages are small (48 KB for the default CAEPadded, 18 MB for the context encoder), i.e. the
collective is latency-bound on xGMI; it is issued as a single bucket on the compute stream between the
captured forward/backward graph and the captured optimizer graph.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SPECTRUM_DIST_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run (ranks sharing one GPU)
            backend = os.environ.get("SPECTRUM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer.  ``all_reduce`` returns the factor the optimizer must apply
    to the summed gradient (1 / world_size)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.calls = 0
        self.bytes = 0
        # gloo (CPU rehearsals, ranks sharing one GPU) reduces host memory: device buffers are staged through a
        # pinned host copy.  RCCL ("nccl") reduces the device buffer in place on the stream.
        self._host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._host: Optional[torch.Tensor] = None

    def _reduce(self, flat: torch.Tensor, fn):
        if self._host_staged and flat.is_cuda:
            if self._host is None or self._host.numel() != flat.numel():
                self._host = torch.empty(flat.numel(), dtype=flat.dtype, pin_memory=True)
            self._host.copy_(flat.view(-1))  # synchronises with the producing stream
            fn(self._host)
            flat.view(-1).copy_(self._host, non_blocking=True)
        else:
            fn(flat)

    def all_reduce(self, flat: torch.Tensor) -> float:
        if self.world > 1:
            self._reduce(flat, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
        self.calls += 1
        self.bytes += flat.numel() * flat.element_size()
        return 1.0 / self.world

    def broadcast_(self, flat: torch.Tensor, src: int = 0):
        """Make every replica start from rank ``src``'s values (weights, BatchNorm state)."""
        if self.world > 1:
            self._reduce(flat, lambda t: dist.broadcast(t, src=src, group=self.group))


def shard_bounds(n: int, rank: int, world: int) -> tuple:
    """Contiguous, equal-size shards; the tail that does not divide evenly is dropped so that every rank
    runs the same number of steps (a rank with fewer batches would leave the others waiting in the
    collective)."""
    per = n // world
    return rank * per, (rank + 1) * per
