"""Data-parallel training: one process per GPU, gradients summed over RCCL (``torch.distributed`` backend
"nccl" on ROCm) and averaged inside the Adam kernel.

The reference has no distributed code; the exchange step is this build's.  Every rank holds a replica
(weights, Adam moments), computes the mean-reduced loss on its shard of the minibatch and contributes its
flat gradient buffer to ONE sum all-reduce per optimizer step; dividing by the world size inside the Adam
launch (``grad_scale``) makes N ranks at batch B identical to one rank at batch N*B for the mean-type
losses used here (BatchNorm statistics stay per-rank unless ``sync_batchnorm`` is added).

Gradient messages are small (48 KB for the default CAEPadded, 18 MB for the context encoder), i.e. the
collective is latency-bound on xGMI.  The flat buffer goes out in buckets of ``bucket_bytes``.  A model whose buffer is
ONE bucket reduces it on the compute stream after the backward pass; a larger one hands its buckets to a communication
stream DURING the backward pass (``BucketPipeline``): a bucket goes out as soon as the last gradient inside it has been
written - the decoder's first, since the pass runs from the output back - and the optimizer launch waits for the last
bucket, so all but the first layers' exchange hides behind the rest of the pass.  With RCCL
the collective is recorded INTO the step's HIP graph (after a probe that records and replays a small all-reduce and
that every rank must pass); otherwise (gloo rehearsals, a failed probe) the step is two graphs with the all-reduce
issued between them.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

from .nn import recording

_log = logging.getLogger(__name__)


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


class BucketPipeline:
    """The buckets of one flat gradient buffer on their way out during a backward pass.

    ``done(variables)`` is called (on the compute stream's host thread) when a layer's gradients are final on the
    compute stream; a bucket all of whose variables are done is reduced: on the communication stream behind an event
    (RCCL), or at once through host memory (gloo rehearsals).  ``finish()`` sends what is left and makes the compute
    stream wait for every bucket.  Elements of the buffer that belong to no variable (alignment gaps) count as done."""

    def __init__(self, reducer: "GradReducer", flat: torch.Tensor, variables):
        self.red, self.flat = reducer, flat.view(-1)
        per = max(4, (reducer.bucket_bytes // self.flat.element_size()) & ~3)
        n = self.flat.numel()
        self.bounds = [(lo, min(lo + per, n)) for lo in range(0, n, per)]
        self.pending = [0] * len(self.bounds)  # variables of each bucket not final yet
        self.where = {}                        # id(variable) -> buckets it overlaps
        for v in variables:
            ks = [k for k, (lo, hi) in enumerate(self.bounds) if v.offset < hi and v.offset + v.size > lo]
            self.where[id(v)] = ks
            for k in ks:
                self.pending[k] += 1
        self.sent = [False] * len(self.bounds)
        self.seen = set()
        self.order = []  # buckets in the order they went out
        self.comm = reducer.comm_stream(flat.device) if (flat.is_cuda and not reducer._host_staged) else None
        self.last = None

    def _send(self, k: int):
        lo, hi = self.bounds[k]
        self.sent[k] = True
        self.order.append(k)
        b = self.flat[lo:hi]
        if self.comm is None:
            self.red._reduce(b, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.red.group))
            return
        main = torch.cuda.current_stream(self.flat.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.red.group)
            self.last = torch.cuda.Event()
            self.last.record(self.comm)

    def done(self, variables):
        for v in variables:
            if id(v) in self.seen or id(v) not in self.where:
                continue
            self.seen.add(id(v))
            for k in self.where[id(v)]:
                self.pending[k] -= 1
        for k in range(len(self.bounds) - 1, -1, -1):  # the buffer's tail holds the layers the pass reaches first
            if not self.sent[k] and self.pending[k] == 0:
                self._send(k)

    def finish(self):
        for k in range(len(self.bounds) - 1, -1, -1):
            if not self.sent[k]:
                self._send(k)
        if self.last is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(self.last)


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer.  ``all_reduce`` returns the factor the optimizer must apply
    to the summed gradient (1 / world_size)."""

    def __init__(self, group=None, bucket_bytes: int = 8 << 20, force_collective: Optional[bool] = None,
                 sync_batchnorm: bool = False):
        self.group = group
        # BatchNormalization layers of a model this reducer is attached to take their batch statistics (and the two
        # sums of their backward pass) over the rows of every replica: N replicas at batch B equal one at batch N * B
        self.sync_batchnorm = bool(sync_batchnorm)
        self.overlap = True          # multi-bucket buffers go out during the backward pass (BucketPipeline)
        self._comm = {}              # device -> communication stream
        self._pipelines = {}         # data_ptr of a flat buffer -> its pipeline of the pass that just ran
        self.last_pipeline = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.calls = 0
        self.bytes = 0
        self.bucket_bytes = int(bucket_bytes)
        # issue the collective even in a one-rank group (tests of the recorded-collective path on a single GPU)
        self.force = (os.environ.get("SPECTRUM_DP_FORCE_COLLECTIVE") == "1") if force_collective is None \
            else bool(force_collective)
        self._capturable: Optional[bool] = None
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

    def _collective(self) -> bool:
        return dist.is_initialized() and (self.world > 1 or self.force)

    def buckets(self, flat: torch.Tensor):
        """Contiguous views of ``flat`` of at most ``bucket_bytes`` each (16-byte aligned cuts)."""
        v = flat.view(-1)
        per = max(4, (self.bucket_bytes // v.element_size()) & ~3)
        return [v[i:i + per] for i in range(0, v.numel(), per)]

    def comm_stream(self, device):
        key = torch.device(device).index
        if key not in self._comm:
            self._comm[key] = torch.cuda.Stream(device=device)
        return self._comm[key]

    def pipeline(self, flat: torch.Tensor, variables) -> Optional[BucketPipeline]:
        """A ``BucketPipeline`` for the backward pass about to run, or None when the buffer goes out in one piece (one
        bucket, no collective, overlap switched off, or a stream recording that cannot hold the collective)."""
        if not (self.overlap and self._collective()) or flat.numel() * flat.element_size() <= self.bucket_bytes:
            return None
        if flat.is_cuda and torch.cuda.is_current_stream_capturing() and (self._host_staged or not self.capturable(flat.device)):
            return None
        p = BucketPipeline(self, flat, variables)
        self._pipelines[flat.data_ptr()] = p
        return p

    def all_reduce(self, flat: torch.Tensor) -> float:
        p = self._pipelines.pop(flat.data_ptr(), None)
        if p is not None:  # the buckets left during the backward pass: send the rest, join the communication stream
            early = len(p.order)
            p.finish()
            self.last_pipeline = (tuple(p.order), early)  # (buckets in the order sent, how many left during the pass)
            self.note_reduced(flat)
            return 1.0 / self.world
        if self._collective():
            if self._host_staged and flat.is_cuda:
                self._reduce(flat, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
            else:
                for b in self.buckets(flat):
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
        self.note_reduced(flat)
        return 1.0 / self.world

    def note_reduced(self, flat: torch.Tensor, times: int = 1):
        """Book-keeping of reductions (called for each replay of a graph that holds the collective; ``times = -1``
        takes back the call that only recorded it)."""
        self.calls += times
        self.bytes += times * flat.numel() * flat.element_size()

    def capturable(self, device) -> bool:
        """Whether the all-reduce can be recorded into a HIP graph: RCCL only, and only after a probe - a small
        all-reduce recorded once and replayed twice must give world^2 on every rank.  ``SPECTRUM_DP_CAPTURE=0`` keeps
        the collective outside the graphs."""
        if self._capturable is not None:
            return self._capturable
        ok = False
        if (self._collective() and dist.get_backend(self.group) == "nccl" and torch.device(device).type == "cuda"
                and os.environ.get("SPECTRUM_DP_CAPTURE", "1") != "0"):
            flag = torch.ones(1, device=device)
            try:
                t = torch.ones(256, device=device)
                dist.all_reduce(t, group=self.group)  # the communicator must exist before anything is recorded
                torch.cuda.synchronize(device)
                t.fill_(1.0)
                g = torch.cuda.CUDAGraph()
                with recording(g, capture_error_mode="thread_local"):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                g.replay()
                g.replay()
                torch.cuda.synchronize(device)
                if float(t[0]) != float(self.world) ** 2 or float(t[-1]) != float(self.world) ** 2:
                    flag.zero_()
                self._probe_graph = g  # keep the recorded collective alive as long as its communicator
            except Exception as e:  # noqa: BLE001 - any failure means: keep the collective outside the graphs
                flag.zero_()
                _log.warning("GradReducer: recording the all-reduce into a HIP graph failed on this rank (%s: %s); "
                             "every rank keeps the collective between two graphs", type(e).__name__, e)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)  # one rank failing switches every rank
            ok = bool(flag.item() > 0)
            if not ok:
                _log.info("GradReducer: all-reduce stays outside the step graph (probe failed on some rank)")
        self._capturable = ok
        return ok

    def agree(self, ok: bool, device) -> bool:
        """True only when ``ok`` holds on EVERY rank (a MIN all-reduce, as in the probe): a decision each rank would
        otherwise take alone - such as falling back after a failed whole-step recording - must be the same everywhere,
        or some ranks would replay a graph holding the collective while others issue it eagerly."""
        if not self._collective():
            return bool(ok)
        flag = torch.full((1,), 1.0 if ok else 0.0, device=device)
        if self._host_staged and flag.is_cuda:
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0)

    def all_reduce_small(self, t: torch.Tensor):
        """In-place sum of a small device tensor over the replicas (BatchNorm's backward sums), on the current stream."""
        if self._collective():
            if self._host_staged and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def combine_moments(self, moments: torch.Tensor, eps: float):
        """``moments`` float64 [groups, 3, C] = (rows, mean, sum of squared deviations) of this replica.  Returns
        (mean float32 [groups, C], invstd float32 [groups, C], biased variance float64, rows of all replicas per group) of
        the rows of EVERY replica: count-weighted mean, and deviations + count * (mean - combined mean)^2, in float64."""
        if self._collective():
            if self._host_staged:  # gloo gathers host tensors, into a list
                h = moments.cpu() if moments.is_cuda else moments.contiguous()
                parts = [torch.empty_like(h) for _ in range(self.world)]
                dist.all_gather(parts, h, group=self.group)
                allm = torch.stack(parts).to(moments.device)
            else:
                allm = torch.empty((self.world,) + tuple(moments.shape), dtype=moments.dtype, device=moments.device)
                dist.all_gather_into_tensor(allm, moments.contiguous(), group=self.group)
        else:
            allm = moments.unsqueeze(0)
        n, mean_r, m2_r = allm[:, :, 0], allm[:, :, 1], allm[:, :, 2]
        total = n.sum(0)
        mean = (n * mean_r).sum(0) / total
        m2 = (m2_r + n * (mean_r - mean) ** 2).sum(0)
        var = m2 / total
        return mean.float(), torch.rsqrt(var + eps).float(), var, total

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


def run_dp_step(st, use_graph: bool, compute, pairs, device):
    """One data-parallel optimizer step of a multi-model training loop (the GAN steps).

    ``compute()`` enqueues everything up to and including the backward passes (no collective, no update);
    ``pairs`` = [(model, optimizer)] in update order; a model without a reducer is updated locally.  ``st`` carries
    ``warm`` and ``dp_graphs`` between calls.  First call eager; then either ONE recorded graph (compute, the bucketed
    all-reduces, the updates) when every reducer can be recorded, or two graphs with the all-reduces issued between
    them on the same stream."""
    def reduce():
        return [m._reducer.all_reduce(m._grads) if m._reducer is not None else 1.0 for m, _ in pairs]

    def update(scales):
        for (m, opt), s in zip(pairs, scales):
            opt.apply_model(m, s)

    if not use_graph or not st.warm:
        compute()
        update(reduce())
        st.warm = True
        return
    if getattr(st, "dp_graphs", None) is None:
        torch.cuda.synchronize()
        reds = [m._reducer for m, _ in pairs if m._reducer is not None]
        if all(r.capturable(device) for r in reds):
            g = torch.cuda.CUDAGraph()
            with recording(g, capture_error_mode="thread_local"):
                compute()
                update(reduce())
            for m, _ in pairs:
                if m._reducer is not None:
                    m._reducer.note_reduced(m._grads, -1)  # recording is not reducing
            st.dp_graphs = (g,)
        else:
            scales = [1.0 / m._reducer.world if m._reducer is not None else 1.0 for m, _ in pairs]
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with recording(g1, capture_error_mode="thread_local"):
                compute()
            with recording(g2, capture_error_mode="thread_local"):
                update(scales)
            st.dp_graphs = (g1, g2)
    st.dp_graphs[0].replay()
    if len(st.dp_graphs) == 2:
        reduce()
        st.dp_graphs[1].replay()
    else:
        for m, _ in pairs:
            if m._reducer is not None:
                m._reducer.note_reduced(m._grads)
