"""Layer engine behind the tf.keras-shaped model classes.

Every layer has an explicit ``forward`` and ``backward`` that enqueue kernels of
``libspectrum_hip.so`` on the current HIP stream; there is no autograd tracer.
Parameters of a model live in ONE flat float32 device buffer (gradients, Adam
moments likewise), so the optimizer step is a single launch and the
data-parallel exchange is a single all-reduce.  Activations and gradients live
in a per-model buffer pool keyed by (layer, role, call slot, shape): after the
first step at a given batch size nothing is allocated, which is what makes a
training step capturable into a HIP graph.

The classes mirror the tf.keras layers the reference builds its models from
(``autoencoder.py``, ``unet.py``, ``context_encoder.py``): same constructor
arguments, NHWC activations, Keras weight layouts and weight ordering.
"""
from __future__ import annotations

import contextlib
import gc

import os

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

DEFAULT_SEED = 1833  # global_vars.py:11


# --------------------------------------------------------------------------- small utilities
class DTensor(torch.Tensor):
    """Device tensor whose ``numpy()`` copies to the host, as tf.Tensor.numpy() does
    (run_context_encoder.py:406-410 prints ``loss.numpy()``)."""

    def numpy(self):  # noqa: D401
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor).cpu())


def as_dtensor(t: torch.Tensor) -> "DTensor":
    return t.as_subclass(DTensor)


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, (tuple, list)):
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _act_name(activation) -> Tuple[Optional[str], float]:
    """Normalise a keras ``activation=`` argument: a name, None, or a LeakyReLU layer instance."""
    if activation is None:
        return None, 0.0
    if isinstance(activation, LeakyReLU):
        return "lrelu", activation.alpha
    if isinstance(activation, str):
        a = activation.lower()
        if a in ("linear", "none"):
            return None, 0.0
        if a in ("relu", "elu", "tanh", "sigmoid"):
            return a, 1.0
        raise ValueError(f"unsupported activation {activation!r}")
    raise TypeError(f"unsupported activation object {activation!r}")


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("spectrum_vae_amd needs a GPU: the HIP kernels are the only execution path")
    return torch.device("cuda", torch.cuda.current_device())


class Variable:
    """One weight tensor: a view into the owning model's flat parameter buffer."""

    def __init__(self, name: str, shape: Sequence[int], init: str, trainable: bool = True):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.init = init
        self.trainable = trainable
        self.size = int(np.prod(self.shape))
        self.offset = -1
        self.data: Optional[torch.Tensor] = None  # view, set by Model._materialize
        self.grad: Optional[torch.Tensor] = None
        self.data_lp: Optional[torch.Tensor] = None  # bfloat16 copy (models whose compute dtype is bfloat16)
        self.owner = None

    def initial_value(self, rng: np.random.Generator) -> np.ndarray:
        if self.init == "zeros":
            return np.zeros(self.shape, np.float32)
        if self.init == "ones":
            return np.ones(self.shape, np.float32)
        if self.init == "glorot_uniform":
            if len(self.shape) == 2:
                fan_in, fan_out = self.shape
            else:
                rf = int(np.prod(self.shape[:-2]))
                fan_in, fan_out = self.shape[-2] * rf, self.shape[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, self.shape).astype(np.float32)
        if self.init == "normal0.02":  # tf.random_normal_initializer(0., 0.02), unet.py:36,65,121
            return (rng.standard_normal(self.shape) * 0.02).astype(np.float32)
        raise ValueError(self.init)

    def numpy(self) -> np.ndarray:
        return self.data.detach().cpu().numpy().copy()

    def assign(self, value):
        v = torch.as_tensor(np.asarray(value, dtype=np.float32)).reshape(self.shape)
        self.data.copy_(v.to(self.data.device))
        if self.owner is not None:
            self.owner._lp_dirty = True  # the bf16 copy of the weights is refreshed before the next forward pass

    def __repr__(self):
        return f"<Variable {self.name} {self.shape}>"


class BufferPool:
    """Persistent device buffers keyed by (owner, role, slot, shape, dtype)."""

    def __init__(self):
        self.bufs: Dict[tuple, torch.Tensor] = {}
        self._ws: Optional[torch.Tensor] = None
        self._ws_side: Dict[int, torch.Tensor] = {}
        self._retired: List[torch.Tensor] = []
        self.device: Optional[torch.device] = None
        # private mode (deferred reductions pending): every workspace request gets its own slice of an arena
        self._arena: Optional[torch.Tensor] = None
        self._bump: Optional[int] = None
        self._arena_peak = 0
        self._pass_total = 0

    def get(self, key, shape, dtype=torch.float32) -> torch.Tensor:
        k = (key, tuple(shape), dtype)
        t = self.bufs.get(k)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self.bufs[k] = t
        return t

    def workspace(self, nbytes: int) -> torch.Tensor:
        """Shared scratch for reductions; grows, never shrinks.  Kernels using it run on one stream."""
        nbytes = max(int(nbytes), 256)
        if self._bump is not None:
            return self._private(nbytes)
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None:
                # recorded HIP graphs hold the old buffer's raw address: it must stay allocated (and unused by anyone
                # else) for as long as they can be replayed, i.e. for the life of the pool
                self._retired.append(self._ws)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def begin_private(self):
        """From here to ``end_private`` every ``workspace`` request returns memory of its own (slices of one arena):
        kernels whose partial sums are reduced later, in one launch, must not share scratch.  The arena is sized from the
        largest pass seen so far, so from the second pass on the slices sit at fixed addresses (recordable)."""
        if self._arena is None or self._arena.numel() < self._arena_peak:
            if self._arena is not None:
                self._retired.append(self._arena)  # recorded graphs / queued kernels may still point into it
            self._arena = torch.empty(max(self._arena_peak, 1 << 20), dtype=torch.uint8, device=self.device)
        self._bump = 0
        self._pass_total = 0

    def end_private(self):
        self._arena_peak = max(self._arena_peak, self._pass_total)
        self._bump = None
        if self._arena.numel() < self._arena_peak and not torch.cuda.is_current_stream_capturing():
            # size the arena for the next pass now: that pass may be the one being recorded, which must not allocate
            self._retired.append(self._arena)
            self._arena = torch.empty(self._arena_peak, dtype=torch.uint8, device=self.device)

    def _private(self, nbytes: int) -> torch.Tensor:
        off = self._bump
        size = (nbytes + 255) & ~255
        if off + size > self._arena.numel():
            # outgrown during a first pass: earlier slices stay valid in the retired arena; the next pass starts with
            # an arena sized for everything this pass asked for
            self._retired.append(self._arena)
            self._arena = torch.empty(max(2 * self._arena.numel(), size), dtype=torch.uint8, device=self.device)
            off = 0
        self._bump = off + size
        self._pass_total += size
        return self._arena[off:off + nbytes]

    def workspace_side(self, lane: int, nbytes: int) -> torch.Tensor:
        """Scratch of the kernels issued on side stream ``lane`` (weight-gradient branch of the backward pass)."""
        nbytes = max(int(nbytes), 256)
        cur = self._ws_side.get(lane)
        if cur is None or cur.numel() < nbytes:
            if cur is not None:  # may still be in use by queued side-stream kernels: keep it alive
                self._retired.append(cur)
            cur = self._ws_side[lane] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return cur

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class CallCtx:
    """State of one forward call: what each layer saved for its backward."""

    def __init__(self, pool: BufferPool, slot: int, training: bool, dtype: torch.dtype = torch.float32):
        self.pool = pool
        self.slot = slot
        self.training = training
        self.dtype = dtype  # storage type of activations and their gradients in this call
        self.saved: Dict[int, dict] = {}
        # False: backward only propagates to the inputs (a generator's gradient THROUGH a discriminator);
        # weight/bias gradients are skipped, BatchNorm's go to scratch.
        self.param_grads = True
        # Extra HIP streams for the weight-gradient kernels: they only feed the optimizer, so they run beside the
        # input-gradient chain (and beside each other, round-robin) instead of in it.  The owner joins the
        # streams before the optimizer step.
        self.side: Optional[List[torch.cuda.Stream]] = None
        self.side_rr = 0
        # A list: fused conv layers leave their kernel-gradient partial sums in per-layer workspaces and append a job
        # here; the owner finishes all of them with one launch after the backward pass.  None: reduce per layer.
        # Set by Model.forward_raw(input_mask=...): the first conv layer reads the model input as x * (1 - mask)
        self.input_mask: Optional[torch.Tensor] = None
        self.wgrad_jobs: Optional[list] = None
        self.wgrad_finish: list = []
        # (partials, n, scale) when the forward launch also evaluated the loss and left per-workgroup sums to fold
        self.loss_fold: Optional[tuple] = None
        # [(device scalar, weight)]: terms the forward pass adds to the training loss (Keras ``add_loss``)
        self.extra_losses: list = []
        # BatchNorm statistics groups: the batch is this many equal consecutive parts, each normalised on its own (a
        # discriminator called on [real; generated] in one pass instead of twice)
        self.bn_groups = 1
        # a parallel.GradReducer with sync_batchnorm: BatchNorm statistics run over the rows of every replica
        self.sync_bn = None
        # callable(variables): a layer's gradients are final on the compute stream (a data-parallel bucket pipeline
        # sends the buckets they complete); set for one backward pass
        self.grad_ready = None
        # A list: fused 3x3 conv layers append their input-gradient job instead of launching it, and the owner issues the
        # whole chain as one launch (ops.conv2d_dgrad_chain).  Any layer that READS a gradient tensor launches what is
        # pending first (flush_dgrad); conv layers of the chain and layers that only pass the tensor on do not.
        self.dgrad_jobs: Optional[list] = None
        # the jobs the forward launch of this call already ran (Model._net_forward), as dgrad_signature() of the list
        self.dgrad_prelaunched = None
        # Model.forward_raw: runs of fused 3x3 convs inside a Sequential may go as one launch each (Sequential.forward)
        self.fuse_runs = False
        self.dgrad_last = None  # the list the last flush_dgrad of this call was asked to launch

    @staticmethod
    def dgrad_signature(jobs):
        """What identifies a job list to a launch: geometry, activation and the addresses of every tensor."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        return tuple((bytes(j["g"]), j["act"], j["alpha"], ptr(j["w"]), ptr(j["dy"]), ptr(j["y"]), ptr(j.get("idx")),
                      ptr(j["dx"])) for j in jobs)

    def flush_dgrad(self):
        """Launch the input-gradient jobs collected so far (one launch when the library has the chained form for them,
        else one per layer, in order); nothing to launch when the forward launch of this call ran exactly this list."""
        jobs = self.dgrad_jobs
        if not jobs:
            return
        self.dgrad_jobs = []
        self.dgrad_last = jobs
        done, self.dgrad_prelaunched = self.dgrad_prelaunched, None
        if done is not None and done == self.dgrad_signature(jobs):
            return
        for i in range(0, len(jobs), 8):  # a launch carries at most eight layers
            part = jobs[i:i + 8]
            if len(part) > 1 and ops.conv2d_dgrad_chain_supported(part):
                ops.conv2d_dgrad_chain(part)
                continue
            for j in part:
                ops.conv2d_bwd(j["g"], None, j["w"], j["dy"], j["y"], j.get("idx"), j["dx"], None, None, j["act"],
                               j["alpha"], None)

    def buf(self, layer, role, shape, dtype=None):
        return self.pool.get((id(layer), role, self.slot), shape, self.dtype if dtype is None else dtype)

    @property
    def lp(self) -> bool:
        return self.dtype != torch.float32

    def group_view(self, g: int) -> "CallCtx":
        """The part of this call that belongs to statistics group ``g``, as a call of its own: saved activations sliced
        to the group's samples, BatchNorm statistics to the group's row.  A backward pass through the view is the
        backward pass of a separate forward call on those samples."""
        groups = self.bn_groups
        v = CallCtx(self.pool, ("grp", g, self.slot), self.training, self.dtype)
        v.param_grads = self.param_grads
        v.sync_bn = self.sync_bn
        v.dgrad_jobs = self.dgrad_jobs
        v.fuse_runs = self.fuse_runs
        for key, d in self.saved.items():
            nd = {}
            for name, t in d.items():
                if not isinstance(t, torch.Tensor):
                    nd[name] = t
                elif name in ("mean", "invstd"):
                    nd[name] = t.view(groups, -1)[g]
                else:
                    b = t.shape[0] // groups
                    nd[name] = t[g * b:(g + 1) * b]
            v.saved[key] = nd
        return v

    def need_f32(self, layer):
        if self.lp:
            raise NotImplementedError(f"{type(layer).__name__} ({layer.name}) has no bfloat16 kernels: build the model "
                                      "with the float32 policy")


@contextlib.contextmanager
def recording(graph, **kw):
    """``torch.cuda.graph`` with the cyclic garbage collector paused: a finalizer that releases device objects (another
    model's graphs, events) in the middle of a stream capture aborts the process."""
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            gc.enable()


# --------------------------------------------------------------------------- base layer
class Layer:
    _counters: Dict[str, int] = {}

    def __init__(self, name: Optional[str] = None):
        cls = type(self).__name__
        if name is None:
            n = Layer._counters.get(cls, 0)
            Layer._counters[cls] = n + 1
            name = f"{cls.lower()}_{n}" if n else cls.lower()
        self.name = name
        self.built = False
        self.input_shape_: Optional[tuple] = None
        self.output_shape_: Optional[tuple] = None
        self._own_vars: List[Variable] = []

    # ---- building (pure host logic; runs without a GPU)
    def build(self, in_shape: tuple) -> tuple:
        """Create Variables for an input of ``in_shape`` (no batch axis); return the output shape."""
        in_shape = tuple(int(s) for s in in_shape)
        if self.built:
            if in_shape != self.input_shape_:
                raise ValueError(f"{self.name}: built for input {self.input_shape_}, called with {in_shape}")
            return self.output_shape_
        self.input_shape_ = in_shape
        self.output_shape_ = tuple(int(s) for s in self._build(in_shape))
        self.built = True
        return self.output_shape_

    def _build(self, in_shape):
        return in_shape

    def add_weight(self, suffix, shape, init, trainable=True) -> Variable:
        v = Variable(f"{self.name}/{suffix}", shape, init, trainable)
        self._own_vars.append(v)
        return v

    def sublayers(self) -> List["Layer"]:
        return []

    def all_variables(self) -> List[Variable]:
        out = list(self._own_vars)
        for l in self.sublayers():
            out.extend(l.all_variables())
        return out

    def count_params(self) -> int:
        return sum(v.size for v in self.all_variables())

    # ---- execution (HIP only)
    def forward(self, ctx: CallCtx, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def backward(self, ctx: CallCtx, dy: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        raise NotImplementedError


class InputLayer(Layer):
    """layers.Input(shape=...) placeholder at the head of the reference's Sequentials."""

    def __init__(self, shape=None, name=None):
        super().__init__(name)
        if isinstance(shape, int):
            shape = (shape,)
        self.shape = tuple(shape) if shape is not None else None

    def _build(self, in_shape):
        if self.shape is not None and tuple(self.shape) != tuple(in_shape):
            raise ValueError(f"Input layer expects {self.shape}, got {in_shape}")
        return in_shape

    def forward(self, ctx, x):
        return x

    def backward(self, ctx, dy, need_dx=True):
        return dy


def Input(shape=None, **_):
    return InputLayer(shape)


# --------------------------------------------------------------------------- convolution
class _ConvBase(Layer):
    transposed = False

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                 kernel_initializer="glorot_uniform", input_shape=None, name=None):
        super().__init__(name)
        self.filters = int(filters)  # Keras-2 casts float filter counts (autoencoder.py:508 divides with /=)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding.lower()
        self.act, self.alpha = _act_name(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.declared_input_shape = input_shape
        self.up = (1, 1)  # nearest-upsampling folded into this conv's gather (set by Sequential's planner)
        self.pool = None  # a following 2x2 MaxPooling2D computed in this conv's epilogue (set by the planner)
        self.fused_bwd = False  # whole backward (act', bias/kernel/input gradient) in the fused kernels
        self._geoms: Dict[int, ops.ConvGeom] = {}

    def _build(self, in_shape):
        if len(in_shape) != 3:
            raise ValueError(f"{self.name}: expected (H, W, C) input, got {in_shape}")
        h, w, cin = in_shape
        kh, kw = self.kernel_size
        sh, sw = self.strides
        cout = self.filters
        if self.transposed:
            ho = ops.deconv_out_size(h, kh, sh, self.padding)
            wo = ops.deconv_out_size(w, kw, sw, self.padding)
            self.kernel = self.add_weight("kernel", (kh, kw, cout, cin), self.kernel_initializer)
        else:
            g = ops.conv_geom(1, h * self.up[0], w * self.up[1], cin, kh, kw, sh, sw, self.padding, cout, self.up)
            ho, wo = g.Ho, g.Wo
            self.kernel = self.add_weight("kernel", (kh, kw, cin, cout), self.kernel_initializer)
        self.bias = self.add_weight("bias", (cout,), "zeros") if self.use_bias else None
        return (ho, wo, cout)

    def geom(self, n: int) -> ops.ConvGeom:
        g = self._geoms.get(n)
        if g is None:
            h, w, cin = self.input_shape_
            kh, kw = self.kernel_size
            sh, sw = self.strides
            if self.transposed:
                g = ops.deconv_geom(n, h, w, cin, kh, kw, sh, sw, self.padding, self.filters)
            else:
                g = ops.conv_geom(n, h * self.up[0], w * self.up[1], cin, kh, kw, sh, sw, self.padding, self.filters,
                                  self.up)
            self._geoms[n] = g
        return g

    def macs_per_sample(self) -> int:
        """Multiply-accumulates of the forward pass for one sample (the GEMM view of the layer)."""
        kh, kw = self.kernel_size
        if self.transposed:
            h, w, cin = self.input_shape_
            return h * w * cin * kh * kw * self.filters
        ho, wo, cout = self.output_shape_
        return ho * wo * cout * kh * kw * self.input_shape_[2]

    def plan_fusion(self, pool=None):
        """Decide (host-side, from the geometry alone) which fused kernels this layer uses."""
        if self.transposed:
            return False
        g = self.geom(1)
        if pool is not None and ops.conv2d_fused_supported(g, True):
            self.pool = pool
        self.fused_bwd = ops.conv2d_fused_supported(g, self.pool is not None)
        return self.pool is not None

    def _scratch(self, ctx, g, op):
        """The shared scratch sized for operation ``op`` of this layer (None when the geometry needs none): lets a
        result of few tiles cut its reduction over many workgroups."""
        cache = self.__dict__.setdefault("_scratch_bytes", {})
        key = (g.N, op)
        nbytes = cache.get(key)
        if nbytes is None:
            nbytes = cache[key] = ops.conv2d_scratch(g, op)
        return ctx.pool.workspace(nbytes) if nbytes else None

    def _w(self, ctx):
        """The filter tensor in the storage type of this call."""
        return self.kernel.data_lp if ctx.lp else self.kernel.data

    def forward(self, ctx, x):
        n = x.shape[0]
        g = self.geom(n)
        b = self.bias.data if self.bias is not None else None
        if ctx.lp and (self.pool is not None or self.up != (1, 1)):
            raise NotImplementedError(f"{self.name}: the fused 3x3 / 5x5 stride-1 family is float32 only")
        mask = ctx.input_mask  # the model input is to be read as x * (1 - mask): the first layer consumes the request
        ctx.input_mask = None
        if mask is not None and (self.transposed or ctx.lp or not self.masked_ok(g)):
            xm = ctx.buf(self, "x_masked", tuple(x.shape), torch.float32)  # geometry outside the folded form: one pass
            ops.mask_input(x, mask, xm)
            x, mask = xm, None
        if self.pool is not None:
            ho, wo, cout = self.output_shape_
            y = ctx.buf(self, "y", (n, ho // 2, wo // 2, cout))
            idx = ctx.buf(self, "idx", (n, ho // 2, wo // 2, cout), torch.uint8)
            ops.conv2d_pool_fwd(g, x, self.kernel.data, b, y, idx, self.act, self.alpha)
            if ctx.training:
                ctx.saved[id(self)] = {"x": x, "y": y, "idx": idx}
            return y
        y = ctx.buf(self, "y", (n,) + self.output_shape_)
        if self.transposed:
            ops.conv2d_transpose_fwd(g, x, self._w(ctx), b, y, self.act, self.alpha,
                                     self._scratch(ctx, g, ops.OP_TRANSPOSE_FWD))
        elif mask is not None:
            ops.conv2d_fwd_masked(g, x, mask, self.kernel.data, b, y, self.act, self.alpha, self._scratch(ctx, g, ops.OP_FWD))
        else:
            ops.conv2d_fwd(g, x, self._w(ctx), b, y, self.act, self.alpha, self._scratch(ctx, g, ops.OP_FWD))
        if ctx.training:
            ctx.saved[id(self)] = {"x": x, "y": y, "mask": mask}
        return y

    def masked_ok(self, g) -> bool:
        """The conv can read its input through a mask while gathering it (implicit-GEMM geometries, float32)."""
        cache = self.__dict__.setdefault("_masked_ok", {})  # per batch size: index limits depend on N
        ok = cache.get(g.N)
        if ok is None:
            ok = cache[g.N] = self.pool is None and self.up == (1, 1) and ops.conv2d_masked_supported(g)
        return ok

    def backward(self, ctx, dy, need_dx=True):
        s = ctx.saved[id(self)]
        x, y = s["x"], s["y"]
        n = x.shape[0]
        g = self.geom(n)
        ho, wo, cout = self.output_shape_
        fused = self.fused_bwd and ctx.param_grads and not ctx.lp
        s1 = self.__dict__.setdefault("_s1_bwd", {})  # per batch size
        if fused and n not in s1:
            s1[n] = ops.conv2d_s1_mfma_bwd(g)
        if ctx.dgrad_jobs and not (fused and not s1[n] and ctx.wgrad_jobs is not None and not ctx.side
                                   and self.kernel_size == (3, 3)):
            ctx.flush_dgrad()  # this layer's launches read dy now: the pending launches that produce it go first
        if fused:
            dx = ctx.buf(self, "dx", tuple(x.shape)) if need_dx else None
            db = self.bias.grad if self.bias is not None else None
            if s1[n] and not ctx.side:
                # 16..128-channel layers: one call forms dz once and runs both gradients on the matrix cores; the
                # workspace (dz + the kernel gradient's partial sums) is this layer's own until the deferred sums ran
                ws = ctx.buf(self, "wgrad_ws", (max(ops.conv2d_wgrad_workspace(g), 16),), torch.uint8)
                ops.conv2d_bwd(g, x, self.kernel.data, dy, y, s.get("idx"), dx, self.kernel.grad, db, self.act, self.alpha,
                               ws)
                return dx
            if ctx.wgrad_jobs is not None and not ctx.side:
                ws = ctx.buf(self, "wgrad_ws", (max(ops.conv2d_wgrad_workspace(g), 16),), torch.uint8)
                if self.kernel_size == (3, 3):
                    # launched by the owner, together with the other layers' jobs, once the whole dy chain exists
                    ctx.wgrad_jobs.append(dict(g=g, x=x, dy=dy, y=y, idx=s.get("idx"), act=self.act, alpha=self.alpha,
                                               dw=self.kernel.grad, db=db, ws=ws))
                else:
                    ops.conv2d_bwd_partial(g, x, dy, y, s.get("idx"), self.act, self.alpha, db is not None, ws)
                    ctx.wgrad_finish.append((g, ws, self.kernel.grad, db))
                if need_dx and ctx.dgrad_jobs is not None and self.kernel_size == (3, 3):
                    ctx.dgrad_jobs.append(dict(g=g, w=self.kernel.data, dy=dy, y=y, idx=s.get("idx"), dx=dx, act=self.act,
                                               alpha=self.alpha))
                elif need_dx:
                    ops.conv2d_bwd(g, None, self.kernel.data, dy, y, s.get("idx"), dx, None, None, self.act, self.alpha,
                                   None)
                return dx
            if not ctx.side or not need_dx:
                ws = ctx.pool.workspace(ops.conv2d_wgrad_workspace(g))
                ops.conv2d_bwd(g, x, self.kernel.data, dy, y, s.get("idx"), dx, self.kernel.grad, db, self.act,
                               self.alpha, ws)
                return dx
            # weight gradients on the side stream (after dy is complete), input gradient on the main stream
            ready = torch.cuda.Event()
            ready.record()
            lane = ctx.side_rr % len(ctx.side)
            ctx.side_rr += 1
            side = ctx.side[lane]
            ws = ctx.pool.workspace_side(lane, ops.conv2d_wgrad_workspace(g))
            with torch.cuda.stream(side):
                side.wait_event(ready)
                ops.conv2d_bwd(g, x, self.kernel.data, dy, y, s.get("idx"), None, self.kernel.grad, db, self.act,
                               self.alpha, ws)
            ops.conv2d_bwd(g, None, self.kernel.data, dy, y, s.get("idx"), dx, None, None, self.act, self.alpha, None)
            return dx
        if self.pool is not None:
            # unfused backward of a fused conv+pool forward: expand the pooled gradient first
            full = ctx.buf(self, "dy_full", (n, ho, wo, cout))
            yfull = ctx.buf(self, "y_full", (n, ho, wo, cout))
            ops.maxpool2d_bwd(dy, s["idx"], full, n, ho, wo, cout, 2, 2, 0, 0, ho // 2, wo // 2)
            ops.maxpool2d_bwd(y, s["idx"], yfull, n, ho, wo, cout, 2, 2, 0, 0, ho // 2, wo // 2)
            dy, y = full, yfull  # act'(0) * 0 == 0 at the non-argmax positions for every activation used
        rows = n * ho * wo
        dz = dy  # the activation backward runs in place on the incoming gradient buffer
        want_db = self.bias is not None and ctx.param_grads
        if s.get("dz_done"):
            pass  # the BatchNormalization that follows this layer emitted dz (and the bias gradient) from its own launch
        elif self.act is not None or want_db:
            ws = ctx.pool.workspace(ops.colreduce_workspace(rows, cout)) if want_db else None
            if self.act is not None:
                ops.act_bwd(dy, y, dz, self.bias.grad if want_db else None, True, rows, cout, self.act, self.alpha, ws)
            else:
                ops.bias_grad(dz, self.bias.grad, True, rows, cout, ws)
        if ctx.param_grads:
            ws = ctx.pool.workspace(ops.conv2d_wgrad_workspace(g))
            if self.transposed:
                ops.conv2d_transpose_wgrad(g, x, dz, self.kernel.grad, True, ws)
            elif s.get("mask") is not None:
                ops.conv2d_wgrad_masked(g, x, s["mask"], dz, self.kernel.grad, True, ws)
            else:
                ops.conv2d_wgrad(g, x, dz, self.kernel.grad, True, ws)
        if not need_dx:
            return None
        dx = ctx.buf(self, "dx", tuple(x.shape))
        if self.transposed:
            ops.conv2d_transpose_dgrad(g, dz, self._w(ctx), dx, self._scratch(ctx, g, ops.OP_TRANSPOSE_DGRAD))
        else:
            ops.conv2d_dgrad(g, dz, self._w(ctx), dx, self._scratch(ctx, g, ops.OP_DGRAD))
        if s.get("mask") is not None:  # d(x * (1 - mask)) / dx
            ops.mask_input(dx, s["mask"], dx)
        return dx


class Conv2D(_ConvBase):
    """layers.Conv2D (autoencoder.py:443,510,521; unet.py:40; context_encoder.py:57,72,172,185,230,242,334,346)."""


class Conv2DTranspose(_ConvBase):
    """layers.Conv2DTranspose (autoencoder.py:371,382; unet.py:69,123; context_encoder.py:92,105,119,260,276,290)."""
    transposed = True


# --------------------------------------------------------------------------- pooling / resampling
class MaxPooling2D(Layer):
    """layers.MaxPooling2D with the default stride == pool size (autoencoder.py:33,98,167,319,451)."""

    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", name=None):
        super().__init__(name)
        self.pool = _pair(pool_size)
        if strides is not None and _pair(strides) != self.pool:
            raise NotImplementedError("MaxPooling2D: only stride == pool size (the reference's use) is built")
        self.padding = padding.lower()
        self.folded = False  # True when the preceding Conv2D computes this pooling in its epilogue

    def _build(self, in_shape):
        h, w, c = in_shape
        ph, pw = self.pool
        if self.padding == "same":
            ho, self.pt, _ = ops.same_pads(h, ph, ph)
            wo, self.pl, _ = ops.same_pads(w, pw, pw)
        else:
            ho, wo, self.pt, self.pl = (h - ph) // ph + 1, (w - pw) // pw + 1, 0, 0
        if ho <= 0 or wo <= 0:
            raise ValueError(f"{self.name}: pooling {self.pool} of a {h}x{w} map leaves nothing")
        return (ho, wo, c)

    def forward(self, ctx, x):
        if self.folded:
            return x
        ctx.need_f32(self)
        n = x.shape[0]
        h, w, c = self.input_shape_
        ho, wo, _ = self.output_shape_
        y = ctx.buf(self, "y", (n, ho, wo, c))
        idx = ctx.buf(self, "idx", (n, ho, wo, c), torch.uint8) if ctx.training else None
        ops.maxpool2d_fwd(x, y, idx, n, h, w, c, self.pool[0], self.pool[1], self.pt, self.pl, ho, wo)
        if ctx.training:
            ctx.saved[id(self)] = {"idx": idx}
        return y

    def backward(self, ctx, dy, need_dx=True):
        if not need_dx:
            return None
        if self.folded:
            return dy
        n = dy.shape[0]
        h, w, c = self.input_shape_
        ho, wo, _ = self.output_shape_
        dx = ctx.buf(self, "dx", (n, h, w, c))
        ops.maxpool2d_bwd(dy, ctx.saved[id(self)]["idx"], dx, n, h, w, c, self.pool[0], self.pool[1], self.pt,
                          self.pl, ho, wo)
        return dx


class UpSampling2D(Layer):
    """layers.UpSampling2D, nearest (autoencoder.py:364,503)."""

    def __init__(self, size=(2, 2), name=None):
        super().__init__(name)
        self.size = _pair(size)
        self.folded = False  # True when the following Conv2D gathers through the repetition itself

    def _build(self, in_shape):
        h, w, c = in_shape
        return (h * self.size[0], w * self.size[1], c)

    def forward(self, ctx, x):
        if self.folded:
            return x
        ctx.need_f32(self)
        n = x.shape[0]
        h, w, c = self.input_shape_
        y = ctx.buf(self, "y", (n,) + self.output_shape_)
        ops.upsample2d_fwd(x, y, n, h, w, c, self.size[0], self.size[1])
        return y

    def backward(self, ctx, dy, need_dx=True):
        if self.folded or not need_dx:
            return dy if need_dx else None
        n = dy.shape[0]
        h, w, c = self.input_shape_
        dx = ctx.buf(self, "dx", (n, h, w, c))
        ops.upsample2d_bwd(dy, dx, n, h, w, c, self.size[0], self.size[1])
        return dx


# --------------------------------------------------------------------------- normalisation / activations
class BatchNormalization(Layer):
    """layers.BatchNormalization() with Keras-2 defaults: axis -1, momentum 0.99, epsilon 1e-3
    (autoencoder.py:318,450,468; unet.py:51; context_encoder.py:64,88,179,237,267,283,341).

    ``unbiased_moving_var``: the fused kernel Keras 2 uses on 4-D inputs feeds the unbiased batch
    variance into the moving variance; 2-D inputs use the biased one.  Only inference after training
    sees the difference."""

    def __init__(self, momentum=0.99, epsilon=1e-3, unbiased_moving_var: Optional[bool] = None, name=None):
        super().__init__(name)
        self.momentum = float(momentum)
        self.epsilon = float(epsilon)
        self._unbiased = unbiased_moving_var

    def _build(self, in_shape):
        c = in_shape[-1]
        self.gamma = self.add_weight("gamma", (c,), "ones")
        self.beta = self.add_weight("beta", (c,), "zeros")
        self.moving_mean = self.add_weight("moving_mean", (c,), "zeros", trainable=False)
        self.moving_variance = self.add_weight("moving_variance", (c,), "ones", trainable=False)
        self.unbiased = (len(in_shape) == 3) if self._unbiased is None else bool(self._unbiased)
        return in_shape

    # set by Sequential's planner:
    fused_act = None   # (act, alpha) of the activation layer that follows: applied in this layer's launches
    prev_conv = None   # the conv in front whose activation backward (+ bias gradient) this layer's backward carries

    def forward(self, ctx, x):
        c = self.input_shape_[-1]
        rows = x.numel() // c
        y = ctx.buf(self, "y", tuple(x.shape))
        act, alpha = self.fused_act if self.fused_act is not None else (None, 0.0)
        if ctx.training:
            g = ctx.bn_groups
            if x.shape[0] % g:
                raise ValueError(f"{self.name}: a batch of {x.shape[0]} does not split into {g} statistics groups")
            mean = ctx.buf(self, "mean", (g, c), torch.float32)
            invstd = ctx.buf(self, "invstd", (g, c), torch.float32)
            ws = ctx.pool.workspace(ops.batchnorm_workspace(rows, c, g))
            if ctx.sync_bn is not None:
                # statistics over the rows of every replica: local moments, the replicas' exchange, then the apply launch
                mom = ctx.buf(self, "moments", (g, 3, c), torch.float64)
                ops.batchnorm_sync_moments(x, mom, rows, c, g, ws)
                m_all, inv_all, var, total = ctx.sync_bn.combine_moments(mom, self.epsilon)
                mean.copy_(m_all)
                invstd.copy_(inv_all)
                for k in range(g):  # one moving-statistics update per group, in order
                    v = var[k] * total[k] / (total[k] - 1.0) if self.unbiased else var[k]
                    self.moving_mean.data.mul_(self.momentum).add_(m_all[k], alpha=1.0 - self.momentum)
                    self.moving_variance.data.mul_(self.momentum).add_(v.float(), alpha=1.0 - self.momentum)
                ops.batchnorm_sync_fwd_apply(x, self.gamma.data, self.beta.data, mean, invstd, y, rows, c, g, act, alpha)
                ctx.saved[id(self)] = {"x": x, "y": y, "mean": mean, "invstd": invstd,
                                       "all_rows": rows // g * ctx.sync_bn.world}
                return y
            ops.batchnorm_act_fwd_train(x, self.gamma.data, self.beta.data, y, mean, invstd, self.moving_mean.data,
                                        self.moving_variance.data, rows, c, self.epsilon, self.momentum, self.unbiased,
                                        g, act, alpha, ws)
            ctx.saved[id(self)] = {"x": x, "y": y, "mean": mean, "invstd": invstd}
        else:
            ops.batchnorm_act_fwd_infer(x, self.gamma.data, self.beta.data, self.moving_mean.data,
                                        self.moving_variance.data, y, rows, c, self.epsilon, act, alpha)
        return y

    def backward(self, ctx, dy, need_dx=True):
        s = ctx.saved[id(self)]
        x = s["x"]
        c = self.input_shape_[-1]
        rows = x.numel() // c
        g = s["mean"].numel() // c
        dx = ctx.buf(self, "dx", tuple(x.shape))
        ws = ctx.pool.workspace(ops.batchnorm_workspace(rows, c, g))
        if ctx.param_grads:
            dg, db, acc = self.gamma.grad, self.beta.grad, True
        else:
            dg, db, acc = (ctx.buf(self, "dg_scratch", (c,), torch.float32),
                           ctx.buf(self, "db_scratch", (c,), torch.float32), False)
        act_out, alpha_out = self.fused_act if self.fused_act is not None else (None, 0.0)
        conv = self.prev_conv
        act_in, alpha_in, dbias = None, 0.0, None
        if (conv is not None and not conv.fused_bwd and conv.pool is None and id(conv) in ctx.saved
                and ctx.saved[id(conv)]["y"].data_ptr() == x.data_ptr()):
            act_in, alpha_in = conv.act, conv.alpha
            if conv.bias is not None and ctx.param_grads:
                dbias = conv.bias.grad
            ctx.saved[id(conv)]["dz_done"] = True
        if "all_rows" in s:  # statistics over every replica: so are the two sums of the input gradient
            sums = ctx.buf(self, "bwd_sums", (g, 2, c), torch.float32)
            yact = s["y"] if act_out is not None else None
            ops.batchnorm_sync_bwd_sums(x, dy, yact, act_out, alpha_out, s["mean"], s["invstd"], sums, dg, db, acc, rows, c,
                                        g, ws)
            ctx.sync_bn.all_reduce_small(sums)
            ops.batchnorm_sync_bwd_apply(x, dy, yact, act_out, alpha_out, self.gamma.data, s["mean"], s["invstd"], sums,
                                         s["all_rows"], dx, act_in, alpha_in, dbias, True, rows, c, g, ws)
            return dx
        ops.batchnorm_act_bwd(x, dy, s["y"] if act_out is not None else None, act_out, alpha_out, self.gamma.data,
                              s["mean"], s["invstd"], dx, dg, db, acc, act_in, alpha_in, dbias, True, rows, c, g, ws)
        return dx


class Activation(Layer):
    def __init__(self, activation, name=None):
        super().__init__(name)
        self.act, self.alpha = _act_name(activation)

    folded = False  # applied inside the BatchNormalization in front of it (set by Sequential's planner)

    def forward(self, ctx, x):
        if self.act is None or self.folded:
            return x
        y = ctx.buf(self, "y", tuple(x.shape))
        ops.act_fwd(x, y, self.act, self.alpha)
        if ctx.training:
            ctx.saved[id(self)] = {"y": y}
        return y

    def backward(self, ctx, dy, need_dx=True):
        if self.act is None or not need_dx or self.folded:
            return dy if need_dx else None
        y = ctx.saved[id(self)]["y"]
        c = y.shape[-1]
        ops.act_bwd(dy, y, dy, None, False, y.numel() // c, c, self.act, self.alpha, None)
        return dy


class LeakyReLU(Activation):
    """layers.LeakyReLU(alpha) as a layer (unet.py:53,79) or as an ``activation=`` argument
    (context_encoder.py:61,177,234,246,339)."""

    def __init__(self, alpha=0.3, name=None):
        Layer.__init__(self, name)
        self.act, self.alpha = "lrelu", float(alpha)


# --------------------------------------------------------------------------- dense / reshaping
class Flatten(Layer):
    def _build(self, in_shape):
        return (int(np.prod(in_shape)),)

    def forward(self, ctx, x):
        return x.view(x.shape[0], -1)

    def backward(self, ctx, dy, need_dx=True):
        return dy.view((dy.shape[0],) + self.input_shape_) if need_dx else None


class Reshape(Layer):
    def __init__(self, target_shape, name=None):
        super().__init__(name)
        self.target_shape = tuple(int(s) for s in target_shape)

    def _build(self, in_shape):
        if int(np.prod(in_shape)) != int(np.prod(self.target_shape)):
            raise ValueError(f"{self.name}: cannot reshape {in_shape} into {self.target_shape}")
        return self.target_shape

    def forward(self, ctx, x):
        return x.view((x.shape[0],) + self.target_shape)

    def backward(self, ctx, dy, need_dx=True):
        return dy.view((dy.shape[0],) + self.input_shape_) if need_dx else None


class Dense(Layer):
    """layers.Dense (autoencoder.py:38-40,107,110,176,179,222-241,330-352,466-492): x W + b, W (in, out)."""

    def __init__(self, units, activation=None, name=None):
        super().__init__(name)
        self.units = int(units)
        self.act, self.alpha = _act_name(activation)

    def _build(self, in_shape):
        if len(in_shape) != 1:
            raise ValueError(f"{self.name}: Dense is built for flattened inputs, got {in_shape}")
        self.kernel = self.add_weight("kernel", (in_shape[0], self.units), "glorot_uniform")
        self.bias = self.add_weight("bias", (self.units,), "zeros")
        return (self.units,)

    def macs_per_sample(self) -> int:
        return self.input_shape_[0] * self.units

    def forward(self, ctx, x):
        ctx.need_f32(self)
        m, k = x.shape
        y = ctx.buf(self, "y", (m, self.units))
        ops.dense_fwd(x, self.kernel.data, self.bias.data, y, m, k, self.units, self.act, self.alpha)
        if ctx.training:
            ctx.saved[id(self)] = {"x": x, "y": y}
        return y

    def backward(self, ctx, dy, need_dx=True):
        s = ctx.saved[id(self)]
        x, y = s["x"], s["y"]
        m, k = x.shape
        n = self.units
        ws = ctx.pool.workspace(ops.colreduce_workspace(m, n))
        pg = ctx.param_grads
        if self.act is not None:
            ops.act_bwd(dy, y, dy, self.bias.grad if pg else None, True, m, n, self.act, self.alpha, ws)
        elif pg:
            ops.bias_grad(dy, self.bias.grad, True, m, n, ws)
        if pg:
            nbytes = ops.dense_wgrad_workspace(m, k, n)
            # the layer's own scratch (a deferred bracket sums the partial sums at the end of the backward pass)
            wws = ctx.buf(self, "wgrad_ws", (nbytes,), torch.uint8) if nbytes else None
            ops.dense_wgrad(x, dy, self.kernel.grad, True, m, k, n, wws)
        if not need_dx:
            return None
        dx = ctx.buf(self, "dx", (m, k))
        ops.dense_dgrad(dy, self.kernel.data, dx, m, k, n)
        return dx


class Dropout(Layer):
    """layers.Dropout (autoencoder.py:108,177,223,331): inverted dropout in training, identity otherwise.
    The keep mask is a pure function of (seed, call counter, element index), so backward regenerates it."""

    def __init__(self, rate, seed: Optional[int] = None, name=None):
        super().__init__(name)
        self.rate = float(rate)
        self.seed = DEFAULT_SEED if seed is None else int(seed)
        self.calls = 0
        # position of this layer in its model (set when the model lays out its parameters): the per-layer random
        # stream is derived from it, so two builds of the same model draw the same masks
        self.stream_id = 0

    def forward(self, ctx, x):
        if not ctx.training or self.rate <= 0.0:
            return x
        ctx.need_f32(self)
        y = ctx.buf(self, "y", tuple(x.shape))
        offset = self.calls * x.numel()
        self.calls += 1
        ops.dropout_fwd(x, y, self.rate, self.seed + 7919 * self.stream_id, offset)
        ctx.saved[id(self)] = {"offset": offset}
        return y

    def backward(self, ctx, dy, need_dx=True):
        if not need_dx:
            return None
        if self.rate <= 0.0:
            return dy
        dx = ctx.buf(self, "dx", tuple(dy.shape))
        ops.dropout_bwd(dy, dx, self.rate, self.seed + 7919 * self.stream_id, ctx.saved[id(self)]["offset"])
        return dx


class ReduceSumH(Layer):
    """layers.Lambda(lambda x: tf.reduce_sum(x, axis=1)) (autoencoder.py:36)."""

    def _build(self, in_shape):
        return (in_shape[1], in_shape[2])

    def forward(self, ctx, x):
        ctx.need_f32(self)
        n = x.shape[0]
        h, w, c = self.input_shape_
        y = ctx.buf(self, "y", (n, w, c))
        ops.reduce_sum_h_fwd(x, y, n, h, w, c)
        return y

    def backward(self, ctx, dy, need_dx=True):
        if not need_dx:
            return None
        n = dy.shape[0]
        h, w, c = self.input_shape_
        dx = ctx.buf(self, "dx", (n, h, w, c))
        ops.reduce_sum_h_bwd(dy, dx, n, h, w, c)
        return dx


# --------------------------------------------------------------------------- containers
class Sequential(Layer):
    """tf.keras.Sequential: a list of layers run in order, with ``summary()``."""

    def __init__(self, layers: Optional[Sequence[Layer]] = None, name=None, fold_upsampling: bool = True,
                 fuse: bool = True):
        super().__init__(name)
        self.layers: List[Layer] = []
        self.fold_upsampling = fold_upsampling
        self.fuse = fuse  # conv+pool forward fusion and fused conv backward where the kernels cover the shape
        for l in layers or []:
            self.add(l)

    def add(self, layer: Layer):
        if self.built:
            raise RuntimeError("cannot add layers to a built Sequential")
        self.layers.append(layer)

    def sublayers(self):
        return self.layers

    def declared_input_shape(self) -> Optional[tuple]:
        if self.layers and isinstance(self.layers[0], InputLayer) and self.layers[0].shape is not None:
            return tuple(self.layers[0].shape)
        if self.layers and getattr(self.layers[0], "declared_input_shape", None) is not None:
            return tuple(self.layers[0].declared_input_shape)
        return None

    def _build(self, in_shape):
        shape = in_shape
        for i, l in enumerate(self.layers):
            if (self.fold_upsampling and isinstance(l, UpSampling2D) and i + 1 < len(self.layers)
                    and type(self.layers[i + 1]) is Conv2D and not self.layers[i + 1].built):
                # UpSampling2D directly followed by a Conv2D: the conv reads through the repetition, the
                # upsampled tensor is never written (and its backward never read).
                l.folded = True
                l.build(shape)
                self.layers[i + 1].up = l.size
                continue  # the conv sees the *stored* (small) tensor's shape
            shape = l.build(shape)
        if self.fuse and os.environ.get("SPECTRUM_BN_FUSE", "1") != "0":
            for i, l in enumerate(self.layers):
                if not isinstance(l, BatchNormalization):
                    continue
                nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
                if type(nxt) in (Activation, LeakyReLU) and nxt.act is not None:
                    l.fused_act = (nxt.act, nxt.alpha)  # BatchNormalization -> activation in one launch (unet.py:51-53)
                    nxt.folded = True
                prv = self.layers[i - 1] if i > 0 else None
                if isinstance(prv, _ConvBase) and prv.act is not None:
                    l.prev_conv = prv  # conv(activation) -> BatchNormalization (context_encoder.py:61-64), backward side
        if self.fuse:
            for i, l in enumerate(self.layers):
                if type(l) is not Conv2D:
                    continue
                nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
                pool = nxt if (isinstance(nxt, MaxPooling2D) and nxt.pool == (2, 2) and nxt.pt == 0 and nxt.pl == 0
                               and l.output_shape_[0] % 2 == 0 and l.output_shape_[1] % 2 == 0) else None
                if l.plan_fusion(pool):
                    pool.folded = True
        return shape

    def _conv_runs(self):
        """Maximal runs of two or more 3x3 stride-1 convs of the fused family (their folded pooling / upsampling layers
        and rate-0 dropouts between them): {index of the first layer: (index behind the run, its conv layers)}."""
        runs = self.__dict__.get("_runs")
        if runs is not None:
            return runs
        runs, i, n = {}, 0, len(self.layers)
        while i < n:
            convs, j, last = [], i, i
            while j < n:
                l = self.layers[j]
                if (type(l) is Conv2D and l.fused_bwd and l.kernel_size == (3, 3) and l.up[0] == l.up[1]
                        and l.strides == (1, 1) and l.padding == "same"):
                    convs.append(l)
                    last = j + 1
                    if l.pool is not None:  # its folded pooling layer belongs to the run
                        last = j + 2
                        j += 1
                elif not ((isinstance(l, UpSampling2D) and l.folded) or (isinstance(l, Dropout) and l.rate == 0)):
                    break
                j += 1
            if len(convs) >= 2:
                # a folded upsampling in front of the first conv is inside the run; one behind the last conv is not
                runs[i] = (last, convs)
                i = last
            else:
                i += 1
        self.__dict__["_runs"] = runs
        return runs

    def _run_forward(self, ctx, x, convs):
        """The convs of a run as one launch (ops.convnet_fwd); None when the library has no such launch for them."""
        n = x.shape[0]
        owner = convs[0].kernel.owner
        if owner is None or any(l.kernel.owner is not owner for l in convs):
            return None
        plans = self.__dict__.setdefault("_run_plans", {})
        key = (id(convs[0]), n, ctx.slot if not isinstance(ctx.slot, int) else "train%d" % ctx.slot)
        plan = plans.get(key)
        if plan is None and any(ops.conv2d_s1_mfma_bwd(l.geom(n)) for l in convs):
            plan = plans[key] = False  # wide layers: their own matrix-core kernels beat the launch that carries a whole run
        if plan is None:
            specs, bufs = [], []
            for l in convs:
                ho, wo, cout = l.output_shape_
                shp = (n, ho // 2, wo // 2, cout) if l.pool is not None else (n, ho, wo, cout)
                y = ctx.buf(l, "y", shp)
                idx = ctx.buf(l, "idx", shp, torch.uint8) if l.pool is not None else None
                specs.append(dict(Cin=l.input_shape_[2], Cout=cout, K=3, up=l.up[0], pool=int(l.pool is not None),
                                  act=l.act, alpha=l.alpha, w_off=l.kernel.offset,
                                  b_off=l.bias.offset if l.bias is not None else -1, y=y, idx=idx))
                bufs.append((l, y, idx))
            arr = ops.net_layers(specs)
            h, w = convs[0].input_shape_[0], convs[0].input_shape_[1]
            plan = plans[key] = (arr, bufs, h, w) if ops.convnet_supported(arr, n, h, w) else False
        if plan is False:
            return None
        arr, bufs, h, w = plan
        ops.convnet_fwd(arr, n, h, w, x, owner._params)
        prev = x
        for l, y, idx in bufs:
            if ctx.training:
                ctx.saved[id(l)] = {"x": prev, "y": y, "idx": idx} if idx is not None else {"x": prev, "y": y}
            prev = y
        return prev

    def forward(self, ctx, x):
        runs = self._conv_runs() if (ctx.fuse_runs and not ctx.lp and self.built) else {}
        skip_to = 0
        for i, l in enumerate(self.layers):
            if i < skip_to:
                continue
            if i in runs and ctx.input_mask is None and x.dtype == torch.float32:
                y = self._run_forward(ctx, x, runs[i][1])
                if y is not None:
                    x, skip_to = y, runs[i][0]
                    continue
            if ctx.input_mask is not None and type(l) is not Conv2D and not isinstance(l, (Sequential, InputLayer)):
                # a pending input mask (Model.forward_raw) and a first layer that cannot read through it: one pass
                mask, ctx.input_mask = ctx.input_mask, None
                xm = ctx.buf(self, "x_masked", tuple(x.shape), torch.float32)
                ops.mask_input(x, mask, xm)
                if ctx.training:
                    ctx.saved[("input_mask", id(self))] = mask
                x = xm
            x = l.forward(ctx, x)
        return x

    def backward(self, ctx, dy, need_dx=True):
        first = 0
        while first < len(self.layers) and not self.layers[first].all_variables():
            first += 1  # layers before the first weighted one only matter if the caller wants dx
        for i in range(len(self.layers) - 1, -1, -1):
            want = need_dx or i > first
            l = self.layers[i]
            if ctx.dgrad_jobs and not (isinstance(l, (Sequential, InputLayer, _ConvBase)) or getattr(l, "folded", False)):
                ctx.flush_dgrad()  # this layer reads dy: the launches that produce it go first
            dy = l.backward(ctx, dy, need_dx=want)
            if ctx.grad_ready is not None:
                vs = self.layers[i].all_variables()
                if vs:
                    ctx.grad_ready(vs)
            if dy is None:
                return None
        mask = ctx.saved.get(("input_mask", id(self)))
        if mask is not None:  # d(x * (1 - mask)) / dx
            ops.mask_input(dy, mask, dy)
        return dy

    # ---- keras-style introspection
    def summary_rows(self):
        rows = []
        for l in self.layers:
            if isinstance(l, InputLayer):
                continue
            shp = l.output_shape_
            if isinstance(l, UpSampling2D) and l.folded:
                shp = l.output_shape_
            elif isinstance(l, Conv2D) and l.up != (1, 1):
                shp = l.output_shape_
            rows.append((f"{l.name} ({type(l).__name__})", (None,) + tuple(shp), l.count_params()))
        return rows

    def summary(self, print_fn=print):
        if not self.built:
            raise ValueError("This model has not yet been built. Call build_graph() first.")
        rows = self.summary_rows()
        print_fn(f'Model: "{self.name}"')
        print_fn("_" * 65)
        print_fn(f"{'Layer (type)':<29}{'Output Shape':<26}{'Param #':<10}")
        print_fn("=" * 65)
        for n, s, p in rows:
            print_fn(f"{n:<29}{str(s):<26}{p:<10}")
        total = self.count_params()
        trainable = sum(v.size for v in self.all_variables() if v.trainable)
        print_fn("=" * 65)
        print_fn(f"Total params: {total}")
        print_fn(f"Trainable params: {trainable}")
        print_fn(f"Non-trainable params: {total - trainable}")
        print_fn("_" * 65)

    def __call__(self, x, training: bool = False):
        owner = getattr(self, "_owner_model", None)
        if owner is None:
            raise RuntimeError(f"{self.name} is not attached to a built Model")
        return owner._run_sub(self, x, training)
