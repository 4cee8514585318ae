"""tf.keras.Model look-alike: build / call / compile / fit / predict on the HIP layer engine.

Reference surface being mirrored (what the three drivers call): ``model.build_graph((1,H,W,C))``,
``model.encoder.summary()``, ``model(x, training=True)``, ``model.trainable_variables``,
``model.compile(optimizer=, loss=)``, ``model.fit(X, Y, epochs=, shuffle=, validation_data=, callbacks=,
batch_size=)`` -> ``.history["loss"|"val_loss"]``, ``model.predict(X, batch_size=32)`` -> NumPy
(run_model.py:65-72,140-148,281-300,356-357; run_context_encoder.py:130-139,242-258,399,491).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import losses as losses_mod
from . import ops
from .losses import to_device_f32
from .nn import (DEFAULT_SEED, BufferPool, CallCtx, DTensor, Layer, Sequential, Variable, as_dtensor, default_device,
                 recording)

ALIGN = 8  # floats; every variable starts on a 16-byte boundary of the flat buffer AND of its bf16 copy


class History:
    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []

    def record(self, epoch, **vals):
        self.epoch.append(epoch)
        for k, v in vals.items():
            self.history.setdefault(k, []).append(float(v))


class EarlyStopping:
    """tf.keras.callbacks.EarlyStopping(monitor=, patience=) (run_model.py:278)."""

    def __init__(self, monitor="val_loss", patience=0, min_delta=0.0):
        self.monitor, self.patience, self.min_delta = monitor, int(patience), abs(float(min_delta))
        self.best = math.inf
        self.wait = 0
        self.stopped_epoch = 0

    def on_train_begin(self):
        self.best, self.wait, self.stopped_epoch = math.inf, 0, 0

    def on_epoch_end(self, epoch: int, logs: Dict[str, float]) -> bool:
        cur = logs.get(self.monitor)
        if cur is None:
            return False
        self.wait += 1
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            return True
        return False


class Model(Layer):
    """Base of every model class.  Subclasses create their sublayers in ``__init__`` and implement
    ``_build`` (shape propagation), ``forward`` and ``backward``; a Sequential-only model can simply
    set ``self._stack``."""

    def __init__(self, name: Optional[str] = None, seed: int = DEFAULT_SEED):
        super().__init__(name)
        self._seed = seed
        self._pool = BufferPool()
        from . import mixed_precision
        # storage type of activations / activation gradients / the filter copy the convs read (the policy in force
        # when the model is constructed); weights, gradients and optimizer state are float32 either way
        self._act_dtype: torch.dtype = mixed_precision.global_policy().torch_dtype
        self._params_lp: Optional[torch.Tensor] = None
        self._lp_dirty = True
        self._params: Optional[torch.Tensor] = None
        self._grads: Optional[torch.Tensor] = None
        self._grads_clean = False  # True only right after an optimizer launch that cleared the gradient buffer
        self._state: Optional[torch.Tensor] = None
        self._calls: List[CallCtx] = []
        self._infer_slot = 0
        self.optimizer = None
        self.loss = None
        self._reducer = None  # data-parallel gradient reducer (parallel.GradReducer)
        self._graphs: Dict[tuple, object] = {}
        self.use_graphs = True
        # Run the whole forward as ONE launch when the model is a plain chain of fusable 3x3 convs (see
        # ``_conv_chain``); the per-layer buffers are filled exactly as the per-layer path fills them.
        self.use_net_fused = True
        # one launch per backward pass for all kernel / bias gradient slab sums (SPECTRUM_DEFER_REDUCES=0: per layer)
        self.defer_reduces = os.environ.get("SPECTRUM_DEFER_REDUCES", "1") != "0"
        self.fuse_loss = True  # MeanSquaredError evaluated inside the one-launch forward, folded by the optimizer launch
        # workgroups per layer when >= 3 layers share the batched kernel-gradient launch (128 each measured best at batch
        # 1024: tools/wgrad_ab.py sweeps other splits)
        self.wgrad_batch_blocks_large = 128  # layers of >= 256 pixels
        self.wgrad_batch_blocks = 128
        # The input gradients of consecutive few-channel 3x3 layers in one launch (CallCtx.dgrad_jobs).  Off unless the
        # model's backward() reads gradient tensors only inside Sequentials (which launch what is pending first) or
        # behind a ctx.flush_dgrad() of its own; such a class says so with chain_dgrad_ok = True.
        self.chain_dgrad = bool(getattr(type(self), "chain_dgrad_ok", False))
        self.fuse_fwd_dgrad = True   # ... and that launch behind the forward pass, in the forward launch
        self._dgrad_plans = {}       # (batch, slot) -> the job list of the last backward pass that was one chain
        self.defer_wgrad_reduce = True  # one reduction launch for the kernel gradients of all fused conv layers
        self._chain_cache: Dict[int, object] = {}
        self._slabs_wanted = False   # set around the backward pass of a training step whose optimizer launch can sum slabs
        self._pending_slabs = None   # the table that backward pass left (ops.conv2d_wgrad_batch_slabs)
        self._rows = None            # set around a training step whose forward launch gathers its own batch (b["rows"])
        self.gather_in_forward = True
        # Optional: run the weight-gradient kernels, the gradient zero-fill and the optimizer's schedule update on
        # extra HIP streams beside the forward / input-gradient chain (fit / train_on_device_batch only).  Off by
        # default: at the shapes measured so far every conv launch already occupies the LDS of all CUs, so the
        # branches do not overlap and the extra dependencies only add latency.
        self.use_side_stream = False
        self.n_side_streams = 3
        self._side: Optional[List[torch.cuda.Stream]] = None
        self.stop_training = False

    # ------------------------------------------------------------------ structure
    def sublayers(self) -> List[Layer]:
        out = []
        for v in self.__dict__.values():
            if isinstance(v, Layer) and v is not self:
                out.append(v)
            elif isinstance(v, (list, tuple)):
                out.extend(l for l in v if isinstance(l, Layer))
        return out

    def build(self, input_shape):
        """Keras ``Model.build`` takes the shape WITH the batch axis."""
        shape = tuple(input_shape)
        if len(shape) >= 2 and (shape[0] is None or len(shape) in (2, 3, 4)):
            shape = shape[1:]
        return Layer.build(self, shape)

    def build_graph(self, test_shape):
        """Shape-propagation smoke test of the reference (autoencoder.py:538-547 and siblings)."""
        self.build(tuple(test_shape))

    @property
    def variables(self) -> List[Variable]:
        return self.all_variables()

    @property
    def trainable_variables(self) -> List[Variable]:
        return [v for v in self.all_variables() if v.trainable]

    @property
    def non_trainable_variables(self) -> List[Variable]:
        return [v for v in self.all_variables() if not v.trainable]

    def summary(self, print_fn=print):
        if not self.built:
            raise ValueError("This model has not yet been built. Call build_graph() first.")
        print_fn(f'Model: "{self.name}"')
        for l in self.sublayers():
            if isinstance(l, Sequential):
                l.summary(print_fn)
        print_fn(f"Total params: {self.count_params()}")

    # ------------------------------------------------------------------ parameters on the device
    def _materialize(self, device: Optional[torch.device] = None):
        if self._params is not None:
            return
        if not self.built:
            raise ValueError("build the model (build_graph / first call) before running it")
        device = device or default_device()
        self._pool.device = device
        tv, ntv = self.trainable_variables, self.non_trainable_variables
        rng = np.random.default_rng(self._seed)

        def layout(vs):
            off = 0
            for v in vs:
                v.offset = off
                off += -(-v.size // ALIGN) * ALIGN
            return max(off, ALIGN)

        n_t, n_s = layout(tv), layout(ntv)
        host_p = np.zeros(n_t, np.float32)
        for v in tv:
            host_p[v.offset:v.offset + v.size] = v.initial_value(rng).ravel()
        host_s = np.zeros(n_s, np.float32)
        for v in ntv:
            host_s[v.offset:v.offset + v.size] = v.initial_value(rng).ravel()
        self._params = torch.from_numpy(host_p).to(device)
        self._grads = torch.zeros(n_t, dtype=torch.float32, device=device)
        self._state = torch.from_numpy(host_s).to(device)
        if self._act_dtype != torch.float32:
            self._params_lp = torch.empty(n_t, dtype=self._act_dtype, device=device)
        for v in tv:
            v.data = self._params[v.offset:v.offset + v.size].view(v.shape)
            v.grad = self._grads[v.offset:v.offset + v.size].view(v.shape)
            if self._params_lp is not None:
                v.data_lp = self._params_lp[v.offset:v.offset + v.size].view(v.shape)
            v.owner = self
        self._lp_dirty = True
        for v in ntv:
            v.data = self._state[v.offset:v.offset + v.size].view(v.shape)
            v.owner = self
        for i, l in enumerate(self._all_layers()):
            if isinstance(l, Sequential):
                l._owner_model = self
            if hasattr(l, "stream_id"):
                l.stream_id = i

    @property
    def compute_dtype(self) -> str:
        return "bfloat16" if self._act_dtype == torch.bfloat16 else "float32"

    def _refresh_lp(self):
        """Re-derive the bf16 copy of the weights from the float32 masters (one launch over the flat buffer)."""
        if self._params_lp is not None:
            ops.cast(self._params, self._params_lp)
        self._lp_dirty = False

    def _to_act(self, t: torch.Tensor, key) -> torch.Tensor:
        """A float32 tensor in this model's activation type (a pooled buffer holds the converted copy)."""
        if t.dtype == self._act_dtype:
            return t
        out = self._pool.get(("act_in", key), tuple(t.shape), self._act_dtype)
        ops.cast(t, out)
        return out

    def _all_layers(self) -> List[Layer]:
        out, stack = [], [self]
        while stack:
            l = stack.pop()
            out.append(l)
            stack.extend(l.sublayers())
        return out

    def get_weights(self) -> List[np.ndarray]:
        self._materialize()
        return [v.numpy() for v in self.variables]

    def set_weights(self, weights: Sequence[np.ndarray]):
        self._materialize()
        vs = self.variables
        if len(weights) != len(vs):
            raise ValueError(f"expected {len(vs)} arrays, got {len(weights)}")
        for v, w in zip(vs, weights):
            if tuple(np.shape(w)) != v.shape:
                raise ValueError(f"{v.name}: expected shape {v.shape}, got {np.shape(w)}")
            v.assign(w)

    def set_trainable_weights(self, weights: Sequence[np.ndarray]):
        self._materialize()
        tv = self.trainable_variables
        if len(weights) != len(tv):
            raise ValueError(f"expected {len(tv)} arrays, got {len(weights)}")
        for v, w in zip(tv, weights):
            v.assign(w)

    def state_dict(self) -> dict:
        self._materialize()
        return {"params": self._params.cpu().clone(), "state": self._state.cpu().clone()}

    def load_state_dict(self, sd: dict):
        self._materialize()
        self._params.copy_(sd["params"].to(self._params.device))
        self._state.copy_(sd["state"].to(self._state.device))
        self._lp_dirty = True

    # ------------------------------------------------------------------ execution
    def _prep_input(self, x) -> torch.Tensor:
        if isinstance(x, (list, tuple)):
            raise TypeError(f"{self.name}: list inputs are not accepted here")
        t = to_device_f32(x, self._pool.device or default_device())
        if not self.built:
            self.build(tuple(t.shape))
        self._materialize(t.device)
        if tuple(t.shape[1:]) != self.input_shape_:
            raise ValueError(f"{self.name}: input {tuple(t.shape[1:])} does not match the built shape {self.input_shape_}")
        return t

    def __call__(self, x, training: bool = False):
        """Eager call: returns a copy of the output (internal buffers are reused by later calls).
        A training-mode call is remembered until ``zero_grad``/the optimizer step so ``backward`` can run."""
        t = self._prep_input(x)
        y = self.forward_raw(t, training)
        if y.dtype != torch.float32:  # callers get float32, as a Keras mixed-precision model's float32 output layer gives
            out = torch.empty(tuple(y.shape), dtype=torch.float32, device=y.device)
            ops.cast(y, out)
            return as_dtensor(out)
        return as_dtensor(y.clone())

    def call(self, x, training: bool = False):
        return self.__call__(x, training=training)

    # ------------------------------------------------------------------ whole-network forward
    def _sequentials(self) -> Optional[List[Sequential]]:
        """The Sequential stacks whose composition IS this model's forward, or None when the forward is not a chain."""
        st = getattr(self, "_stack", None)
        return [st] if isinstance(st, Sequential) else None

    def _conv_chain(self):
        """Conv layers in execution order when every layer of the forward is a fused 3x3 stride-1 SAME Conv2D (with its
        folded UpSampling2D / MaxPooling2D neighbours), else None."""
        from .nn import Conv2D, Dropout, InputLayer, MaxPooling2D, UpSampling2D
        seqs = self._sequentials()
        if not seqs:
            return None
        chain = []
        for sq in seqs:
            for l in sq.layers:
                if isinstance(l, InputLayer) or (isinstance(l, (MaxPooling2D, UpSampling2D)) and l.folded):
                    continue
                if isinstance(l, Dropout) and l.rate == 0:
                    continue
                if (type(l) is Conv2D and l.fused_bwd and l.kernel_size == (3, 3) and l.up[0] == l.up[1]
                        and l.kernel.owner is self):
                    chain.append(l)
                    continue
                return None
        return chain or None

    def _net_forward(self, ctx, x: torch.Tensor, mse=None):
        """One-launch forward; returns None when the model (or this batch size) is outside the fused family.
        ``mse = (y_true, dpred, partials)``: the launch also evaluates the squared-error loss against ``y_true``
        (gradient into ``dpred``, per-workgroup sums into ``partials``; ``ctx.loss_fold`` then describes the pending
        fold ``(partials, n, 1 / count)``)."""
        n = x.shape[0]
        key = (n, ctx.slot if not isinstance(ctx.slot, int) else "train%d" % ctx.slot)
        plan = self._chain_cache.get(key)
        if plan is None:
            chain = self._conv_chain()
            plan = False
            if chain:
                specs, bufs, prev = [], [], x
                for l in chain:
                    ho, wo, cout = l.output_shape_
                    shp = (n, ho // 2, wo // 2, cout) if l.pool is not None else (n, ho, wo, cout)
                    y = ctx.buf(l, "y", shp)
                    idx = ctx.buf(l, "idx", shp, torch.uint8) if l.pool is not None else None
                    specs.append(dict(Cin=l.input_shape_[2], Cout=cout, K=3, up=l.up[0], pool=int(l.pool is not None),
                                      act=l.act, alpha=l.alpha, w_off=l.kernel.offset,
                                      b_off=l.bias.offset if l.bias is not None else -1, y=y, idx=idx))
                    bufs.append((l, y, idx))
                arr = ops.net_layers(specs)
                h, w = chain[0].input_shape_[0], chain[0].input_shape_[1]
                if ops.convnet_supported(arr, n, h, w):
                    plan = (arr, bufs, h, w)
            self._chain_cache[key] = plan
        if plan is False:
            return None
        arr, bufs, h, w = plan
        # The input-gradient jobs the backward pass of an earlier step with these buffers issued (same batch size and slot
        # -> same pool buffers -> same addresses): when that pass follows this forward call directly, as in a training
        # step, the forward launch runs them too and the backward pass finds them done (CallCtx.flush_dgrad compares the
        # list it collects with this one and launches it itself if they differ).
        jobs = self._dgrad_plans.get(key) if (ctx.training and mse is not None and self.fuse_fwd_dgrad) else None
        if jobs is not None and not (jobs[0]["dy"].data_ptr() == mse[1].data_ptr()
                                     and ops.convnet_fwd_mse_dgrad_supported(arr, n, h, w, jobs)):
            jobs = None
        if self._rows is not None:
            # the launch gathers the batch into x (fit / rows_step_supported chose this form for this chain)
            if not (mse is not None and mse[0] is x and ops.convnet_rows_supported(arr, n, h, w)):
                raise RuntimeError("the forward launch cannot gather this batch; call rows_step_supported() first")
            y_true, dpred, partials = mse
            r = self._rows
            rows = ops.batch_rows(r["X"], r["index"], x, self.optimizer._ensure_state(x.device), r["base"], r["stride"])
            if jobs is not None:
                cnt = ops.convnet_fwd_mse_dgrad(arr, n, h, w, None, rows, self._params, y_true, dpred, 1.0, partials, jobs)
                ctx.dgrad_prelaunched = ctx.dgrad_signature(jobs)
            else:
                cnt = ops.convnet_fwd_mse_rows(arr, n, h, w, rows, self._params, y_true, dpred, 1.0, partials)
            ctx.loss_fold = (partials, cnt, 1.0 / y_true.numel())
        elif mse is not None and bufs[-1][2] is None and tuple(mse[0].shape) == tuple(bufs[-1][1].shape):
            y_true, dpred, partials = mse
            if jobs is not None:
                cnt = ops.convnet_fwd_mse_dgrad(arr, n, h, w, x, None, self._params, y_true, dpred, 1.0, partials, jobs)
                ctx.dgrad_prelaunched = ctx.dgrad_signature(jobs)
            else:
                cnt = ops.convnet_fwd_mse(arr, n, h, w, x, self._params, y_true, dpred, 1.0, partials)
            ctx.loss_fold = (partials, cnt, 1.0 / y_true.numel())
        else:
            ops.convnet_fwd(arr, n, h, w, x, self._params)
        if ctx.training:
            ctx.net = (plan, x)
        prev = x
        for l, y, idx in bufs:
            if ctx.training:
                ctx.saved[id(l)] = {"x": prev, "y": y, "idx": idx} if idx is not None else {"x": prev, "y": y}
            prev = y
        return prev

    def forward_raw(self, x: torch.Tensor, training: bool, side=None, mse=None, bn_groups: int = 1,
                    input_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward on a prepared device tensor; the result aliases the model's buffer pool.  ``side``: stream
        the backward pass of this call may use for its weight-gradient kernels (the caller joins it).  ``mse``: see
        ``_net_forward``; honoured only when the whole forward is the one-launch kernel (check ``ctx.loss_fold``).  The
        gradient buffer in it must reach ``backward_raw`` as the launch left it: the launch may already have run the
        input-gradient chain from it.
        ``input_mask`` (float32, shape of x): the model sees x * (1 - input_mask) (run_context_encoder.py:242-243); when
        the first layer is a convolution of the implicit-GEMM family the product is formed inside its gathers (forward
        and kernel gradient) and the masked tensor never exists, otherwise it costs one elementwise pass."""
        if training:
            ctx = CallCtx(self._pool, len(self._calls), True, self._act_dtype)
            ctx.side = side
            self._calls.append(ctx)
        else:
            ctx = CallCtx(self._pool, "infer", False, self._act_dtype)
        fold_mask = input_mask is not None and self._act_dtype == torch.float32
        if input_mask is not None and not fold_mask:  # bf16 policy: one elementwise pass in float32, ahead of the cast
            xm = self._pool.get(("input_masked", ctx.slot), tuple(x.shape), torch.float32)
            ops.mask_input(x, input_mask, xm)
            x = xm
        if self._act_dtype != torch.float32:
            if self._lp_dirty:
                self._refresh_lp()
            x = self._to_act(x, ctx.slot)
        ctx.loss_fold = None
        ctx.fuse_runs = bool(self.use_net_fused)
        if training and self._reducer is not None and getattr(self._reducer, "sync_batchnorm", False):
            ctx.sync_bn = self._reducer
        ctx.bn_groups = int(bn_groups)  # the batch is this many independent BatchNorm groups (see CallCtx)
        self._last_ctx = ctx
        if fold_mask:  # the first layer that runs takes the request (a Conv2D folds it into its gathers)
            ctx.input_mask = input_mask
            y = self.forward(ctx, x)
            if ctx.input_mask is not None:
                raise RuntimeError("forward_raw: no layer consumed input_mask (the model does not start with a layer stack)")
            return y
        if self.use_net_fused and not ctx.lp:
            y = self._net_forward(ctx, x, mse)
            if y is not None:
                return y
        return self.forward(ctx, x)

    def backward_raw(self, dy: torch.Tensor, call: int = -1, need_dx: bool = False):
        """Backward of the ``call``-th remembered training call; parameter gradients accumulate.  The per-layer
        "sum the partial slabs into dw / db" steps of the strided / transposed conv family are queued and run as ONE
        launch at the end of the pass (every layer then works in scratch of its own)."""
        ctx = self._calls[call]
        # data-parallel, gradient buffer of several buckets: the buckets leave for the all-reduce as the pass completes
        # them, which needs every gradient written by its own layer's launches (no batched or deferred reductions)
        pipe = None
        if self._reducer is not None and ctx.param_grads and ctx.side is None and len(self._calls) == 1:
            pipe = self._reducer.pipeline(self._grads, self.trainable_variables)
        if pipe is not None:
            ctx.grad_ready = pipe.done
            try:
                return self._backward_raw(ctx, dy, need_dx)
            finally:
                ctx.grad_ready = None
        if not (self.defer_reduces and ctx.param_grads and ctx.side is None):
            return self._backward_raw(ctx, dy, need_dx)
        ops.deferred_reduce_begin()
        self._pool.begin_private()
        try:
            dx = self._backward_raw(ctx, dy, need_dx)
        except BaseException:
            ops.deferred_reduce_flush(False)
            self._pool.end_private()
            raise
        ops.deferred_reduce_flush(True)
        self._pool.end_private()
        return dx

    def _backward_raw(self, ctx, dy: torch.Tensor, need_dx: bool):
        self._grads_clean = False
        if self.defer_wgrad_reduce and ctx.param_grads and ctx.wgrad_jobs is None and ctx.grad_ready is None:
            ctx.wgrad_jobs, ctx.wgrad_finish = [], []
            if self.chain_dgrad and ctx.dgrad_jobs is None:
                ctx.dgrad_jobs = []  # consecutive 3x3 layers' input gradients as one launch
            dx = self.backward(ctx, dy, need_dx=need_dx)
            whole = bool(ctx.dgrad_jobs) and ctx.dgrad_last is None  # nothing was launched on the way
            ctx.flush_dgrad()
            if whole and ctx.loss_fold is not None and getattr(ctx, "net", None) is not None:
                # the pass was forward launch -> loss gradient -> this one chain: the next step's forward launch takes it
                n = dy.shape[0]
                key = (n, ctx.slot if not isinstance(ctx.slot, int) else "train%d" % ctx.slot)
                self._dgrad_plans[key] = ctx.dgrad_last
            ctx.dgrad_jobs = None
            nb = self.wgrad_batch_blocks if len(ctx.wgrad_jobs) >= 3 else 0
            for j in ctx.wgrad_jobs:  # layers with large images: fewer workgroups, each staging more samples
                j["blocks"] = (self.wgrad_batch_blocks_large if j["g"].H * j["g"].W >= 256 else nb) if nb else 0
            if not ctx.wgrad_finish and self._slabs_wanted and self._slabs_cover(ctx.wgrad_jobs, nb):
                # the optimizer launch that follows reads its gradients as sums of the slabs: no summing launch
                self._pending_slabs = ops.conv2d_wgrad_batch_slabs(ctx.wgrad_jobs, blocks=nb)
            elif not ctx.wgrad_finish:  # nothing else waits for the finishing launch: the batch sums its own slabs
                ops.conv2d_wgrad_batch(ctx.wgrad_jobs, reduce=True, blocks=nb)
            else:
                ops.conv2d_wgrad_batch(ctx.wgrad_jobs, reduce=False, blocks=nb)
                ops.conv2d_wgrad_finish(ctx.wgrad_finish + [(j["g"], j["ws"], j["dw"], j["db"], j["blocks"]) for j in ctx.wgrad_jobs])
            ctx.wgrad_jobs = None
            return dx
        return self.backward(ctx, dy, need_dx=need_dx)

    def _slabs_cover(self, jobs, nb: int) -> bool:
        """The kernel-gradient jobs of this backward pass can stay slabs for the optimizer launch: the library has that
        form for all of them and together they are every trainable variable of the model (nothing else writes
        gradients), decided once per batch size."""
        if not jobs:
            return False
        key = ("slabs", jobs[0]["g"].N, len(jobs))
        ok = self._chain_cache.get(key)
        if ok is None:
            covered = sum(j["dw"].numel() + (j["db"].numel() if j.get("db") is not None else 0) for j in jobs)
            ok = (covered == sum(v.size for v in self.trainable_variables)
                  and ops.conv2d_wgrad_batch_slabs_supported(jobs, blocks=nb))
            self._chain_cache[key] = ok
        return ok

    def zero_grad(self):
        self._materialize()
        self._grads.zero_()
        self._calls.clear()

    def _run_sub(self, seq: Sequential, x, training: bool):
        """``model.encoder(x)`` / ``model.decoder(z)``: run one sub-stack on its own."""
        t = to_device_f32(x, self._pool.device or default_device())
        self._materialize(t.device)
        ctx = CallCtx(self._pool, ("sub", id(seq)), False, self._act_dtype)
        if training:
            raise NotImplementedError("sub-model calls are inference-only; train through the owning model")
        if ctx.lp:
            if self._lp_dirty:
                self._refresh_lp()
            t = self._to_act(t, ctx.slot)
        y = seq.forward(ctx, t)
        if y.dtype != torch.float32:
            out = torch.empty(tuple(y.shape), dtype=torch.float32, device=y.device)
            ops.cast(y, out)
            return as_dtensor(out)
        return as_dtensor(y.clone())

    # default forward/backward for models that are one stack of layers
    def forward(self, ctx, x):
        return self._stack.forward(ctx, x)

    def backward(self, ctx, dy, need_dx=True):
        return self._stack.backward(ctx, dy, need_dx)

    def _build(self, in_shape):
        return self._stack.build(in_shape)

    # ------------------------------------------------------------------ compile / fit / predict
    def compile(self, optimizer=None, loss=None, **_):
        self.optimizer = optimizer
        self.loss = losses_mod.get(loss) if loss is not None else None
        self._graphs.clear()

    def distribute(self, reducer):
        """Attach a data-parallel gradient reducer (see ``parallel.GradReducer``)."""
        self._reducer = reducer
        self._graphs.clear()

    def rows_step_supported(self, X: torch.Tensor, bs: int, same_xy: bool = True) -> bool:
        """Whether a training step at batch size ``bs`` can read its batch straight out of the resident float32 tensor
        ``X`` through a row index (``b["rows"]`` for ``train_on_device_batch``): the one-launch forward with the fused
        squared-error loss on an autoencoder target, one stream, no gradient reducer, an optimizer with a device-side
        update count."""
        if not (self.gather_in_forward and same_xy and X.is_cuda and X.dtype == torch.float32 and self.use_net_fused):
            return False
        if self._reducer is not None or self._side_stream() is not None or self._act_dtype != torch.float32:
            return False
        if not (type(self.loss) is losses_mod.MeanSquaredError and self.fuse_loss and hasattr(self.optimizer, "_ensure_state")):
            return False
        if tuple(X.shape[1:]) != tuple(self.input_shape_) or tuple(self.output_shape_) != tuple(self.input_shape_):
            return False
        chain = self._conv_chain()
        if not chain:
            return False
        specs = [dict(Cin=l.input_shape_[2], Cout=l.output_shape_[2], K=3, up=l.up[0], pool=int(l.pool is not None),
                      act=l.act, alpha=l.alpha, w_off=l.kernel.offset, b_off=l.bias.offset if l.bias is not None else -1,
                      y=None, idx=None) for l in chain]
        h, w = chain[0].input_shape_[0], chain[0].input_shape_[1]
        return chain[-1].pool is None and ops.convnet_rows_supported(ops.net_layers(specs), bs, h, w)

    def step_rows(self, X: torch.Tensor, index: torch.Tensor, stride: int) -> dict:
        """The ``b["rows"]`` entry for steps that walk ``index`` (int64 rows of ``X``) ``stride`` positions at a time.
        ``restart_rows`` marks the step about to run as position 0."""
        base = torch.zeros(1, dtype=torch.int64, device=X.device)
        return {"X": X, "index": index, "base": base, "stride": int(stride)}

    def restart_rows(self, rows: dict):
        """The next step reads ``index[0:batch]``: remembers the optimizer's update count (a device-to-device copy)."""
        rows["base"].copy_(self.optimizer._ensure_state(rows["X"].device)[0:1])

    def _step_buffers(self, bs: int, y_shape: tuple):
        p = self._pool
        return {
            "x": p.get(("fit", "x"), (bs,) + self.input_shape_),
            "y": p.get(("fit", "y"), (bs,) + tuple(y_shape)),
            "dpred": p.get(("fit", "dpred"), (bs,) + self.output_shape_, self._act_dtype),
            "loss": p.get(("fit", "loss"), (1,)),
            "acc": p.get(("fit", "acc"), (1,)),
        }

    def _side_stream(self):
        if not self.use_side_stream or not self._params.is_cuda:
            return None
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self._params.device) for _ in range(self.n_side_streams)]
        return self._side

    def _fwd_bwd(self, b: dict, same_xy: bool, weight: float, prepare_opt: bool = False):
        """zero grads -> forward -> loss (+dL/dpred) -> backward, enqueued on the current stream, with the
        work that is off the critical path (zero-fill, schedule update, weight gradients) forked onto the
        side stream and joined before returning."""
        side = self._side_stream()
        if side is None:
            if not self._grads_clean:  # the previous step's optimizer launch left the buffer cleared
                self.zero_grad_raw()
            self._grads_clean = False
            self._calls.clear()
            if prepare_opt:
                self.optimizer.prepare(self._params.device)
        else:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            zeroed = torch.cuda.Event()
            for i, sd in enumerate(side):
                with torch.cuda.stream(sd):
                    sd.wait_event(fork)
                    if i == 0:
                        self.zero_grad_raw()
                        zeroed.record(sd)
                        if prepare_opt:
                            self.optimizer.prepare(self._params.device)
                    else:
                        sd.wait_event(zeroed)
        yt = b["x"] if same_xy else b["y"]
        mse = None
        if type(self.loss) is losses_mod.MeanSquaredError and self.fuse_loss and self._act_dtype == torch.float32:
            mse = (yt, b["dpred"], self._pool.get(("fit", "loss_partials"), (ops.loss_workspace_floats(),)))
        self._rows = b.get("rows")
        try:
            pred = self.forward_raw(b["x"], True, side, mse)
        finally:
            self._rows = None
        fold = self._calls[-1].loss_fold
        extras = self._calls[-1].extra_losses
        if fold is not None:  # loss and its gradient left with the forward launch; the optimizer launch folds the sums
            fold = fold + (b["loss"], b["acc"], weight)
        elif extras:  # terms the forward pass contributed (a VAE's KL): loss = compiled loss + sum(weight * term)
            self.loss.value_and_grad(yt, pred, b["loss"], b["dpred"], 1.0)
            for term, w in extras:
                ops.axpy(b["loss"], term, w)
            ops.axpy(b["acc"], b["loss"], weight)
        elif getattr(self.loss, "fuses_running_sum", False):
            self.loss.value_and_grad(yt, pred, b["loss"], b["dpred"], 1.0, running=b["acc"], running_weight=weight)
        else:
            self.loss.value_and_grad(yt, pred, b["loss"], b["dpred"], 1.0)
            ops.axpy(b["acc"], b["loss"], weight)
        if side is not None:
            main.wait_event(zeroed)
        # without a reducer (which needs the summed gradients in the flat buffer) the optimizer launch may take the
        # kernel gradients as slabs of partial sums
        self._slabs_wanted = side is None and self._reducer is None and hasattr(self.optimizer, "apply_model")
        self._pending_slabs = None
        try:
            self.backward_raw(b["dpred"], -1, need_dx=False)
        finally:
            self._slabs_wanted = False
        if side is not None:
            for sd in side:
                done = torch.cuda.Event()
                done.record(sd)
                main.wait_event(done)
        self._calls.clear()
        return fold

    def _train_step_body(self, b: dict, same_xy: bool, weight: float):
        if self._reducer is None:
            side = self._side_stream() is not None  # with side streams the schedule update runs beside the forward pass
            fold = self._fwd_bwd(b, same_xy, weight, prepare_opt=side)
            slabs, self._pending_slabs = self._pending_slabs, None
            self.optimizer.apply_model(self, 1.0, prepared=side, zero_grads=True, fold=fold, slabs=slabs)
            return
        fold = self._fwd_bwd(b, same_xy, weight)
        scale = self._reducer.all_reduce(self._grads)
        self.optimizer.apply_model(self, scale, zero_grads=True, fold=fold)

    def zero_grad_raw(self):
        ops.fill(self._grads, 0.0)
        self._grads_clean = False  # only the optimizer launch of a training step vouches for a cleared buffer
        self._calls.clear()

    def train_on_device_batch(self, b: dict, same_xy: bool, weight: float):
        """One optimizer step on the batch already sitting in ``b['x']`` / ``b['y']``.  The first step at a
        batch size runs eagerly (it allocates every buffer), the second is captured into HIP graphs, later
        ones replay them.  With a data-parallel reducer the step is two graphs (forward/backward, optimizer)
        with the gradient all-reduce issued between them on the same stream."""
        bs = b["x"].shape[0]
        key = (bs, same_xy, float(weight), id(self.optimizer), id(self.loss), id(self._reducer))
        if b.get("rows") is not None:
            r = b["rows"]
            key += (r["X"].data_ptr(), r["index"].data_ptr(), r["base"].data_ptr(), r["stride"])
        if not (self.use_graphs and self.graph_safe()):
            self._train_step_body(b, same_xy, weight)
            return
        g = self._graphs.get(key)
        if g is None:
            self._train_step_body(b, same_xy, weight)
            self._graphs[key] = "warm"
            return
        if g == "warm":
            torch.cuda.synchronize()
            g = None
            if self._reducer is not None and self._reducer.capturable(self._params.device):
                # RCCL: the all-reduce is recorded with the rest of the step (one graph launch per step)
                calls_before = self._reducer.calls
                try:
                    graph = torch.cuda.CUDAGraph()
                    with recording(graph, capture_error_mode="thread_local"):
                        self._train_step_body(b, same_xy, weight)
                    g = (graph, None, "collective inside")
                except RuntimeError as e:  # recording the whole step failed where the probe passed
                    import logging
                    logging.getLogger(__name__).warning("recording the data-parallel step failed on this rank: %s", e)
                    torch.cuda.synchronize()
                    self._calls.clear()
                if self._reducer.calls != calls_before:  # the recorded body reached the all-reduce: recording is not reducing
                    self._reducer.note_reduced(self._grads, -1)
                # the fallback must be taken by every rank or by none: one rank replaying a graph that holds the
                # collective while another issues it eagerly between two graphs would not meet
                if not self._reducer.agree(g is not None, self._params.device):
                    g = None
                    self._reducer._capturable = False
            if g is not None:
                pass
            elif self._reducer is None:
                graph = torch.cuda.CUDAGraph()
                with recording(graph):
                    self._train_step_body(b, same_xy, weight)
                g = (graph,)
            else:
                scale = 1.0 / self._reducer.world
                g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # thread-local capture: the process group's watchdog thread polls events while this thread records
                with recording(g_fb, capture_error_mode="thread_local"):
                    fold = self._fwd_bwd(b, same_xy, weight)
                with recording(g_opt, capture_error_mode="thread_local"):
                    self.optimizer.apply_model(self, scale, zero_grads=True, fold=fold)
                g = (g_fb, g_opt)
            self._graphs[key] = g
        if not self._grads_clean:
            # the recorded step relies on its own optimizer launch having cleared the buffer; a manual backward
            # (or anything else that touched the gradients) since the last replay means clearing it here first
            ops.fill(self._grads, 0.0)
        g[0].replay()
        if len(g) == 3:
            self._reducer.note_reduced(self._grads)
        elif len(g) == 2:
            self._reducer.all_reduce(self._grads)
            g[1].replay()
        self._grads_clean = True

    def graph_safe(self) -> bool:
        """A step can be frozen into a HIP graph when no layer needs fresh host-side arguments per call."""
        from .nn import BatchNormalization, Dropout
        layers = self._all_layers()
        if any(isinstance(l, Dropout) and l.rate > 0 for l in layers):
            return False
        red = self._reducer
        if red is not None and getattr(red, "sync_batchnorm", False) and any(isinstance(l, BatchNormalization) for l in layers):
            # the statistics exchange sits inside the forward and backward passes: recordable only where the collective is
            return red.capturable(self._params.device)
        return True

    def _to_resident(self, a, device):
        """Training data lives in HBM in its original precision (float64 stays float64; the batch gather
        casts to float32, as Keras does on the way into the model)."""
        if isinstance(a, torch.Tensor):
            t = a.as_subclass(torch.Tensor)
        else:
            t = torch.from_numpy(np.ascontiguousarray(a))
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float32)
        return t.to(device).contiguous()

    def fit(self, x, y=None, epochs=1, shuffle=True, validation_data=None, callbacks=None, batch_size=32,
            verbose=1, seed: Optional[int] = None):
        if self.optimizer is None or self.loss is None:
            raise RuntimeError("compile(optimizer=..., loss=...) must be called before fit")
        device = default_device()
        same_xy = y is None or y is x
        X = self._to_resident(x, device)
        Y = X if same_xy else self._to_resident(y, device)
        if not self.built:
            self.build((None,) + tuple(X.shape[1:]))
        self._materialize(device)
        n = X.shape[0]
        if Y.shape[0] != n:
            raise ValueError("x and y have different numbers of samples")
        if n == 0:
            raise ValueError("fit: empty training set")
        bs = int(batch_size)
        hist = History()
        cbs = list(callbacks or [])
        for cb in cbs:
            cb.on_train_begin()
        gen = torch.Generator(device=device)
        gen.manual_seed(self._seed if seed is None else seed)
        y_shape = tuple(Y.shape[1:])
        # the forward launch gathers its own batch where it can (every batch size of the epoch must take that form)
        rows = None
        if shuffle and self.rows_step_supported(X, min(bs, n), same_xy) and (n % bs == 0 or self.rows_step_supported(X, n % bs, same_xy)):
            rows = self.step_rows(X, torch.empty(n, dtype=torch.int64, device=device), bs)
        for epoch in range(int(epochs)):
            perm = torch.randperm(n, device=device, generator=gen) if shuffle else None
            if rows is not None:
                rows["index"].copy_(perm)
                self.restart_rows(rows)
            acc = None
            for start in range(0, n, bs):
                cur = min(bs, n - start)
                b = self._step_buffers(cur, y_shape)
                if start == 0:
                    acc = self._pool.get(("fit", "acc"), (1,))
                    ops.fill(acc, 0.0)
                b["acc"] = acc
                if rows is not None:
                    b["rows"] = rows
                else:
                    idx = perm[start:start + cur] if perm is not None else None
                    src_x = X if idx is not None else X[start:start + cur]
                    ops.gather_rows(src_x, idx, b["x"])
                    if not same_xy:
                        ops.gather_rows(Y if idx is not None else Y[start:start + cur], idx, b["y"])
                self.train_on_device_batch(b, same_xy, float(cur))
            logs = {"loss": float(acc.item()) / n}  # one host sync per epoch
            if validation_data is not None:
                vx, vy = validation_data[0], validation_data[1]
                logs["val_loss"] = self.evaluate(vx, None if vy is vx else vy, batch_size=bs)
            hist.record(epoch, **logs)
            if verbose:
                msg = " - ".join(f"{k}: {v:.6f}" for k, v in logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {msg}")
            stop = False
            for cb in cbs:
                stop = cb.on_epoch_end(epoch, logs) or stop
            if stop or self.stop_training:
                break
        self.history = hist
        return hist

    def evaluate(self, x, y=None, batch_size=32) -> float:
        """Mean loss over a data set in inference mode (``training=False``), sample-weighted over batches."""
        device = default_device()
        same_xy = y is None or y is x
        X = self._to_resident(x, device)
        Y = X if same_xy else self._to_resident(y, device)
        if not self.built:
            self.build((None,) + tuple(X.shape[1:]))
        self._materialize(device)
        n, bs = X.shape[0], int(batch_size)
        acc = self._pool.get(("eval", "acc"), (1,))
        loss = self._pool.get(("eval", "loss"), (1,))
        ops.fill(acc, 0.0)
        for start in range(0, n, bs):
            cur = min(bs, n - start)
            xb = self._pool.get(("eval", "x"), (cur,) + self.input_shape_)
            ops.gather_rows(X[start:start + cur], None, xb)
            if same_xy:
                yb = xb
            else:
                yb = self._pool.get(("eval", "y"), (cur,) + tuple(Y.shape[1:]))
                ops.gather_rows(Y[start:start + cur], None, yb)
            pred = self.forward_raw(xb, False)
            self.loss.value_and_grad(yb, pred, loss, None, 1.0)
            for term, w in self._last_ctx.extra_losses:
                ops.axpy(loss, term, w)
            ops.axpy(acc, loss, float(cur))
        return float(acc.item()) / max(n, 1)

    def predict(self, x, batch_size=32, verbose=0) -> np.ndarray:
        device = default_device()
        X = self._to_resident(x, device)
        if not self.built:
            self.build((None,) + tuple(X.shape[1:]))
        self._materialize(device)
        n, bs = X.shape[0], int(batch_size)
        out = torch.empty((n,) + self.output_shape_, dtype=torch.float32, device=device)
        for start in range(0, n, bs):
            cur = min(bs, n - start)
            xb = self._pool.get(("predict", "x"), (cur,) + self.input_shape_)
            ops.gather_rows(X[start:start + cur], None, xb)
            yb = self.forward_raw(xb, False)
            if yb.dtype != torch.float32:
                ops.cast(yb, out[start:start + cur])
            else:
                out[start:start + cur].copy_(yb)
        return out.cpu().numpy()


    def reconstruction_scores(self, x, batch_size=32) -> torch.Tensor:
        """Per-sample reconstruction error of an auto-encoder in inference mode, left on the device:
        ``tf.reduce_mean(losses.mse(model.predict(x), x), axis=(1, 2))`` (run_model.py:356-362) without the
        reconstructions leaving HBM."""
        device = default_device()
        X = self._to_resident(x, device)
        if not self.built:
            self.build((None,) + tuple(X.shape[1:]))
        self._materialize(device)
        if tuple(self.output_shape_) != tuple(X.shape[1:]):
            raise ValueError(f"reconstruction_scores: output {self.output_shape_} is not the input shape {tuple(X.shape[1:])}")
        n, bs = X.shape[0], int(batch_size)
        scores = torch.empty((n,), dtype=torch.float32, device=device)
        for start in range(0, n, bs):
            cur = min(bs, n - start)
            xb = self._pool.get(("predict", "x"), (cur,) + self.input_shape_)
            ops.gather_rows(X[start:start + cur], None, xb)
            ops.sample_mse(xb, self.forward_raw(xb, False), scores[start:start + cur])
        return scores

class SequentialModel(Model):
    """A Model that is exactly one Sequential stack."""

    def __init__(self, layers: Sequence[Layer], name=None, seed: int = DEFAULT_SEED):
        super().__init__(name, seed)
        self._stack = Sequential(list(layers), name=(name or "model") + "_stack")
