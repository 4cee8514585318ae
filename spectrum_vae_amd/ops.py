"""Tensor-level wrappers over the C ABI: shape checks on the host, raw device
pointers and the current HIP stream into ``libspectrum_hip.so``.

torch supplies device memory and streams only; every wrapper refuses tensors
that are not contiguous float32 on a GPU, so a silent CPU path cannot exist.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT, BF16, F32, BatchRows, ConvGeom, DgradJob, GradSlabs, NetLayer, WgradBatchJob, WgradJob, check, lib


def set_option(name: str, value: int):
    """Plan-time option of the library (``sv_set_option``): which of several equivalent kernel bodies / launch shapes a
    planner picks; -1 = the planner's own choice."""
    check(lib.sv_set_option(name.encode(), int(value)), "sv_set_option")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib.sv_get_option(name.encode(), C.byref(v)), "sv_get_option")
    return int(v.value)


class option:
    """``with ops.option("dgrad_nhwc", 0): ...`` - an option for the duration of a block (tests, measurements)."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, int(value)

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], dtype=torch.float32, name: str = "tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.SpectrumHipError(f"{name} is on {t.device}: the HIP path needs device tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    """Storage-type code of an activation tensor: the kernels take float32 or bfloat16 activations / filters."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _aptr(t: Optional[torch.Tensor], like: Optional[torch.Tensor] = None, name: str = "tensor") -> Optional[int]:
    """Pointer of an activation-typed tensor (float32 or bfloat16); ``like``: the tensor whose type it must share."""
    if t is None:
        return None
    if like is not None and t.dtype != like.dtype:
        raise TypeError(f"{name}: expected {like.dtype}, got {t.dtype}")
    return _ptr(t, t.dtype, name)


# --------------------------------------------------------------------------- geometry (host logic, GPU-free)
def same_pads(in_size: int, k: int, s: int) -> Tuple[int, int, int]:
    """tf SAME: out = ceil(in/s), total = max((out-1)s + k - in, 0), before = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv_geom(n, h, w, cin, kh, kw, sh, sw, padding, cout, up=(1, 1)) -> ConvGeom:
    """Geometry of a Conv2D on a logical input of (h, w) (already multiplied by ``up``)."""
    if padding == "same":
        ho, pt, _ = same_pads(h, kh, sh)
        wo, pl, _ = same_pads(w, kw, sw)
    elif padding == "valid":
        ho, wo, pt, pl = (h - kh) // sh + 1, (w - kw) // sw + 1, 0, 0
    else:
        raise ValueError(f"padding {padding!r}")
    if ho <= 0 or wo <= 0:
        raise ValueError(f"conv output would be empty: input {h}x{w}, kernel {kh}x{kw}, padding {padding}")
    return ConvGeom(n, h, w, cin, kh, kw, sh, sw, pt, pl, ho, wo, cout, up[0], up[1])


def deconv_out_size(in_size: int, k: int, s: int, padding: str) -> int:
    return in_size * s if padding == "same" else in_size * s + max(k - s, 0)


def deconv_geom(n, h, w, cin, kh, kw, sh, sw, padding, cout) -> ConvGeom:
    """Geometry struct for a Conv2DTranspose with input (h, w, cin) and ``cout`` filters:
    the forward conv it is the input-gradient of maps (Hout, Wout, cout) -> (h, w, cin)."""
    ho = deconv_out_size(h, kh, sh, padding)
    wo = deconv_out_size(w, kw, sw, padding)
    g = conv_geom(n, ho, wo, cout, kh, kw, sh, sw, padding, cin)
    if (g.Ho, g.Wo) != (h, w):
        raise ValueError("transposed-conv geometry does not round-trip")
    return g


# --------------------------------------------------------------------------- conv family
OP_FWD, OP_DGRAD, OP_TRANSPOSE_FWD, OP_TRANSPOSE_DGRAD = 0, 1, 2, 3


def conv2d_scratch(g: ConvGeom, op: int) -> int:
    """Bytes of optional scratch with which operation ``op`` spreads a long reduction over more workgroups (0: none)."""
    return int(lib.sv_conv2d_scratch_bytes(C.byref(g), op))


def _ws_args(ws):
    return (_ptr(ws, torch.uint8), ws.numel()) if ws is not None else (None, 0)


def conv2d_fwd(g: ConvGeom, x, w, bias, y, act: str, alpha: float, ws=None):
    """x, w, y float32 or all bfloat16 (bias stays float32)."""
    check(lib.sv_conv2d_fwd_ws(C.byref(g), _aptr(x), _aptr(w, x, "w"), _ptr(bias), _aptr(y, x, "y"), ACT[act], alpha,
                               *_ws_args(ws), _dt(x), _stream()), "sv_conv2d_fwd_ws")


def conv2d_masked_supported(g: ConvGeom) -> bool:
    """Whether ``conv2d_fwd_masked`` / ``conv2d_wgrad_masked`` cover this geometry."""
    return bool(lib.sv_conv2d_masked_supported(C.byref(g)))


def conv2d_fwd_masked(g: ConvGeom, x, mask, w, bias, y, act: str, alpha: float, ws=None):
    """``conv2d_fwd`` of x * (1 - mask) without the masked tensor (float32)."""
    if mask.shape != x.shape:
        raise ValueError(f"conv2d_fwd_masked: mask {tuple(mask.shape)} vs x {tuple(x.shape)}")
    check(lib.sv_conv2d_fwd_masked(C.byref(g), _ptr(x), _ptr(mask), _ptr(w), _ptr(bias), _ptr(y), ACT[act], alpha,
                                   *_ws_args(ws), F32, _stream()), "sv_conv2d_fwd_masked")


def conv2d_wgrad_masked(g: ConvGeom, x, mask, dz, dw, accumulate: bool, ws: torch.Tensor):
    """``conv2d_wgrad`` with x read as x * (1 - mask) (float32)."""
    check(lib.sv_conv2d_wgrad_masked(C.byref(g), _ptr(x), _ptr(mask), _ptr(dz), _ptr(dw), int(accumulate),
                                     _ptr(ws, torch.uint8), ws.numel(), F32, _stream()), "sv_conv2d_wgrad_masked")


def conv2d_dgrad(g: ConvGeom, dz, w, dx, ws=None):
    check(lib.sv_conv2d_dgrad_ws(C.byref(g), _aptr(dz), _aptr(w, dz, "w"), _aptr(dx, dz, "dx"), *_ws_args(ws), _dt(dz),
                                 _stream()), "sv_conv2d_dgrad_ws")


def conv2d_wgrad_workspace(g: ConvGeom) -> int:
    return int(lib.sv_conv2d_wgrad_workspace_bytes(C.byref(g)))


def conv2d_wgrad(g: ConvGeom, x, dz, dw, accumulate: bool, ws: torch.Tensor):
    """x, dz float32 or both bfloat16; dw is always float32."""
    check(lib.sv_conv2d_wgrad(C.byref(g), _aptr(x), _aptr(dz, x, "dz"), _ptr(dw), int(accumulate), _ptr(ws, torch.uint8),
                              ws.numel(), _dt(x), _stream()), "sv_conv2d_wgrad")


def static_scratch_bytes() -> int:
    return int(lib.sv_static_scratch_bytes())


def conv2d_wgrad_static(g: ConvGeom, x, dz, dw, accumulate: bool):
    """``conv2d_wgrad`` with the partial sums in the library's built-in device array: no workspace, no allocation."""
    check(lib.sv_conv2d_wgrad_static(C.byref(g), _ptr(x), _ptr(dz), _ptr(dw), int(accumulate), F32, _stream()),
          "sv_conv2d_wgrad_static")


def conv2d_fused_supported(g: ConvGeom, pooled: bool) -> bool:
    return bool(lib.sv_conv2d_fused_supported(C.byref(g), int(pooled)))


def conv2d_s1_mfma_bwd(g: ConvGeom) -> bool:
    """Whether ``conv2d_bwd`` takes this geometry through the matrix-core kernels of the 16..128-channel family (then
    dx and dw are best asked for in one call)."""
    return bool(lib.sv_conv2d_s1_mfma_bwd(C.byref(g)))


def conv2d_pool_fwd(g: ConvGeom, x, w, bias, y_pooled, idx, act: str, alpha: float):
    check(lib.sv_conv2d_pool_fwd(C.byref(g), _ptr(x), _ptr(w), _ptr(bias), _ptr(y_pooled), _ptr(idx, torch.uint8),
                                 ACT[act], alpha, F32, _stream()), "sv_conv2d_pool_fwd")


def net_layers(specs) -> "C.Array":
    """ctypes array of ``sv_net_layer`` from dicts with Cin, Cout, K, up, pool, act, alpha, w_off, b_off, y, idx
    (y / idx: tensors or None)."""
    arr = (NetLayer * len(specs))()
    for a, d in zip(arr, specs):
        a.Cin, a.Cout, a.K, a.up, a.pool = d["Cin"], d["Cout"], d["K"], d["up"], d["pool"]
        a.act, a.alpha, a.reserved = ACT[d["act"]], float(d["alpha"] or 0.0), 0
        a.w_off, a.b_off = d["w_off"], d["b_off"]
        a.y = _ptr(d.get("y"))
        a.idx = _ptr(d.get("idx"), torch.uint8)
    return arr


def convnet_supported(arr, n: int, h: int, w: int) -> bool:
    return bool(lib.sv_convnet_supported(arr, len(arr), n, h, w))


def convnet_fwd(arr, n: int, h: int, w: int, x, params):
    check(lib.sv_convnet_fwd(arr, len(arr), n, h, w, _ptr(x), _ptr(params), F32, _stream()), "sv_convnet_fwd")


def convnet_rows_supported(arr, n: int, h: int, w: int) -> bool:
    """Whether ``convnet_fwd_mse_rows`` takes this chain (the forward launch gathers its own batch)."""
    return bool(lib.sv_convnet_rows_supported(arr, len(arr), n, h, w))


def batch_rows(x_rows, row_index, x_batch, counter=None, counter_base=None, batch_stride: int = 0) -> BatchRows:
    """``sv_batch_rows`` over device tensors: ``x_rows`` float32 [rows, ...], ``row_index`` int64 [positions],
    ``x_batch`` float32 [N, ...] receiving the batch; ``counter`` / ``counter_base``: int64 device scalars (the
    optimizer's update count and its value at position 0), ``batch_stride`` positions per step."""
    if x_rows.dtype != torch.float32 or x_batch.dtype != torch.float32 or row_index.dtype != torch.int64:
        raise TypeError("batch_rows: float32 rows / batch and an int64 index")
    if x_rows[0].numel() != x_batch[0].numel():
        raise ValueError("batch_rows: row sizes differ")
    if (counter is None) != (counter_base is None):
        raise ValueError("batch_rows: counter and counter_base go together")
    r = BatchRows()
    r.x_rows, r.n_rows = _ptr(x_rows), x_rows.shape[0]
    r.row_index, r.n_index = _ptr(row_index, torch.int64), row_index.numel()
    r.counter = _ptr(counter, torch.int64) if counter is not None else None
    r.counter_base = _ptr(counter_base, torch.int64) if counter_base is not None else None
    r.batch_stride, r.x_batch = int(batch_stride), _ptr(x_batch)
    return r


def convnet_fwd_mse_rows(arr, n: int, h: int, w: int, rows: BatchRows, params, y_true, dpred, loss_scale, partials) -> int:
    """``convnet_fwd_mse`` on the batch the launch gathers itself (``rows``: see ``batch_rows``)."""
    if y_true.numel() != dpred.numel() or partials.numel() < loss_workspace_floats():
        raise ValueError("convnet_fwd_mse_rows: y_true / dpred sizes differ or the partial-sum scratch is too small")
    cnt = C.c_int(0)
    check(lib.sv_convnet_fwd_mse_rows(arr, len(arr), n, h, w, C.byref(rows), _ptr(params), _ptr(y_true), _ptr(dpred),
                                      loss_scale, _ptr(partials), C.byref(cnt), F32, _stream()), "sv_convnet_fwd_mse_rows")
    return int(cnt.value)


def convnet_fwd_mse_dgrad_supported(arr, n: int, h: int, w: int, jobs) -> bool:
    """Whether ``convnet_fwd_mse_dgrad`` takes this layer table with these input-gradient jobs behind it."""
    if not jobs or len(jobs) > 8:
        return False
    return bool(lib.sv_convnet_fwd_mse_dgrad_supported(arr, len(arr), n, h, w, _dgrad_jobs(jobs), len(jobs)))


def convnet_fwd_mse_dgrad(arr, n: int, h: int, w: int, x, rows, params, y_true, dpred, loss_scale, partials, jobs) -> int:
    """``convnet_fwd_mse`` (``rows`` None) or ``convnet_fwd_mse_rows`` (``rows`` a ``batch_rows`` table) with the
    input-gradient jobs of the backward pass (``conv2d_dgrad_chain``'s list) run by the same launch."""
    if y_true.numel() != dpred.numel() or partials.numel() < loss_workspace_floats():
        raise ValueError("convnet_fwd_mse_dgrad: y_true / dpred sizes differ or the partial-sum scratch is too small")
    cnt = C.c_int(0)
    check(lib.sv_convnet_fwd_mse_dgrad(arr, len(arr), n, h, w, _ptr(x), C.byref(rows) if rows is not None else None,
                                       _ptr(params), _ptr(y_true), _ptr(dpred), loss_scale, _ptr(partials), C.byref(cnt),
                                       _dgrad_jobs(jobs), len(jobs), F32, _stream()), "sv_convnet_fwd_mse_dgrad")
    return int(cnt.value)


def convnet_fwd_mse(arr, n: int, h: int, w: int, x, params, y_true, dpred, loss_scale, partials) -> int:
    """Forward + squared-error loss gradient in one launch; returns how many entries of ``partials`` hold the
    per-workgroup sums of squared errors (fold them with ``fold_scalar`` or the optimizer launch)."""
    if y_true.numel() != dpred.numel() or partials.numel() < loss_workspace_floats():
        raise ValueError("convnet_fwd_mse: y_true / dpred sizes differ or the partial-sum scratch is too small")
    cnt = C.c_int(0)
    check(lib.sv_convnet_fwd_mse(arr, len(arr), n, h, w, _ptr(x), _ptr(params), _ptr(y_true), _ptr(dpred), loss_scale,
                                 _ptr(partials), C.byref(cnt), F32, _stream()), "sv_convnet_fwd_mse")
    return int(cnt.value)


def deferred_reduce_begin():
    check(lib.sv_deferred_reduce_begin(), "sv_deferred_reduce_begin")


def deferred_reduce_flush(launch: bool = True):
    check(lib.sv_deferred_reduce_flush(int(launch), _stream()), "sv_deferred_reduce_flush")


def fold_scalar(partials, n: int, scale: float, out, accumulate=False, running=None, running_weight=0.0):
    check(lib.sv_fold_scalar(_ptr(partials), int(n), float(scale), _ptr(out), int(accumulate), _ptr(running),
                             float(running_weight), _stream()), "sv_fold_scalar")


def scalar_program(statements):
    """``statements``: [(out, accumulate, [(term, weight), ...]), ...] - up to 4 scalar statements
    out = [out +] sum weight * term, run in order by ONE launch (a GAN step's loss bookkeeping)."""
    import ctypes as C
    n = len(statements)
    flat = [tw for _, _, terms in statements for tw in terms]
    outs = (C.c_void_p * n)(*[_ptr(o) for o, _, _ in statements])
    acc = (C.c_int * n)(*[int(bool(a)) for _, a, _ in statements])
    nt = (C.c_int * n)(*[len(t) for _, _, t in statements])
    terms = (C.c_void_p * max(len(flat), 1))(*[_ptr(t) for t, _ in flat])
    weights = (C.c_float * max(len(flat), 1))(*[float(w) for _, w in flat])
    check(lib.sv_scalar_program(n, outs, acc, nt, terms, weights, _stream()), "sv_scalar_program")


def conv2d_bwd_partial(g: ConvGeom, x, dy, y, idx, act: str, alpha: float, with_bias: bool, ws: torch.Tensor):
    check(lib.sv_conv2d_bwd_partial(C.byref(g), _ptr(x), _ptr(dy), _ptr(y), _ptr(idx, torch.uint8), ACT[act], alpha,
                                    int(with_bias), _ptr(ws, torch.uint8), ws.numel(), F32, _stream()),
          "sv_conv2d_bwd_partial")


def _wgrad_batch_jobs(jobs, blocks: int):
    arr = (WgradBatchJob * len(jobs))()
    for a, d in zip(arr, jobs):
        a.g = d["g"]
        a.x, a.dy, a.y, a.idx = _ptr(d["x"]), _ptr(d["dy"]), _ptr(d["y"]), _ptr(d.get("idx"), torch.uint8)
        a.dw, a.db = _ptr(d["dw"]), _ptr(d.get("db"))
        a.workspace, a.workspace_bytes = _ptr(d["ws"], torch.uint8), d["ws"].numel()
        a.act, a.alpha, a.accumulate, a.blocks = ACT[d["act"]], float(d["alpha"] or 0.0), 1, d.get("blocks", blocks)
    return arr


def conv2d_wgrad_batch(jobs, reduce: bool = True, blocks: int = 0):
    """jobs: list of dicts g, x, dy, y, idx, act, alpha, dw, db, ws; gradients accumulate into dw / db (reduce=False:
    the partial sums stay in the workspaces for ``conv2d_wgrad_finish``)."""
    if not jobs:
        return
    arr = _wgrad_batch_jobs(jobs, blocks)
    check(lib.sv_conv2d_wgrad_batch(arr, len(jobs), int(reduce), F32, _stream()), "sv_conv2d_wgrad_batch")


def conv2d_wgrad_batch_slabs_supported(jobs, blocks: int = 0) -> bool:
    """Whether ``conv2d_wgrad_batch_slabs`` takes these jobs (all of them in the few-channel 3x3 family, at most 8)."""
    if not jobs or len(jobs) > 8:
        return False
    return bool(lib.sv_conv2d_wgrad_batch_slabs_supported(_wgrad_batch_jobs(jobs, blocks), len(jobs)))


def conv2d_wgrad_batch_slabs(jobs, blocks: int = 0):
    """The kernel / bias gradients of the jobs as slabs of partial sums left in the jobs' workspaces; returns the
    ``sv_grad_slabs`` table for ``adam_step(..., slabs=)`` (dw / db are not touched)."""
    arr = _wgrad_batch_jobs(jobs, blocks)
    out = (GradSlabs * len(jobs))()
    check(lib.sv_conv2d_wgrad_batch_slabs(arr, len(jobs), out, F32, _stream()), "sv_conv2d_wgrad_batch_slabs")
    return out


def conv2d_wgrad_finish(jobs):
    """jobs: list of (geom, workspace, dw, db or None[, blocks]); gradients accumulate into dw / db."""
    if not jobs:
        return
    arr = (WgradJob * len(jobs))()
    for a, job in zip(arr, jobs):
        g, ws, dw, db = job[:4]
        a.g = g
        a.workspace, a.dw, a.db = _ptr(ws, torch.uint8), _ptr(dw), _ptr(db)
        a.accumulate, a.blocks = 1, (job[4] if len(job) > 4 else 0)
    check(lib.sv_conv2d_wgrad_finish(arr, len(jobs), F32, _stream()), "sv_conv2d_wgrad_finish")


def _dgrad_jobs(jobs):
    arr = (DgradJob * len(jobs))()
    for a, d in zip(arr, jobs):
        a.g = d["g"]
        a.w, a.dy, a.y, a.idx, a.dx = _ptr(d["w"]), _ptr(d["dy"]), _ptr(d["y"]), _ptr(d.get("idx"), torch.uint8), _ptr(d["dx"])
        a.act, a.alpha = ACT[d["act"]], d["alpha"] or 0.0
    return arr


def conv2d_dgrad_chain_supported(jobs) -> bool:
    """Whether ``conv2d_dgrad_chain`` takes these jobs (all in the few-channel 3x3 family, one batch size, at most 8)."""
    if not jobs or len(jobs) > 8:
        return False
    return bool(lib.sv_conv2d_dgrad_chain_supported(_dgrad_jobs(jobs), len(jobs)))


def conv2d_dgrad_chain(jobs):
    """The input gradients of consecutive layers (dicts g, w, dy, y, idx, dx, act, alpha; a job's dy may be an earlier
    job's dx) in one launch."""
    check(lib.sv_conv2d_dgrad_chain(_dgrad_jobs(jobs), len(jobs), F32, _stream()), "sv_conv2d_dgrad_chain")


def conv2d_bwd(g: ConvGeom, x, w, dy, y, idx, dx, dw, db, act: str, alpha: float, ws: Optional[torch.Tensor]):
    check(lib.sv_conv2d_bwd(C.byref(g), _ptr(x), _ptr(w), _ptr(dy), _ptr(y), _ptr(idx, torch.uint8), _ptr(dx), _ptr(dw),
                            _ptr(db), ACT[act], alpha, _ptr(ws, torch.uint8) if ws is not None else None,
                            ws.numel() if ws is not None else 0, F32, _stream()), "sv_conv2d_bwd")


def conv2d_transpose_fwd(g: ConvGeom, x, w, bias, y, act: str, alpha: float, ws=None):
    check(lib.sv_conv2d_transpose_fwd_ws(C.byref(g), _aptr(x), _aptr(w, x, "w"), _ptr(bias), _aptr(y, x, "y"), ACT[act],
                                         alpha, *_ws_args(ws), _dt(x), _stream()), "sv_conv2d_transpose_fwd_ws")


def conv2d_transpose_dgrad(g: ConvGeom, dz, w, dx, ws=None):
    check(lib.sv_conv2d_transpose_dgrad_ws(C.byref(g), _aptr(dz), _aptr(w, dz, "w"), _aptr(dx, dz, "dx"), *_ws_args(ws),
                                           _dt(dz), _stream()), "sv_conv2d_transpose_dgrad_ws")


def conv2d_transpose_wgrad(g: ConvGeom, x, dz, dw, accumulate: bool, ws: torch.Tensor):
    check(lib.sv_conv2d_transpose_wgrad(C.byref(g), _aptr(x), _aptr(dz, x, "dz"), _ptr(dw), int(accumulate),
                                        _ptr(ws, torch.uint8), ws.numel(), _dt(x), _stream()),
          "sv_conv2d_transpose_wgrad")


# --------------------------------------------------------------------------- elementwise / reductions
def act_fwd(x, y, act: str, alpha: float):
    check(lib.sv_act_fwd(_aptr(x), _aptr(y, x, "y"), x.numel(), ACT[act], alpha, _dt(x), _stream()), "sv_act_fwd")


def colreduce_workspace(rows: int, c: int) -> int:
    return int(lib.sv_colreduce_workspace_bytes(rows, c))


def act_bwd(dy, y, dz, db, db_accumulate: bool, rows: int, c: int, act: str, alpha: float, ws):
    check(lib.sv_act_bwd(_aptr(dy), _aptr(y, dy, "y"), _aptr(dz, dy, "dz"), _ptr(db), int(db_accumulate), rows, c,
                         ACT[act], alpha, _ptr(ws, torch.uint8) if ws is not None else None,
                         ws.numel() if ws is not None else 0, _dt(dy), _stream()), "sv_act_bwd")


def bias_grad(dz, db, accumulate: bool, rows: int, c: int, ws):
    check(lib.sv_bias_grad(_aptr(dz), _ptr(db), int(accumulate), rows, c, _ptr(ws, torch.uint8), ws.numel(), _dt(dz),
                           _stream()), "sv_bias_grad")


def maxpool2d_fwd(x, y, idx, n, h, w, c, ph, pw, pt, pl, ho, wo):
    check(lib.sv_maxpool2d_fwd(_ptr(x), _ptr(y), _ptr(idx, torch.uint8), n, h, w, c, ph, pw, pt, pl, ho, wo, F32,
                               _stream()), "sv_maxpool2d_fwd")


def maxpool2d_bwd(dy, idx, dx, n, h, w, c, ph, pw, pt, pl, ho, wo):
    check(lib.sv_maxpool2d_bwd(_ptr(dy), _ptr(idx, torch.uint8), _ptr(dx), n, h, w, c, ph, pw, pt, pl, ho, wo, F32,
                               _stream()), "sv_maxpool2d_bwd")


def upsample2d_fwd(x, y, n, h, w, c, uh, uw):
    check(lib.sv_upsample2d_fwd(_ptr(x), _ptr(y), n, h, w, c, uh, uw, F32, _stream()), "sv_upsample2d_fwd")


def upsample2d_bwd(dy, dx, n, h, w, c, uh, uw):
    check(lib.sv_upsample2d_bwd(_ptr(dy), _ptr(dx), n, h, w, c, uh, uw, F32, _stream()), "sv_upsample2d_bwd")


def batchnorm_fwd_train(x, gamma, beta, y, save_mean, save_invstd, moving_mean, moving_var, rows, c, eps, momentum,
                        unbiased: bool, ws):
    check(lib.sv_batchnorm_fwd_train(_aptr(x), _ptr(gamma), _ptr(beta), _aptr(y, x, "y"), _ptr(save_mean),
                                     _ptr(save_invstd), _ptr(moving_mean), _ptr(moving_var), rows, c, eps, momentum,
                                     int(unbiased), _ptr(ws, torch.uint8), ws.numel(), _dt(x), _stream()),
          "sv_batchnorm_fwd_train")


def batchnorm_fwd_infer(x, gamma, beta, moving_mean, moving_var, y, rows, c, eps):
    check(lib.sv_batchnorm_fwd_infer(_aptr(x), _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                     _aptr(y, x, "y"), rows, c, eps, _dt(x), _stream()), "sv_batchnorm_fwd_infer")


def batchnorm_bwd(x, dy, gamma, save_mean, save_invstd, dx, dgamma, dbeta, accumulate: bool, rows, c, ws):
    check(lib.sv_batchnorm_bwd(_aptr(x), _aptr(dy, x, "dy"), _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                               _aptr(dx, x, "dx"), _ptr(dgamma), _ptr(dbeta), int(accumulate), rows, c,
                               _ptr(ws, torch.uint8), ws.numel(), _dt(x), _stream()), "sv_batchnorm_bwd")


def batchnorm_workspace(rows: int, c: int, groups: int = 1) -> int:
    return int(lib.sv_batchnorm_workspace_bytes(rows, c, groups))


def batchnorm_act_fwd_train(x, gamma, beta, y, save_mean, save_invstd, moving_mean, moving_var, rows, c, eps, momentum,
                            unbiased: bool, groups: int, act, alpha: float, ws):
    """BatchNorm (``groups`` independent row groups) with a following activation in the same launches."""
    if save_mean.numel() != groups * c or save_invstd.numel() != groups * c:
        raise ValueError("batchnorm: saved statistics must be [groups][C]")
    check(lib.sv_batchnorm_act_fwd_train(_aptr(x), _ptr(gamma), _ptr(beta), _aptr(y, x, "y"), _ptr(save_mean),
                                         _ptr(save_invstd), _ptr(moving_mean), _ptr(moving_var), rows, c, eps, momentum,
                                         int(unbiased), groups, ACT[act], alpha, _ptr(ws, torch.uint8), ws.numel(), _dt(x),
                                         _stream()), "sv_batchnorm_act_fwd_train")


def batchnorm_act_fwd_infer(x, gamma, beta, moving_mean, moving_var, y, rows, c, eps, act, alpha: float):
    check(lib.sv_batchnorm_act_fwd_infer(_aptr(x), _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                         _aptr(y, x, "y"), rows, c, eps, ACT[act], alpha, _dt(x), _stream()),
          "sv_batchnorm_act_fwd_infer")


def batchnorm_act_bwd(x, dy, y_act, act_out, alpha_out: float, gamma, save_mean, save_invstd, dx, dgamma, dbeta,
                      accumulate: bool, act_in, alpha_in: float, dbias_in, dbias_accumulate: bool, rows, c, groups: int, ws):
    """Backward of ``batchnorm_act_fwd_train``; ``act_in`` / ``dbias_in``: also the activation backward (and bias
    gradient) of the conv that produced ``x``."""
    if save_mean.numel() != groups * c or save_invstd.numel() != groups * c:
        raise ValueError("batchnorm: saved statistics must be [groups][C]")
    check(lib.sv_batchnorm_act_bwd(_aptr(x), _aptr(dy, x, "dy"), _aptr(y_act, x, "y_act") if y_act is not None else None,
                                   ACT[act_out], alpha_out, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                                   _aptr(dx, x, "dx"), _ptr(dgamma), _ptr(dbeta), int(accumulate), ACT[act_in], alpha_in,
                                   _ptr(dbias_in), int(dbias_accumulate), rows, c, groups, _ptr(ws, torch.uint8), ws.numel(),
                                   _dt(x), _stream()), "sv_batchnorm_act_bwd")


def batchnorm_sync_moments(x, moments, rows, c, groups: int, ws):
    """Per group and channel (rows, mean, sum of squared deviations) of this replica's rows into ``moments``
    (float64 [groups, 3, C]); see ``parallel.GradReducer.combine_moments`` for the exchange."""
    if moments.dtype != torch.float64 or moments.numel() != groups * 3 * c:
        raise ValueError("batchnorm_sync_moments: moments must be float64 [groups][3][C]")
    check(lib.sv_batchnorm_sync_moments(_aptr(x), _ptr(moments, torch.float64), rows, c, groups, _ptr(ws, torch.uint8),
                                        ws.numel(), _dt(x), _stream()), "sv_batchnorm_sync_moments")


def batchnorm_sync_fwd_apply(x, gamma, beta, mean, invstd, y, rows, c, groups: int, act, alpha: float):
    check(lib.sv_batchnorm_sync_fwd_apply(_aptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _aptr(y, x, "y"),
                                          rows, c, groups, ACT[act], alpha, _dt(x), _stream()),
          "sv_batchnorm_sync_fwd_apply")


def batchnorm_sync_bwd_sums(x, dy, y_act, act_out, alpha_out: float, mean, invstd, sums, dgamma, dbeta, accumulate: bool,
                            rows, c, groups: int, ws):
    if sums.dtype != torch.float32 or sums.numel() != groups * 2 * c:
        raise ValueError("batchnorm_sync_bwd_sums: sums must be float32 [groups][2][C]")
    check(lib.sv_batchnorm_sync_bwd_sums(_aptr(x), _aptr(dy, x, "dy"),
                                         _aptr(y_act, x, "y_act") if y_act is not None else None, ACT[act_out], alpha_out,
                                         _ptr(mean), _ptr(invstd), _ptr(sums), _ptr(dgamma), _ptr(dbeta), int(accumulate),
                                         rows, c, groups, _ptr(ws, torch.uint8), ws.numel(), _dt(x), _stream()),
          "sv_batchnorm_sync_bwd_sums")


def batchnorm_sync_bwd_apply(x, dy, y_act, act_out, alpha_out: float, gamma, mean, invstd, sums, all_rows_per_group: int,
                             dx, act_in, alpha_in: float, dbias_in, dbias_accumulate: bool, rows, c, groups: int, ws):
    check(lib.sv_batchnorm_sync_bwd_apply(_aptr(x), _aptr(dy, x, "dy"),
                                          _aptr(y_act, x, "y_act") if y_act is not None else None, ACT[act_out], alpha_out,
                                          _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(sums), int(all_rows_per_group),
                                          _aptr(dx, x, "dx"), ACT[act_in], alpha_in, _ptr(dbias_in), int(dbias_accumulate),
                                          rows, c, groups, _ptr(ws, torch.uint8), ws.numel(), _dt(x), _stream()),
          "sv_batchnorm_sync_bwd_apply")


def concat_channels_fwd(a, b, y, rows, ca, cb):
    check(lib.sv_concat_channels_fwd(_aptr(a), _aptr(b, a, "b"), _aptr(y, a, "y"), rows, ca, cb, _dt(a), _stream()),
          "sv_concat_channels_fwd")


def concat_channels_bwd(dy, da, db, rows, ca, cb):
    check(lib.sv_concat_channels_bwd(_aptr(dy), _aptr(da, dy, "da"), _aptr(db, dy, "db"), rows, ca, cb, _dt(dy),
                                     _stream()), "sv_concat_channels_bwd")


def dense_fwd(x, w, bias, y, m, k, n, act: str, alpha: float):
    check(lib.sv_dense_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), m, k, n, ACT[act], alpha, F32, _stream()),
          "sv_dense_fwd")


def dense_dgrad(dz, w, dx, m, k, n):
    check(lib.sv_dense_dgrad(_ptr(dz), _ptr(w), _ptr(dx), m, k, n, F32, _stream()), "sv_dense_dgrad")


def dense_wgrad_workspace(m: int, k: int, n: int) -> int:
    return int(lib.sv_dense_wgrad_workspace_bytes(m, k, n))


def dense_wgrad(x, dz, dw, accumulate: bool, m, k, n, ws: Optional[torch.Tensor] = None):
    """``ws``: scratch of ``dense_wgrad_workspace(m, k, n)`` bytes that lets the reduction over the batch be cut over
    many workgroups (its own until a deferred-reduce bracket is flushed)."""
    if ws is None:
        check(lib.sv_dense_wgrad(_ptr(x), _ptr(dz), _ptr(dw), int(accumulate), m, k, n, F32, _stream()),
              "sv_dense_wgrad")
    else:
        check(lib.sv_dense_wgrad_ws(_ptr(x), _ptr(dz), _ptr(dw), int(accumulate), m, k, n, _ptr(ws, torch.uint8),
                                    ws.numel(), F32, _stream()), "sv_dense_wgrad_ws")


def dropout_fwd(x, y, rate, seed, offset):
    check(lib.sv_dropout_fwd(_ptr(x), _ptr(y), x.numel(), rate, seed, offset, F32, _stream()), "sv_dropout_fwd")


def dropout_bwd(dy, dx, rate, seed, offset):
    check(lib.sv_dropout_bwd(_ptr(dy), _ptr(dx), dy.numel(), rate, seed, offset, F32, _stream()), "sv_dropout_bwd")


def reduce_sum_h_fwd(x, y, n, h, w, c):
    check(lib.sv_reduce_sum_h_fwd(_ptr(x), _ptr(y), n, h, w, c, F32, _stream()), "sv_reduce_sum_h_fwd")


def reduce_sum_h_bwd(dy, dx, n, h, w, c):
    check(lib.sv_reduce_sum_h_bwd(_ptr(dy), _ptr(dx), n, h, w, c, F32, _stream()), "sv_reduce_sum_h_bwd")


def reparam_kl_fwd(mean, logvar, eps_in, z, eps_out, kl_out, partials, seed=0, offset=0, counter=None):
    """``counter``: optional int64[1] device tensor holding the position in the random stream (read, then advanced by
    the launch), which keeps a recorded step drawing fresh values."""
    b, d = mean.shape
    if tuple(logvar.shape) != (b, d) or tuple(z.shape) != (b, d) or (eps_in is not None and eps_in.numel() != b * d):
        raise ValueError("reparam_kl_fwd: mean / logvar / z / epsilon shapes differ")
    check(lib.sv_reparam_kl_fwd(_ptr(mean), _ptr(logvar), _ptr(eps_in), _ptr(z), _ptr(eps_out), _ptr(kl_out),
                                _ptr(partials), b, d, seed, offset, _ptr(counter, torch.int64), _stream()),
          "sv_reparam_kl_fwd")


def reparam_kl_bwd(mean, logvar, eps, dz, dmean, dlogvar, kl_weight: float):
    b, d = mean.shape
    check(lib.sv_reparam_kl_bwd(_ptr(mean), _ptr(logvar), _ptr(eps), _ptr(dz), _ptr(dmean), _ptr(dlogvar), kl_weight,
                                b, d, _stream()), "sv_reparam_kl_bwd")


# --------------------------------------------------------------------------- losses / optimiser
def loss_workspace_floats() -> int:
    return int(lib.sv_loss_workspace_bytes()) // 4


def loss_mse(y_true, y_pred, loss_out, partials, dpred=None, loss_scale=1.0, weight=None, weight_stride=1,
             weight_scale=1.0, accumulate=False, running=None, running_weight=0.0):
    c = y_pred.shape[-1]
    n_pix = y_pred.numel() // c
    if y_true.numel() != y_pred.numel():
        raise ValueError(f"MSE: y_true {tuple(y_true.shape)} vs y_pred {tuple(y_pred.shape)}")
    check(lib.sv_loss_mse(_ptr(y_true), _aptr(y_pred), _ptr(weight), weight_stride, weight_scale, _ptr(loss_out),
                          int(accumulate), _aptr(dpred, y_pred, "dpred"), loss_scale, n_pix, c, _ptr(partials),
                          _ptr(running), float(running_weight), _dt(y_pred), _stream()),
          "sv_loss_mse")


def loss_mse_map(y_true, y_pred, out, weight=None, weight_stride=1, weight_scale=1.0):
    c = y_pred.shape[-1]
    check(lib.sv_loss_mse_map(_ptr(y_true), _aptr(y_pred), _ptr(weight), weight_stride, weight_scale, _ptr(out),
                              y_pred.numel() // c, c, _dt(y_pred), _stream()), "sv_loss_mse_map")


def sample_mse(y_true, y_pred, out):
    """out[n] = mean over sample n of (y_pred - y_true)^2 (run_model.py:361-362)."""
    n = y_pred.shape[0]
    if y_true.numel() != y_pred.numel() or out.numel() != n:
        raise ValueError(f"sample_mse: y_true {tuple(y_true.shape)}, y_pred {tuple(y_pred.shape)}, out {tuple(out.shape)}")
    check(lib.sv_sample_mse(_ptr(y_true), _aptr(y_pred), _ptr(out), n, y_pred.numel() // n, _dt(y_pred), _stream()),
          "sv_sample_mse")


def roc_auc_counts(scores_label1, scores_label0, counts):
    """counts[0] = #{s1 > s0}, counts[1] = #{s1 == s0} over all pairs (int64 tensor of 2 on the device)."""
    check(lib.sv_roc_auc_counts(_ptr(scores_label1), scores_label1.numel(), _ptr(scores_label0), scores_label0.numel(),
                                _ptr(counts, torch.int64, "counts"), _stream()), "sv_roc_auc_counts")


def loss_mae(y_true, y_pred, loss_out, partials, dpred=None, loss_scale=1.0, accumulate=False):
    if y_true.numel() != y_pred.numel():
        raise ValueError(f"MAE: y_true {tuple(y_true.shape)} vs y_pred {tuple(y_pred.shape)}")
    check(lib.sv_loss_mae(_ptr(y_true), _aptr(y_pred), _ptr(loss_out), int(accumulate), _aptr(dpred, y_pred, "dpred"),
                          loss_scale, y_pred.numel(), _ptr(partials), _dt(y_pred), _stream()), "sv_loss_mae")


def loss_bce(pred, label: float, from_logits: bool, loss_out, partials, dpred=None, loss_scale=1.0,
             accumulate=False):
    check(lib.sv_loss_bce(_aptr(pred), label, int(from_logits), _ptr(loss_out), int(accumulate),
                          _aptr(dpred, pred, "dpred"), loss_scale, pred.numel(), _ptr(partials), _dt(pred), _stream()),
          "sv_loss_bce")


def adam_prepare(state, lr0, beta1, beta2, decay_steps, decay_rate, staircase):
    check(lib.sv_adam_keras_prepare(_ptr(state, torch.int64), lr0, beta1, beta2, int(decay_steps or 0), decay_rate,
                                    int(staircase), _stream()), "sv_adam_keras_prepare")


def adam_step(theta, grad, m, v, state, lr0, beta1, beta2, eps, decay_steps, decay_rate, staircase, grad_scale=1.0,
              zero_grad=False, theta_lp=None, fold=None, slabs=None):
    """Schedule + update in one launch; ``theta_lp``: optional bfloat16 copy of theta refreshed in the same pass;
    ``fold``: optional pending scalar fold ``(partials, n, scale, out, running, running_weight)`` carried by the launch;
    ``slabs``: the table ``conv2d_wgrad_batch_slabs`` returned - the gradient of the elements it covers is grad + the sum
    of the slabs."""
    if not (theta.numel() == grad.numel() == m.numel() == v.numel()):
        raise ValueError("adam: buffer sizes differ")
    if theta_lp is not None and theta_lp.numel() != theta.numel():
        raise ValueError("adam: the bf16 copy has a different size")
    head = (_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), theta.numel(), _ptr(state, torch.int64), lr0,
            beta1, beta2, eps, int(decay_steps or 0), decay_rate, int(staircase), grad_scale,
            int(zero_grad), _ptr(theta_lp, torch.bfloat16) if theta_lp is not None else None,
            *((_ptr(fold[0]), int(fold[1]), float(fold[2]), _ptr(fold[3]), _ptr(fold[4]),
               float(fold[5])) if fold is not None else (None, 0, 0.0, None, None, 0.0)))
    if slabs is not None:
        check(lib.sv_adam_keras_step_slabs(*head, slabs, len(slabs), _stream()), "sv_adam_keras_step_slabs")
    else:
        check(lib.sv_adam_keras_step(*head, _stream()), "sv_adam_keras_step")


def adam_apply(theta, grad, m, v, state, beta1, beta2, eps, grad_scale=1.0, zero_grad=False):
    if not (theta.numel() == grad.numel() == m.numel() == v.numel()):
        raise ValueError("adam: buffer sizes differ")
    check(lib.sv_adam_keras_apply(_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), theta.numel(), _ptr(state, torch.int64),
                                  beta1, beta2, eps, grad_scale, int(zero_grad), _stream()), "sv_adam_keras_apply")


# --------------------------------------------------------------------------- batching helpers
def gather_rows(src, idx, dst):
    """dst[i] = float32(src[idx[i]]); src float32 or float64, idx int64 or None (identity)."""
    if src.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"gather_rows: source dtype {src.dtype}")
    n_rows = dst.shape[0]
    row = dst.numel() // n_rows
    if src.numel() // src.shape[0] != row:
        raise ValueError("gather_rows: row sizes differ")
    if idx is None and src.shape[0] < n_rows:
        raise ValueError("gather_rows: source has fewer rows than destination")
    check(lib.sv_gather_rows(_ptr(src, src.dtype), int(src.dtype == torch.float64), _ptr(idx, torch.int64), _ptr(dst),
                             n_rows, row, _stream()), "sv_gather_rows")


# --------------------------------------------------------------------------- data preparation
def process_regions(x, positions, offsets, n_regions: int, n_haps: int, n_mut: int, n_snps: int, n_channels: int,
                    norm_len: int, out, distances, status):
    check(lib.sv_process_regions(_ptr(x), _ptr(positions, torch.int64), _ptr(offsets, torch.int64), n_regions, n_haps,
                                 n_mut, n_snps, n_channels, norm_len, _ptr(out), _ptr(distances, torch.float64),
                                 _ptr(status, torch.int32), _stream()), "sv_process_regions")


def sort_min_diff(images, channel: int, order):
    n, h, s, c = images.shape
    check(lib.sv_sort_min_diff(_ptr(images), n, h, s, c, channel, _ptr(order, torch.int64), _stream()),
          "sv_sort_min_diff")


def reorder_rows(images, order, out):
    n, h = order.shape
    check(lib.sv_reorder_rows(_ptr(images), _ptr(order, torch.int64), _ptr(out), n, h, images[0, 0].numel(), _stream()),
          "sv_reorder_rows")


def mask_input(x, mask, y):
    """y = x * (1 - mask); x and mask float32, y float32 or bfloat16."""
    if not (x.numel() == mask.numel() == y.numel()):
        raise ValueError("mask_input: sizes differ")
    check(lib.sv_mask_input_dt(_ptr(x), _ptr(mask), _aptr(y), x.numel(), _dt(y), _stream()), "sv_mask_input_dt")


def random_block_mask(mask, pct_mask: float, seed: int, offset: int = 0):
    """Fill the NHWC float32 tensor ``mask`` with random block masks (equal over channels), one per image."""
    if mask.dim() != 4:
        raise ValueError("random_block_mask: an NHWC tensor is expected")
    n, h, w, c = mask.shape
    check(lib.sv_random_block_mask(_ptr(mask), n, h, w, c, float(pct_mask), int(seed), int(offset), _stream()),
          "sv_random_block_mask")


def fill(t, value: float):
    check(lib.sv_fill_f32(_ptr(t), value, t.numel(), _stream()), "sv_fill_f32")


def axpy(y, x, a: float):
    """y += a * x (both float32 or both bfloat16)."""
    if x.numel() != y.numel():
        raise ValueError("axpy: sizes differ")
    check(lib.sv_axpy(_aptr(y), _aptr(x, y, "x"), a, x.numel(), _dt(y), _stream()), "sv_axpy")


def copy(src, dst):
    """dst = src, same type and size (a device-to-device copy on the current stream)."""
    if src.numel() != dst.numel() or src.dtype != dst.dtype:
        raise ValueError("copy: size or type differs")
    dst.as_subclass(torch.Tensor).view(-1).copy_(src.as_subclass(torch.Tensor).reshape(-1), non_blocking=True)


def cast(src, dst):
    """dst = src converted between float32 and bfloat16 (either direction)."""
    if src.numel() != dst.numel():
        raise ValueError("cast: sizes differ")
    if src.dtype == torch.float32 and dst.dtype == torch.bfloat16:
        check(lib.sv_cast_f32_bf16(_ptr(src), _ptr(dst, torch.bfloat16), src.numel(), _stream()), "sv_cast_f32_bf16")
    elif src.dtype == torch.bfloat16 and dst.dtype == torch.float32:
        check(lib.sv_cast_bf16_f32(_ptr(src, torch.bfloat16), _ptr(dst), src.numel(), _stream()), "sv_cast_bf16_f32")
    else:
        raise TypeError(f"cast: {src.dtype} -> {dst.dtype}")
