"""Naive float64 NumPy restatement of the tf.keras-2 primitives on the hot path.

Test infrastructure (see ``oracle/__init__.py``).  Every function loops over
kernel taps / output pixels explicitly so that it shares no code path with the
torch restatement in ``oracle/torch_ref.py`` or with the HIP kernels.

Layouts are the reference's (tf.keras defaults): activations NHWC, Conv2D
kernels HWIO ``(kh, kw, Cin, Cout)``, Conv2DTranspose kernels
``(kh, kw, Cout, Cin)``, Dense kernels ``(in, out)``.

Call sites in the reference that these restate:
  Conv2D               autoencoder.py:443,510,521  unet.py:40  context_encoder.py:57,72,172,185,230,242,334,346
  Conv2DTranspose      autoencoder.py:371,382  unet.py:69,123  context_encoder.py:92,105,119,260,276,290
  MaxPooling2D         autoencoder.py:33,98,167,319,451
  UpSampling2D         autoencoder.py:364,503
  BatchNormalization   autoencoder.py:318,450,468  unet.py:51  context_encoder.py:64,88,179,237,267,283,341
  Dense                autoencoder.py:38-40,107,110,176,179,222-241,330-352,466-492
  Sampling             autoencoder.py:264-272
  losses               run_model.py:140,281,361-362  run_context_encoder.py:165-198  run_unet.py:69-161
  legacy Adam + decay  run_model.py:77-89  run_unet.py:48-66  run_context_encoder.py:144-162
"""
from __future__ import annotations

import math

import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------- geometry
def same_pads(in_size: int, k: int, s: int):
    """tf SAME padding: out = ceil(in/s); total = max((out-1)s + k - in, 0);
    before = total // 2, remainder after."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def valid_out(in_size: int, k: int, s: int) -> int:
    return (in_size - k) // s + 1


def conv_geometry(h, w, kh, kw, sh, sw, padding):
    """Returns (Ho, Wo, pad_top, pad_left)."""
    if padding == "same":
        ho, pt, _ = same_pads(h, kh, sh)
        wo, pl, _ = same_pads(w, kw, sw)
    elif padding == "valid":
        ho, wo, pt, pl = valid_out(h, kh, sh), valid_out(w, kw, sw), 0, 0
    else:
        raise ValueError(padding)
    return ho, wo, pt, pl


def deconv_out_size(in_size: int, k: int, s: int, padding: str) -> int:
    """keras conv_utils.deconv_output_length with output_padding=None."""
    if padding == "same":
        return in_size * s
    return in_size * s + max(k - s, 0)


# --------------------------------------------------------------------------- activations
def act_fwd(z, act, alpha=0.2):
    z = np.asarray(z, F64)
    if act in (None, "linear", "none"):
        return z.copy()
    if act == "relu":
        return np.maximum(z, 0.0)
    if act == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0.0)))
    if act == "lrelu":
        return np.where(z >= 0, z, alpha * z)
    if act == "tanh":
        return np.tanh(z)
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    raise ValueError(act)


def act_bwd_from_output(y, act, alpha=0.2):
    """d act / d z expressed through the activation OUTPUT y (what the HIP path stores)."""
    y = np.asarray(y, F64)
    if act in (None, "linear", "none"):
        return np.ones_like(y)
    if act == "relu":
        return (y > 0).astype(F64)
    if act == "elu":
        return np.where(y > 0, 1.0, y + 1.0)
    if act == "lrelu":
        return np.where(y > 0, 1.0, alpha)  # TensorFlow's LeakyReluGrad tests features > 0
    if act == "tanh":
        return 1.0 - y * y
    if act == "sigmoid":
        return y * (1.0 - y)
    raise ValueError(act)


# --------------------------------------------------------------------------- conv
def conv2d(x, w, b=None, stride=(1, 1), padding="same"):
    """Cross-correlation, NHWC x HWIO -> NHWC (pre-activation)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    assert cin == cin2
    sh, sw = stride
    ho, wo, pt, pl = conv_geometry(h, wd, kh, kw, sh, sw, padding)
    y = np.zeros((n, ho, wo, cout), F64)
    for oh in range(ho):
        for ow in range(wo):
            for i in range(kh):
                ih = oh * sh + i - pt
                if ih < 0 or ih >= h:
                    continue
                for j in range(kw):
                    iw = ow * sw + j - pl
                    if iw < 0 or iw >= wd:
                        continue
                    y[:, oh, ow, :] += x[:, ih, iw, :] @ w[i, j]
    if b is not None:
        y += np.asarray(b, F64)
    return y


def conv2d_grads(x, w, dz, stride=(1, 1), padding="same"):
    """Gradients of conv2d wrt x, w, b given dz = dL/d(pre-activation)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    dz = np.asarray(dz, F64)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    sh, sw = stride
    ho, wo, pt, pl = conv_geometry(h, wd, kh, kw, sh, sw, padding)
    assert dz.shape == (n, ho, wo, cout)
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for oh in range(ho):
        for ow in range(wo):
            for i in range(kh):
                ih = oh * sh + i - pt
                if ih < 0 or ih >= h:
                    continue
                for j in range(kw):
                    iw = ow * sw + j - pl
                    if iw < 0 or iw >= wd:
                        continue
                    dx[:, ih, iw, :] += dz[:, oh, ow, :] @ w[i, j].T
                    dw[i, j] += x[:, ih, iw, :].T @ dz[:, oh, ow, :]
    db = dz.sum(axis=(0, 1, 2))
    return dx, dw, db


def conv2d_transpose(x, w, b=None, stride=(1, 1), padding="same"):
    """Keras Conv2DTranspose: scatter every input pixel through the kernel
    (kh, kw, Cout, Cin), then crop by the SAME 'before' pad of the forward conv
    whose input is the output of this layer."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert cin == cin2
    sh, sw = stride
    ho = deconv_out_size(h, kh, sh, padding)
    wo = deconv_out_size(wd, kw, sw, padding)
    # geometry of the forward conv (ho, wo) -> (h, wd)
    fh, fw, pt, pl = conv_geometry(ho, wo, kh, kw, sh, sw, padding)
    assert (fh, fw) == (h, wd), ((fh, fw), (h, wd))
    full_h = (h - 1) * sh + kh
    full_w = (wd - 1) * sw + kw
    full = np.zeros((n, max(full_h, ho + pt), max(full_w, wo + pl), cout), F64)
    for ih in range(h):
        for iw in range(wd):
            for i in range(kh):
                for j in range(kw):
                    full[:, ih * sh + i, iw * sw + j, :] += x[:, ih, iw, :] @ w[i, j].T
    y = full[:, pt:pt + ho, pl:pl + wo, :].copy()
    if b is not None:
        y += np.asarray(b, F64)
    return y


def conv2d_transpose_grads(x, w, dz, stride=(1, 1), padding="same"):
    """Gradients of conv2d_transpose wrt x, w, b (dz wrt pre-activation output)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    dz = np.asarray(dz, F64)
    n, h, wd, cin = x.shape
    kh, kw, cout, _ = w.shape
    sh, sw = stride
    ho = deconv_out_size(h, kh, sh, padding)
    wo = deconv_out_size(wd, kw, sw, padding)
    _, _, pt, pl = conv_geometry(ho, wo, kh, kw, sh, sw, padding)
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for ih in range(h):
        for iw in range(wd):
            for i in range(kh):
                oh = ih * sh + i - pt
                if oh < 0 or oh >= ho:
                    continue
                for j in range(kw):
                    ow = iw * sw + j - pl
                    if ow < 0 or ow >= wo:
                        continue
                    # y[oh,ow,co] += x[ih,iw,ci] * w[i,j,co,ci]
                    dx[:, ih, iw, :] += dz[:, oh, ow, :] @ w[i, j]
                    dw[i, j] += dz[:, oh, ow, :].T @ x[:, ih, iw, :]
    db = dz.sum(axis=(0, 1, 2))
    return dx, dw, db


# --------------------------------------------------------------------------- pooling / resampling
def maxpool2d(x, pool=(2, 2), padding="valid"):
    """MaxPooling2D with stride == pool size.  Returns (y, argmax index inside the
    window in row-major scan order, first max wins)."""
    x = np.asarray(x, F64)
    n, h, wd, c = x.shape
    ph, pw = pool
    if padding == "same":
        ho, pt, _ = same_pads(h, ph, ph)
        wo, pl, _ = same_pads(wd, pw, pw)
    else:
        ho, wo, pt, pl = valid_out(h, ph, ph), valid_out(wd, pw, pw), 0, 0
    y = np.full((n, ho, wo, c), -np.inf, F64)
    arg = np.zeros((n, ho, wo, c), np.int64)
    for oh in range(ho):
        for ow in range(wo):
            for i in range(ph):
                ih = oh * ph + i - pt
                if ih < 0 or ih >= h:
                    continue
                for j in range(pw):
                    iw = ow * pw + j - pl
                    if iw < 0 or iw >= wd:
                        continue
                    v = x[:, ih, iw, :]
                    better = v > y[:, oh, ow, :]
                    y[:, oh, ow, :] = np.where(better, v, y[:, oh, ow, :])
                    arg[:, oh, ow, :] = np.where(better, i * pw + j, arg[:, oh, ow, :])
    return y, arg


def maxpool2d_bwd(dy, arg, in_shape, pool=(2, 2), padding="valid"):
    dy = np.asarray(dy, F64)
    n, h, wd, c = in_shape
    ph, pw = pool
    if padding == "same":
        _, pt, _ = same_pads(h, ph, ph)
        _, pl, _ = same_pads(wd, pw, pw)
    else:
        pt = pl = 0
    dx = np.zeros(in_shape, F64)
    _, ho, wo, _ = dy.shape
    for oh in range(ho):
        for ow in range(wo):
            for i in range(ph):
                for j in range(pw):
                    ih, iw = oh * ph + i - pt, ow * pw + j - pl
                    if ih < 0 or ih >= h or iw < 0 or iw >= wd:
                        continue
                    sel = arg[:, oh, ow, :] == (i * pw + j)
                    dx[:, ih, iw, :] += np.where(sel, dy[:, oh, ow, :], 0.0)
    return dx


def upsample2d(x, size=(2, 2)):
    x = np.asarray(x, F64)
    return np.repeat(np.repeat(x, size[0], axis=1), size[1], axis=2)


def upsample2d_bwd(dy, size=(2, 2)):
    dy = np.asarray(dy, F64)
    n, h, w, c = dy.shape
    uh, uw = size
    return dy.reshape(n, h // uh, uh, w // uw, uw, c).sum(axis=(2, 4))


# --------------------------------------------------------------------------- batch norm
def batchnorm_train(x, gamma, beta, eps=1e-3):
    """Normalise over every axis but the last with the biased batch variance
    (eps inside the sqrt).  Returns (y, mean, biased var)."""
    x = np.asarray(x, F64)
    red = tuple(range(x.ndim - 1))
    mean = x.mean(axis=red)
    var = x.var(axis=red)
    y = (x - mean) / np.sqrt(var + eps) * gamma + beta
    return y, mean, var


def batchnorm_moving_update(moving_mean, moving_var, mean, var, count, momentum=0.99,
                            unbiased_var=True):
    """moving <- momentum*moving + (1-momentum)*batch.  The fused 4-D kernel of
    Keras-2 feeds the UNBIASED batch variance into the moving variance."""
    v = var * (count / max(count - 1, 1)) if unbiased_var else var
    return (momentum * moving_mean + (1 - momentum) * mean,
            momentum * moving_var + (1 - momentum) * v)


def batchnorm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-3):
    x = np.asarray(x, F64)
    return (x - moving_mean) / np.sqrt(moving_var + eps) * gamma + beta


def batchnorm_train_bwd(x, gamma, dy, eps=1e-3):
    x = np.asarray(x, F64)
    dy = np.asarray(dy, F64)
    red = tuple(range(x.ndim - 1))
    m = np.prod([x.shape[a] for a in red])
    mean = x.mean(axis=red)
    var = x.var(axis=red)
    inv = 1.0 / np.sqrt(var + eps)
    xh = (x - mean) * inv
    dbeta = dy.sum(axis=red)
    dgamma = (dy * xh).sum(axis=red)
    dx = gamma * inv / m * (m * dy - dbeta - xh * dgamma)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- dense / misc
def dense(x, w, b=None):
    y = np.asarray(x, F64) @ np.asarray(w, F64)
    if b is not None:
        y = y + np.asarray(b, F64)
    return y


def dense_grads(x, w, dz):
    x, w, dz = (np.asarray(a, F64) for a in (x, w, dz))
    return dz @ w.T, x.T @ dz, dz.sum(axis=0)


def dropout(x, keep_mask, rate):
    return np.asarray(x, F64) * keep_mask / (1.0 - rate)


def sampling(z_mean, z_log_var, eps):
    """autoencoder.py:264-272 with epsilon supplied by the caller."""
    return np.asarray(z_mean, F64) + np.exp(0.5 * np.asarray(z_log_var, F64)) * eps


def kl_standard_normal(z_mean, z_log_var):
    """-1/2 sum(1 + logvar - mu^2 - exp(logvar)) per sample, mean over the batch.
    No counterpart in the reference (it defines Sampling but no KL term)."""
    mu, lv = np.asarray(z_mean, F64), np.asarray(z_log_var, F64)
    per = -0.5 * np.sum(1 + lv - mu * mu - np.exp(lv), axis=1)
    return per.mean()


# --------------------------------------------------------------------------- losses
def mse(y_true, y_pred):
    """keras MeanSquaredError, reduction AUTO: mean over the last axis, then mean
    over the remaining elements == mean over all elements."""
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    return float(np.mean(d * d))


def mse_grad(y_true, y_pred):
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    return 2.0 * d / d.size


def mse_weighted(y_true, y_pred, sample_weight):
    """MeanSquaredError()(y, p, sample_weight=w) with w of shape (B,H,W):
    sum_bhw(w * mean_c (y-p)^2) / (B*H*W)   (SUM_OVER_BATCH_SIZE divides by the
    number of per-pixel losses, not by sum(w))."""
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    per = np.mean(d * d, axis=-1) * np.asarray(sample_weight, F64)
    return float(per.sum() / per.size)


def mse_weighted_grad(y_true, y_pred, sample_weight):
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    c = d.shape[-1]
    npx = d.size // c
    return 2.0 * d * np.asarray(sample_weight, F64)[..., None] / (c * npx)


def mse_per_pixel(y_true, y_pred, sample_weight=None):
    """reduction='none': (B,H,W) map (run_context_encoder.py:493-501, run_model.py:361)."""
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    per = np.mean(d * d, axis=-1)
    if sample_weight is not None:
        per = per * np.asarray(sample_weight, F64)
    return per


def mae(y_true, y_pred):
    return float(np.mean(np.abs(np.asarray(y_pred, F64) - np.asarray(y_true, F64))))


def mae_grad(y_true, y_pred):
    d = np.asarray(y_pred, F64) - np.asarray(y_true, F64)
    return np.sign(d) / d.size


KERAS_EPS = 1e-7


def bce_probs(y_true, p, dtype=F64):
    """BinaryCrossentropy(from_logits=False): clip p to [eps, 1-eps], eps = 1e-7;
    -mean(t*log(p+eps) + (1-t)*log(1-p+eps)) (keras backend adds eps inside the logs).
    ``dtype=np.float32`` evaluates the clip and the logs in float32 as TensorFlow does: for saturated
    predictions 1-p+eps is a handful of float32 ulps, so the two precisions differ visibly there."""
    t = np.asarray(y_true, dtype)
    eps = dtype(KERAS_EPS)
    one = dtype(1.0)
    p = np.clip(np.asarray(p, dtype), eps, one - eps)
    per = -(t * np.log(p + eps) + (one - t) * np.log(one - p + eps))
    return float(np.mean(per.astype(F64)))


def bce_probs_grad(y_true, p):
    t = np.asarray(y_true, F64)
    p0 = np.asarray(p, F64)
    inside = (p0 >= KERAS_EPS) & (p0 <= 1.0 - KERAS_EPS)
    pc = np.clip(p0, KERAS_EPS, 1.0 - KERAS_EPS)
    g = -(t / (pc + KERAS_EPS) - (1 - t) / (1 - pc + KERAS_EPS)) / p0.size
    return np.where(inside, g, 0.0)


def bce_logits(y_true, z):
    """BinaryCrossentropy(from_logits=True): max(z,0) - z*t + log1p(exp(-|z|)), mean."""
    t = np.asarray(y_true, F64)
    z = np.asarray(z, F64)
    return float(np.mean(np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))))


def bce_logits_grad(y_true, z):
    t = np.asarray(y_true, F64)
    z = np.asarray(z, F64)
    return (1.0 / (1.0 + np.exp(-z)) - t) / z.size


# --------------------------------------------------------------------------- optimiser
def exponential_decay(lr0, step, decay_steps, decay_rate, staircase=True):
    """tf.keras.optimizers.schedules.ExponentialDecay; ``step`` is the number of
    updates applied so far (0 for the first update)."""
    p = step / decay_steps
    if staircase:
        p = math.floor(p)
    return lr0 * decay_rate ** p


def adam_keras_step(theta, g, m, v, it, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras legacy Adam (ResourceApplyAdam): t = it+1;
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= alpha*m/(sqrt(v)+eps).
    Returns new (theta, m, v)."""
    t = it + 1
    alpha = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    theta = theta - alpha * m / (np.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------- initialisers
def glorot_uniform_limit(shape):
    """keras glorot_uniform: fans = receptive field * channels."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    return math.sqrt(6.0 / (fan_in + fan_out))


# --------------------------------------------------------------------------- device-side random block masks
_M64 = (1 << 64) - 1


def mix64(x: int) -> int:
    """The counter hash of the HIP library's generators (splitmix64 finaliser), on Python ints."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def random_block_mask(n, h, w, c, pct_mask=0.25, seed=0, offset=0, max_draws=1 << 16):
    """create_random_image_mask / batch_random_mask (run_context_encoder.py:24-58) with the reference's
    ``np.random.randint`` draws replaced by the library's counter hash, so the device kernel can be checked bit for bit:
    draw d of image i takes ``(mix64(seed ^ mix64(offset + (i << 20) + 2 d [+ 1])) >> 11) % (w - seg)``."""
    out = np.zeros((n, h, w, c))
    seg = w // 8
    for i in range(n):
        mask = np.zeros((h, w, c))
        mask[seg:seg * 3, seg:seg * 3, :] = 1
        base = offset + (i << 20)
        d = 0
        while (np.sum(mask[:, :, 0]) / (w ** 2)) < pct_mask and d < max_draws:
            rx = (mix64(seed ^ mix64((base + 2 * d) & _M64)) >> 11) % (w - seg)
            ry = (mix64(seed ^ mix64((base + 2 * d + 1) & _M64)) >> 11) % (w - seg)
            mask[rx:rx + seg, ry:ry + seg, :] = 1
            d += 1
        out[i] = mask
    return out
