"""GPU parity at model level: the HIP engine against the torch-CPU restatement with IDENTICAL weights.

Bars (the north star's): activations and gradients at relative error <= 1e-4 (fp32), loss trajectory
within 1e-3 over 100 fixed-seed optimizer steps.  Relative error of a tensor is
max|hip - ref| / max(max|ref|, floor) with a small floor so all-zero gradients do not divide by zero.
"""
import os

import numpy as np
import pytest
import torch

from oracle import models as omodels
from oracle import torch_ref
from oracle.torch_ref import KerasAdam, RefSequential

pytestmark = pytest.mark.gpu


def rel(got, want, floor=1e-3):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all()
    return float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), floor))


def data(n, seed=1833, c=3):
    from spectrum_vae_amd.data import synth_training_tensor
    return synth_training_tensor(n, seed)[..., :c]


def load_into_ref(ref, model):
    ref.load_params([v.numpy() for v in model.trainable_variables])


def grads_of(model):
    return [v.grad.detach().cpu().numpy().copy() for v in model.trainable_variables]


def _ref_pass(ref, x, tgt, dtype, dropout_masks=None):
    xt = torch.from_numpy(np.asarray(x, np.float32)).to(dtype)
    for p in ref.params:
        p.grad = None
    masks = None if dropout_masks is None else {k: v.to(dtype) for k, v in dropout_masks.items()}
    pr = ref(xt, training=True) if masks is None else ref(xt, training=True, dropout_masks=masks)
    loss = torch_ref.mse(torch.from_numpy(tgt).to(dtype), pr)
    loss.backward()
    return pr.detach().numpy(), float(loss), [p.grad.numpy().copy() for p in ref.params]


def engine_dropout_masks(model, ref_layers, batch):
    """Keep masks (0 / 1, float32) the model's Dropout layers will draw in their NEXT training call, keyed by the
    position of the matching dropout entry in the oracle's layer list: the engine's mask is a pure function of
    (seed, offset, element index) - sv_dropout_fwd on a tensor of ones returns keep / (1 - rate)."""
    from spectrum_vae_amd import ops
    from spectrum_vae_amd.nn import Dropout
    drops = [l for seq in model._sequentials() for l in seq.layers if isinstance(l, Dropout) and l.rate > 0]
    slots = [i for i, l in enumerate(ref_layers) if l["type"] == "dropout" and l["rate"] > 0]
    assert len(drops) == len(slots) and drops, "every dropout of the model has its entry in the oracle's list"
    masks = {}
    for l, li in zip(drops, slots):
        shape = (batch,) + tuple(l.input_shape_)
        ones = torch.ones(shape, device="cuda")
        y = torch.empty_like(ones)
        ops.dropout_fwd(ones, y, l.rate, l.seed + 7919 * l.stream_id, l.calls * ones.numel())
        keep = (y != 0).float().cpu()
        assert abs(float(keep.mean()) - (1.0 - l.rate)) < 0.15, "the mask keeps about 1 - rate of the units"
        masks[li] = keep
    return masks


def check_fwd_bwd(model, ref, x, tol=1e-4, target=None, ref64=None, dropout_layers=None):
    """Training forward, loss, gradient of every weight, BatchNorm state, inference forward.

    The bar is the float32 CPU restatement at ``tol``.  Gradients upstream of a BatchNorm are sums with heavy
    cancellation, where float32 round-off alone exceeds 1e-4 (the CPU restatement itself misses a float64
    evaluation by that much); when ``ref64`` (the same model in float64) is given, a gradient also passes if
    it is at least as close to the float64 truth as 3x the float32 CPU restatement's own distance from it."""
    from spectrum_vae_amd import losses
    from spectrum_vae_amd.losses import to_device_f32
    xd = to_device_f32(x)
    model.build((None,) + tuple(x.shape[1:]))
    model._materialize()
    load_into_ref(ref, model)
    tgt = np.asarray(x if target is None else target, np.float32)
    model.zero_grad()
    # dropout: the masks of the training call about to run, replayed in the CPU restatement
    masks = engine_dropout_masks(model, dropout_layers, x.shape[0]) if dropout_layers is not None else None
    pred = model.forward_raw(xd, True)
    out = torch.empty(1, device="cuda")
    dpred = torch.empty_like(pred)
    losses.MeanSquaredError().value_and_grad(to_device_f32(tgt), pred, out, dpred)
    pred_host = pred.cpu().numpy()
    model.backward_raw(dpred, -1, need_dx=False)
    pr, loss, g32 = _ref_pass(ref, x, tgt, torch.float32, masks)
    g64 = None
    if ref64 is not None:
        state32 = [s.clone() for s in ref.state]
        load_into_ref(ref64, model)
        _, _, g64 = _ref_pass(ref64, x, tgt, torch.float64, masks)
    assert rel(pred_host, pr) <= tol, "training forward"
    assert abs(float(out) - loss) <= tol * max(1.0, abs(loss)), "loss"
    for i, (v, g) in enumerate(zip(model.trainable_variables, grads_of(model))):
        floor = 1e-6 + 1e-3 * float(np.abs(g32[i]).max())
        e = rel(g, g32[i], floor=floor)
        if e > tol and g64 is not None:
            e_hip, e_cpu = rel(g, g64[i], floor=floor), rel(g32[i], g64[i], floor=floor)
            assert e_hip <= max(tol, 3 * e_cpu), f"grad of {v.name}: {e_hip:.2e} vs float64 (CPU f32: {e_cpu:.2e})"
        else:
            assert e <= tol, f"grad of {v.name}: {e:.2e}"
    for v, s in zip(model.non_trainable_variables, ref.state):
        assert rel(v.numpy(), s.numpy()) <= tol, f"state {v.name}"
    y_inf = model.predict(x, batch_size=max(1, x.shape[0] // 2 + 1))
    xt = torch.from_numpy(np.asarray(x, np.float32))
    assert rel(y_inf, ref(xt, training=False).detach().numpy()) <= tol, "inference forward"
    model._calls.clear()


RUN_MODEL_DEFAULTS = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
                          conv_operations=1, fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0,
                          batch_norm=False)


@pytest.mark.parametrize("cfg", [
    RUN_MODEL_DEFAULTS,
    dict(RUN_MODEL_DEFAULTS, batch_norm=True, activation="relu", conv_operations=2, kernel_size=5),
    dict(RUN_MODEL_DEFAULTS, conv_layers=4, conv_layer_multiplier=1, initial_filters=16, conv_operations=3),
    dict(RUN_MODEL_DEFAULTS, latent_dimensions=8, fc_layers=2, fc_layer_size=64, batch_norm=True),
    dict(),  # constructor defaults
])
def test_caepadded_matches_oracle(cfg):
    from spectrum_vae_amd import autoencoder
    x = data(6)
    model = autoencoder.CAEPadded(**cfg)
    enc, dec = omodels.caepadded(**cfg)
    ref = RefSequential(enc + dec, (32, 32, 3))
    ref64 = RefSequential(enc + dec, (32, 32, 3), dtype=torch.float64)
    model.build((None, 32, 32, 3))
    assert model.count_params() - sum(v.size for v in model.non_trainable_variables) == ref.num_params()
    check_fwd_bwd(model, ref, x, ref64=ref64)


def test_caepadded_per_layer_activations():
    """Every conv output of the default model, layer by layer."""
    from spectrum_vae_amd import autoencoder
    from spectrum_vae_amd.losses import to_device_f32
    from spectrum_vae_amd.nn import Conv2D
    x = data(4)
    model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    model.build((None, 32, 32, 3))
    model._materialize()
    enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
    ref = RefSequential(enc + dec, (32, 32, 3))
    load_into_ref(ref, model)
    model.forward_raw(to_device_f32(x), True)
    ctx = model._calls[-1]
    taps = {}
    ref(torch.from_numpy(x), training=True, taps=taps)
    # a conv whose 2x2 max-pool is fused into its epilogue saves the POOLED activation: compare with the
    # oracle's pooling output there, with the conv output elsewhere
    ref_outs = []
    for i, l in enumerate(ref.layers):
        if l["type"] == "conv2d":
            pooled_next = i + 1 < len(ref.layers) and ref.layers[i + 1]["type"] == "maxpool"
            ref_outs.append((taps[i].detach().numpy(), taps[i + 1].detach().numpy() if pooled_next else None))
    hip_convs = [l for l in model.encoder.layers + model.decoder.layers if isinstance(l, Conv2D)]
    assert len(ref_outs) == len(hip_convs) == 6
    assert [l.pool is not None for l in hip_convs] == [True, True, True, False, False, False]
    for i, (l, (full, pooled)) in enumerate(zip(hip_convs, ref_outs)):
        got = ctx.saved[id(l)]["y"].cpu().numpy()
        assert rel(got, pooled if l.pool is not None else full) <= 1e-4, f"conv {i}"
    model._calls.clear()


def test_cae1d_and_classifiers():
    from spectrum_vae_amd import autoencoder
    x = data(4)
    m = autoencoder.CAE1D(conv_layers=2, conv_operations=1, latent_dimensions=0)
    enc, dec = omodels.cae1d(conv_layers=2, conv_operations=1, latent_dimensions=0)
    ref = RefSequential(enc + dec, (32, 32, 3))
    m.build((None, 32, 32, 3))
    tgt = np.random.default_rng(0).standard_normal((4,) + m.output_shape_).astype(np.float32)
    check_fwd_bwd(m, ref, x, target=tgt)
    for cls, layers in ((autoencoder.CNNRiley, omodels.cnnriley()),
                        (autoencoder.CNNPadded, omodels.cnnpadded(dropout=0.0)),
                        (autoencoder.CNN1D, omodels.cnn1d(conv_layers=2, dropout=0.0))):
        kw = {} if cls is autoencoder.CNNRiley else {"dropout": 0.0}
        if cls is autoencoder.CNN1D:
            kw["conv_layers"] = 2
        m = cls(**kw)
        ref = RefSequential(layers, (32, 32, 3))
        tgt = np.random.default_rng(1).standard_normal((4, 1)).astype(np.float32)
        check_fwd_bwd(m, ref, x, target=tgt)
    a = autoencoder.AE(dropout=0.0)
    enc, dec = omodels.ae(dropout=0.0)
    check_fwd_bwd(a, RefSequential(enc + dec, (32, 3)), data(4)[:, 0])


def test_classifier_and_cae1d_with_dropout_replayed():
    """Training passes with dropout ON: the engine's keep masks (sv_dropout_fwd's (seed, offset) stream) are replayed in
    the CPU restatement, so forward, loss and every gradient are compared with the units actually dropped.
    CNNPadded at its constructor defaults (autoencoder.py:129-181, dropout 0.5) and CAE1D at its constructor defaults
    (autoencoder.py:275-394: three blocks of two (1, 3) valid convs, latent 8, dropout 0.5) on the narrowest input its
    width arithmetic accepts, 36 columns (36 -> 16 -> 6 -> 1)."""
    from spectrum_vae_amd import autoencoder
    x = data(8)
    m = autoencoder.CNNPadded()
    layers = omodels.cnnpadded()
    tgt = np.random.default_rng(2).standard_normal((8, 1)).astype(np.float32)
    check_fwd_bwd(m, RefSequential(layers, (32, 32, 3)), x, target=tgt, dropout_layers=layers)
    # a second call draws the next masks of the stream: different units, still in agreement
    first = engine_dropout_masks(m, layers, 8)
    check_fwd_bwd(m, RefSequential(layers, (32, 32, 3)), x, target=tgt, dropout_layers=layers)
    assert any(not torch.equal(a, b) for a, b in zip(first.values(), engine_dropout_masks(m, layers, 8).values()))
    x36 = np.concatenate([data(8), data(8, seed=5)[:, :, :4]], axis=2)
    c = autoencoder.CAE1D(input_shape=(32, 36, 3))
    enc, dec = omodels.cae1d(input_shape=(32, 36, 3))
    c.build((None, 32, 36, 3))
    assert c.output_shape_ == (32, 36, 3)
    tgt = np.random.default_rng(3).standard_normal((8, 32, 36, 3)).astype(np.float32)
    check_fwd_bwd(c, RefSequential(enc + dec, (32, 36, 3)), x36, target=tgt, dropout_layers=enc + dec)


def test_cae1d_at_its_constructor_defaults_fails_as_the_reference_does():
    """autoencoder.py:277-294 with nothing overridden: 32 columns shrink 32 -> 14 -> 5 -> (1 after two valid (1, 3)
    convs), which the third (1, 2) pooling cannot halve - Keras raises while building the encoder; so does this model."""
    from spectrum_vae_amd import autoencoder
    with pytest.raises(ValueError):
        autoencoder.CAE1D().build((None, 32, 32, 3))


def test_unet_matches_oracle():
    from spectrum_vae_amd import unet
    x = data(4, c=1)
    y = data(4)
    model = unet.ContextEncoder()
    ref = omodels.RefUNet()
    model.build((None, 32, 32, 1))
    assert sum(v.size for v in model.trainable_variables) == ref.num_params() == 133_379
    check_fwd_bwd(model, ref, x, target=y)


def test_context_encoder_and_discriminator_match_oracle():
    from spectrum_vae_amd import context_encoder
    x = data(4)
    ce = context_encoder.ContextEncoder()
    enc, dec = omodels.context_encoder()
    ref = RefSequential(enc + dec, (32, 32, 3))
    ce.build((None, 32, 32, 3))
    assert sum(v.size for v in ce.trainable_variables) == ref.num_params() == 4_497_899
    check_fwd_bwd(ce, ref, x)
    d = context_encoder.Discriminator()
    rd = RefSequential(omodels.discriminator(), (32, 32, 3))
    d.build((None, 32, 32, 3))
    assert sum(v.size for v in d.trainable_variables) == rd.num_params() == 202_497
    check_fwd_bwd(d, rd, x, target=np.full((4, 1), 0.3, np.float32))
    oh = context_encoder.ContextEncoderOneHot()
    enc, dec = omodels.context_encoder_onehot()
    xo = np.random.default_rng(3).standard_normal((4, 4, 32, 3)).astype(np.float32)
    ro = RefSequential(enc + dec, (4, 32, 3))
    oh.build((None, 4, 32, 3))
    tgt = np.random.default_rng(4).standard_normal((4,) + oh.output_shape_).astype(np.float32)
    check_fwd_bwd(oh, ro, xo, target=tgt)
    do = context_encoder.DiscriminatorOneHot()
    check_fwd_bwd(do, RefSequential(omodels.discriminator_onehot(), (4, 16, 3)), xo[:, :, :16],
                  target=np.full((4, 1), 0.7, np.float32))


def test_trajectory_100_steps_run_model_config():
    """100 optimizer steps of the run_model.py configuration (MSE, Adam lr 5e-4, decay .96/10 steps, batch 32,
    fixed batch order): loss trajectory within 1e-3 of the CPU restatement.  Steps 3..100 replay a HIP graph."""
    from spectrum_vae_amd import autoencoder, losses, ops
    from spectrum_vae_amd.run_model import build_optimizer
    X = data(32 * 10)
    model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    model.build((None, 32, 32, 3))
    model._materialize()
    enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
    ref = RefSequential(enc + dec, (32, 32, 3))
    load_into_ref(ref, model)
    opt_ref = KerasAdam(ref.params, 5e-4, decay_steps=10, decay_rate=0.96, staircase=True)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    Xd = torch.from_numpy(X).cuda()
    b = model._step_buffers(32, (32, 32, 3))
    hip, cpu = [], []
    for step in range(100):
        sl = slice((step % 10) * 32, (step % 10 + 1) * 32)
        ops.gather_rows(Xd[sl], None, b["x"])
        model.train_on_device_batch(b, True, 32.0)
        hip.append(float(b["loss"].item()))
        xb = torch.from_numpy(X[sl])
        cpu.append(torch_ref.train_step_mse(ref, opt_ref, xb, xb))
    hip, cpu = np.array(hip), np.array(cpu)
    assert len(model._graphs) == 1 and not isinstance(next(iter(model._graphs.values())), str), "graph not captured"
    assert np.max(np.abs(hip - cpu) / np.maximum(np.abs(cpu), 1e-6)) <= 1e-3, (hip[-5:], cpu[-5:])
    assert cpu[-1] < cpu[0]
    assert model.optimizer.iterations == 100
    for v, p in zip(model.trainable_variables, ref.params):
        assert rel(v.numpy(), p.detach().numpy()) <= 1e-3, v.name


def test_replayed_steps_survive_a_manual_backward_in_between():
    """The recorded step leaves out the gradient clear (its optimizer launch does it).  A manual backward between
    two replays leaves gradients in the buffer; the next replayed step must not add to them."""
    from spectrum_vae_amd import autoencoder, losses, ops
    from spectrum_vae_amd.run_model import build_optimizer
    X = torch.from_numpy(data(32 * 4)).cuda()

    def run(disturb):
        torch.manual_seed(5)
        np.random.seed(5)
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.build((None, 32, 32, 3))
        model._materialize()
        model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
        b = model._step_buffers(32, (32, 32, 3))
        out = []
        for step in range(8):
            ops.gather_rows(X[(step % 4) * 32:(step % 4 + 1) * 32], None, b["x"])
            model.train_on_device_batch(b, True, 32.0)
            out.append(float(b["loss"].item()))
            if disturb and step in (3, 5):
                pred = model.forward_raw(X[:7], True)
                model.backward_raw(torch.ones_like(pred), -1, need_dx=False)
                model._calls.clear()
        return out, [v.numpy().copy() for v in model.trainable_variables]

    (l0, w0), (l1, w1) = run(False), run(True)
    assert l0 == l1
    for a, c in zip(w0, w1):
        assert np.array_equal(a, c)


def test_data_parallel_step_with_one_rank_equals_the_plain_step():
    """With a gradient reducer attached the step is two recorded graphs with the all-reduce issued between them.  A
    reducer over one rank (no process group) sums nothing and scales by 1, so losses and weights must equal the
    single-graph step bit for bit."""
    from spectrum_vae_amd import autoencoder, losses, ops, parallel
    from spectrum_vae_amd.run_model import build_optimizer
    X = torch.from_numpy(data(32 * 4)).cuda()

    def run(reduced):
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.build((None, 32, 32, 3))
        model._materialize()
        model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
        red = None
        if reduced:
            red = parallel.GradReducer()
            assert red.world == 1
            model.distribute(red)
        b = model._step_buffers(32, (32, 32, 3))
        out = []
        for step in range(8):
            ops.gather_rows(X[(step % 4) * 32:(step % 4 + 1) * 32], None, b["x"])
            model.train_on_device_batch(b, True, 32.0)
            out.append(float(b["loss"].item()))
        g = next(iter(model._graphs.values()))
        assert len(g) == (2 if reduced else 1)
        if reduced:
            assert red.calls == 8 and red.bytes == 8 * 4 * model._grads.numel()
        return out, [v.numpy().copy() for v in model.trainable_variables]

    (l0, w0), (l1, w1) = run(False), run(True)
    assert l0 == l1
    for a, c in zip(w0, w1):
        assert np.array_equal(a, c)


def test_fit_predict_history_and_early_stopping():
    from spectrum_vae_amd import EarlyStopping, losses
    from spectrum_vae_amd.run_model import build_optimizer, default_model, pad_height
    X = pad_height(data(200), 3)  # float64, as the reference's np.concatenate with np.zeros produces
    assert X.dtype == np.float64
    Xtr, Xte = X[:150], X[150:]
    model = default_model(X.shape[1:], verbose=False)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    h = model.fit(Xtr, Xtr, epochs=4, shuffle=True, validation_data=(Xte, Xte),
                  callbacks=[EarlyStopping(monitor="loss", patience=3)], batch_size=32, verbose=0)
    assert len(h.history["loss"]) == 4 and len(h.history["val_loss"]) == 4
    assert h.history["loss"][-1] < h.history["loss"][0]
    # 150 samples = 4 full batches + a partial one of 22 per epoch: 5 updates per epoch
    assert model.optimizer.iterations == 20
    pred = model.predict(Xte)
    assert pred.shape == Xte.shape and pred.dtype == np.float32
    # val_loss of the last epoch equals the MSE of predict() on the validation set
    assert abs(np.mean((pred - Xte) ** 2) - h.history["val_loss"][-1]) <= 1e-5
    # encoder / decoder sub-models are callable on their own and compose to the model
    z = model.encoder(Xte[:8])
    assert tuple(z.shape) == (8, 4, 4, 32)
    assert rel(model.decoder(z).numpy(), pred[:8]) <= 1e-6
    # early stopping on a loss that cannot improve
    model2 = default_model(X.shape[1:], verbose=False)
    from spectrum_vae_amd.optimizers import Adam
    model2.compile(optimizer=Adam(0.0), loss=losses.MeanSquaredError())
    h2 = model2.fit(Xtr, Xtr, epochs=10, shuffle=False, callbacks=[EarlyStopping(monitor="loss", patience=3)],
                    batch_size=50, verbose=0)
    assert len(h2.history["loss"]) == 4  # epoch 0 sets the best, epochs 1-3 do not improve


def roundoff_floor(g64, g32, k=32.0):
    """Per element, ``k`` times the largest difference between the float32 and the float64 gradient of the CPU twin
    inside the element's tensor: what float32 arithmetic alone moves a gradient of that tensor by.  Returns
    (|g64| flattened, floor flattened)."""
    mags, floors = [], []
    for a, b in zip(g64, g32):
        a = a.detach().numpy().astype(np.float64)
        b = b.detach().numpy().astype(np.float64)
        mags.append(np.abs(a).ravel())
        floors.append(np.full(a.size, k * np.abs(a - b).max()))
    return np.concatenate(mags), np.concatenate(floors)


def moments_of(model, opt):
    return opt._slots_for(model)[0].cpu().numpy().copy()


def engine_grads(model, opt, m_before):
    """The gradient the engine's optimizer launch consumed, recovered from its first moments:
    m_after = beta_1 * m_before + (1 - beta_1) * g."""
    m_after = opt._slots_for(model)[0].cpu().numpy()
    g = (m_after - opt.beta_1 * m_before) / (1.0 - opt.beta_1)
    return [g[v.offset:v.offset + v.size].reshape(v.shape) for v in model.trainable_variables]


def activation_flip(g_engine, g64, g32, tol=1e-4):
    """True when some tensor of the engine's gradient misses the float64 twin's by more than ``tol`` of the tensor's
    largest entry AND by more than 10x what the float32 twin misses it by (BatchNorm over a handful of positions makes
    some tensors ill-conditioned for every float32 evaluation; the floor of ``roundoff_floor`` covers those).
    Round-off stays two orders below ``tol`` (1e-6 measured); what reaches it is a ReLU whose input is a few ulps from
    zero taking the other branch in float32 - one unit's whole contribution appears or vanishes, in that layer's
    tensors and, diluted, in every layer upstream.  TensorFlow's float32 graph is exposed to the same flips."""
    flipped = False
    for i, (a, b, c) in enumerate(zip(g_engine, g64, g32)):
        b, c = b.detach().numpy(), c.detach().numpy()
        top = float(np.abs(b).max())
        e_hip, e_cpu = float(np.abs(a - b).max()) / top, float(np.abs(c - b).max()) / top
        if e_hip > max(tol, 10.0 * e_cpu):
            flipped = True
            if os.environ.get("SV_TEST_VERBOSE"):
                print(f"tensor {i} {tuple(a.shape)}: engine {e_hip:.1e}, float32 twin {e_cpu:.1e} of the largest entry")
    return flipped


def assert_first_updates_agree(variables, w0, ref_params, frac=0.98, significant=None):
    """Adam's first update is -lr * g / (|g| + eps'): it has the size of the learning rate whatever |g| is, so a
    weight whose true gradient is zero (a conv bias feeding a BatchNorm: the mean subtraction cancels it) moves
    by +-lr with the sign of float32 round-off noise - in TensorFlow as much as here.  Element-wise over ALL weights
    together at least ``frac`` of them must take the same step as the CPU restatement to within 5 %; ``significant``
    (``roundoff_floor`` of the twin's float64 and float32 gradients) names the weights whose gradient stands above
    that noise, and of those EVERY one must - to within 5 % of its update plus 0.5 % of the largest update in its
    tensor (the size of the learning rate): after the first step an update is -alpha * m / (sqrt(v) + eps) with m a
    difference of gradients of either sign, which can cancel to a few ulps of the float32 weight itself."""
    d_hip = np.concatenate([(v.numpy() - a).ravel() for v, a in zip(variables, w0)])
    d_ref = np.concatenate([(p.detach().numpy() - a).ravel() for p, a in zip(ref_params, w0)])
    ok = np.abs(d_hip - d_ref) <= 0.05 * np.abs(d_ref) + 1e-9
    assert ok.mean() >= frac, f"only {ok.mean():.4f} of the first updates agree"
    if significant is not None:
        mag, floor = significant
        sig = mag > floor
        assert sig.mean() >= 0.2, f"only {sig.mean():.3f} of the gradients stand above the float32 round-off floor"
        sizes = np.cumsum([0] + [int(np.prod(a.shape)) for a in w0])
        step = np.concatenate([np.full(sizes[t + 1] - sizes[t], np.abs(d_ref[sizes[t]:sizes[t + 1]]).max())
                               for t in range(len(w0))])
        bad = sig & (np.abs(d_hip - d_ref) > 0.05 * np.abs(d_ref) + 0.005 * step)
        if bad.any() and os.environ.get("SV_TEST_VERBOSE"):
            for i in np.nonzero(bad)[0][:12]:
                t = int(np.searchsorted(sizes, i, side="right") - 1)
                print("tensor", t, tuple(w0[t].shape), "g", mag[i], "floor", floor[i], "tensor max g", mag[sizes[t]:sizes[t + 1]].max(),
                      "d_hip", d_hip[i], "d_ref", d_ref[i])
        assert not bad.any(), (f"{int(bad.sum())} of {int(sig.sum())} weights with a gradient above the round-off floor "
                               f"disagree; worst |g| / floor {float((mag[bad] / floor[bad]).max()):.1f}")


def resync_twin(model, opt, ref, ref_opt):
    """Copy the engine's state (weights, BatchNorm moving statistics, Adam moments and update count) into the CPU twin.
    Trajectory tests of the GAN steps do this after every compared step: a weight whose gradient is round-off noise
    takes Adam steps of +-lr with the sign of that noise (in TensorFlow as much as here), so free-running copies drift
    apart chaotically; from a common state every single step - losses and the update of every weight - must still agree
    tightly, and the state that evolves (moving statistics, moments, schedule) is the engine's own."""
    ref.load_params([v.numpy() for v in model.trainable_variables])
    state = ref.state if not hasattr(ref, "seq") else ref.seq.state
    with torch.no_grad():
        for s, v in zip(state, model.non_trainable_variables):
            s.copy_(torch.as_tensor(v.numpy(), dtype=s.dtype).reshape(s.shape))
        m, vv = opt._slots_for(model)
        m, vv = m.cpu(), vv.cpu()
        for i, var in enumerate(model.trainable_variables):
            sl = slice(var.offset, var.offset + var.size)
            ref_opt.m[i].copy_(m[sl].view(var.shape).to(ref_opt.m[i].dtype))
            ref_opt.v[i].copy_(vv[sl].view(var.shape).to(ref_opt.v[i].dtype))
    ref_opt.it = opt.iterations


def ref_ce_step(ce, disc, opt_ce, opt_d, X, mask, l2w=0.999):
    """CPU restatement of run_context_encoder.train_step (run_context_encoder.py:225-260).  Without optimizers: the two
    gradient lists (context encoder, discriminator) instead of an update."""
    for p in ce.params + disc.params:
        p.grad = None
    pred = ce(X * (1 - mask), training=True)
    l2 = torch_ref.mse_weighted(X, pred, mask[:, :, :, 0] * 10)
    p_real = disc(X, training=True)
    p_fake = disc(pred, training=True)
    dl = (torch_ref.bce_probs(torch.ones_like(p_real), p_real) + torch_ref.bce_probs(torch.zeros_like(p_fake), p_fake)) / 2
    total = l2 * l2w + dl * (1 - l2w)
    d_grads = torch.autograd.grad(dl, disc.params, retain_graph=True)
    c_grads = torch.autograd.grad(l2, ce.params)
    ref_ce_step.last_grads = (list(c_grads), list(d_grads))
    if opt_ce is None:
        return ref_ce_step.last_grads
    opt_d.apply_gradients(d_grads)
    opt_ce.apply_gradients(c_grads)
    return float(dl), float(l2), float(total)


def test_context_encoder_train_step_trajectory():
    """8 steps of run_context_encoder.train_step against the float64 CPU twin, resynchronised after every step (see
    ``resync_twin``): every step's three losses within 1e-4, every step's weight updates element-wise, the
    discriminator's BatchNorm statistics after its pass over [real; generated]."""
    from spectrum_vae_amd import context_encoder
    from spectrum_vae_amd.run_context_encoder import batch_random_mask, build_optimizer, train_step
    rng = np.random.default_rng(7)
    X = data(64)
    mask = batch_random_mask(X, 0.25, rng).astype(np.float32)
    assert (mask[:, 4:12, 4:12, :] == 1).all() and (mask[..., 0].mean(axis=(1, 2)) >= 0.25).all()
    ce, disc = context_encoder.ContextEncoder(), context_encoder.Discriminator()
    ce.build((None, 32, 32, 3)); disc.build((None, 32, 32, 3))
    ce._materialize(); disc._materialize()
    e, d = omodels.context_encoder()
    rce = RefSequential(e + d, (32, 32, 3), dtype=torch.float64)
    rd = RefSequential(omodels.discriminator(), (32, 32, 3), dtype=torch.float64)
    load_into_ref(rce, ce); load_into_ref(rd, disc)
    # float32 copies of the twin: the difference between their gradients and the float64 ones is the round-off floor
    e32, d32_ = omodels.context_encoder()
    rce32 = RefSequential(e32 + d32_, (32, 32, 3), dtype=torch.float32)
    rd32 = RefSequential(omodels.discriminator(), (32, 32, 3), dtype=torch.float32)
    LR = 2e-4
    oce, od = build_optimizer(learning_rate=LR * 10), build_optimizer(learning_rate=LR)
    roce = KerasAdam(rce.params, LR * 10, beta1=0.5, decay_steps=50, decay_rate=0.96)
    rod = KerasAdam(rd.params, LR, beta1=0.5, decay_steps=50, decay_rate=0.96)
    Xt, Mt = torch.from_numpy(X).double(), torch.from_numpy(mask).double()
    flips = 0
    for step in range(8):
        sl = slice((step % 4) * 16, (step % 4 + 1) * 16)
        w0 = [v.numpy() for v in ce.trainable_variables + disc.trainable_variables]
        mc, md = moments_of(ce, oce), moments_of(disc, od)
        got = [float(v) for v in train_step(ce, disc, oce, od, X[sl], mask[sl])]
        g_eng = engine_grads(ce, oce, mc) + engine_grads(disc, od, md)
        nce = len(rce.params)
        rce32.load_params(w0[:nce]); rd32.load_params(w0[nce:])
        c32, d32 = ref_ce_step(rce32, rd32, None, None, Xt[sl].float(), Mt[sl].float())
        want = ref_ce_step(rce, rd, roce, rod, Xt[sl], Mt[sl])
        c64, d64 = ref_ce_step.last_grads
        for name, g, w in zip(("disc", "l2", "total"), got, want):
            assert abs(g - w) <= 1e-4 * max(1.0, abs(w)), (step, name, g, w)
        # the update of every weight tensor: same direction and size as the twin's - of every weight whose gradient
        # stands above round-off, unless a ReLU took the other branch in this step (see activation_flip)
        flipped = activation_flip(g_eng, c64 + d64, c32 + d32)
        flips += int(flipped)
        assert_first_updates_agree(ce.trainable_variables + disc.trainable_variables, w0, rce.params + rd.params,
                                   significant=None if flipped else roundoff_floor(c64 + d64, c32 + d32))
        for v, s in zip(disc.non_trainable_variables, rd.state):  # D's BatchNorm saw the real and the generated batch
            assert rel(v.numpy(), s.numpy()) <= 1e-4, (step, v.name)
        resync_twin(ce, oce, rce, roce)
        resync_twin(disc, od, rd, rod)
    assert oce.iterations == 8 and od.iterations == 8 and flips <= 1


def ref_unet_step(gen, disc, opt_g, opt_d, X, y, lam=100.0):
    """CPU restatement of run_unet.train_step (run_unet.py:164-203) with the build's generator/discriminator."""
    for p in gen.params + disc.params:
        p.grad = None
    g_out = gen(X, training=True)
    z_real = disc(torch.cat([X, y], dim=-1), training=True)
    z_fake = disc(torch.cat([X, g_out], dim=-1), training=True)
    gan = torch_ref.bce_logits(torch.ones_like(z_fake), z_fake)
    l1 = torch_ref.mae(y, g_out)
    g_total = gan + lam * l1
    d_loss = torch_ref.bce_logits(torch.ones_like(z_real), z_real) + torch_ref.bce_logits(torch.zeros_like(z_fake), z_fake)
    d_grads = torch.autograd.grad(d_loss, disc.params, retain_graph=True)
    g_grads = torch.autograd.grad(g_total, gen.params)
    ref_unet_step.last_grads = (list(g_grads), list(d_grads))
    if opt_g is None:  # the two gradient lists (generator, discriminator) instead of an update
        return ref_unet_step.last_grads
    opt_d.apply_gradients(d_grads)
    opt_g.apply_gradients(g_grads)
    return float(d_loss), float(g_total), float(gan), float(l1)


class RefUNetDisc:
    """Oracle twin of unet.Discriminator (the build's own structure; no reference definition exists)."""

    def __init__(self, dtype=torch.float32):
        k = (4, 4)
        blocks = [omodels.unet_down_block(64, k, (2, 2), apply_batchnorm=False), omodels.unet_down_block(128, k, (2, 2)),
                  omodels.unet_down_block(256, k, (2, 2))]
        head = [dict(type="conv2d", filters=1, kernel=k, strides=(1, 1), padding="valid", activation=None,
                     use_bias=True, init="normal0.02"), dict(type="flatten")]
        self.seq = RefSequential([l for b in blocks for l in b] + head, (32, 32, 4), dtype=dtype)
        self.params, self.state = self.seq.params, self.seq.state

    def __call__(self, x, training=False):
        return self.seq(x, training)


def test_unet_gan_train_step():
    """8 steps of run_unet.train_step against the float64 CPU twin, resynchronised after every step (see
    ``resync_twin``): the four losses within 1e-4 and the weight updates element-wise at every step."""
    from spectrum_vae_amd import unet
    from spectrum_vae_amd.run_unet import build_optimizer, train_step
    full = data(48)
    X, y = full[..., :1].copy(), full
    gen, disc = unet.Generator(), unet.Discriminator()
    gen.build((None, 32, 32, 1)); disc.build((None, 32, 32, 4))
    gen._materialize(); disc._materialize()
    rg, rd = omodels.RefUNet(dtype=torch.float64), RefUNetDisc(torch.float64)
    rg32, rd32 = omodels.RefUNet(dtype=torch.float32), RefUNetDisc(torch.float32)  # their gradients measure the round-off floor
    rg.load_params([v.numpy() for v in gen.trainable_variables])
    rd.seq.load_params([v.numpy() for v in disc.trainable_variables])
    og, od = build_optimizer(learning_rate=2e-4), build_optimizer(learning_rate=2e-4)
    rog = KerasAdam(rg.params, 2e-4, beta1=0.5, decay_steps=10, decay_rate=0.96)
    rod = KerasAdam(rd.params, 2e-4, beta1=0.5, decay_steps=10, decay_rate=0.96)
    Xt, yt = torch.from_numpy(X).double(), torch.from_numpy(y).double()
    flips = 0
    for step in range(8):
        sl = slice((step % 3) * 16, (step % 3 + 1) * 16)
        w0 = [v.numpy() for v in gen.trainable_variables + disc.trainable_variables]
        mg, md = moments_of(gen, og), moments_of(disc, od)
        got = [float(v) for v in train_step(gen, disc, og, od, X[sl], y[sl])]
        g_eng = engine_grads(gen, og, mg) + engine_grads(disc, od, md)
        ng = len(rg.params)
        rg32.load_params(w0[:ng]); rd32.seq.load_params(w0[ng:])
        g32, d32 = ref_unet_step(rg32, rd32, None, None, Xt[sl].float(), yt[sl].float())
        want = ref_unet_step(rg, rd, rog, rod, Xt[sl], yt[sl])
        g64, d64 = ref_unet_step.last_grads
        for name, g, w in zip(("disc", "gen_total", "gan", "l1"), got, want):
            assert abs(g - w) <= 1e-4 * max(1.0, abs(w)), (step, name, g, w)
        flipped = activation_flip(g_eng, g64 + d64, g32 + d32)
        flips += int(flipped)
        assert_first_updates_agree(gen.trainable_variables + disc.trainable_variables, w0, rg.params + rd.params,
                                   significant=None if flipped else roundoff_floor(g64 + d64, g32 + d32))
        resync_twin(gen, og, rg, rog)
        resync_twin(disc, od, rd.seq, rod)
    assert flips <= 1


def test_keras_surface_apply_gradients_and_losses():
    """optimizer.apply_gradients(zip(grads, model.trainable_variables)) and the callable loss objects."""
    from spectrum_vae_amd import autoencoder, losses
    from spectrum_vae_amd.losses import to_device_f32
    from spectrum_vae_amd.optimizers import Adam
    x = data(8)
    m = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    pred = m(x, training=True)
    assert tuple(pred.shape) == (8, 32, 32, 3) and isinstance(pred.numpy(), np.ndarray)
    loss = losses.MeanSquaredError()(x, pred)
    assert abs(float(loss.numpy()) - float(np.mean((pred.numpy() - x) ** 2))) < 1e-6
    dpred = to_device_f32(2 * (pred.numpy() - x) / x.size)
    before = [v.numpy() for v in m.trainable_variables]
    m.backward_raw(dpred, -1, need_dx=False)
    opt = Adam(1e-3)
    opt.apply_gradients(zip([v.grad for v in m.trainable_variables], m.trainable_variables))
    assert opt.iterations == 1
    after = [v.numpy() for v in m.trainable_variables]
    assert all(np.abs(a - b).max() > 0 for a, b in zip(after, before))
    # per-variable path: a subset of the variables with free-standing gradient tensors
    sub = m.trainable_variables[:2]
    opt2 = Adam(1e-2)
    opt2.apply_gradients([(torch.ones(v.shape, device="cuda"), v) for v in sub])
    for v, a in zip(sub, after[:2]):
        assert np.allclose(v.numpy(), a - 1e-2 * 1.0 / (1.0 + 1e-7 / np.sqrt(1 - 0.999)), atol=1e-6)
    per_pixel = losses.mse(pred, x)
    assert tuple(per_pixel.shape) == (8, 32, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 64, 1100])
def test_whole_network_forward_matches_per_layer_path_and_oracle(n):
    """The one-launch forward fills every per-layer buffer (activations, argmax) exactly like the per-layer path."""
    from spectrum_vae_amd import autoencoder
    from spectrum_vae_amd.losses import to_device_f32
    x = data(n, seed=7 + n)
    outs = {}
    for fused in (True, False):
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.use_net_fused = fused
        model.build((None, 32, 32, 3))
        model._materialize()
        chain = model._conv_chain()
        assert chain is not None and len(chain) == 6
        model.forward_raw(to_device_f32(x), True)
        ctx = model._calls[-1]
        outs[fused] = [(ctx.saved[id(l)]["y"].cpu().numpy().copy(),
                        ctx.saved[id(l)]["idx"].cpu().numpy().copy() if "idx" in ctx.saved[id(l)] else None) for l in chain]
        if fused:
            assert model._chain_cache and all(v is not False for v in model._chain_cache.values())
            pred_infer = model.predict(x[: min(n, 40)], batch_size=16)
            enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
            ref = RefSequential(enc + dec, (32, 32, 3))
            load_into_ref(ref, model)
            want = ref(torch.from_numpy(x[: min(n, 40)]), training=False).detach().numpy()
            assert rel(pred_infer, want) < 1e-4
    for (yf, idf), (yu, idu) in zip(outs[True], outs[False]):
        assert rel(yf, yu) < 1e-5
        if idf is not None:
            assert (idf != idu).mean() < 1e-4  # a differing slot needs a tie within rounding


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 5, 64, 1100, 2100])
def test_forward_kernel_built_for_the_default_layer_table_equals_the_run_time_table_kernel(n, sv_opts):
    """The run_model.py default conv stack has a forward kernel compiled for its layer table (extents and LDS layout as
    constants); the option netfwd_generic selects the kernel that reads the table at run time.  Same arithmetic in the same
    order: every layer output, the argmax bytes, the loss gradient and the loss partial sums agree bit for bit.  2100
    samples are more passes than the loss workspace has slots, so that launch takes the run-time-table kernel by itself."""
    from spectrum_vae_amd import autoencoder, ops
    from spectrum_vae_amd.losses import to_device_f32
    x = to_device_f32(data(n, seed=51 + n))
    got = {}
    for generic in (False, True):
        sv_opts("netfwd_generic", 1 if generic else -1)
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.build((None, 32, 32, 3))
        model._materialize()
        torch.manual_seed(3)
        model._params.copy_(torch.randn_like(model._params) * 0.2)
        part = torch.zeros(ops.loss_workspace_floats(), device="cuda")
        dpred = torch.empty(n, 32, 32, 3, device="cuda")
        pred = model.forward_raw(x, True, None, (x, dpred, part))
        ctx = model._calls[-1]
        outs = [ctx.saved[id(l)]["y"].clone() for l in model._conv_chain()]
        idxs = [ctx.saved[id(l)]["idx"].clone() for l in model._conv_chain() if "idx" in ctx.saved[id(l)]]
        got[generic] = (pred.clone(), dpred, float(part.sum()), outs, idxs)
    a, b = got[False], got[True]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert abs(a[2] - b[2]) <= 1e-6 * abs(b[2])  # the same squared errors, summed over differently many workgroups
    assert all(torch.equal(u, v) for u, v in zip(a[3], b[3])) and all(torch.equal(u, v) for u, v in zip(a[4], b[4]))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 64, 1100])
def test_forward_launch_with_loss_matches_the_separate_loss(n):
    """MSE evaluated while the one-launch forward writes its output: gradient and value against the loss kernel, the value
    folded once by sv_fold_scalar and once by the optimizer launch (which must also still apply its update)."""
    from spectrum_vae_amd import autoencoder, losses, ops
    from spectrum_vae_amd.losses import to_device_f32
    from spectrum_vae_amd.optimizers import Adam
    x = data(n, seed=23 + n)
    model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    model.build((None, 32, 32, 3))
    model._materialize()
    xt = to_device_f32(x)
    pred = model.forward_raw(xt, True).clone()
    want_loss, want_d = torch.zeros(1, device="cuda"), torch.empty_like(pred)
    losses.MeanSquaredError().value_and_grad(xt, pred, want_loss, want_d, 1.0)
    model._calls.clear()
    part = torch.full((ops.loss_workspace_floats(),), float("nan"), device="cuda")
    got_d = torch.empty_like(pred)
    pred2 = model.forward_raw(xt, True, None, (xt, got_d, part))
    fold = model._calls[-1].loss_fold
    assert fold is not None and fold[1] >= 1 and torch.equal(pred2, pred)
    assert rel(got_d.cpu().numpy(), want_d.cpu().numpy()) < 1e-6
    got = torch.zeros(1, device="cuda")
    ops.fold_scalar(part, fold[1], fold[2], got)
    assert abs(float(got) - float(want_loss)) <= 2e-6 * abs(float(want_loss))
    # the same fold riding on the optimizer launch, with a running sum
    out2, run = torch.zeros(1, device="cuda"), torch.full((1,), 3.0, device="cuda")
    opt = Adam(1e-3)
    model._grads.normal_()
    before = model._params.clone()
    opt.apply_model(model, 1.0, zero_grads=True, fold=(part, fold[1], fold[2], out2, run, 0.5))
    assert float(out2) == float(got) and abs(float(run) - (3.0 + 0.5 * float(got))) < 1e-6
    assert opt.iterations == 1 and not torch.equal(before, model._params) and not model._grads.any()


@pytest.mark.gpu
def test_fit_with_loss_inside_the_forward_launch_equals_the_separate_loss():
    from spectrum_vae_amd import autoencoder, losses
    from spectrum_vae_amd.optimizers import Adam
    x = data(96, seed=31)
    hist = {}
    for fuse in (True, False):
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.fuse_loss = fuse
        model.build((None, 32, 32, 3))
        model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
        torch.manual_seed(5)
        h = model.fit(x, x, epochs=3, batch_size=32, shuffle=False, verbose=0)
        hist[fuse] = (np.array(h.history["loss"]), model._params.cpu().numpy().copy())
    assert np.allclose(hist[True][0], hist[False][0], rtol=1e-5)
    assert np.abs(hist[True][1] - hist[False][1]).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 7, 130])
def test_matrix_core_bodies_agree_with_the_vector_bodies(n, sv_opts):
    """The 16/32-channel layers run on MFMA tiles in the fused forward, the input-gradient and the kernel-gradient
    kernels; the *_valu options select the vector bodies at plan time.  Both must give the same prediction
    and gradients to rounding, also for batches that leave a pass or a wave partly empty."""
    from spectrum_vae_amd import autoencoder, losses
    from spectrum_vae_amd.losses import to_device_f32
    x = data(n, seed=23 + n)
    res = {}
    for mode in ("mfma", "valu"):
        for name in ("netfwd_valu", "dgrad_valu", "wgrad_valu"):
            sv_opts(name, 1 if mode == "valu" else -1)
        model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
        model.build((None, 32, 32, 3))
        model._materialize()
        model.compile(optimizer=None, loss=losses.MeanSquaredError())
        xd = to_device_f32(x)
        model.zero_grad()
        pred = model.forward_raw(xd, True)
        loss = torch.zeros(1, device=xd.device)
        dpred = torch.empty_like(pred)
        model.loss.value_and_grad(xd, pred, loss, dpred, 1.0)
        model.backward_raw(dpred, -1, need_dx=False)
        res[mode] = (pred.cpu().numpy().copy(), model._grads.cpu().numpy().copy())
    assert rel(res["mfma"][0], res["valu"][0]) < 1e-5
    assert rel(res["mfma"][1], res["valu"][1]) < 2e-5
    assert not np.array_equal(res["mfma"][1], res["valu"][1])  # the switches do select different bodies


def test_bench_shape_step_matches_oracle():
    """The headline configuration itself: run_model.py defaults at batch 1024 through ``train_on_device_batch``'s
    replayed graphs (fused forward, matrix-core bodies, 128 / 256 kernel-gradient workgroups per layer).  Loss and
    every gradient within 1e-4 of the CPU restatement, then 20 optimizer steps with the loss trajectory within 1e-3."""
    from spectrum_vae_amd import autoencoder, losses, ops, parallel
    from spectrum_vae_amd.run_model import build_optimizer
    n = 1024
    X = data(2 * n, seed=41)
    model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    model.build((None, 32, 32, 3))
    model._materialize()
    assert model.wgrad_batch_blocks_large == 128 and model.wgrad_batch_blocks == 128
    enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
    ref = RefSequential(enc + dec, (32, 32, 3))
    load_into_ref(ref, model)
    w_start = [v.numpy() for v in model.trainable_variables]

    class Snoop(parallel.GradReducer):
        """A one-rank reducer that keeps a copy of what it is handed: the gradient buffer exactly as the replayed
        forward/backward graph left it (the optimizer graph that follows clears it)."""
        snaps = []

        def all_reduce(self, flat):
            self.snaps.append(flat.clone())
            return super().all_reduce(flat)

    # ---- (1) gradients out of the replayed step.  lr = 0 keeps the weights where the oracle's are.
    from spectrum_vae_amd.optimizers import Adam
    model.compile(optimizer=Adam(0.0), loss=losses.MeanSquaredError())
    red = Snoop()
    model.distribute(red)
    Xd = torch.from_numpy(X).cuda()
    b = model._step_buffers(n, (32, 32, 3))
    ops.gather_rows(Xd[:n], None, b["x"])
    for _ in range(4):  # eager, capture, replay, replay
        model.train_on_device_batch(b, True, float(n))
    g = next(iter(model._graphs.values()))
    assert not isinstance(g, str) and len(g) == 2
    flat = red.snaps[-1].cpu().numpy()
    hip_loss = float(b["loss"].item())
    _, loss, g32 = _ref_pass(ref, X[:n], X[:n], torch.float32)
    assert abs(hip_loss - loss) <= 1e-4 * abs(loss)
    for v, want in zip(model.trainable_variables, g32):
        got = flat[v.offset:v.offset + v.size].reshape(v.shape)
        assert rel(got, want, floor=1e-6 + 1e-3 * float(np.abs(want).max())) <= 1e-4, v.name
    for v, w0 in zip(model.trainable_variables, w_start):
        assert np.array_equal(v.numpy(), w0)  # lr = 0

    # ---- (2) trajectory through the single-graph step
    model.distribute(None)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    opt_ref = KerasAdam(ref.params, 5e-4, decay_steps=10, decay_rate=0.96, staircase=True)
    hip, cpu = [], []
    for step in range(20):
        sl = slice((step % 2) * n, (step % 2 + 1) * n)
        ops.gather_rows(Xd[sl], None, b["x"])
        model.train_on_device_batch(b, True, float(n))
        hip.append(float(b["loss"].item()))
        xb = torch.from_numpy(X[sl])
        cpu.append(torch_ref.train_step_mse(ref, opt_ref, xb, xb))
    hip, cpu = np.array(hip), np.array(cpu)
    g = next(iter(model._graphs.values()))
    assert not isinstance(g, str) and len(g) == 1, "the trajectory steps were not replayed from one graph"
    assert np.max(np.abs(hip - cpu) / np.abs(cpu)) <= 1e-3, (hip, cpu)
    for v, p in zip(model.trainable_variables, ref.params):
        assert rel(v.numpy(), p.detach().numpy()) <= 1e-3, v.name


def test_recorded_step_survives_a_larger_scratch_request():
    """A recorded step holds the raw address of the pool's scratch buffer.  A later call that needs more scratch (a
    training-mode forward at a much larger batch) replaces the buffer; the old one must stay alive, or replays would
    write partial sums into memory the allocator has handed to someone else."""
    from spectrum_vae_amd import autoencoder, losses, ops
    from spectrum_vae_amd.losses import to_device_f32
    from spectrum_vae_amd.run_model import build_optimizer
    cfg = dict(RUN_MODEL_DEFAULTS, batch_norm=True)  # BatchNorm reductions go through the shared scratch
    X = data(8 * 4)
    big = to_device_f32(data(600, seed=9))

    def run(disturb):
        model = autoencoder.CAEPadded(**cfg)
        model.build((None, 32, 32, 3))
        model._materialize()
        model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
        Xd = torch.from_numpy(X).cuda()
        b = model._step_buffers(8, (32, 32, 3))
        out, hog = [], []
        for step in range(8):
            ops.gather_rows(Xd[(step % 4) * 8:(step % 4 + 1) * 8], None, b["x"])
            model.train_on_device_batch(b, True, 8.0)
            out.append(float(b["loss"].item()))
            if disturb and step == 3:
                ws0 = model._pool._ws
                state = model._state.clone()
                model.forward_raw(big, True)      # asks for far more scratch than the batch-8 step did
                model._calls.clear()
                model._state.copy_(state)         # undo the moving-statistics update of the extra call
                assert model._pool._ws is not ws0 and any(t is ws0 for t in model._pool._retired)
                # whatever the allocator hands out next must not be the retired scratch
                hog = [torch.full((ws0.numel() // 4,), 7.0, device="cuda") for _ in range(4)]
        for h in hog:
            assert bool((h == 7.0).all())
        return out, [v.numpy().copy() for v in model.trainable_variables]

    (l0, w0), (l1, w1) = run(False), run(True)
    assert l0 == l1
    for a, c in zip(w0, w1):
        assert np.array_equal(a, c)


@pytest.mark.gpu
def test_context_encoder_driver_runs_on_device_masks():
    """run_context_encoder.main with the training set and its masks resident in HBM (masks generated there)."""
    from spectrum_vae_amd import run_context_encoder
    res, per_sample = run_context_encoder.main(n_train=320, epochs=2, batch_size=64)
    assert len(res) == 2 and per_sample.shape == (64,)
    assert all(np.isfinite(list(r.values())).all() for r in res)
    assert res[1]["recon"] < res[0]["recon"] * 1.5
