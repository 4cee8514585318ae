"""Evaluation after training on the device: per-sample reconstruction error and ROC AUC (run_model.py:349-398), through
the C ABI, against NumPy float64 and scikit-learn's roc_curve + auc (the functions the reference calls)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 32, 32, 3), (5, 32, 32, 3), (33, 7, 5, 3), (1100, 32, 32, 3), (3, 1, 1, 1)])
def test_sample_mse_matches_numpy(shape):
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(shape[0])
    t = rng.standard_normal(shape).astype(np.float32)
    p = rng.standard_normal(shape).astype(np.float32)
    out = torch.empty(shape[0], device="cuda")
    ops.sample_mse(torch.from_numpy(t).cuda(), torch.from_numpy(p).cuda(), out)
    # losses.mse = mean over channels, then reduce_mean over (H, W), in float64
    ref = ((p.astype(np.float64) - t) ** 2).mean(axis=-1).mean(axis=(1, 2))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=0)
    pb = torch.from_numpy(p).cuda().to(torch.bfloat16)
    ops.sample_mse(torch.from_numpy(t).cuda(), pb, out)
    refb = ((pb.float().cpu().numpy().astype(np.float64) - t) ** 2).mean(axis=(1, 2, 3))
    np.testing.assert_allclose(out.cpu().numpy(), refb, rtol=2e-6, atol=0)


@pytest.mark.parametrize("n1,n0,levels", [(1, 1, 2), (7, 5, 4), (300, 1500, 50), (2049, 1025, 0), (64, 5000, 7)])
def test_auc_counts_are_exact_and_equal_sklearn(n1, n0, levels):
    sklearn_metrics = pytest.importorskip("sklearn.metrics")
    from spectrum_vae_amd import ops, run_model
    rng = np.random.default_rng(n1 * 7 + n0)
    draw = (lambda n: rng.random(n).astype(np.float32)) if levels == 0 else \
        (lambda n: (rng.integers(0, levels, n) / levels).astype(np.float32))
    a, b = draw(n1), draw(n0)
    counts = torch.full((2,), -1, dtype=torch.int64, device="cuda")
    ops.roc_auc_counts(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), counts)
    gt = int((a[:, None] > b[None, :]).sum())
    eq = int((a[:, None] == b[None, :]).sum())
    assert counts.tolist() == [gt, eq]
    auc = run_model.roc_auc_device(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    fpr, tpr, _ = sklearn_metrics.roc_curve(np.concatenate((np.ones(n1), np.zeros(n0))), np.concatenate((a, b)))
    assert abs(auc - sklearn_metrics.auc(fpr, tpr)) < 1e-12
    assert abs(auc - run_model.roc_auc(a, b)) < 1e-12


def test_labeled_evaluation_on_device_equals_the_host_path():
    """model.predict -> losses.mse -> mean over (1, 2) -> AUC on the host against the device-resident path."""
    from spectrum_vae_amd import run_model
    from spectrum_vae_amd.data import synth_training_tensor
    model = run_model.default_model((32, 32, 3), verbose=False)
    normal = synth_training_tensor(70, 5)[:, :30]       # 30 rows: pad_height adds two rows of zeros
    mutator = synth_training_tensor(45, 6)[:, :30] * 0.5
    res = run_model.evaluate_labeled(model, normal, mutator, conv_layers=3, batch_size=32)
    host_n = run_model.reconstruction_scores(model, run_model.pad_height(normal, 3), batch_size=32)
    host_m = run_model.reconstruction_scores(model, run_model.pad_height(mutator, 3), batch_size=32)
    np.testing.assert_allclose(res["normal_loss"], host_n, rtol=1e-5)
    np.testing.assert_allclose(res["mutator_loss"], host_m, rtol=1e-5)
    assert abs(res["auc"] - run_model.roc_auc(res["normal_loss"], res["mutator_loss"])) < 1e-12
    assert 0.0 <= res["auc"] <= 1.0
