"""Counterpart of the reference's ``run_model.py``: trains ``autoencoder.CAEPadded`` with MSE and the
Keras-form Adam on the HIP engine.  Function names, arguments and defaults follow run_model.py:22-89;
the wandb sweep (run_model.py:92-227) becomes a local random search over the same grid with JSONL logging,
and the matplotlib/seaborn plots are dropped.
"""
from __future__ import annotations

import argparse
import json
from typing import Optional, Tuple

import numpy as np

from . import autoencoder, global_vars, losses
from .data import sort_min_diff, synth_training_tensor
from .model import EarlyStopping
from .optimizers import Adam, ExponentialDecay

# run_model.py:164-219
SWEEP_PARAMETERS = {
    "optimizer": {"value": "adam"}, "latent_dimensions": {"value": 0},
    "initial_filters": {"values": [8, 16]}, "conv_layers": {"values": [3, 4]},
    "conv_layer_multiplier": {"values": [1, 2]}, "conv_operations": {"values": [1, 2, 3]},
    "fc_layers": {"value": 0}, "fc_layer_size": {"value": 128}, "dropout": {"value": 0.1},
    "activation": {"values": ["relu", "elu"]}, "learning_rate": {"value": 1e-4},
    "decay_rate": {"value": 0.92}, "decay_steps": {"value": 10}, "batch_size": {"values": [16, 32]},
    "kernel_size": {"values": [3, 5]}, "epochs": {"value": 10}, "sort_data": {"value": True},
    "batch_norm": {"values": [True, False]},
}


def sort_batch(batch: np.ndarray) -> np.ndarray:
    """run_model.py:22-29.  The reference hands the whole 3-D sample to ``util.sort_min_diff``, which sklearn's
    NearestNeighbors rejects (2-D only); the similarity is taken on the haplotype channel (channel 0), as
    ``run_context_encoder.batch_sort`` (run_context_encoder.py:101-112) does."""
    sorted_batch = np.zeros(batch.shape)
    for i in range(batch.shape[0]):
        arr = batch[i]
        sorted_batch[i] = arr[sort_min_diff(arr[:, :, 0]), :, :]
    return sorted_batch


def build_model(input_shape: Tuple[int] = (global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS),
                initial_filters: int = 8, conv_layers: int = 2, conv_layer_multiplier: int = 2,
                activation: str = "elu", kernel_size: int = 5, conv_operations: int = 2, latent_dimensions: int = 8,
                fc_layers: int = 2, fc_layer_size: int = 64, dropout: float = 0.5, batch_norm: bool = False,
                verbose: bool = True):
    """run_model.py:32-74."""
    model = autoencoder.CAEPadded(input_shape=input_shape, initial_filters=initial_filters, conv_layers=conv_layers,
                                  conv_layer_multiplier=conv_layer_multiplier, activation=activation,
                                  kernel_size=kernel_size, conv_operations=conv_operations,
                                  fc_layer_size=fc_layer_size, fc_layers=fc_layers,
                                  latent_dimensions=latent_dimensions, dropout=dropout, batch_norm=batch_norm)
    model.build_graph((1, input_shape[0], input_shape[1], input_shape[2]))
    if verbose:
        model.encoder.summary()
        model.decoder.summary()
    return model


def build_optimizer(learning_rate: float = 1e-4, decay_rate: float = 0.96, decay_steps: int = 10) -> Adam:
    """run_model.py:77-89: ExponentialDecay(staircase) + legacy Adam."""
    return Adam(learning_rate=ExponentialDecay(learning_rate, decay_steps=decay_steps, decay_rate=decay_rate,
                                               staircase=True))


def pad_height(train: np.ndarray, conv_layers: int) -> np.ndarray:
    """run_model.py:104-111,239-246: zero rows until the height divides 2**conv_layers (float64 zeros, so the
    concatenated training array is float64, as in the reference)."""
    h = train.shape[1]
    padding = 0
    while (h + padding) % (2 ** conv_layers) != 0:
        padding += 1
    zeros = np.zeros((train.shape[0], padding, train.shape[2], train.shape[3]))
    return np.concatenate((train, zeros), axis=1)


def default_model(input_shape, verbose=True):
    """The non-sweep configuration of run_model.py:263-276."""
    return build_model(input_shape=input_shape, conv_layers=3, conv_layer_multiplier=2, initial_filters=8,
                       activation="elu", kernel_size=3, conv_operations=1, fc_layers=0, fc_layer_size=256,
                       dropout=0.2, latent_dimensions=0, batch_norm=False, verbose=verbose)


def train_model(config: dict, train: np.ndarray, seed: int = global_vars.DEFAULT_SEED, verbose: int = 0):
    """Body of one sweep trial (run_model.py:92-152) for a sampled ``config``."""
    rng = np.random.default_rng(seed)
    if config.get("sort_data"):
        train = sort_batch(train)
    data = pad_height(train, config["conv_layers"])
    perm = rng.permutation(data.shape[0])
    split = int(round(0.8 * data.shape[0]))
    X_train, X_test = data[perm[:split]], data[perm[split:]]
    model = build_model(input_shape=X_train.shape[1:], conv_layers=config["conv_layers"],
                        conv_layer_multiplier=config["conv_layer_multiplier"],
                        initial_filters=config["initial_filters"], activation=config["activation"],
                        kernel_size=config["kernel_size"], conv_operations=config["conv_operations"],
                        latent_dimensions=0, fc_layers=0, fc_layer_size=0, dropout=0,
                        batch_norm=config["batch_norm"], verbose=False)
    model.compile(optimizer=build_optimizer(config["learning_rate"], config["decay_rate"], config["decay_steps"]),
                  loss=losses.MeanSquaredError())
    history = model.fit(X_train, X_train, epochs=config["epochs"], shuffle=True, validation_data=(X_test, X_test),
                        batch_size=config["batch_size"], verbose=verbose, seed=seed)
    return model, history


def sample_config(rng: np.random.Generator) -> dict:
    """One draw of the ``method: random`` sweep."""
    return {k: (v["value"] if "value" in v else v["values"][int(rng.integers(len(v["values"])))])
            for k, v in SWEEP_PARAMETERS.items()}


def reconstruction_scores(model, data: np.ndarray, batch_size: int = 32) -> np.ndarray:
    """run_model.py:356-362: per-sample reconstruction error = mean over (H, W) of the per-pixel MSE."""
    recon = model.predict(data, batch_size=batch_size)
    per_pixel = losses.mse(recon, data)
    return per_pixel.numpy().mean(axis=(1, 2))


def roc_auc(neg_scores: np.ndarray, pos_scores: np.ndarray) -> float:
    """AUC of run_model.py:390-398 (labels 1 for the normal set, 0 for the mutator set, score = loss),
    computed by the rank-sum identity."""
    scores = np.concatenate((neg_scores, pos_scores))
    labels = np.concatenate((np.ones(len(neg_scores)), np.zeros(len(pos_scores))))
    order = np.argsort(scores, kind="mergesort")
    ranks = np.empty(len(scores))
    ranks[order] = np.arange(1, len(scores) + 1)
    # average ranks over ties
    s_sorted = scores[order]
    i = 0
    while i < len(scores):
        j = i
        while j + 1 < len(scores) and s_sorted[j + 1] == s_sorted[i]:
            j += 1
        ranks[order[i:j + 1]] = (i + j + 2) / 2.0
        i = j + 1
    n1, n0 = labels.sum(), (1 - labels).sum()
    return float((ranks[labels == 1].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))


def reconstruction_scores_device(model, data, batch_size: int = 32):
    """``reconstruction_scores`` with the forward pass, the squared error and the per-sample mean all on the device;
    returns a float32 device tensor of one score per sample."""
    return model.reconstruction_scores(data, batch_size=batch_size)


def roc_auc_device(neg_scores, pos_scores) -> float:
    """``roc_auc`` for score tensors that live on the device: the pair counts (first set's score above / equal to the
    second set's) are integers formed on the device, only the two counts come back."""
    import torch

    from . import ops
    counts = torch.empty(2, dtype=torch.int64, device=neg_scores.device)
    ops.roc_auc_counts(neg_scores.contiguous(), pos_scores.contiguous(), counts)
    gt, eq = (int(v) for v in counts.tolist())
    return (gt + 0.5 * eq) / (neg_scores.numel() * pos_scores.numel())


def evaluate_labeled(model, normal_data: np.ndarray, mutator_data: np.ndarray, conv_layers: int = 3,
                     batch_size: int = 32) -> dict:
    """The labelled evaluation at the end of run_model.py:349-398 (arrays as ``data/labeled.npz`` holds them under "neg" /
    "pos"): pad the height, score both sets by reconstruction error, AUC with the normal set as the positive class."""
    normal = reconstruction_scores_device(model, pad_height(normal_data, conv_layers), batch_size)
    mutator = reconstruction_scores_device(model, pad_height(mutator_data, conv_layers), batch_size)
    return {"auc": roc_auc_device(normal, mutator), "normal_loss": normal.cpu().numpy(),
            "mutator_loss": mutator.cpu().numpy()}


def main(args, n_train: int = 4096, log_path: Optional[str] = None):
    rng = np.random.default_rng(global_vars.DEFAULT_SEED)
    train = synth_training_tensor(n_train, global_vars.DEFAULT_SEED)
    if args.run_sweep:
        results = []
        for trial in range(args.search_size):
            cfg = sample_config(rng)
            _, hist = train_model(cfg, train, seed=global_vars.DEFAULT_SEED + trial)
            rec = {"trial": trial, "config": cfg, "validation_loss": hist.history["loss"]}
            results.append(rec)
            if log_path:
                with open(log_path, "a") as f:
                    f.write(json.dumps(rec) + "\n")
            print(json.dumps({"trial": trial, "final_loss": hist.history["loss"][-1]}))
        return results
    CONV_LAYERS, BATCH_SIZE = 3, 32
    data = pad_height(sort_batch(train), CONV_LAYERS)
    perm = rng.permutation(data.shape[0])
    split = int(round(0.8 * data.shape[0]))
    X_train, X_test = data[perm[:split]], data[perm[split:]]
    model = default_model(data.shape[1:])
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    history = model.fit(X_train, X_train, epochs=20, shuffle=True, validation_data=(X_test, X_test),
                        callbacks=[EarlyStopping(monitor="loss", patience=3)], batch_size=BATCH_SIZE)
    decoded = model.predict(X_test)
    return model, history, decoded


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("-run_sweep", action="store_true")
    p.add_argument("-search_size", type=int, default=20)
    main(p.parse_args())
