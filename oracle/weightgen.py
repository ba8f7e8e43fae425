"""Deterministic synthetic checkpoints and inputs (TEST INFRASTRUCTURE ONLY).

The reference's trained checkpoints are GitHub release assets and are not
available offline, so parity is pinned on seeded random weights (SURVEY.md
§8c).  To keep fixtures small the weights are never stored: both
``tools/gen_golden.py`` (which loads them into the imported reference) and the
tests (which feed them to the oracle / the HIP path) regenerate the same
training-form state_dict from a seed with numpy's PCG64, which is
bit-reproducible across platforms.

BatchNorm statistics are randomised so that every fold of
models/fastenhancer/default/model.py:532-608 is exercised, and the final conv's
``scale`` is set so that the mask is O(1) (with the default init the enhanced
waveform is ~1e-2 of the input and an absolute tolerance would be vacuous).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .fe_oracle import FEConfig, linear_filterbank, linear_filterbank_tk, positional_embedding, training_state_dict_spec


def make_training_state_dict(cfg: FEConfig, seed: int) -> Dict[str, np.ndarray]:
    """Training-form state_dict with seeded values."""
    rng = np.random.Generator(np.random.PCG64(seed))
    spec = training_state_dict_spec(cfg)
    pre, post = (linear_filterbank_tk if cfg.time_kernel or cfg.dprnn or cfg.dpt or cfg.ln or cfg.noncausal else linear_filterbank)(cfg.F1, cfg.rf_freq)
    pe = positional_embedding(cfg.rf_channels, cfg.rf_freq)
    sd: Dict[str, np.ndarray] = {}
    for key, shape in spec.items():
        if key.endswith("num_batches_tracked"):
            sd[key] = np.asarray(100, dtype=np.int64)
            continue
        leaf = key.split(".")[-1]
        if leaf == "running_var":
            v = rng.uniform(0.75, 1.25, shape)
        elif leaf == "running_mean":
            v = 0.1 * rng.standard_normal(shape)
        elif key == "rf_pre.0.weight":
            v = pre + 0.01 * rng.standard_normal(shape)
        elif key == "rf_post.0.weight":
            v = post + 0.01 * rng.standard_normal(shape)
        elif key == "time_pe":                          # the dptransformer variant's positional bias [NH, L+1]
            v = positional_embedding(cfg.rf_heads, cfg.lookbehind + 1).T + 0.05 * rng.standard_normal(shape)
        elif leaf == "pe":
            v = pe + 0.05 * rng.standard_normal(shape)
        elif leaf == "original0":                       # weight-norm gain g
            v = rng.uniform(0.6, 1.1, shape)
        elif leaf == "scale":
            v = np.full(shape, 2.0 * (cfg.channels / 24.0) ** 0.6)  # keeps the enhanced RMS ~ the input RMS
            if cfg.ln:                                              # (normalised activations; the output goes with mask^(1/0.3))
                v = v / 2.7
        elif "bias" in leaf:                            # BN beta, GRU biases, final conv bias
            v = 0.1 * rng.standard_normal(shape)
        elif len(shape) == 1:                           # BN gamma
            v = rng.uniform(0.75, 1.25, shape)
        else:                                           # conv / linear / GRU weights
            fan_in = int(np.prod(shape[1:]))
            if key == "dec_post.3.weight":
                fan_in = 1
            v = rng.standard_normal(shape) * (1.2 / np.sqrt(fan_in))
        sd[key] = np.asarray(v, dtype=np.float32)
    return sd


def make_input(B: int, n_samples: int, seed: int, sr: int = 16000) -> np.ndarray:
    """Synthetic noisy input of SURVEY.md §8(d): 0.1*N(0,1) + 0.3*sin(2*pi*f_b*t),
    f_b = 100 + 13*b Hz, clipped to [-1,1], float32 [B, n_samples]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n_samples, dtype=np.float64) / sr
    f = 100.0 + 13.0 * np.arange(B, dtype=np.float64)
    x = 0.1 * rng.standard_normal((B, n_samples)) + 0.3 * np.sin(2 * np.pi * f[:, None] * t[None, :])
    return np.clip(x, -1.0, 1.0).astype(np.float32)
