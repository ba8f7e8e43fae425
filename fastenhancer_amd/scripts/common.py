"""Shared host logic of the command-line callers: config / checkpoint lookup as the reference does it
(`logs/<name>/config.yaml`, latest `[0-9]{5,}.pth`: utils/hparams.py:88-150, wrappers/ns.py:288-314) and WAV I/O
(scipy; librosa / soundfile are not required — the sampling rate must already match, no resampling)."""
from __future__ import annotations

import importlib
import os
import re
from typing import Optional, Tuple

import numpy as np
import torch
import yaml
from scipy.io import wavfile


def load_hparams(config: Optional[str] = None, name: Optional[str] = None, log_root: str = "logs") -> dict:
    """-c <yaml> or -n <name> (-> logs/<name>/config.yaml), plain dict (only model / model_kwargs /
    data.sampling_rate are used)."""
    if config is None:
        if name is None:
            raise ValueError("Either --name or --config should be given.")
        config = os.path.join(log_root, name, "config.yaml")
    with open(config) as f:
        return yaml.safe_load(f)


def latest_checkpoint(base_dir: str) -> Optional[str]:
    """wrappers/ns.py:288-300: newest file matching [0-9]{5,}.pth in logs/<name>."""
    if not os.path.isdir(base_dir):
        return None
    files = [int(f[:-4]) for f in os.listdir(base_dir) if re.fullmatch(r"[0-9]{5,}\.pth", f)]
    if not files:
        return None
    return os.path.join(base_dir, f"{max(files):0>5d}.pth")


def build_model(hps: dict, device: str, offline: bool, checkpoint: Optional[str] = None):
    """importlib.import_module(f"models.{hps.model}.model") of wrappers/ns.py:29-32, on the HIP path."""
    module = importlib.import_module(f"fastenhancer_amd.models.{hps['model']}.model")
    cls = module.Model if offline else module.ONNXModel
    model = cls(**hps["model_kwargs"]).to(device).eval()
    if checkpoint is not None:
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt, strict=True)
        model.remove_weight_reparameterizations()
    return model


def read_wav(path: str, sr: int) -> np.ndarray:
    """mono float32 in [-1, 1]; raises if the file's rate differs from the model's (no resampler here)."""
    rate, data = wavfile.read(path)
    if rate != sr:
        raise ValueError(f"{path}: sampling rate {rate} != model sampling rate {sr} (resample the file first)")
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)      # librosa.load(mono=True)
    return x


def write_wav(path: str, sr: int, x: np.ndarray):
    wavfile.write(path, sr, np.asarray(x, dtype=np.float32))
