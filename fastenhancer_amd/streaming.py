"""The wav -> wav streaming step and its driver loop.

``StreamingModel`` mirrors ``class Model`` of scripts/export_onnx.py:38-58 (what the
reference exports to ONNX and what scripts/test_onnx.py feeds hop by hop):

    wav_out, cache_stft, cache_istft, *cache_model = M(wav_in, cache_stft, cache_istft, *cache_model)

``enhance_stream`` mirrors the loop of scripts/test_onnx.py:11-60 for B >= 1 streams."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from .config import FEConfig
from .engine import Engine


class StreamingModel:
    def __init__(self, model):
        """model: a fastenhancer_amd ONNXModel (weights already loaded)."""
        self.model = model
        self.cfg: FEConfig = model.cfg

    @property
    def engine(self) -> Engine:
        return self.model.engine

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        cache_list = self.model.stft.initialize_cache(x)
        cache_list.extend(self.model.initialize_cache(x))
        return cache_list

    def forward(self, wav_in: Tensor, cache_stft: Tensor, cache_istft: Tensor, *cache_model: Tensor):
        """wav_in [B, H]; functional: returns new cache tensors, inputs untouched."""
        eng = self.engine
        B = wav_in.size(0)
        state = eng.pack_state([cache_stft, cache_istft, *cache_model], B)
        wav_out = eng.step(wav_in.contiguous().float(), state, T=1)
        return (wav_out, *eng.split_state(state, B))

    __call__ = forward


@torch.no_grad()
def enhance_stream(model, wav: Tensor, frames_per_call: int = 1) -> Tensor:
    """Driver loop of scripts/test_onnx.py:11-60 for wav [B, L] on the model's device:
    clip to [-1,1], right-pad n_fft zeros, run hop by hop (or ``frames_per_call`` hops per
    launch), drop the n_fft-hop samples of latency, clip."""
    cfg: FEConfig = model.cfg
    eng: Engine = model.engine
    N, H = cfg.n_fft, cfg.hop_size
    wav = wav.to(eng.device, torch.float32).clamp(-1, 1)
    B, length = wav.shape
    n_hops = len(range(0, length + N - H, H))
    total = n_hops * H
    padded = torch.zeros(B, max(total, length + N), dtype=torch.float32, device=eng.device)
    padded[:, :length] = wav
    out = torch.empty(B, total, dtype=torch.float32, device=eng.device)
    state = eng.new_state(B)
    t = 0
    while t < n_hops:
        T = min(frames_per_call, n_hops - t)
        eng.step(padded[:, t * H:(t + T) * H], state, out[:, t * H:(t + T) * H], T=T)
        t += T
    s = N - H
    return out[:, s:s + length].clamp(-1.0, 1.0)
