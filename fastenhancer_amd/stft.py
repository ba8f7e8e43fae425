"""Mirrors of the reference's STFT modules, as the `.stft` attribute of the model mirrors:

* ``ONNXSTFT``       functional/audio_modules.py:167-303 - streaming, one hop per call, caches threaded by the caller:
                     ``spec, cache = stft(wav_in, cache)`` / ``wav_out, cache = stft.inverse(spec, cache)``; what
                     scripts/export_onnx.py:55-57 composes around ``ONNXModel.forward``.
* ``CompressedSTFT`` functional/audio_modules.py:124-164 - offline, centered: ``spec = stft(noisy)`` /
                     ``wav = stft.inverse(spec_complex)``; what ``Model.forward`` composes (model.py:728-735).

Both launch the stand-alone kernels of csrc/stft_kernels.hip.h through the C ABI (fe_stft_step, fe_istft_step,
fe_stft_offline, fe_istft_offline); inside fe_step / fe_offline the same transforms are fused into the frame kernel."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor


class ONNXSTFT:
    def __init__(self, owner, cfg):
        self._owner = owner                     # the model mirror: provides .engine (lazily, on its device)
        self.n_fft, self.hop_size, self.cache_len = cfg.n_fft, cfg.hop_size, cfg.cache_len
        self.normalized = False

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        """functional/audio_modules.py:238-241"""
        return [torch.zeros(x.size(0), self.cache_len, dtype=torch.float32, device=x.device) for _ in range(2)]

    def forward(self, x: Tensor, cache: Tensor) -> tp.Tuple[Tensor, Tensor]:
        """x [B, hop_size], cache [B, n_fft-hop_size] -> spec [B, n_fft//2+1, 1, 2], new cache (:243-257)"""
        return self._owner.engine.stft_step(x, cache)

    __call__ = forward

    def inverse(self, x: Tensor, cache: Tensor) -> tp.Tuple[Tensor, Tensor]:
        """x [B, n_fft//2+1, 1, 2], cache [B, n_fft-hop_size] -> wav [B, hop_size], new cache (:259-303)"""
        return self._owner.engine.istft_step(x, cache)


class CompressedSTFT:
    def __init__(self, owner, cfg, discard_last_freq_bin: bool):
        self._owner = owner
        self.n_fft, self.hop_size, self.win_size = cfg.n_fft, cfg.hop_size, cfg.win_size
        self.compression = cfg.input_compression
        self.discard_last_freq_bin = discard_last_freq_bin
        self.eps = 1.0e-5

    def forward(self, x: Tensor) -> Tensor:
        """x [B, T_wav] or [B, 1, T_wav] -> compressed spectrum [B, F, T, 2] (:146-155)"""
        return self._owner.engine.stft_offline(x, discard_last=self.discard_last_freq_bin, compress=True)

    __call__ = forward

    def inverse(self, x: Tensor) -> Tensor:
        """x [B, F, T] complex (or [B, F, T, 2] real pairs) in the compressed domain -> wav [B, hop_size*(T-1)] (:157-164)"""
        if x.is_complex():
            x = torch.view_as_real(x)
        return self._owner.engine.istft_offline(x, compress=True)
