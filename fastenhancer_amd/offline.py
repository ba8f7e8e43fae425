"""Offline wav -> wav path: Model.forward (models/fastenhancer/default/model.py:728-735) with CompressedSTFT
(functional/audio_modules.py:70-164).  One launch of the frame kernel in offline mode: centered reflect-padded
STFT, zero initial GRU state carried through all T frames, masked compressed spectrum out, torch.istft-style
envelope-normalised overlap-add."""
from __future__ import annotations

from torch import Tensor


def offline_forward(model, noisy: Tensor):
    """returns (wav_hat [B, H*(Tw//H)], spec_hat [B, F0, T, 2]) like the reference's Model.forward"""
    return model.engine.offline(noisy.to(model.engine.device))
