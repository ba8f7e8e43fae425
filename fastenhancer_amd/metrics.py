"""SI-SDR as the reference's evaluation script computes it (scripts/metrics_ns.py:43-52): no mean subtraction,
eps = 1e-7 inside both the projection and the log, masked mean over the valid samples.  Used for the parity report
("SISDR identical to 2 d.p." between the HIP path and the reference path); PESQ / STOI stay external."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def si_snr(wav_hat: Tensor, wav: Tensor, mask: Optional[Tensor] = None, eps: float = 1e-7) -> Tensor:
    """wav_hat (s1), wav (s2): [B, L]; mask [B, L] of valid samples (or None) -> SI-SDR in dB per utterance [B]."""
    wav_hat, wav = wav_hat.double(), wav.double()
    if mask is None:
        mask = torch.ones_like(wav)
    mask = mask.double()
    wav_hat, wav = wav_hat * mask, wav * mask
    s1_s2 = (wav_hat * wav).sum(-1, keepdim=True)
    s2_s2 = (wav * wav).sum(-1, keepdim=True)
    s_target = s1_s2 / (s2_s2 + eps) * wav
    e_noise = wav_hat - s_target
    target_norm = (s_target * s_target).sum(-1)
    noise_norm = (e_noise * e_noise).sum(-1)
    return 10.0 * torch.log10(target_norm / (noise_norm + eps) + eps)
