"""SI-SDR as the reference's evaluation script computes it (scripts/metrics_ns.py:38-52): no mean subtraction,
eps = 1e-7 inside both the projection and the log, in the dtype of its inputs.  The function itself does NOT mask the
signals - its caller multiplies clean / enhanced by the length mask first (scripts/metrics_ns.py:128,134) - and the
per-utterance value is constant over time, so the masked mean of line 52 returns it unchanged.  Pinned on outputs of the
reference's own function (tests/golden/si_snr.npz, tools/gen_golden.py).  PESQ / STOI stay external."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def si_snr(wav_hat: Tensor, wav: Tensor, mask: Optional[Tensor] = None, eps: float = 1e-7) -> Tensor:
    """wav_hat (s1), wav (s2): [B, L]; mask [B, L] of valid samples (or None = all valid) -> SI-SDR in dB per utterance [B]."""
    if mask is None:
        mask = torch.ones_like(wav)
    s1_s2 = (wav_hat * wav).sum(-1, keepdim=True)
    s2_s2 = (wav * wav).sum(-1, keepdim=True)
    s_target = s1_s2 / (s2_s2 + eps) * wav
    e_noise = wav_hat - s_target
    target_norm = (s_target * s_target).sum(-1, keepdim=True)
    noise_norm = (e_noise * e_noise).sum(-1, keepdim=True)
    snr = torch.log10(target_norm / (noise_norm + eps) + eps)          # [B, 1]
    return 10.0 * (snr * mask).sum(dim=1) / mask.sum(dim=1)


def masked_si_snr(wav_hat: Tensor, wav: Tensor, lengths: Tensor) -> Tensor:
    """The evaluation loop's use (scripts/metrics_ns.py:125-137): zero both signals past each utterance's length, then si_snr."""
    L = wav.size(-1)
    mask = (torch.arange(L, device=wav.device)[None, :] < lengths[:, None]).to(wav.dtype)
    return si_snr(wav_hat * mask, wav * mask, mask)
