"""Host-side mirror of ``models.fastenhancer.noncausal.model`` of the reference (models/fastenhancer/noncausal/model.py;
configs/fastenhancer_dns/huge_noncausal.yaml, huge_noncausal_24khz.yaml, configs/fastenhancer_48khz/huge_noncausal.yaml): FastEnhancer
whose blocks run a BIDIRECTIONAL GRU over time (``nn.GRU(C2, C2, bidirectional=True)``, :186) followed by ``rnn_fc: Linear(2 C2 -> C2)``
(:187).  The reference module defines the offline ``Model`` only (:348; ``forward(noisy) -> (wav_hat, spec_hat)``, :628-635): there is no
``ONNXModel``, no cache list and no streaming step - a reverse-time scan needs the whole utterance.

On the GPU the model runs on the time-batched (layer-by-layer) engine of csrc/tb_kernels.hip.h: encoder pass over all frames, per block
a forward and a reverse scan over time (only ``W_hh h`` serial) + a batched attention pass, decoder pass, overlap-add."""
from __future__ import annotations

import typing as tp

from torch import Tensor

from ....config import noncausal_config
from ....stft import CompressedSTFT
from ..default import model as _default


class Model(_default.ONNXModel):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=noncausal_config(**model_kwargs))

    def get_stft(self):
        """noncausal/model.py:483-492: CompressedSTFT(compression=input_compression, discard_last_freq_bin=True)"""
        return CompressedSTFT(self, self.cfg, discard_last_freq_bin=True)

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        raise AttributeError("the noncausal model has no caches (models/fastenhancer/noncausal/model.py defines the offline Model only)")

    def forward(self, noisy: Tensor):
        """noncausal/model.py:628-635: noisy [B, T_wav] -> (wav_hat [B, H*(Tw//H)], spec_hat [B, F0, T, 2])"""
        if isinstance(noisy, (list, tuple)):      # utterances of different lengths, one batched call: (list of wavs, list of specs)
            return self.engine.offline_ragged(list(noisy))
        return self.engine.offline(noisy.to(self.engine.device))

    __call__ = forward
