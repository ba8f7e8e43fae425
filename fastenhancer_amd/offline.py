"""Offline wav -> wav path (Model.forward, models/fastenhancer/default/model.py:728-735 with
CompressedSTFT, functional/audio_modules.py:70-164).  Placeholder until fe_offline lands."""
from __future__ import annotations

from torch import Tensor


def offline_forward(model, noisy: Tensor):
    raise NotImplementedError("offline Model.forward is not built yet (SURVEY.md §8f rank 2)")
