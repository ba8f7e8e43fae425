"""Host-side mirror of ``models.fastenhancer.time_kernel.model`` of the reference
(models/fastenhancer/time_kernel/model.py; configs/ablation/time_kernel_b.yaml): FastEnhancer whose encoder / decoder
k = 3 convs are causal Conv2d with ``kernel_size_time`` taps over time.  Streaming, each such conv carries its input of
the previous ``kernel_size_time - 1`` frames as a cache tensor (B, C1, kt-1, F1); the model's cache list is
``encoder caches, GRU states, decoder caches`` (:746-754).  ``ONNXModel`` / ``Model`` take the yaml ``model_kwargs``
verbatim (``kernel_size_freq``, ``kernel_size_time``, ``final_scale`` ...); everything else - call surface, checkpoint
loading with the deployment folds, the HIP engine underneath - is the default model's mirror."""
from __future__ import annotations

from ....config import time_kernel_config
from ..default import model as _default


class ONNXModel(_default.ONNXModel):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=time_kernel_config(**model_kwargs))


class Model(_default.Model):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=time_kernel_config(**model_kwargs))
