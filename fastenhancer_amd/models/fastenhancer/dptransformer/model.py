"""Host-side mirror of ``models.fastenhancer.dptransformer.model`` of the reference
(models/fastenhancer/dptransformer/model.py; configs/ablation/dpt_{t,b,s,m}.yaml): FastEnhancer whose blocks are dual-path
transformers - a causal attention over the last ``lookbehind`` frames of each sub-band (per block a K and a V cache
``[B*F2, NH, lookbehind, C2/NH]``, oldest frame first, and a learned positional bias ``pe [NH, lookbehind+1]`` shared by the
blocks) in place of the time GRU, then the sub-band attention of the default block.  ``ONNXModel`` / ``Model`` take the yaml
``model_kwargs`` verbatim (``dpt_kwargs``, ``final_scale``, ``final_scale_init`` ...) and load the reference's checkpoints
(module names ``dpt_pre`` / ``dpt_block.k.time_attn`` / ``freq_attn`` / ``dpt_post`` / ``pe``, training or fused form).
Streaming (caches given, one frame per call in the reference; any T here = T such steps) attends to the zero-initialised
caches like the reference; called without caches the frames before the start are masked, as in the reference's offline path."""
from __future__ import annotations

from ....config import dpt_config
from ..default import model as _default


class ONNXModel(_default.ONNXModel):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=dpt_config(**model_kwargs))


class Model(_default.Model):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=dpt_config(**model_kwargs))
