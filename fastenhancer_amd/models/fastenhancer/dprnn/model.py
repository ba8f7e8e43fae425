"""Host-side mirror of ``models.fastenhancer.dprnn.model`` of the reference
(models/fastenhancer/dprnn/model.py; configs/ablation/dprnn_{t,b,s,m,l}.yaml): FastEnhancer whose blocks are dual-path
RNNs - the time GRU of the RNNFormer block, then a bidirectional GRU over the sub-bands (``channels_frnn`` hidden units
per direction, zero initial state every frame) in place of the attention; no positional embedding.  ``ONNXModel`` /
``Model`` take the yaml ``model_kwargs`` verbatim (``dprnn_kwargs``, ``final_scale`` ...) and load the reference's
checkpoints (module names ``dprnn_pre`` / ``dprnn_block.k.trnn`` / ``frnn`` / ``dprnn_post``, training or fused form);
everything else - call surface, cache list (one GRU state per block), the HIP engine underneath - is the default model's
mirror."""
from __future__ import annotations

from ....config import dprnn_config
from ..default import model as _default


class ONNXModel(_default.ONNXModel):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=dprnn_config(**model_kwargs))


class Model(_default.Model):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=dprnn_config(**model_kwargs))
