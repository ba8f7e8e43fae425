"""Host-side mirror of ``models.fastenhancer.ln.model`` of the reference (models/fastenhancer/ln/model.py;
configs/ablation/ln_b.yaml): FastEnhancer with ``GroupNorm(1, C)`` after every conv (statistics over the channels and
sub-bands of a frame - nothing folds into the convs, which carry their own biases) and the reference's ``LayerNorm`` over
(F2, C2) after the blocks' fc layers (as written there: ``diff.addcmul(inv_std * weight, bias)``, i.e. the centred input plus
``inv_std * weight * bias``).  ``ONNXModel`` / ``Model`` take the yaml ``model_kwargs`` verbatim; everything else - call
surface, cache list (one GRU state per block), the HIP engine underneath - is the default model's mirror."""
from __future__ import annotations

from ....config import ln_config
from ..default import model as _default


class ONNXModel(_default.ONNXModel):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=ln_config(**model_kwargs))


class Model(_default.Model):
    def __init__(self, **model_kwargs):
        super().__init__(_cfg=ln_config(**model_kwargs))
