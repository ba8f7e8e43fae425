"""Host-side mirror of ``models.lisennet.model`` of the reference (models/lisennet/model.py): ``ONNXModel`` (streaming, spec -> spec
with 1 + 3 + 2 n_blocks + 1 caches) and ``Model`` (offline wav -> wav), built from the yaml ``model_kwargs``
(configs/others/lisennet.yaml).  All arithmetic runs in libfastenhancer_hip.so (lisennet_frame_kernel); inference only."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from ...config import LiSenNetConfig
from ...engine import Engine
from ...weights import check_shapes, lisennet_default_state_dict, lisennet_expected_shapes, lisennet_state_dict
from ...stft import CompressedSTFT, ONNXSTFT


class ONNXModel:
    def __init__(self, **model_kwargs):
        self.cfg = LiSenNetConfig.from_model_kwargs(**model_kwargs)
        self.input_compression = self.cfg.input_compression
        self.n_freqs = self.cfg.n_fft // 2 + 1
        self.stft = self.get_stft()
        self.device = torch.device("cpu")
        self._sd: tp.Dict[str, Tensor] = lisennet_default_state_dict(self.cfg)
        self._engine: tp.Optional[Engine] = None
        self.training = False

    def get_stft(self):
        """models/lisennet/model.py:342-349"""
        return ONNXSTFT(self, self.cfg)

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("fastenhancer_amd models are inference-only")
        return self

    def to(self, device):
        self.device = torch.device(device)
        self._engine = None
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, state_dict: tp.Mapping[str, Tensor], strict: bool = True):
        check_shapes(lisennet_state_dict(state_dict, self.cfg), lisennet_expected_shapes(self.cfg), strict=strict)
        self._sd = {k: torch.as_tensor(v).detach().clone() for k, v in state_dict.items()}
        self._engine = None
        return self

    def remove_weight_reparameterizations(self):
        """models/lisennet/model.py:476-477: nothing to remove"""

    def flatten_parameters(self):
        pass

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            eng = Engine(self.cfg, self.device)
            eng.load_state_dict(self._sd)
            self._engine = eng
        return self._engine

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        """models/lisennet/model.py:380-396, sized for the B = x.size(0) streams of the batch."""
        return [torch.zeros(*s, dtype=torch.float32, device=x.device) for s in self.cfg.cache_shapes(x.size(0))]

    def forward(self, spec_noisy: Tensor, *args: Tensor):
        """input/output: [B, n_fft//2+1, T, 2]; returns (spec_hat, *cache_out) (models/lisennet/model.py:434-474)."""
        B = spec_noisy.size(0)
        shapes = self.cfg.cache_shapes(B)
        if len(args) == 0:
            args = tuple(torch.zeros(*s, dtype=torch.float32, device=spec_noisy.device) for s in shapes)
        assert len(args) == len(shapes), f"expected {len(shapes)} caches, got {len(args)}"
        st = torch.cat([a.reshape(-1).float() for a in args]).contiguous()
        spec_hat = self.engine.spec_step(spec_noisy.contiguous().float(), st)
        outs, o = [], 0
        for s in shapes:
            n = 1
            for d in s:
                n *= d
            outs.append(st[o:o + n].view(*s))
            o += n
        return (spec_hat, *outs)

    __call__ = forward


class Model(ONNXModel):
    """Offline wav -> wav (models/lisennet/model.py:480-531): forward(noisy) -> (wav_hat, spec_hat [B, 257, T, 2]).
    NB its phase features take `current - previous` (torch.diff) where ONNXModel takes `previous - current`; the kernel follows
    each, as the reference does."""

    def get_stft(self):
        """models/lisennet/model.py:481-489: CompressedSTFT keeping all 257 bins"""
        return CompressedSTFT(self, self.cfg, discard_last_freq_bin=False)

    def forward(self, noisy: Tensor):
        if isinstance(noisy, (list, tuple)):      # utterances of different lengths, one batched call: (list of wavs, list of specs)
            return self.engine.offline_ragged(list(noisy))
        return self.engine.offline(noisy.to(self.engine.device))

    __call__ = forward
