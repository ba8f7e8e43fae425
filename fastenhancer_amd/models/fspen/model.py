"""Host-side mirror of ``models.fspen.model`` of the reference (models/fspen/model.py): ``ONNXModel`` (streaming, spec -> spec
with num_blocks * groups inter-GRU caches) and ``Model`` (offline wav -> wav), built from the yaml ``model_kwargs``
(configs/others/fspen.yaml).  All arithmetic runs in libfastenhancer_hip.so (fspen_frame_kernel); inference only."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from ...config import FSPENConfig
from ...engine import Engine
from ...weights import check_shapes, fspen_default_state_dict, fspen_expected_fused_shapes, fspen_fold_state_dict
from ...stft import CompressedSTFT, ONNXSTFT


class ONNXModel:
    def __init__(self, **model_kwargs):
        self.cfg = FSPENConfig.from_model_kwargs(**model_kwargs)
        self.input_compression = self.cfg.input_compression
        self.stft = self.get_stft()
        self.device = torch.device("cpu")
        self._sd: tp.Dict[str, Tensor] = fspen_default_state_dict(self.cfg)
        self._engine: tp.Optional[Engine] = None
        self.training = False

    def get_stft(self):
        """models/fspen/model.py:279-286"""
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
        fused = fspen_fold_state_dict(state_dict, self.cfg)
        check_shapes(fused, fspen_expected_fused_shapes(self.cfg), strict=strict)
        self._sd = {k: torch.as_tensor(v).detach().clone() for k, v in state_dict.items()}
        self._engine = None
        return self

    def remove_weight_reparameterizations(self):
        self._sd = fspen_fold_state_dict(self._sd, self.cfg)

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
        """models/fspen/model.py:293-297 (-> :111-116), sized for the B = x.size(0) streams of the batch."""
        c = self.cfg
        return [torch.zeros(1, x.size(0) * (c.freq // c.groups), c.dpe_channels, dtype=torch.float32, device=x.device)
                for _ in range(c.n_caches)]

    def forward(self, spec_noisy: Tensor, *args: Tensor):
        """input/output: [B, n_fft//2+1, T, 2]; returns (spec_hat, *cache_out) (models/fspen/model.py:409-429)."""
        B = spec_noisy.size(0)
        c = self.cfg
        rows = B * (c.freq // c.groups)
        if len(args) == 0:
            st = torch.zeros(c.n_caches, rows, c.dpe_channels, dtype=torch.float32, device=spec_noisy.device)
        else:
            assert len(args) == c.n_caches, f"expected {c.n_caches} caches, got {len(args)}"
            st = torch.stack([a.reshape(rows, c.dpe_channels) for a in args], dim=0).contiguous().float()
        spec_hat = self.engine.spec_step(spec_noisy.contiguous().float(), st)
        return (spec_hat, *[st[i].view(1, rows, c.dpe_channels) for i in range(c.n_caches)])

    __call__ = forward


class Model(ONNXModel):
    """Offline wav -> wav (models/fspen/model.py:432-449): forward(noisy) -> (wav_hat, spec_hat [B, 257, T, 2])."""

    def get_stft(self):
        """models/fspen/model.py:433-441: CompressedSTFT keeping all 257 bins"""
        return CompressedSTFT(self, self.cfg, discard_last_freq_bin=False)

    def forward(self, noisy: Tensor):
        if isinstance(noisy, (list, tuple)):      # utterances of different lengths, one batched call: (list of wavs, list of specs)
            return self.engine.offline_ragged(list(noisy))
        return self.engine.offline(noisy.to(self.engine.device))

    __call__ = forward
