"""Host-side mirror of ``models.bsrnn.model`` of the reference (models/bsrnn/model.py): ``ONNXModel`` (streaming,
spec -> spec with 2*num_layers LSTM caches) and ``Model`` (offline wav -> wav), built from the yaml ``model_kwargs``.
All arithmetic runs in libfastenhancer_hip.so (bsrnn_frame_kernel); inference only."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from ...config import BSRNNConfig
from ...engine import Engine
from ...weights import bsrnn_default_state_dict, bsrnn_expected_fused_shapes, bsrnn_fold_state_dict, check_shapes
from ...stft import CompressedSTFT, ONNXSTFT


class ONNXModel:
    def __init__(self, **model_kwargs):
        self.cfg = BSRNNConfig.from_model_kwargs(**model_kwargs)
        self.input_compression = self.cfg.input_compression
        self.num_layers = self.cfg.num_layers
        self.stft = self.get_stft()
        self.device = torch.device("cpu")
        self._sd: tp.Dict[str, Tensor] = bsrnn_default_state_dict(self.cfg)
        self._engine: tp.Optional[Engine] = None
        self.training = False

    def get_stft(self):
        """models/bsrnn/model.py:323-329"""
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
        fused = bsrnn_fold_state_dict(state_dict, self.cfg)
        check_shapes(fused, bsrnn_expected_fused_shapes(self.cfg), strict=strict)
        self._sd = {k: torch.as_tensor(v).detach().clone() for k, v in state_dict.items()}
        self._engine = None
        return self

    def remove_weight_reparameterizations(self):
        self._sd = bsrnn_fold_state_dict(self._sd, self.cfg)

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
        """models/bsrnn/model.py:409-416 (onnx form), sized for the B = x.size(0) streams of the batch."""
        B = x.size(0)
        return [torch.zeros(B * self.cfg.n_bands, self.cfg.hidden, dtype=torch.float32, device=x.device)
                for _ in range(2 * self.cfg.num_layers)]

    def forward(self, spec_noisy: Tensor, *args: Tensor):
        """input/output: [B, n_fft//2+1, T_spec, 2]; returns (spec_hat, *cache_out) (models/bsrnn/model.py:418-448).
        Unlike the reference's LSTMCell path (T=1 only), any T >= 1 is accepted."""
        B = spec_noisy.size(0)
        cfg = self.cfg
        n = 2 * cfg.num_layers
        if len(args) == 0:
            st = torch.zeros(n, B * cfg.n_bands, cfg.hidden, dtype=torch.float32, device=spec_noisy.device)
        else:
            assert len(args) == n, f"expected {n} caches, got {len(args)}"
            st = torch.stack([c.reshape(B * cfg.n_bands, cfg.hidden) for c in args], dim=0).contiguous().float()
        spec_hat = self.engine.spec_step(spec_noisy.contiguous().float(), st)
        return (spec_hat, *[st[i] for i in range(n)])

    __call__ = forward


class Model(ONNXModel):
    """Offline wav -> wav (models/bsrnn/model.py:463-483): forward(noisy) -> (wav_hat, spec_hat [B, 257, T, 2])."""

    def get_stft(self):
        """models/bsrnn/model.py:467-475: CompressedSTFT keeping all 257 bins"""
        return CompressedSTFT(self, self.cfg, discard_last_freq_bin=False)

    def forward(self, noisy: Tensor):
        if isinstance(noisy, (list, tuple)):      # utterances of different lengths, one batched call: (list of wavs, list of specs)
            return self.engine.offline_ragged(list(noisy))
        return self.engine.offline(noisy.to(self.engine.device))

    __call__ = forward
