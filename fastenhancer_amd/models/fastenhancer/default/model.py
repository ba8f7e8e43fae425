"""Host-side mirror of ``models.fastenhancer.default.model`` of the reference
(models/fastenhancer/default/model.py): the classes ``ONNXModel`` (streaming,
spec -> spec, caches threaded by the caller) and ``Model`` (offline wav -> wav),
constructed from the yaml ``model_kwargs`` verbatim, e.g.

    module = importlib.import_module(f"fastenhancer_amd.models.{hps.model}.model")
    model = module.ONNXModel(**hps.model_kwargs)            # scripts/export_onnx.py:32-35
    model.load_state_dict(ckpt["model"], strict=True)       # wrappers/ns.py:313
    model.remove_weight_reparameterizations()               # scripts/export_onnx.py:78

All arithmetic runs in libfastenhancer_hip.so on the GPU the model was moved to;
these classes only hold the checkpoint and marshal torch tensors across the C ABI.
They are inference-only (``.eval()``; no autograd, no training forward)."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from ....config import FEConfig
from ....engine import Engine
from ....stft import CompressedSTFT, ONNXSTFT
from ....weights import default_state_dict, fold_state_dict


class ONNXModel:
    def __init__(self, _cfg: tp.Optional[FEConfig] = None, **model_kwargs):
        # (_cfg: a ready FEConfig - how the time_kernel variant's mirror, whose yaml keys differ, constructs this class)
        self.cfg = _cfg if _cfg is not None else FEConfig.from_model_kwargs(**model_kwargs)
        self.input_compression = self.cfg.input_compression
        self.rf_ch, self.rf_freq = self.cfg.rf_channels, self.cfg.rf_freq
        self.stft = self.get_stft()
        self.device = torch.device("cpu")
        self._sd: tp.Dict[str, Tensor] = default_state_dict(self.cfg)
        self._engine: tp.Optional[Engine] = None
        self.training = False

    def get_stft(self):
        """model.py:523-530: the streaming model carries an ONNXSTFT"""
        return ONNXSTFT(self, self.cfg)

    # ---- nn.Module-like plumbing -----------------------------------------------------------
    def eval(self):
        self.training = False
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

    def state_dict(self) -> tp.Dict[str, Tensor]:
        return dict(self._sd)

    def load_state_dict(self, state_dict: tp.Mapping[str, Tensor], strict: bool = True):
        from ....weights import check_fused
        fused = fold_state_dict(state_dict, self.cfg)
        check_fused(fused, self.cfg, strict=strict)
        self._sd = {k: torch.as_tensor(v).detach().clone() for k, v in state_dict.items()}
        self._engine = None
        return self

    def remove_weight_reparameterizations(self):
        """model.py:532-608.  Folding happens when the blob is built; make it visible in state_dict()."""
        self._sd = fold_state_dict(self._sd, self.cfg)

    def flatten_parameters(self):
        pass

    def parameters(self):
        return iter(self._sd.values())

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            eng = Engine(self.cfg, self.device)
            eng.load_state_dict(self._sd)     # raises without a GPU: no CPU fallback
            self._engine = eng
        return self._engine

    # ---- reference API ---------------------------------------------------------------------
    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        """model.py:614-618, sized for the B = x.size(0) streams of the batch (b-major).  (time_kernel variant: + the
        causal convs' frame caches, in its order encoder / GRU / decoder - time_kernel/model.py:746-754)"""
        B, c = x.size(0), self.cfg
        if c.dpt:     # dptransformer variant: h_k, h_v per block (dptransformer/model.py:194-198, 740-744)
            return [torch.zeros(B * c.rf_freq, c.rf_heads, c.lookbehind, c.rf_channels // c.rf_heads, dtype=torch.float32, device=x.device)
                    for _ in range(2 * c.rf_blocks)]
        hs = [torch.zeros(1, B * self.rf_freq, self.rf_ch, dtype=torch.float32, device=x.device) for _ in range(c.rf_blocks)]
        if not c.time_kernel:
            return hs
        cc = lambda: [torch.zeros(B, c.channels, c.kernel_size_time - 1, c.F1, dtype=torch.float32, device=x.device) for _ in range(c.n_layers)]
        return cc() + hs + cc()

    def forward(self, spec_noisy: Tensor, *args: Tensor):
        """input/output: [B, n_fft//2+1, T_spec, 2]; returns (spec_hat, *cache_out)  (model.py:677-710).
        Functional like the reference: the caches passed in are not modified."""
        B = spec_noisy.size(0)
        cfg, eng = self.cfg, self.engine
        n_caches = 2 * cfg.rf_blocks if cfg.dpt else cfg.rf_blocks + (2 * cfg.n_layers if cfg.time_kernel else 0)
        if len(args) == 0:
            h = torch.zeros(eng.model_state_floats(B), dtype=torch.float32, device=eng.device)
            if cfg.dpt:
                # the dptransformer variant without caches masks the frames before the start (dptransformer/model.py:216-218)
                # instead of attending to zero caches: marked by +inf in the first element of every K slot (fe_config.lookbehind).
                # (The caches returned keep all L slots, the not-yet-filled ones still marked; the reference returns min(T, L) slots.)
                n = B * cfg.rf_freq * cfg.rf_channels * cfg.lookbehind
                hd = cfg.rf_channels // cfg.rf_heads
                for k in range(cfg.rf_blocks):
                    h[2 * k * n:(2 * k + 1) * n].view(-1, hd)[:, 0] = float("inf")
        else:
            assert len(args) == n_caches, f"expected {n_caches} caches, got {len(args)}"
            h = torch.cat([t.to(eng.device, torch.float32) for t in eng.model_state_order(list(args))]).contiguous()
        spec_hat = eng.spec_step(spec_noisy.to(eng.device).contiguous().float(), h)
        # the updated caches as views of h, in the reference's list order
        dummy = torch.empty(2 * B * cfg.cache_len, dtype=torch.float32, device=h.device)
        return (spec_hat, *eng.split_state(torch.cat([dummy, h]), B)[2:])

    __call__ = forward


class Model(ONNXModel):
    """Offline wav -> wav model (model.py:713-735): forward(noisy [B, T_wav]) -> (wav_hat, spec_hat)."""

    def get_stft(self):
        """model.py:717-726: CompressedSTFT(compression=input_compression, discard_last_freq_bin=True)"""
        return CompressedSTFT(self, self.cfg, discard_last_freq_bin=True)

    def forward(self, noisy: Tensor):
        """One fused launch sequence (fe_offline): centered STFT, all T frames, envelope-normalised overlap-add;
        returns (wav_hat [B, H*(Tw//H)], spec_hat [B, F0, T, 2]).  ``self.stft`` / ``self.stft.inverse`` give the
        front / back end alone."""
        if isinstance(noisy, (list, tuple)):      # utterances of different lengths, one batched call: (list of wavs, list of specs)
            return self.engine.offline_ragged(list(noisy))
        return self.engine.offline(noisy.to(self.engine.device))

    __call__ = forward
