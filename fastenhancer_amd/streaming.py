"""The wav -> wav streaming step and its driver loop.

``StreamingModel`` mirrors ``class Model`` of scripts/export_onnx.py:38-58 (what the
reference exports to ONNX and what scripts/test_onnx.py feeds hop by hop):

    wav_out, cache_stft, cache_istft, *cache_model = M(wav_in, cache_stft, cache_istft, *cache_model)

``enhance_stream`` mirrors the loop of scripts/test_onnx.py:11-60 for B >= 1 streams."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from .config import FEConfig
from .engine import Engine


class StreamingModel:
    def __init__(self, model):
        """model: a fastenhancer_amd ONNXModel (weights already loaded)."""
        self.model = model
        self.cfg: FEConfig = model.cfg
        self._buf: tp.List[tp.Optional[Tensor]] = [None, None]     # two state buffers of the C ABI, used alternately
        self._views: tp.List[tp.Optional[tp.List[Tensor]]] = [None, None]
        self._B = 0
        # dptransformer variant: its K / V caches are rings in the state, the tensors handed out are rotated copies - the
        # tensors of the last call are recognised by identity instead of by address
        self._handed: tp.Tuple[int, tp.List[Tensor]] = (-1, [])

    @property
    def engine(self) -> Engine:
        return self.model.engine

    @staticmethod
    def _same_device(a: torch.device, b) -> bool:
        """torch.device('cuda') != torch.device('cuda:0'): compare type and index with None -> the current device"""
        b = torch.device(b)
        if a.type != b.type:
            return False
        if a.type != "cuda":
            return True
        cur = torch.cuda.current_device()
        return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)

    def _ensure(self, B: int, device, fresh: bool = False):
        if fresh or self._buf[0] is None or self._B != B or not self._same_device(self._buf[0].device, device):
            eng = self.engine
            self._buf = [eng.new_state(B), eng.new_state(B)]
            self._views = [eng.split_state(self._buf[0], B, head0=True), eng.split_state(self._buf[1], B, head0=True)]      # (fresh zero states)
            self._B = B
            self._handed = (-1, [])          # the rotated copies of the previous buffers no longer name a buffer

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        """scripts/export_onnx.py:43-46: [cache_stft, cache_istft] ++ model caches (zeros).  On the model's GPU the
        tensors are views of ONE state buffer of the C ABI, so forward() need not re-pack them.  Every call allocates
        fresh buffers (like the reference, which returns new tensors): the caches of a session still in progress on
        this object keep their memory and simply stop being recognised as views (they are packed like foreign tensors)."""
        if x.is_cuda:
            self._ensure(x.size(0), x.device, fresh=True)
            return list(self._views[0])
        cache_list = self.model.stft.initialize_cache(x)
        cache_list.extend(self.model.initialize_cache(x))
        return cache_list

    def _which(self, caches) -> int:
        """index of the state buffer the given cache tensors are exactly the views of, else -1"""
        if self._handed[0] >= 0 and len(caches) == len(self._handed[1]) and all(c is w for c, w in zip(caches, self._handed[1])):
            return self._handed[0]
        for i in (0, 1):
            v = self._views[i]
            if v is not None and len(v) == len(caches) and all(
                    c.data_ptr() == w.data_ptr() and c.shape == w.shape and c.dtype == w.dtype and c.stride() == w.stride()
                    for c, w in zip(caches, v)):
                return i
        return -1

    def forward(self, wav_in: Tensor, cache_stft: Tensor, cache_istft: Tensor, *cache_model: Tensor):
        """wav_in [B, H]; the caches passed in are never written (like the reference).  ALIASING CONTRACT, unlike the
        reference: the caches returned are views of one of TWO internal state buffers used alternately, so the tensors
        returned by call n are overwritten by call n + 2 - a caller that keeps a snapshot for roll-back must .clone() it
        (INTEGRATION.md).  (dptransformer models return rotated copies of their K / V rings: in-place edits of those
        are not seen by the next call; pass the edited tensors back in, which packs them like foreign tensors.)
        Caches that are the views handed out by initialize_cache() / the previous call (the driver loop of
        scripts/test_onnx.py) cost one device copy into the other state buffer; foreign tensors are packed first."""
        eng = self.engine
        B = wav_in.size(0)
        caches = [cache_stft, cache_istft, *cache_model]
        self._ensure(B, eng.device)
        src = self._which(caches)
        dst = 1 - src if src >= 0 else 0
        if src >= 0:
            self._buf[dst].copy_(self._buf[src])
        else:
            self._buf[dst].copy_(eng.pack_state([t.to(eng.device) for t in caches], B))
        wav_out = eng.step(wav_in.to(eng.device, torch.float32), self._buf[dst], T=1)
        if getattr(self.cfg, "dpt", False):
            out = eng.split_state(self._buf[dst], B)
            self._handed = (dst, out)
            return (wav_out, *out)
        return (wav_out, *self._views[dst])

    __call__ = forward


@torch.no_grad()
def enhance_stream(model, wav: Tensor, frames_per_call: int = 1) -> Tensor:
    """Driver loop of scripts/test_onnx.py:11-60 for wav [B, L] on the model's device:
    clip to [-1,1], right-pad n_fft zeros, run hop by hop (or ``frames_per_call`` hops per
    launch), drop the n_fft-hop samples of latency, clip."""
    cfg: FEConfig = model.cfg
    eng: Engine = model.engine
    N, H = cfg.n_fft, cfg.hop_size
    if not wav.is_cuda:
        # audio in host memory (how scripts/test_onnx.py holds it): hop blocks go over PCIe under the kernels of their neighbours
        # (fe_step_host); the result comes back in pinned host memory
        wav = wav.to(torch.float32).clamp(-1, 1)
        B, length = wav.shape
        n_hops = len(range(0, length + N - H, H))
        T = max(1, int(frames_per_call))
        n_calls = -(-n_hops // T)
        padded = torch.zeros(B, max(n_calls * T * H, length + N), dtype=torch.float32).pin_memory()
        padded[:, :length] = wav
        out = torch.empty(B, n_calls * T * H, dtype=torch.float32).pin_memory()
        state = eng.new_state(B)
        eng.step_host(padded[:, :n_calls * T * H], state, out, T=T)
        torch.cuda.current_stream(eng.device).synchronize()
        s = N - H
        return out[:, s:s + length].clamp(-1.0, 1.0)
    wav = wav.to(eng.device, torch.float32).clamp(-1, 1)
    B, length = wav.shape
    n_hops = len(range(0, length + N - H, H))
    total = n_hops * H
    padded = torch.zeros(B, max(total, length + N), dtype=torch.float32, device=eng.device)
    padded[:, :length] = wav
    out = torch.empty(B, total, dtype=torch.float32, device=eng.device)
    state = eng.new_state(B)
    t = 0
    while t < n_hops:
        T = min(frames_per_call, n_hops - t)
        eng.step(padded[:, t * H:(t + T) * H], state, out[:, t * H:(t + T) * H], T=T)
        t += T
    s = N - H
    return out[:, s:s + length].clamp(-1.0, 1.0)
