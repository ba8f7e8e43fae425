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
        # per batch size: the geometry of the reference's cache list inside one state buffer of the C ABI, as
        # (offset in floats, shape, stride) per tensor - what split_state() returns for a fresh state
        self._geom: tp.Dict[int, tp.List[tp.Tuple[int, torch.Size, tp.Tuple[int, ...]]]] = {}
        # dptransformer variant: its K / V caches are rings in the state, the tensors handed out are rotated COPIES - the
        # tensors of the last call are recognised by identity and name the state buffer they were gathered from
        self._handed: tp.Tuple[tp.Optional[Tensor], tp.List[Tensor], tp.List[int]] = (None, [], [])

    @property
    def engine(self) -> Engine:
        return self.model.engine

    def _geometry(self, B: int, state: Tensor):
        g = self._geom.get(B)
        if g is None:
            p0 = state.data_ptr()
            g = [((v.data_ptr() - p0) // 4, v.shape, v.stride()) for v in self.engine.split_state(state, B, head0=True)]
            self._geom[B] = g
        return g

    def _views(self, state: Tensor, B: int) -> tp.List[Tensor]:
        return [state.as_strided(shape, stride, off) for off, shape, stride in self._geometry(B, state)]

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        """scripts/export_onnx.py:43-46: [cache_stft, cache_istft] ++ model caches (zeros).  On the model's GPU the
        tensors are views of ONE freshly allocated state buffer of the C ABI, so forward() need not re-pack them."""
        if x.is_cuda:
            B = x.size(0)
            return self._views(self.engine.new_state(B), B)
        cache_list = self.model.stft.initialize_cache(x)
        cache_list.extend(self.model.initialize_cache(x))
        return cache_list

    def _source(self, caches: tp.List[Tensor], B: int, n: int) -> tp.Optional[Tensor]:
        """the state buffer the given cache tensors are exactly the views of (same memory, same layout), else None"""
        # dptransformer: the K / V tensors handed out are rotated COPIES of the rings, recognised by identity - and only while nobody has
        # written to them (torch's version counters, recorded at hand-out): an edited copy (say, one stream's cache zeroed) is packed like
        # any foreign tensor instead of being silently replaced by the internal buffer
        if self._handed[0] is not None and len(caches) == len(self._handed[1]) and \
                all(c is w and c._version == v for c, w, v in zip(caches, self._handed[1], self._handed[2])):
            return self._handed[0]
        base = caches[0]._base
        g = self._geom.get(B)
        if base is None or g is None or len(g) != len(caches) or base.numel() != n or base.dtype != torch.float32 or not base.is_cuda \
                or base.dim() != 1 or base.stride(0) != 1:
            return None
        p0 = base.data_ptr()
        for c, (off, shape, stride) in zip(caches, g):
            if c.data_ptr() != p0 + 4 * off or c.shape != shape or c.stride() != stride or c.dtype != torch.float32:
                return None
        return base

    def forward(self, wav_in: Tensor, cache_stft: Tensor, cache_istft: Tensor, *cache_model: Tensor):
        """wav_in [B, H].  VALUE SEMANTICS, like the reference's Model.forward (scripts/export_onnx.py:48-58): the caches passed
        in are never written, the caches returned are views of a state buffer allocated for THIS call (the caching allocator's
        job) and are never written again - a caller may keep them for roll-back, look-ahead or A/B runs and feed them back in at
        any later time.  Caches that are the views handed out by initialize_cache() / an earlier call (the driver loop of
        scripts/test_onnx.py) cost one device copy into the new buffer; any other tensors (clones, CPU tensors, tensors of the
        reference) are packed into the state layout first.  dptransformer models return rotated copies of their K / V rings (the reference's
        oldest-first order); fed back UNMODIFIED they are recognised and the rings themselves are used, edited in place they are packed
        like any other tensor (their version counters tell).  The last handed-out set keeps its state buffer alive."""
        eng = self.engine
        B = wav_in.size(0)
        caches = [cache_stft, cache_istft, *cache_model]
        n = eng.state_floats(B)
        src = self._source(caches, B, n)
        if src is not None and src.device == eng.device:
            state = torch.empty_like(src)
            state.copy_(src)
        else:
            state = eng.pack_state([t.to(eng.device) for t in caches], B)        # (torch.cat: a fresh buffer)
        wav_out = eng.step(wav_in.to(eng.device, torch.float32), state, T=1)
        if getattr(self.cfg, "dpt", False):
            out = eng.split_state(state, B)
            self._handed = (state, out, [t._version for t in out])
            return (wav_out, *out)
        return (wav_out, *self._views(state, B))

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
