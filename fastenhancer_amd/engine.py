"""Engine: one fe_handle (C ABI) on one GPU + torch-tensor convenience around it.

PyTorch is used for device memory and streams only; all arithmetic of the path
runs in libfastenhancer_hip.so."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_char_p, c_int, c_size_t, c_void_p
from typing import Dict, List, Mapping, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .config import BSRNNConfig, FEConfig, FSPENConfig, LiSenNetConfig
from .weights import (bsrnn_expected_fused_shapes, bsrnn_fold_state_dict, check_fused, check_shapes, fold_state_dict,
                      fspen_expected_fused_shapes, fspen_fold_state_dict, lisennet_expected_shapes, lisennet_state_dict)


def _ptr(t: Optional[Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


def _stream(device: torch.device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """Owns the native handle for one model shape on one device."""

    def __init__(self, cfg: FEConfig, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            # torch.device('cuda') != torch.device('cuda:0'): keep one spelling, the indexed one tensors report
            self.device = torch.device("cuda", torch.cuda.current_device())
        c = _lib.fe_config()
        self.is_bsrnn = isinstance(cfg, BSRNNConfig)
        self.is_fspen = isinstance(cfg, FSPENConfig)
        self.is_lisennet = isinstance(cfg, LiSenNetConfig)
        c.n_fft, c.hop_size, c.win_size = cfg.n_fft, cfg.hop_size, cfg.win_size
        c.input_compression = cfg.input_compression
        if self.is_lisennet:
            c.arch = _lib.FE_ARCH_LISENNET
            c.channels, c.rf_blocks = cfg.num_channels, cfg.n_blocks
        elif self.is_fspen:
            c.arch = _lib.FE_ARCH_FSPEN
            c.channels = cfg.channels[-1]
            c.n_kernels = len(cfg.kernel_size)
            for i, k in enumerate(cfg.kernel_size):
                c.kernel_size[i] = k
            if len(set(cfg.stride)) != 1 or list(cfg.channels) != [4, 16, 32]:
                raise _lib.FEError(f"no FSPEN kernel compiled for channels={list(cfg.channels)} stride={list(cfg.stride)} "
                                   "(configs/others/fspen.yaml is the compiled architecture)")
            c.stride = cfg.stride[0]
            c.rf_channels, c.rf_freq, c.rf_blocks, c.rf_heads = cfg.dpe_channels, cfg.freq, cfg.num_blocks, cfg.groups
        elif self.is_bsrnn:
            c.arch = _lib.FE_ARCH_BSRNN
            c.channels, c.rf_blocks = cfg.num_channels, cfg.num_layers
        else:
            c.arch = _lib.FE_ARCH_FASTENHANCER
            c.channels = cfg.channels
            c.n_kernels = len(cfg.kernel_size)
            for i, k in enumerate(cfg.kernel_size):
                c.kernel_size[i] = k
            c.stride = cfg.stride
            c.rf_channels, c.rf_freq, c.rf_blocks, c.rf_heads = cfg.rf_channels, cfg.rf_freq, cfg.rf_blocks, cfg.rf_heads
            c.kernel_size_time = cfg.kernel_size_time
            c.channels_frnn = cfg.channels_frnn
            c.lookbehind = cfg.lookbehind
            c.ln = 1 if cfg.ln else 0
            c.rf_eps = cfg.rf_eps
            c.bidirectional = 1 if getattr(cfg, "noncausal", False) else 0
        self._h = c_void_p()
        if self.device is not None and self.device.type == "cuda":
            with torch.cuda.device(self.device):
                _lib.check(self.lib.fe_create(byref(c), byref(self._h)), "fe_create")
        else:
            _lib.check(self.lib.fe_create(byref(c), byref(self._h)), "fe_create")
        self.weight_floats = int(self.lib.fe_weight_floats(self._h))
        self.sections = self._read_sections()
        self.flops_per_frame = float(self.lib.fe_flops_per_frame(self._h))
        self.loaded = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self.lib.fe_destroy(h)
            except Exception:
                pass
            self._h = c_void_p()

    # ------------------------------------------------------------------ weights
    def _read_sections(self) -> List[Tuple[str, int, int]]:
        out = []
        for i in range(self.lib.fe_weight_sections(self._h)):
            name, off, cnt = c_char_p(), c_size_t(), c_size_t()
            _lib.check(self.lib.fe_weight_section(self._h, i, byref(name), byref(off), byref(cnt)), "fe_weight_section")
            out.append((name.value.decode(), int(off.value), int(cnt.value)))
        return out

    def make_blob(self, state_dict: Mapping[str, Tensor], strict: bool = True) -> Tensor:
        """reference checkpoint (training or fused form) -> flat fp32 blob on the CPU."""
        if self.is_lisennet:
            fused = lisennet_state_dict(state_dict, self.cfg)
            check_shapes(fused, lisennet_expected_shapes(self.cfg), strict=strict)
        elif self.is_fspen:
            fused = fspen_fold_state_dict(state_dict, self.cfg)
            check_shapes(fused, fspen_expected_fused_shapes(self.cfg), strict=strict)
        elif self.is_bsrnn:
            fused = bsrnn_fold_state_dict(state_dict, self.cfg)
            check_shapes(fused, bsrnn_expected_fused_shapes(self.cfg), strict=strict)
        else:
            fused = fold_state_dict(state_dict, self.cfg)
            check_fused(fused, self.cfg, strict=strict)
        blob = torch.zeros(self.weight_floats, dtype=torch.float32)
        for name, off, cnt in self.sections:
            t = fused[name].contiguous().reshape(-1)
            assert t.numel() == cnt, (name, t.numel(), cnt)
            blob[off:off + cnt] = t
        return blob

    def load_blob(self, blob_dev: Tensor):
        self._require_gpu()
        assert blob_dev.is_cuda and blob_dev.dtype == torch.float32 and blob_dev.is_contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_load_weights(self._h, _ptr(blob_dev), blob_dev.numel(), _stream(self.device)), "fe_load_weights")
        self.loaded = True

    def load_state_dict(self, state_dict: Mapping[str, Tensor], strict: bool = True):
        blob = self.make_blob(state_dict, strict=strict)
        self.load_blob(blob.to(self.device))

    # ------------------------------------------------------------------ state
    def _require_gpu(self):
        if self.device is None or self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.FEError("the FastEnhancer HIP path needs a GPU device (no CPU fallback); got device=%r" % (self.device,))

    def state_floats(self, B: int) -> int:
        return int(self.lib.fe_state_floats(self._h, B))

    def new_state(self, B: int) -> Tensor:
        self._require_gpu()
        return torch.zeros(self.state_floats(B), dtype=torch.float32, device=self.device)

    def split_state(self, state: Tensor, B: int, head0: bool = False) -> List[Tensor]:
        """Views of the opaque state as the reference cache list
        [cache_stft [B,N-H], cache_istft [B,N-H], K x h [1,B*F2,C2]] (scripts/export_onnx.py:43-46)."""
        c = self.cfg
        L = c.cache_len
        out = [state[:B * L].view(B, L), state[B * L:2 * B * L].view(B, L)]
        o = 2 * B * L
        if self.is_lisennet:       # the reference's cache list, each tensor sized for B streams (models/lisennet/model.py:380-396)
            for shp in c.cache_shapes(B):
                n = 1
                for d_ in shp:
                    n *= d_
                out.append(state[o:o + n].view(*shp))
                o += n
            return out
        if self.is_fspen:          # num_blocks * groups inter-GRU states [1, B * freq/groups, C]  (models/fspen/model.py:293-297, :111-116)
            n = B * (c.freq // c.groups) * c.dpe_channels
            for _ in range(c.n_caches):
                out.append(state[o:o + n].view(1, B * (c.freq // c.groups), c.dpe_channels))
                o += n
            return out
        if self.is_bsrnn:          # 2 * num_layers LSTM caches (h, c) of shape [B*31, 2C]  (models/bsrnn/model.py:409-416)
            n = B * c.n_bands * c.hidden
            for _ in range(2 * c.num_layers):
                out.append(state[o:o + n].view(B * c.n_bands, c.hidden))
                o += n
            return out
        if c.dpt:     # K and V caches per block, [B*F2, NH, L, hd] (models/fastenhancer/dptransformer/model.py:194-198)
            # In the state every cache is a ring over its L slots with one head per stream (include/fastenhancer_hip.h,
            # fe_config.lookbehind): the reference's tensors (oldest frame first) are the rings rotated left by head - copies,
            # (gathered on the device without looking at the heads: no device-to-host sync on the per-hop path), or views
            # when the caller knows every head is 0 (head0: a fresh state)
            n = B * c.rf_freq * c.rf_channels * c.lookbehind
            L, hd = c.lookbehind, c.rf_channels // c.rf_heads
            heads = state[o + 2 * c.rf_blocks * n:o + 2 * c.rf_blocks * n + B]
            rot = not head0
            if rot:
                idx = (heads.long()[:, None] + torch.arange(L, device=state.device)[None, :]) % L            # [B, L]
                idx = idx[:, None, None, :, None].expand(B, c.rf_freq, c.rf_heads, L, hd)
            for _ in range(2 * c.rf_blocks):
                t = state[o:o + n].view(B, c.rf_freq, c.rf_heads, L, hd)
                if rot:
                    t = torch.gather(t, 3, idx)
                out.append(t.reshape(B * c.rf_freq, c.rf_heads, L, hd))
                o += n
            return out
        n = B * c.rf_freq * c.rf_channels
        hs = []
        for _ in range(c.rf_blocks):
            hs.append(state[o:o + n].view(1, B * c.rf_freq, c.rf_channels))
            o += n
        if not c.time_kernel:
            return out + hs
        # time_kernel variant: the causal convs' frame caches, kept as [B, kt-1, F1, C1]; the reference tensors
        # (B, C1, kt-1, F1) are permuted views of them, and its cache list is encoder caches, GRU states, decoder caches
        # (models/fastenhancer/time_kernel/model.py:746-754)
        tk = []
        n = B * (c.kernel_size_time - 1) * c.F1 * c.channels
        for _ in range(2 * c.n_layers):
            tk.append(state[o:o + n].view(B, c.kernel_size_time - 1, c.F1, c.channels).permute(0, 3, 1, 2))
            o += n
        return out + tk[:c.n_layers] + hs + tk[c.n_layers:]

    def model_state_order(self, caches: List[Tensor]) -> List[Tensor]:
        """the model's cache list (reference order) -> flat pieces in the order of the C ABI state (h ..., then the conv caches)"""
        c = self.cfg
        if not (self.is_bsrnn or self.is_fspen or self.is_lisennet) and c.dpt:      # reference-order caches = rings with head 0
            B = caches[0].shape[0] // c.rf_freq
            return [t.reshape(-1) for t in caches] + [torch.zeros(B, dtype=torch.float32, device=caches[0].device)]
        if self.is_bsrnn or self.is_fspen or self.is_lisennet or not c.time_kernel:
            return [t.reshape(-1) for t in caches]
        nl, K = c.n_layers, c.rf_blocks
        assert len(caches) == 2 * nl + K, f"expected {2 * nl + K} caches, got {len(caches)}"
        conv = lambda t: t.permute(0, 2, 3, 1).reshape(-1)          # (B, C1, kt-1, F1) -> [B, kt-1, F1, C1]
        return [t.reshape(-1) for t in caches[nl:nl + K]] + [conv(t) for t in caches[:nl]] + [conv(t) for t in caches[nl + K:]]

    def pack_state(self, caches: List[Tensor], B: int) -> Tensor:
        pieces = [t.reshape(-1) for t in caches[:2]] + self.model_state_order(list(caches[2:]))
        return torch.cat([t.to(torch.float32) for t in pieces]).contiguous()

    # ------------------------------------------------------------------ compute
    def step(self, wav_in: Tensor, state: Tensor, wav_out: Optional[Tensor] = None, T: int = 1) -> Tensor:
        """wav_in [B, T*H] (row stride free) -> wav_out [B, T*H]; state updated in place."""
        self._require_gpu()
        B = wav_in.shape[0]
        H = self.cfg.hop_size
        assert wav_in.is_cuda and wav_in.dtype == torch.float32 and wav_in.stride(1) == 1 and wav_in.shape[1] == T * H
        assert state.numel() == self.state_floats(B) and state.is_contiguous()
        if wav_out is None:
            wav_out = torch.empty(B, T * H, dtype=torch.float32, device=wav_in.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_step(self._h, _ptr(wav_in), wav_in.stride(0) if B > 1 else T * H, _ptr(state), _ptr(wav_out),
                                        wav_out.stride(0) if B > 1 else T * H, B, T, _stream(self.device)), "fe_step")
        return wav_out

    def step_host(self, wav_in: Tensor, state: Tensor, wav_out: Optional[Tensor] = None, T: int = 1) -> Tensor:
        """fe_step_host: wav_in [B, n*T*H] in HOST memory (pinned for full speed) -> wav_out [B, n*T*H] in host memory, n calls of T hops
        each with the copies of the neighbouring calls under each kernel; state (device) updated in place.  Asynchronous on the current
        stream: synchronise it before reading wav_out."""
        self._require_gpu()
        B, H = wav_in.shape[0], self.cfg.hop_size
        assert not wav_in.is_cuda and wav_in.dtype == torch.float32 and wav_in.stride(1) == 1 and wav_in.shape[1] % (T * H) == 0
        assert state.numel() == self.state_floats(B) and state.is_contiguous() and state.is_cuda
        n = wav_in.shape[1] // (T * H)
        if wav_out is None:
            wav_out = torch.empty(B, n * T * H, dtype=torch.float32).pin_memory()
        assert not wav_out.is_cuda and wav_out.stride(1) == 1 and wav_out.shape == wav_in.shape
        work = torch.empty(4 * B * T * H, dtype=torch.float32, device=self.device)
        self._host_work = work          # (kept until the next call: the launches are asynchronous)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_step_host(self._h, ctypes.c_void_p(wav_in.data_ptr()), wav_in.stride(0) if B > 1 else n * T * H, _ptr(state),
                                             ctypes.c_void_p(wav_out.data_ptr()), wav_out.stride(0) if B > 1 else n * T * H, B, T, n, _ptr(work),
                                             _stream(self.device)), "fe_step_host")
        return wav_out

    def set_step_kernel(self, kernel: str):
        """fe_set_step_kernel: "waves4" (the 256-thread kernel) | "wg8" (default: the 512-thread per-hop kernel where built) | "wg8_persist"."""
        code = {"waves4": _lib.FE_STEP_KERNEL_WAVES4, "wg8": _lib.FE_STEP_KERNEL_WG8, "wg8_persist": _lib.FE_STEP_KERNEL_WG8_PERSIST}[kernel]
        _lib.check(self.lib.fe_set_step_kernel(self._h, code), "fe_set_step_kernel")

    def set_option(self, name: str, value: int):
        """fe_set_option: the other kernel-selection switches of the handle by name ("bsrnn_role_split", "bsrnn_stream_batch_min",
        "bsrnn_three_launch_step", "bsrnn_ov_profile", "fspen_stream_batch_min", "low_lds_companion", "bsrnn_fused_step", "lisennet_stream_batch_min";
        include/fastenhancer_hip.h)."""
        _lib.check(self.lib.fe_set_option(self._h, name.encode(), int(value)), "fe_set_option")

    def get_option(self, name: str) -> int:
        v = ctypes.c_int(0)
        _lib.check(self.lib.fe_get_option(self._h, name.encode(), ctypes.byref(v)), "fe_get_option")
        return v.value

    def option_names(self):
        return [self.lib.fe_option_name(i).decode() for i in range(self.lib.fe_options())]

    def last_step_kernel(self) -> str:
        """fe_last_step_kernel: what the last compute call of this handle enqueued (kernel families / instantiations + the compiled shape)."""
        return self.lib.fe_last_step_kernel(self._h).decode()

    def set_offline_engine(self, engine: str):
        """fe_set_offline_engine: "auto" | "frame_walk" | "time_batched" (the layer-by-layer engine of csrc/tb_kernels.hip.h)"""
        code = {"auto": _lib.FE_OFFLINE_AUTO, "frame_walk": _lib.FE_OFFLINE_FRAME_WALK, "time_batched": _lib.FE_OFFLINE_TIME_BATCHED}[engine]
        _lib.check(self.lib.fe_set_offline_engine(self._h, code), "fe_set_offline_engine")

    def set_time_pipeline(self, frames_in_flight: int):
        """fe_set_time_pipeline: workgroups per stream in offline / spec launches with T >= 4 (0 = one workgroup per stream)."""
        _lib.check(self.lib.fe_set_time_pipeline(self._h, int(frames_in_flight)), "fe_set_time_pipeline")

    def model_state_floats(self, B: int) -> int:
        """floats of the model's own caches (fe_spec_step's h_dev): the state without the two STFT caches"""
        return self.state_floats(B) - 2 * B * self.cfg.cache_len

    def spec_step(self, spec: Tensor, h: Tensor) -> Tensor:
        """spec [B, N/2+1, T, 2], h = the model caches in C ABI order (K x [B*F2, C2]; time_kernel: + the conv caches),
        updated in place -> spec_hat [B, N/2+1, T, 2]."""
        self._require_gpu()
        B, Fb, T, two = spec.shape
        assert Fb == self.cfg.F0 + 1 and two == 2 and spec.is_contiguous() and h.is_contiguous()
        assert h.numel() == self.model_state_floats(B), (h.numel(), self.model_state_floats(B))
        out = torch.empty_like(spec)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_spec_step(self._h, _ptr(spec), _ptr(h), _ptr(out), B, T, _stream(self.device)), "fe_spec_step")
        return out

    def offline(self, noisy: Tensor) -> Tuple[Tensor, Tensor]:
        """Model.forward (model.py:728-735): noisy [B, Tw] -> (wav_hat [B, H*(Tw//H)], spec_hat [B, N/2, T, 2])."""
        self._require_gpu()
        if noisy.dim() == 3:            # [B, 1, Tw] -> [B, Tw]  (functional/audio_modules.py:73-74)
            noisy = noisy.squeeze(1)
        noisy = noisy.contiguous().float()
        B, Tw = noisy.shape
        cfg = self.cfg
        T = 1 + Tw // cfg.hop_size
        wav = torch.empty(B, cfg.hop_size * (T - 1), dtype=torch.float32, device=noisy.device)
        spec = torch.empty(B, cfg.F0 + (1 if (self.is_bsrnn or self.is_fspen or self.is_lisennet) else 0), T, 2, dtype=torch.float32, device=noisy.device)
        work = torch.empty(int(self.lib.fe_offline_work_floats(self._h, B, Tw)), dtype=torch.float32, device=noisy.device)
        self._last_work = work          # (tools/gpu_tb_check.py looks at the time-batched engine's intermediate buffers)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_offline(self._h, _ptr(noisy), B, Tw, _ptr(wav), _ptr(spec), _ptr(work), _stream(self.device)),
                       "fe_offline")
        return wav, spec

    def offline_ragged(self, noisy: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        """Model.forward over utterances of different lengths in ONE call (fe_offline_ragged; the reference enhances a directory file by
        file, scripts/test_pytorch.py:28-37): noisy = B tensors [Tw_b] (or [1, Tw_b]) -> (B wavs [H * (Tw_b // H)], B specs [F, T_b, 2])."""
        self._require_gpu()
        cfg, dev = self.cfg, self.device
        xs = [t.reshape(-1).to(dev, torch.float32) for t in noisy]
        B = len(xs)
        lens = [int(t.numel()) for t in xs]
        Tw = max(lens)
        Tmax = 1 + Tw // cfg.hop_size
        F = cfg.F0 + (1 if (self.is_bsrnn or self.is_fspen or self.is_lisennet) else 0)
        batch = torch.zeros(B, Tw, dtype=torch.float32, device=dev)
        for b, t in enumerate(xs):
            batch[b, :lens[b]] = t
        n_out = cfg.hop_size * (Tmax - 1)
        wav = torch.zeros(B, n_out, dtype=torch.float32, device=dev)
        spec = torch.empty(B, F, Tmax, 2, dtype=torch.float32, device=dev)
        work = torch.empty(int(self.lib.fe_offline_ragged_work_floats(self._h, B, Tw)), dtype=torch.float32, device=dev)
        self._last_work = work
        lens_c = (ctypes.c_int * B)(*lens)
        with torch.cuda.device(dev):
            _lib.check(self.lib.fe_offline_ragged(self._h, _ptr(batch), Tw, lens_c, B, _ptr(wav), n_out, _ptr(spec), _ptr(work), _stream(dev)),
                       "fe_offline_ragged")
        Tb = [1 + n // cfg.hop_size for n in lens]
        return [wav[b, :cfg.hop_size * (Tb[b] - 1)] for b in range(B)], [spec[b, :, :Tb[b]] for b in range(B)]

    # ------------------------------------------------------------------ stand-alone STFT / iSTFT (the `.stft` modules)
    def stft_step(self, wav_in: Tensor, cache: Tensor) -> Tuple[Tensor, Tensor]:
        """ONNXSTFT.forward: wav_in [B, H], cache [B, N-H] -> (spec [B, N/2+1, 1, 2], cache'); inputs untouched."""
        self._require_gpu()
        B, c = wav_in.shape[0], self.cfg
        wav_in = wav_in.to(self.device, torch.float32)
        cache = cache.to(self.device, torch.float32).contiguous()
        assert wav_in.shape[1] == c.hop_size and wav_in.stride(1) == 1 and tuple(cache.shape) == (B, c.cache_len)
        spec = torch.empty(B, c.n_fft // 2 + 1, 1, 2, dtype=torch.float32, device=self.device)
        cache_out = torch.empty_like(cache)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_stft_step(self._h, _ptr(wav_in), wav_in.stride(0) if B > 1 else c.hop_size, _ptr(cache),
                                             _ptr(cache_out), _ptr(spec), B, _stream(self.device)), "fe_stft_step")
        return spec, cache_out

    def istft_step(self, spec: Tensor, cache: Tensor) -> Tuple[Tensor, Tensor]:
        """ONNXSTFT.inverse: spec [B, N/2+1, 1, 2], cache [B, N-H] -> (wav_out [B, H], cache'); inputs untouched."""
        self._require_gpu()
        B, c = spec.shape[0], self.cfg
        spec = spec.to(self.device, torch.float32).contiguous()
        cache = cache.to(self.device, torch.float32).contiguous()
        assert tuple(spec.shape) == (B, c.n_fft // 2 + 1, 1, 2) and tuple(cache.shape) == (B, c.cache_len)
        wav = torch.empty(B, c.hop_size, dtype=torch.float32, device=self.device)
        cache_out = torch.empty_like(cache)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_istft_step(self._h, _ptr(spec), _ptr(cache), _ptr(cache_out), _ptr(wav), c.hop_size, B,
                                              _stream(self.device)), "fe_istft_step")
        return wav, cache_out

    def stft_offline(self, x: Tensor, discard_last: bool, compress: bool = True) -> Tensor:
        """CompressedSTFT.forward: x [B, Tw] or [B, 1, Tw] -> [B, F, T, 2], T = 1 + Tw // H."""
        self._require_gpu()
        if x.dim() == 3:
            x = x.squeeze(1)
        x = x.to(self.device, torch.float32).contiguous()
        B, Tw = x.shape
        c = self.cfg
        F = c.n_fft // 2 + (0 if discard_last else 1)
        spec = torch.empty(B, F, 1 + Tw // c.hop_size, 2, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_stft_offline(self._h, _ptr(x), B, Tw, F, int(compress), _ptr(spec), _stream(self.device)), "fe_stft_offline")
        return spec

    def istft_offline(self, spec: Tensor, compress: bool = True) -> Tensor:
        """CompressedSTFT.inverse: spec [B, F, T, 2] (compressed domain) -> wav [B, H * (T - 1)]."""
        self._require_gpu()
        spec = spec.to(self.device, torch.float32).contiguous()
        B, F, T, two = spec.shape
        c = self.cfg
        assert two == 2
        wav = torch.empty(B, c.hop_size * (T - 1), dtype=torch.float32, device=self.device)
        frames = torch.empty(B * T * c.n_fft, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_istft_offline(self._h, _ptr(spec), B, T, F, int(compress), _ptr(wav), _ptr(frames),
                                                 _stream(self.device)), "fe_istft_offline")
        return wav

    def poison_lds(self) -> None:
        """fe_debug_poison_lds: NaN into every CU's LDS (test support: what an earlier kernel leaves in LDS must not matter)."""
        self._require_gpu()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_debug_poison_lds(_stream(self.device)), "fe_debug_poison_lds")

    def profile_step(self, wav_in: Tensor, state: Tensor, T: int = 1) -> Tensor:
        """Phase cycle counters (int64[64]) of workgroup 0 for the last frame of the launch."""
        self._require_gpu()
        B = wav_in.shape[0]
        H = self.cfg.hop_size
        clk = torch.zeros(64, dtype=torch.int64, device=wav_in.device)
        out = torch.empty(B, T * H, dtype=torch.float32, device=wav_in.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_profile_step(self._h, _ptr(wav_in), wav_in.stride(0), _ptr(state), _ptr(out), T * H, B, T,
                                                _ptr(clk), _stream(self.device)), "fe_profile_step")
        return clk

    def debug_stages(self) -> List[Tuple[str, int, int, int]]:
        out = []
        for i in range(self.lib.fe_debug_stages(self._h)):
            name, r, c, off = c_char_p(), c_int(), c_int(), c_size_t()
            _lib.check(self.lib.fe_debug_stage(self._h, i, byref(name), byref(r), byref(c), byref(off)), "fe_debug_stage")
            out.append((name.value.decode(), r.value, c.value, int(off.value)))
        return out

    def debug_step(self, wav_in: Tensor, state: Tensor) -> Tuple[Tensor, Dict[str, Tensor]]:
        self._require_gpu()
        B = wav_in.shape[0]
        H = self.cfg.hop_size
        n = int(self.lib.fe_debug_floats(self._h))
        dbg = torch.zeros(B, n, dtype=torch.float32, device=wav_in.device)
        wav_out = torch.empty(B, H, dtype=torch.float32, device=wav_in.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fe_debug_step(self._h, _ptr(wav_in), wav_in.stride(0), _ptr(state), _ptr(wav_out), H, B,
                                              _ptr(dbg), _stream(self.device)), "fe_debug_step")
        taps = {}
        for name, r, c, off in self.debug_stages():
            taps[name] = dbg[:, off:off + r * c].view(B, r, c)
        return wav_out, taps
