"""CPU oracle for the FastEnhancer streaming / offline forward path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker.  The product path (``fastenhancer_amd``) never imports it and has no
CPU fallback.

This is a plain-numpy restatement of the reference algorithm
(aask1357/fastenhancer).  Every function cites the reference file:line it
follows (paths relative to the reference checkout).  The restatement is pinned
against outputs of the imported reference itself: ``tools/gen_golden.py``
(run in the authoring container, where the reference is mounted) dumps
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks this module
against them.  The reference holds no golden vectors of its own,
so those fixtures are the pin.

Layout conventions follow the reference: ``spec`` is ``[B, F, T, 2]`` (re, im
last), model caches are ``[1, B*F2, C2]`` (stream-major), STFT caches ``[B, N-H]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Array = np.ndarray


# --------------------------------------------------------------------------- config
@dataclass
class FEConfig:
    """Mirror of the yaml ``model_kwargs`` of ``fastenhancer.default``
    (reference configs/fastenhancer/b.yaml:2-29; defaults from
    models/fastenhancer/default/model.py:384-403)."""
    channels: int = 64
    kernel_size: Sequence[int] = (8, 3, 3)
    stride: int = 4
    rf_blocks: int = 3
    rf_channels: int = 32
    rf_freq: int = 32
    rf_heads: int = 4
    rf_eps: float = 1e-8
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3
    activation: str = "ReLU"
    mask: Optional[str] = None
    positional_embedding: Optional[str] = "train"
    pre_post_init: Optional[str] = None
    normalize_final_conv: bool = False
    weight_norm: bool = False
    resnet: bool = False
    # models/fastenhancer/time_kernel/model.py (configs/ablation/time_kernel_b.yaml): the encoder / decoder k=3 convs are
    # causal Conv2d with kernel_size_time taps over time (1 = the default model); final_scale "exp": scale.exp()
    kernel_size_time: int = 1
    final_scale_exp: bool = False
    # models/fastenhancer/dprnn/model.py (configs/ablation/dprnn_*.yaml): the blocks' attention is a bidirectional GRU over the
    # sub-band axis with channels_frnn hidden units per direction (0 = the default RNNFormer block); no positional embedding
    channels_frnn: int = 0

    @property
    def dprnn(self) -> bool:
        return self.channels_frnn > 0

    # models/fastenhancer/dptransformer/model.py (configs/ablation/dpt_*.yaml): the blocks' time GRU is a causal attention over
    # the last `lookbehind` frames (K / V caches [B*F2, NH, lookbehind, hd] per block) with a learned positional bias [NH, L+1]
    lookbehind: int = 0

    @property
    def dpt(self) -> bool:
        return self.lookbehind > 0

    # models/fastenhancer/ln/model.py (configs/ablation/ln_b.yaml): GroupNorm(1, C) after every conv (statistics over the
    # channels and sub-bands of a frame) and LayerNorm over (F2, C2) after the blocks' fc layers instead of the BatchNorms -
    # nothing folds into the convs, which carry their own biases
    ln: bool = False
    # models/fastenhancer/noncausal/model.py (configs/fastenhancer_dns/huge_noncausal*.yaml, configs/fastenhancer_48khz/huge_noncausal.yaml):
    # the blocks' time GRU is bidirectional (nn.GRU(C2, C2, bidirectional=True), :186) and rnn_fc maps 2 C2 -> C2 (:187); the
    # module has the offline `Model` only (:348, :628-635) - no caches, no streaming step
    noncausal: bool = False

    @property
    def time_kernel(self) -> bool:
        return self.kernel_size_time > 1

    @staticmethod
    def from_model_kwargs(kw: dict, variant: Optional[str] = None) -> "FEConfig":
        """variant: the last component of the yaml's `model:` key where the kwargs alone do not tell (`ln`)"""
        dp = "dprnn_kwargs" in kw
        dt = "dpt_kwargs" in kw
        rk = dict(kw.get("dprnn_kwargs" if dp else ("dpt_kwargs" if dt else "rnnformer_kwargs"), {}))
        if dt:
            assert "pre_norm" in rk, "DPTConfig.pre_norm defaults to True (dptransformer/model.py:417): not restated"

        for flag in ("attn_bias", "post_act", "pre_norm"):
            assert not rk.get(flag, False), f"rnnformer_kwargs.{flag}=True is not restated"
        assert rk.get("p_dropout", 0.0) == 0.0
        assert not kw.get("stft_normalized", False)
        assert kw.get("window", "hann") == "hann"
        return FEConfig(
            channels=kw.get("channels", 64),
            kernel_size=tuple(kw["kernel_size_freq"]) if "kernel_size_freq" in kw else tuple(kw.get("kernel_size", (8, 3, 3))),
            kernel_size_time=int(kw.get("kernel_size_time", 3)) if "kernel_size_freq" in kw else 1,
            final_scale_exp=(("kernel_size_freq" in kw or dp or dt or variant == "ln") and kw.get("final_scale", "exp") == "exp"),
            ln=(variant == "ln"),
            noncausal=(variant == "noncausal"),
            lookbehind=int(rk.get("lookbehind", 16)) if dt else 0,
            channels_frnn=int(rk.get("channels_frnn", 16)) if dp else 0,
            stride=kw.get("stride", 4),
            rf_blocks=rk.get("num_blocks", 3),
            rf_channels=rk.get("channels", 32),
            rf_freq=rk.get("freq", 32),
            rf_heads=rk.get("num_heads", 4),
            rf_eps=rk.get("eps", 1e-5 if dp else 1e-8),
            n_fft=kw.get("n_fft", 512),
            hop_size=kw.get("hop_size", 256),
            win_size=kw.get("win_size", 512),
            input_compression=kw.get("input_compression", 0.3),
            activation=kw.get("activation", "ReLU"),
            mask=kw.get("mask", None),
            positional_embedding=None if dp else rk.get("positional_embedding", "train"),
            pre_post_init=kw.get("pre_post_init", None),
            normalize_final_conv=kw.get("normalize_final_conv", False),
            weight_norm=kw.get("weight_norm", False),
            resnet=kw.get("resnet", False),
        )

    # derived sizes (vocabulary of models/fastenhancer/default/macs.py:10-15)
    @property
    def F0(self) -> int:
        return self.n_fft // 2

    @property
    def F1(self) -> int:
        return self.n_fft // 2 // self.stride

    @property
    def n_layers(self) -> int:
        return len(self.kernel_size) - 1

    def macs_per_frame(self) -> int:
        """models/fastenhancer/default/macs.py:17-87 with T=1."""
        C1, C2, F1, F2, K = self.channels, self.rf_channels, self.F1, self.rf_freq, self.rf_blocks
        k0 = self.kernel_size[0]
        m = 2 * C1 * k0 * F1
        for k in self.kernel_size[1:]:
            m += C1 * C1 * k * self.kernel_size_time * F1
        m += F1 * F2 * C1 + C1 * C2 * F2
        if self.dpt:        # time attention (qkv, L+1 scores and weighted values per token) + fc, then the sub-band attention + fc
            m += K * (C2 * C2 * 3 * F2 + 2 * (self.lookbehind + 1) * C2 * F2 + C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2)
        elif self.dprnn:    # time GRU + fc, then the BiGRU over the F2 sub-bands (input and hidden products of both directions) + fc
            H = self.channels_frnn
            m += K * (C2 * C2 * 6 * F2 + C2 * C2 * F2 + 2 * 3 * H * (C2 + H) * F2 + 2 * H * C2 * F2)
        elif self.noncausal:   # (no macs.py in the reference for this variant: the default model's count with both GRU directions and rnn_fc over 2 C2)
            m += K * (2 * C2 * C2 * 6 * F2 + 2 * C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2)
        else:
            m += K * (C2 * C2 * 6 * F2 + C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2)
        m += F2 * F1 * C2 + C2 * C1 * F1
        for k in self.kernel_size[1:]:
            m += 2 * C1 * C1 * F1 + C1 * C1 * k * self.kernel_size_time * F1
        m += 2 * C1 * C1 * F1 + C1 * 2 * k0 * F1
        return m

    def flops_per_frame(self) -> float:
        """2*MACs + rfft+irfft (2 * 2.5 * N * log2 N)."""
        return 2.0 * self.macs_per_frame() + 2 * 2.5 * self.n_fft * math.log2(self.n_fft)


# --------------------------------------------------------------------------- windows
def hann_window(n: int, dtype=np.float64) -> Array:
    """torch.hann_window(n) (periodic=True), functional/audio_modules.py:211-214."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def stft_windows(n_fft: int, hop: int, win_size: Optional[int] = None, dtype=np.float32) -> Tuple[Array, Array]:
    """Analysis window and streaming synthesis window.
    functional/audio_modules.py:207-235 (ONNXSTFT.__init__).  The reference builds
    them in float32; we follow the same operation order in float32."""
    assert n_fft % 2 == 0
    if win_size is None:
        win_size = n_fft
    assert n_fft >= win_size
    # torch.hann_window is evaluated in the requested dtype; evaluate in f64 then round:
    # differences are <= 1 ulp and are covered by the golden check.
    window = hann_window(win_size, np.float64).astype(np.float32)
    if win_size < n_fft:
        pad = n_fft - win_size
        window = np.pad(window, (pad // 2, pad - pad // 2))
    K = (n_fft + hop - 1) // hop
    L = hop * (2 * K - 1) + (n_fft - hop)
    win_sq = window.astype(np.float32) ** 2
    acc = np.zeros(L, dtype=np.float32)
    for j in range(2 * K - 1):  # F.fold == overlap-add of 2K-1 shifted copies
        acc[j * hop: j * hop + n_fft] += win_sq
    win_sq_sum = acc[(K - 1) * hop:(K - 1) * hop + n_fft]
    window_istft = window / win_sq_sum
    return window.astype(dtype), window_istft.astype(dtype)


# --------------------------------------------------------------------------- small ops
def silu(x: Array) -> Array:
    return x / (1.0 + np.exp(-x))


def sigmoid(x: Array) -> Array:
    return 1.0 / (1.0 + np.exp(-x))


def conv1d(x: Array, w: Array, b: Optional[Array], pad: int) -> Array:
    """F.conv1d, stride 1.  x [M,Ci,L], w [Co,Ci,k] -> [M,Co,L+2*pad-k+1]."""
    M, Ci, L = x.shape
    Co, Ci2, k = w.shape
    assert Ci == Ci2
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    Lo = x.shape[2] - k + 1
    out = np.zeros((M, Co, Lo), dtype=x.dtype)
    for t in range(k):
        out += np.einsum("oc,mcl->mol", w[:, :, t], x[:, :, t:t + Lo], optimize=True)
    if b is not None:
        out += b[None, :, None]
    return out


def strided_conv1d(x: Array, w: Array, b: Array, stride: int, pad: int) -> Array:
    """StridedConv1d.forward, models/fastenhancer/default/model.py:51-59.
    x [M,C,L]; weight is stored in the reshaped form (Co, C*stride, k/stride)."""
    x = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    M, C, L = x.shape
    x = x.reshape(M, C, L // stride, stride).transpose(0, 3, 1, 2).reshape(M, C * stride, L // stride)
    return conv1d(x, w, b, 0)


def conv_transpose1d(x: Array, w: Array, b: Array, stride: int, pad: int) -> Array:
    """F.conv_transpose1d (models/fastenhancer/default/model.py:91-95).
    x [M,Ci,L], w [Ci,Co,k] -> [M,Co,(L-1)*stride-2*pad+k]."""
    M, Ci, L = x.shape
    Ci2, Co, k = w.shape
    full = np.zeros((M, Co, (L - 1) * stride + k), dtype=x.dtype)
    for j in range(k):
        full[:, :, j:j + (L - 1) * stride + 1:stride] += np.einsum("mcl,co->mol", x, w[:, :, j], optimize=True)
    out = full[:, :, pad:full.shape[2] - pad]
    return out + b[None, :, None]


def gru_step(x: Array, h: Array, w_ih: Array, w_hh: Array, b_ih: Array, b_hh: Array) -> Array:
    """One time step of nn.GRU (gate order r,z,n), used by
    models/fastenhancer/default/model.py:187,271.  x,h [M,C]."""
    C = h.shape[1]
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = sigmoid(gi[:, :C] + gh[:, :C])
    z = sigmoid(gi[:, C:2 * C] + gh[:, C:2 * C])
    n = np.tanh(gi[:, 2 * C:] + r * gh[:, 2 * C:])
    return (1.0 - z) * n + z * h


def mhsa(x: Array, w_qkv: Array, num_heads: int) -> Array:
    """Attention.forward, models/fastenhancer/default/model.py:142-152.
    x [M,F,C] -> [M,F,C]; qkv rows are per-head interleaved [head][q|k|v][hd]."""
    M, F, C = x.shape
    hd = C // num_heads
    qkv = x @ w_qkv.T                                           # [M,F,3C]
    qkv = qkv.reshape(M, F, num_heads, 3 * hd).transpose(0, 2, 1, 3)  # [M,NH,F,3hd]
    q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
    s = (q @ k.transpose(0, 1, 3, 2)) * (hd ** -0.5)            # [M,NH,F,F]
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    o = p @ v                                                   # [M,NH,F,hd]
    return o.transpose(0, 2, 1, 3).reshape(M, F, C)


def group_norm1(x: Array, g: Array, b: Array, eps: float = 1e-5) -> Array:
    """nn.GroupNorm(1, C) on x [M,C,F] (models/fastenhancer/ln/model.py:427 ...): per sample, statistics over (C, F), biased
    variance, per-channel affine."""
    m = x.mean(axis=(1, 2), keepdims=True)
    d = x - m
    v = (d * d).mean(axis=(1, 2), keepdims=True)
    return d / np.sqrt(v + x.dtype.type(eps)) * g[None, :, None] + b[None, :, None]


def layer_norm_fc(x: Array, g: Array, b: Array, eps: float) -> Array:
    """LayerNorm of models/fastenhancer/ln/model.py:16-37 on x [..., F, C], statistics over (F, C) - AS WRITTEN there:
    `w = inv_std.mul(self.weight); x = diff.addcmul(w, self.bias)`, and torch.addcmul(input, t1, t2) = input + t1 * t2, so the
    output is  (x - mean) + inv_std * weight * bias  - the centred input is NOT scaled by inv_std * weight.  Restated as is
    (the deployed checkpoints were trained with it)."""
    m = x.mean(axis=(-2, -1), keepdims=True)
    d = x - m
    v = (d * d).mean(axis=(-2, -1), keepdims=True)
    return d + (g / np.sqrt(v + x.dtype.type(eps))) * b


def causal_time_attention(x: Array, w_qkv: Array, pe: Array, num_heads: int, lookbehind: int,
                          h_k: Optional[Array], h_v: Optional[Array]) -> Tuple[Array, Array, Array]:
    """CausalAttention.forward, models/fastenhancer/dptransformer/model.py:200-236.  x [M,T,C] (M = B*F2), pe [NH,L+1] (column L
    = the current frame), caches [M,NH,L,hd] or None.  With caches every frame attends to the L cached frames and itself
    (zero-initialised caches take part: score = pe, value 0); without, frames before the start are masked out
    (expand_attn_map, :151-171).  The reference's cached branch is written for T = 1; T > 1 here is T such steps."""
    M, T, C = x.shape
    L, hd = lookbehind, C // num_heads
    scale = x.dtype.type(hd ** -0.5)
    qkv = (x @ w_qkv.T).reshape(M, T, num_heads, 3 * hd).transpose(0, 2, 1, 3)          # [M,NH,T,3hd]
    q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
    masked = h_k is None
    if masked:
        h_k = np.zeros((M, num_heads, L, hd), x.dtype)
        h_v = np.zeros((M, num_heads, L, hd), x.dtype)
    kk = np.concatenate([h_k, k], axis=2)                                                 # [M,NH,L+T,hd]
    vv = np.concatenate([h_v, v], axis=2)
    out = np.empty((M, num_heads, T, hd), x.dtype)
    for t in range(T):
        kw, vw = kk[:, :, t:t + L + 1], vv[:, :, t:t + L + 1]                             # the window of frame t
        a = pe[None] + scale * np.einsum("mnd,mnjd->mnj", q[:, :, t], kw)                 # [M,NH,L+1]
        if masked and t < L:
            a[:, :, :L - t] = -np.inf
        a = a - a.max(axis=2, keepdims=True)
        e = np.exp(a)
        a = e / e.sum(axis=2, keepdims=True)
        out[:, :, t] = np.einsum("mnj,mnjd->mnd", a, vw)
    keep = min(L, T) if masked else L       # (without caches the reference returns k[:, :, -L:] of the T frames it saw)
    return out.transpose(0, 2, 1, 3).reshape(M, T, C), kk[:, :, -keep:].copy(), vv[:, :, -keep:].copy()


def bigru_over_rows(x: Array, w: Dict[str, Array], p: str) -> Array:
    """nn.GRU(C, H, bidirectional=True, batch_first=True) over axis 1 with zero initial state
    (DPRNN.forward, models/fastenhancer/dprnn/model.py:239-241).  x [M,F,C] -> [M,F,2H] (forward | reverse)."""
    M, F, _ = x.shape
    H = w[p + "weight_hh_l0"].shape[1]
    out = np.empty((M, F, 2 * H), x.dtype)
    for d, sfx in enumerate(("", "_reverse")):
        h = np.zeros((M, H), x.dtype)
        for f in (range(F) if d == 0 else range(F - 1, -1, -1)):
            h = gru_step(x[:, f], h, w[p + "weight_ih_l0" + sfx], w[p + "weight_hh_l0" + sfx], w[p + "bias_ih_l0" + sfx], w[p + "bias_hh_l0" + sfx])
            out[:, f, d * H:(d + 1) * H] = h
    return out


# --------------------------------------------------------------------------- weight transform
def linear_filterbank(n_freq: int, n_filter: int) -> Tuple[Array, Array]:
    """rf_pre_post_lin with init 'linear*', models/fastenhancer/default/model.py:308-380.
    Returns (pre [n_filter,n_freq], post [n_freq,n_filter]) in float32."""
    f32 = np.float32
    delta = f32((n_freq - 1) / (n_filter - 1))
    f_filter = np.linspace(0, n_freq - 1, n_filter, dtype=np.float64).astype(f32)
    f_freqs = np.linspace(0, n_freq - 1, n_freq, dtype=np.float64).astype(f32)
    down = (f_filter[1:, None] - f_freqs[None, :]) / delta
    up = (f_freqs[None, :] - f_filter[:-1, None]) / delta
    down = np.concatenate([down, np.ones((1, n_freq), f32)], axis=0)
    up = np.concatenate([np.ones((1, n_freq), f32), up], axis=0)
    pre = np.maximum(f32(0), np.minimum(down, up)).astype(f32)
    pre = pre / pre.sum(axis=1, keepdims=True)
    post = pre.T
    post = post / post.sum(axis=1, keepdims=True)
    return np.ascontiguousarray(pre), np.ascontiguousarray(post)


def linear_filterbank_tk(n_freq: int, n_filter: int, sr: int = 16000) -> Tuple[Array, Array]:
    """rf_pre_post_lin with init 'linear*' of the time_kernel variant, models/fastenhancer/time_kernel/model.py:477-489:
    centres and bins on Hz axes, slope 1 / (sr/2 / n_filter) (not the default model's (n_freq-1)/(n_filter-1) bins);
    post = pre^T taken BEFORE pre is normalised, each then row-normalised."""
    f32 = np.float32
    half = sr // 2
    f_filter = np.linspace(0, half, n_filter, dtype=np.float64).astype(f32)
    delta_f = f32(half / n_filter)
    f_freqs = np.linspace(0, half, n_freq, dtype=np.float64).astype(f32)
    down = (f_filter[1:, None] - f_freqs[None, :]) / delta_f
    down = np.concatenate([down, np.ones((1, n_freq), f32)], axis=0)
    up = (f_freqs[None, :] - f_filter[:-1, None]) / delta_f
    up = np.concatenate([np.ones((1, n_freq), f32), up], axis=0)
    pre = np.maximum(f32(0), np.minimum(down, up)).astype(f32)
    post = pre.T
    pre = pre / pre.sum(axis=1, keepdims=True)
    post = post / post.sum(axis=1, keepdims=True)
    return np.ascontiguousarray(pre), np.ascontiguousarray(post)


def causal_conv2d(x: Array, w: Array, b: Array, cache: Optional[Array]) -> Tuple[Array, Array]:
    """CausalConv2d.forward, models/fastenhancer/time_kernel/model.py:119-148: x [B,C,T,F], w [Co,Ci,kt,kf], cache
    [B,Ci,kt-1,F] (None = zeros) -> (out [B,Co,T,F], cache_out = the last kt-1 frames of cat(cache, x))."""
    B, Ci, T, F = x.shape
    Co, _, kt, kf = w.shape
    if cache is None:
        cache = np.zeros((B, Ci, kt - 1, F), x.dtype)
    xc = np.concatenate([cache.astype(x.dtype), x], axis=2)            # [B,Ci,T+kt-1,F]
    out = np.zeros((B, Co, T, F), x.dtype)
    for dt in range(kt):
        xs = xc[:, :, dt:dt + T].transpose(0, 2, 1, 3).reshape(B * T, Ci, F)
        out += conv1d(xs, w[:, :, dt, :], None, (kf - 1) // 2).reshape(B, T, Co, F).transpose(0, 2, 1, 3)
    out += b[None, :, None, None]
    return out, xc[:, :, -(kt - 1):].copy()


def positional_embedding(channels: int, freq: int) -> Array:
    """calculate_positional_embedding, models/fastenhancer/default/model.py:98-110."""
    f = np.arange(1, freq + 1, dtype=np.float32) * np.float32(math.pi / freq)
    c = np.exp(np.linspace(math.log(1), math.log(freq - 1), channels // 2, dtype=np.float64).astype(np.float32))
    grid = f[:, None] * c[None, :]
    return np.concatenate([np.sin(grid), np.cos(grid)], axis=1).astype(np.float32)


def _bn_scale_shift(sd: Dict[str, Array], prefix: str, eps: float) -> Tuple[Array, Array]:
    std = np.sqrt(sd[prefix + ".running_var"].astype(np.float32) + np.float32(eps))
    g = sd[prefix + ".weight"] / std
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * sd[prefix + ".weight"] / std
    return g.astype(np.float32), shift.astype(np.float32)


def _weight_norm(g: Array, v: Array) -> Array:
    """torch weight_norm parametrization (dim=0): w = g * v / ||v||_2 per output row."""
    norm = np.sqrt((v.astype(np.float32) ** 2).reshape(v.shape[0], -1).sum(axis=1)).reshape((-1,) + (1,) * (v.ndim - 1))
    return (v * (g.reshape(norm.shape) / norm)).astype(np.float32)


# models/fastenhancer/dprnn/model.py:412-436 module names <-> this file's (rf_pre / rf_block / rf_post, rnn = trnn)
_DPRNN_NAMES = (("dprnn_pre.", "rf_pre."), ("dprnn_post.", "rf_post."), ("dprnn_block.", "rf_block."), (".trnn_fc.", ".rnn_fc."),
                (".trnn_post_norm.", ".rnn_post_norm."), (".trnn.", ".rnn."))


# models/fastenhancer/dptransformer/model.py:580-610 module names <-> this file's (the model-level `pe` is `time_pe` here)
_DPT_NAMES = (("dpt_pre.", "rf_pre."), ("dpt_post.", "rf_post."), ("dpt_block.", "rf_block."), (".time_fc.", ".rnn_fc."),
              (".time_post_norm.", ".rnn_post_norm."), (".freq_attn.", ".attn."), (".freq_fc.", ".attn_fc."),
              (".freq_post_norm.", ".attn_post_norm."))


def canonical_key(k: str) -> str:
    if k == "pe":
        return "time_pe"
    for a, b in _DPRNN_NAMES + _DPT_NAMES:
        k = k.replace(a, b)
    return k


def reference_key(k: str, cfg: "FEConfig") -> str:
    """this file's state_dict key -> the reference module's (they differ for the dprnn / dptransformer variants only)"""
    if cfg.dprnn:
        for a, b in _DPRNN_NAMES:
            k = k.replace(b, a)
    if cfg.dpt:
        if k == "time_pe":
            return "pe"
        for a, b in _DPT_NAMES:
            k = k.replace(b, a)
    return k


def fold_state_dict(sd: Dict[str, Array], cfg: FEConfig) -> Dict[str, Array]:
    """ONNXModel.remove_weight_reparameterizations,
    models/fastenhancer/default/model.py:532-608 (+ :215-231 for the RNNFormer
    blocks and :74-81 for the final transposed conv).  Accepts the training-form
    state_dict and returns the fused-form dict.  A dict that
    is already fused (has 'enc_pre.0.bias') is returned unchanged (as float32)."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    if cfg.dprnn or cfg.dpt:
        sd = {canonical_key(k): v for k, v in sd.items()}
    if cfg.ln:
        return _fold_state_dict_ln(sd, cfg)
    if "enc_pre.0.bias" in sd:
        return {k: v.astype(np.float32) for k, v in sd.items() if v.dtype.kind == "f"}
    out: Dict[str, Array] = {}
    bn_eps = 1e-5  # nn.BatchNorm1d default (convs); RNNFormer norms use cfg.rf_eps

    def conv_bn(conv: str, bn: str, dst: str):
        g, shift = _bn_scale_shift(sd, bn, bn_eps)
        w = sd[conv + ".weight"].astype(np.float32)
        out[dst + ".weight"] = w * g.reshape((-1,) + (1,) * (w.ndim - 1))
        out[dst + ".bias"] = shift

    conv_bn("enc_pre.0", "enc_pre.1", "enc_pre.0")
    for i in range(cfg.n_layers):
        conv_bn(f"encoder.{i}.0", f"encoder.{i}.1", f"encoder.{i}.0")
    out["rf_pre.0.weight"] = sd["rf_pre.0.weight"].astype(np.float32)
    conv_bn("rf_pre.1", "rf_pre.2", "rf_pre.1")
    if cfg.dpt:
        out["time_pe"] = sd["time_pe"].astype(np.float32)
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if p + "pe" in sd:
            out[p + "pe"] = sd[p + "pe"].astype(np.float32)
        if cfg.dpt:        # DPTBlock.remove_weight_reparameterizations, dptransformer/model.py:323-336
            key0 = p + "time_attn.qkv.parametrizations.weight.original0"
            if key0 in sd:
                out[p + "time_attn.qkv.weight"] = _weight_norm(sd[key0], sd[p + "time_attn.qkv.parametrizations.weight.original1"])
            else:
                out[p + "time_attn.qkv.weight"] = sd[p + "time_attn.qkv.weight"].astype(np.float32)
        else:
            for sfx in ("", "_reverse") if cfg.noncausal else ("",):      # noncausal/model.py:215-222
                for name in ("weight_ih_l0" + sfx, "weight_hh_l0" + sfx):
                    key0 = p + f"rnn.parametrizations.{name}.original0"
                    if key0 in sd:
                        out[p + "rnn." + name] = _weight_norm(sd[key0], sd[p + f"rnn.parametrizations.{name}.original1"])
                    else:
                        out[p + "rnn." + name] = sd[p + "rnn." + name].astype(np.float32)
                out[p + "rnn.bias_ih_l0" + sfx] = sd[p + "rnn.bias_ih_l0" + sfx].astype(np.float32)
                out[p + "rnn.bias_hh_l0" + sfx] = sd[p + "rnn.bias_hh_l0" + sfx].astype(np.float32)
        if cfg.dprnn:      # DPRNN.remove_weight_reparameterizations, dprnn/model.py:172-192
            for name in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse"):
                key0 = p + f"frnn.parametrizations.{name}.original0"
                if key0 in sd:
                    out[p + "frnn." + name] = _weight_norm(sd[key0], sd[p + f"frnn.parametrizations.{name}.original1"])
                else:
                    out[p + "frnn." + name] = sd[p + "frnn." + name].astype(np.float32)
            for name in ("bias_ih_l0", "bias_hh_l0", "bias_ih_l0_reverse", "bias_hh_l0_reverse"):
                out[p + "frnn." + name] = sd[p + "frnn." + name].astype(np.float32)
        else:
            key0 = p + "attn.qkv.parametrizations.weight.original0"
            if key0 in sd:
                out[p + "attn.qkv.weight"] = _weight_norm(sd[key0], sd[p + "attn.qkv.parametrizations.weight.original1"])
            else:
                out[p + "attn.qkv.weight"] = sd[p + "attn.qkv.weight"].astype(np.float32)
        for fc, norm in (("rnn_fc", "rnn_post_norm"), ("frnn_fc", "frnn_post_norm")) if cfg.dprnn else (("rnn_fc", "rnn_post_norm"), ("attn_fc", "attn_post_norm")):
            std = np.sqrt(sd[p + norm + ".running_var"].astype(np.float32) + np.float32(cfg.rf_eps))
            g = sd[p + norm + ".weight"] / std
            out[p + fc + ".weight"] = (sd[p + fc + ".weight"] * g.reshape(-1, 1)).astype(np.float32)
            out[p + fc + ".bias"] = (sd[p + norm + ".bias"] - sd[p + norm + ".running_mean"] * sd[p + norm + ".weight"] / std).astype(np.float32)
    out["rf_post.0.weight"] = sd["rf_post.0.weight"].astype(np.float32)
    conv_bn("rf_post.1", "rf_post.2", "rf_post.1")
    for i in range(cfg.n_layers):
        conv_bn(f"decoder.{i}.0", f"decoder.{i}.1", f"decoder.{i}.0")
        conv_bn(f"decoder.{i}.3", f"decoder.{i}.4", f"decoder.{i}.2")
    conv_bn("dec_post.0", "dec_post.1", "dec_post.0")
    w = sd["dec_post.3.weight"].astype(np.float32)
    scale = sd["dec_post.3.scale"].astype(np.float32) if "dec_post.3.scale" in sd else np.ones(1, np.float32)
    if cfg.final_scale_exp:               # time_kernel/model.py:95,105 (exp_scale)
        scale = np.exp(scale)
    if cfg.normalize_final_conv:
        # F.normalize(w, dim=(0,1,2)): w / max(||w||_2, 1e-12)
        w = w / max(float(np.sqrt((w.astype(np.float32) ** 2).sum())), 1e-12)
    out["dec_post.2.weight"] = (w * scale).astype(np.float32)
    out["dec_post.2.bias"] = sd["dec_post.3.bias"].astype(np.float32)
    return out


def _fold_state_dict_ln(sd: Dict[str, Array], cfg: FEConfig) -> Dict[str, Array]:
    """ONNXModel.remove_weight_reparameterizations of the ln variant (models/fastenhancer/ln/model.py:524-533, 239-245, 116-135):
    only the weight norms of the GRU / qkv matrices and the final conv's normalisation + scale go away; every norm layer stays.
    Fused keys = the training keys, except the final conv: dec_post.3.{weight,scale,bias} -> dec_post.2.{weight,bias}."""
    if "dec_post.2.weight" in sd:
        return {k: v.astype(np.float32) for k, v in sd.items() if v.dtype.kind == "f"}
    out: Dict[str, Array] = {}
    for k, v in sd.items():
        if v.dtype.kind != "f" or ".parametrizations." in k or k.startswith("dec_post.3."):
            continue
        out[k] = v.astype(np.float32)
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        for mod, name in (("rnn", "weight_ih_l0"), ("rnn", "weight_hh_l0"), ("attn.qkv", "weight")):
            key0 = p + f"{mod}.parametrizations.{name}.original0"
            if key0 in sd:
                out[p + f"{mod}.{name}"] = _weight_norm(sd[key0], sd[p + f"{mod}.parametrizations.{name}.original1"])
    w = sd["dec_post.3.weight"].astype(np.float32)
    scale = sd["dec_post.3.scale"].astype(np.float32) if "dec_post.3.scale" in sd else np.ones(1, np.float32)
    if cfg.final_scale_exp:
        scale = np.exp(scale)
    if cfg.normalize_final_conv:
        w = w / max(float(np.sqrt((w.astype(np.float32) ** 2).sum())), 1e-12)
    out["dec_post.2.weight"] = (w * scale).astype(np.float32)
    out["dec_post.2.bias"] = sd["dec_post.3.bias"].astype(np.float32)
    return out


def _training_state_dict_spec_ln(cfg: FEConfig) -> Dict[str, Tuple[int, ...]]:
    C1, C2, F1, F2, S = cfg.channels, cfg.rf_channels, cfg.F1, cfg.rf_freq, cfg.stride
    spec: Dict[str, Tuple[int, ...]] = {}

    def wb(prefix, wshape, bias=True):
        spec[prefix + ".weight"] = wshape
        if bias:
            spec[prefix + ".bias"] = (wshape[0],)

    wb("enc_pre.0", (C1, 2 * S, cfg.kernel_size[0] // S)); wb("enc_pre.1", (C1,))
    for i in range(cfg.n_layers):
        wb(f"encoder.{i}.0", (C1, C1, cfg.kernel_size[i + 1])); wb(f"encoder.{i}.1", (C1,))
    spec["rf_pre.0.weight"] = (F2, F1)
    wb("rf_pre.1", (C2, C1, 1)); wb("rf_pre.2", (C2,))
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if k == 0 and cfg.positional_embedding is not None:
            spec[p + "pe"] = (F2, C2)
        spec[p + "rnn.bias_ih_l0"] = (3 * C2,)
        spec[p + "rnn.bias_hh_l0"] = (3 * C2,)
        for name in ("weight_ih_l0", "weight_hh_l0"):
            if cfg.weight_norm:
                spec[p + f"rnn.parametrizations.{name}.original0"] = (3 * C2, 1)
                spec[p + f"rnn.parametrizations.{name}.original1"] = (3 * C2, C2)
            else:
                spec[p + "rnn." + name] = (3 * C2, C2)
        spec[p + "rnn_fc.weight"] = (C2, C2)
        wb(p + "rnn_post_norm", (C2,))
        if cfg.weight_norm:
            spec[p + "attn.qkv.parametrizations.weight.original0"] = (3 * C2, 1)
            spec[p + "attn.qkv.parametrizations.weight.original1"] = (3 * C2, C2)
        else:
            spec[p + "attn.qkv.weight"] = (3 * C2, C2)
        spec[p + "attn_fc.weight"] = (C2, C2)
        wb(p + "attn_post_norm", (C2,))
    spec["rf_post.0.weight"] = (F1, F2)
    wb("rf_post.1", (C1, C2, 1)); wb("rf_post.2", (C1,))
    for i in range(cfg.n_layers):
        wb(f"decoder.{i}.0", (C1, 2 * C1, 1)); wb(f"decoder.{i}.1", (C1,))
        wb(f"decoder.{i}.3", (C1, C1, cfg.kernel_size[cfg.n_layers - i]), bias=False); wb(f"decoder.{i}.4", (C1,))
    wb("dec_post.0", (C1, 2 * C1, 1), bias=False); wb("dec_post.1", (C1,))
    spec["dec_post.3.weight"] = (C1, 2, cfg.kernel_size[0])
    spec["dec_post.3.bias"] = (2,)
    spec["dec_post.3.scale"] = (1,)
    return spec


def training_state_dict_spec(cfg: FEConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the training-form checkpoint, in
    the reference's state_dict order.  Pinned against the imported reference by
    tools/gen_golden.py."""
    if cfg.ln:
        return _training_state_dict_spec_ln(cfg)
    C1, C2, F1, F2, S = cfg.channels, cfg.rf_channels, cfg.F1, cfg.rf_freq, cfg.stride
    spec: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix: str, c: int):
        spec[prefix + ".weight"] = (c,)
        spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    if cfg.dpt:          # the model's own parameter precedes its submodules in state_dict()
        spec["time_pe"] = (cfg.rf_heads, cfg.lookbehind + 1)
    spec["enc_pre.0.weight"] = (C1, 2 * S, cfg.kernel_size[0] // S)
    bn("enc_pre.1", C1)
    tk = cfg.time_kernel
    one = (1, 1) if tk else (1,)          # the time_kernel variant's 1x1 convs are Conv2d
    for i in range(cfg.n_layers):
        spec[f"encoder.{i}.0.weight"] = (C1, C1, cfg.kernel_size_time, cfg.kernel_size[i + 1]) if tk else (C1, C1, cfg.kernel_size[i + 1])
        bn(f"encoder.{i}.1", C1)
    spec["rf_pre.0.weight"] = (F2, F1)
    spec["rf_pre.1.weight"] = (C2, C1) + one
    bn("rf_pre.2", C2)
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if k == 0 and cfg.positional_embedding is not None:
            spec[p + "pe"] = (F2, C2)
        if cfg.dpt:
            if cfg.weight_norm:
                spec[p + "time_attn.qkv.parametrizations.weight.original0"] = (3 * C2, 1)
                spec[p + "time_attn.qkv.parametrizations.weight.original1"] = (3 * C2, C2)
            else:
                spec[p + "time_attn.qkv.weight"] = (3 * C2, C2)
        else:
            sfxs = ("", "_reverse") if cfg.noncausal else ("",)       # nn.GRU registers the biases first, then the (parametrized) weights
            for sfx in sfxs:
                spec[p + "rnn.bias_ih_l0" + sfx] = (3 * C2,)
                spec[p + "rnn.bias_hh_l0" + sfx] = (3 * C2,)
            for sfx in sfxs:
                if cfg.weight_norm:
                    spec[p + f"rnn.parametrizations.weight_ih_l0{sfx}.original0"] = (3 * C2, 1)
                    spec[p + f"rnn.parametrizations.weight_ih_l0{sfx}.original1"] = (3 * C2, C2)
                    spec[p + f"rnn.parametrizations.weight_hh_l0{sfx}.original0"] = (3 * C2, 1)
                    spec[p + f"rnn.parametrizations.weight_hh_l0{sfx}.original1"] = (3 * C2, C2)
                else:
                    spec[p + "rnn.weight_ih_l0" + sfx] = (3 * C2, C2)
                    spec[p + "rnn.weight_hh_l0" + sfx] = (3 * C2, C2)
        spec[p + "rnn_fc.weight"] = (C2, 2 * C2 if cfg.noncausal else C2)
        bn(p + "rnn_post_norm", C2)
        if cfg.dprnn:      # nn.GRU(bidirectional) registers biases first, then the (parametrized) weights, like the time GRU
            H = cfg.channels_frnn
            for sfx in ("", "_reverse"):
                spec[p + "frnn.bias_ih_l0" + sfx] = (3 * H,)
                spec[p + "frnn.bias_hh_l0" + sfx] = (3 * H,)
            for name, cols in (("weight_ih_l0", C2), ("weight_hh_l0", H), ("weight_ih_l0_reverse", C2), ("weight_hh_l0_reverse", H)):
                if cfg.weight_norm:
                    spec[p + f"frnn.parametrizations.{name}.original0"] = (3 * H, 1)
                    spec[p + f"frnn.parametrizations.{name}.original1"] = (3 * H, cols)
                else:
                    spec[p + "frnn." + name] = (3 * H, cols)
            spec[p + "frnn_fc.weight"] = (C2, 2 * H)
            bn(p + "frnn_post_norm", C2)
            continue
        if cfg.weight_norm:
            spec[p + "attn.qkv.parametrizations.weight.original0"] = (3 * C2, 1)
            spec[p + "attn.qkv.parametrizations.weight.original1"] = (3 * C2, C2)
        else:
            spec[p + "attn.qkv.weight"] = (3 * C2, C2)
        spec[p + "attn_fc.weight"] = (C2, C2)
        bn(p + "attn_post_norm", C2)
    spec["rf_post.0.weight"] = (F1, F2)
    spec["rf_post.1.weight"] = (C1, C2) + one
    bn("rf_post.2", C1)
    for i in range(cfg.n_layers):
        spec[f"decoder.{i}.0.weight"] = (C1, 2 * C1) + one
        bn(f"decoder.{i}.1", C1)
        kf = cfg.kernel_size[cfg.n_layers - i]
        spec[f"decoder.{i}.3.weight"] = (C1, C1, cfg.kernel_size_time, kf) if tk else (C1, C1, kf)
        bn(f"decoder.{i}.4", C1)
    spec["dec_post.0.weight"] = (C1, 2 * C1, 1)
    bn("dec_post.1", C1)
    spec["dec_post.3.weight"] = (C1, 2, cfg.kernel_size[0])
    spec["dec_post.3.bias"] = (2,)
    spec["dec_post.3.scale"] = (1,)
    return spec


# --------------------------------------------------------------------------- the model
class FEOracle:
    """Restatement of ONNXModel (streaming, spec->spec and wav->wav) and Model
    (offline wav->wav) of models/fastenhancer/default/model.py on fused weights."""

    def __init__(self, cfg: FEConfig, fused: Dict[str, Array], dtype=np.float32):
        assert cfg.activation == "SiLU", "only SiLU is restated (all shipped yamls)"
        assert cfg.mask is None, "only mask: null is restated (all shipped yamls)"
        assert not cfg.resnet
        assert cfg.kernel_size[0] % cfg.stride == 0 and (cfg.kernel_size[0] - cfg.stride) % 2 == 0
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in fused.items()}
        win, win_i = stft_windows(cfg.n_fft, cfg.hop_size, cfg.win_size, np.float32)
        self.window = win.astype(dtype)
        self.window_istft = win_i.astype(dtype)

    # ---- caches: ONNXSTFT.initialize_cache (functional/audio_modules.py:238-241)
    #      + ONNXModel.initialize_cache (model.py:614-618, sized for B streams)
    def initialize_cache(self, B: int) -> List[Array]:
        c = self.cfg
        caches = [np.zeros((B, c.n_fft - c.hop_size), self.dtype), np.zeros((B, c.n_fft - c.hop_size), self.dtype)]
        hs = [np.zeros((1, B * c.rf_freq, c.rf_channels), self.dtype) for _ in range(c.rf_blocks)]
        if c.dpt:      # DPTBlock.initialize_cache (dptransformer/model.py:194-198, sized for B streams): h_k, h_v per block
            hs = [np.zeros((B * c.rf_freq, c.rf_heads, c.lookbehind, c.rf_channels // c.rf_heads), self.dtype) for _ in range(2 * c.rf_blocks)]
        if c.time_kernel:     # (B, C1, kt-1, F1) per causal conv (time_kernel/model.py:138-139, sized for B streams)
            tkc = lambda: [np.zeros((B, c.channels, c.kernel_size_time - 1, c.F1), self.dtype) for _ in range(c.n_layers)]
            return caches + tkc() + hs + tkc()
        caches += hs
        return caches

    # ---- a3: ONNXSTFT.forward (functional/audio_modules.py:243-257)
    def stft_step(self, wav_in: Array, cache: Array) -> Tuple[Array, Array]:
        c = self.cfg
        x = np.concatenate([cache, wav_in.astype(self.dtype)], axis=1)   # [B,N]
        cache = x[:, -(c.n_fft - c.hop_size):].copy() if c.n_fft > c.hop_size else x[:, :0].copy()
        X = np.fft.rfft(x * self.window, axis=1)                          # [B,N/2+1]
        spec = np.stack([X.real, X.imag], axis=-1).astype(self.dtype)[:, :, None, :]
        return spec, cache

    # ---- a18: ONNXSTFT.inverse (functional/audio_modules.py:259-303); the ifft+correction
    #      form there is mathematically irfft(Y, N).
    def istft_step(self, spec: Array, cache: Array) -> Tuple[Array, Array]:
        c = self.cfg
        Y = spec[:, :, 0, 0] + 1j * spec[:, :, 0, 1]
        Y[:, 0] = Y[:, 0].real        # the reference formula ignores Im X[0], Im X[N/2]
        Y[:, -1] = Y[:, -1].real
        x = np.fft.irfft(Y, n=c.n_fft, axis=1).astype(self.dtype)
        x = x * self.window_istft
        L = c.n_fft - c.hop_size
        x[:, :L] += cache
        return x[:, :c.hop_size].copy(), x[:, c.hop_size:].copy()

    # ---- a5..a16: ONNXModel.model_forward (model.py:620-675)
    def _model_forward_ln(self, spec: Array, h_list: Optional[List[Array]], taps: Optional[dict]) -> Tuple[Array, List[Array]]:
        """ONNXModel.model_forward of the ln variant (models/fastenhancer/ln/model.py:546-602): conv + bias -> GroupNorm(1, C)
        -> SiLU everywhere the default model has conv -> BatchNorm -> SiLU, LayerNorm over (F2, C2) after the blocks' fc layers."""
        c, w = self.cfg, self.w
        B, F0, T, _ = spec.shape
        C2, F2 = c.rf_channels, c.rf_freq
        tap = (lambda k, v: taps.__setitem__(k, v.copy())) if taps is not None else (lambda k, v: None)
        gn = lambda x, key: group_norm1(x, w[key + ".weight"], w[key + ".bias"])
        bias = lambda key: w[key + ".bias"] if (key + ".bias") in w else None
        cv = lambda x, key: conv1d(x, w[key + ".weight"], bias(key), (w[key + ".weight"].shape[-1] - 1) // 2)
        x = spec.transpose(0, 2, 3, 1).reshape(B * T, 2, F0)
        pad0 = (c.kernel_size[0] - c.stride) // 2
        x = silu(gn(strided_conv1d(x, w["enc_pre.0.weight"], w["enc_pre.0.bias"], c.stride, pad0), "enc_pre.1"))
        enc_outs = [x]
        tap("enc_pre", x)
        for i in range(c.n_layers):
            x = silu(gn(cv(x, f"encoder.{i}.0"), f"encoder.{i}.1"))
            enc_outs.append(x)
            tap(f"encoder.{i}", x)
        x = x @ w["rf_pre.0.weight"].T
        x = gn(cv(x, "rf_pre.1"), "rf_pre.2")
        x = np.ascontiguousarray(x.reshape(B, T, C2, F2).transpose(1, 0, 3, 2))          # [T,B,F2,C2]
        tap("rf_pre", x)
        h_out = []
        for k in range(c.rf_blocks):
            p = f"rf_block.{k}."
            h = np.zeros((B * F2, C2), self.dtype) if h_list is None else h_list[k][0].astype(self.dtype).copy()
            xs = x.reshape(T, B * F2, C2)
            ys = np.empty_like(xs)
            for t in range(T):
                h = gru_step(xs[t], h, w[p + "rnn.weight_ih_l0"], w[p + "rnn.weight_hh_l0"], w[p + "rnn.bias_ih_l0"], w[p + "rnn.bias_hh_l0"])
                ys[t] = h
            h_out.append(h[None].copy())
            y = (ys @ w[p + "rnn_fc.weight"].T).reshape(T, B, F2, C2)
            x = layer_norm_fc(y, w[p + "rnn_post_norm.weight"], w[p + "rnn_post_norm.bias"], c.rf_eps) + x
            if (p + "pe") in w:
                x = x + w[p + "pe"]
            tap(f"rf_block.{k}.rnn", x)
            a = mhsa(x.reshape(T * B, F2, C2), w[p + "attn.qkv.weight"], c.rf_heads) @ w[p + "attn_fc.weight"].T
            x = layer_norm_fc(a.reshape(T, B, F2, C2), w[p + "attn_post_norm.weight"], w[p + "attn_post_norm.bias"], c.rf_eps) + x
            tap(f"rf_block.{k}", x)
        x = x.transpose(1, 0, 3, 2).reshape(B * T, C2, F2)
        x = x @ w["rf_post.0.weight"].T
        x = gn(cv(x, "rf_post.1"), "rf_post.2")
        tap("rf_post", x)
        for i in range(c.n_layers):
            x = np.concatenate([x, enc_outs.pop(-1)], axis=1)
            x = silu(gn(cv(x, f"decoder.{i}.0"), f"decoder.{i}.1"))
            x = silu(gn(cv(x, f"decoder.{i}.3"), f"decoder.{i}.4"))
            tap(f"decoder.{i}", x)
        x = np.concatenate([x, enc_outs.pop(-1)], axis=1)
        x = silu(gn(cv(x, "dec_post.0"), "dec_post.1"))
        x = conv_transpose1d(x, w["dec_post.2.weight"], w["dec_post.2.bias"], c.stride, pad0)
        mask = np.ascontiguousarray(x.reshape(B, T, 2, F0).transpose(0, 3, 1, 2))
        tap("mask", mask)
        return mask, h_out

    def model_forward(self, spec: Array, h_list: Optional[List[Array]], taps: Optional[dict] = None
                      ) -> Tuple[Array, List[Array]]:
        if self.cfg.ln:
            return self._model_forward_ln(spec, h_list, taps)
        c, w = self.cfg, self.w
        B, F0, T, _ = spec.shape
        tk = c.time_kernel
        nl = c.n_layers
        C1, F1 = c.channels, c.F1
        # time_kernel: the cache list is [encoder caches, GRU states, decoder caches] (time_kernel/model.py:746-754)
        enc_c = dec_c = None
        if tk and h_list is not None:
            assert len(h_list) == 2 * nl + c.rf_blocks
            enc_c, dec_c = h_list[:nl], h_list[nl + c.rf_blocks:]
            h_list = h_list[nl:nl + c.rf_blocks]
        enc_c_out, dec_c_out = [], []
        w1 = (lambda key: w[key][:, :, 0, 0][:, :, None]) if tk else (lambda key: w[key])     # Conv2d 1x1 -> Conv1d form

        def k3(x2, key, cache):       # x2 [B*T,C1,F1] -> same; causal time taps when tk
            if not tk:
                kf = w[key + ".weight"].shape[-1]
                return conv1d(x2, w[key + ".weight"], w[key + ".bias"], (kf - 1) // 2), None
            x4 = x2.reshape(B, T, C1, F1).transpose(0, 2, 1, 3)
            y4, cache_out = causal_conv2d(x4, w[key + ".weight"], w[key + ".bias"], cache)
            return y4.transpose(0, 2, 1, 3).reshape(B * T, C1, F1), cache_out
        x = spec.transpose(0, 2, 3, 1).reshape(B * T, 2, F0)
        pad0 = (c.kernel_size[0] - c.stride) // 2
        x = silu(strided_conv1d(x, w["enc_pre.0.weight"], w["enc_pre.0.bias"], c.stride, pad0))
        enc_outs = [x]
        if taps is not None:
            taps["enc_pre"] = x.copy()
        for i in range(c.n_layers):
            x, co = k3(x, f"encoder.{i}.0", None if enc_c is None else enc_c[i])
            x = silu(x)
            enc_c_out.append(co)
            enc_outs.append(x)
            if taps is not None:
                taps[f"encoder.{i}"] = x.copy()
        # rf_pre: Linear over the freq axis then 1x1 conv (model.py:458-465, :646)
        x = x @ w["rf_pre.0.weight"].T                                   # [BT,C1,F2]
        x = conv1d(x, w1("rf_pre.1.weight"), w["rf_pre.1.bias"], 0)      # [BT,C2,F2]
        C2, F2 = c.rf_channels, c.rf_freq
        x = x.reshape(B, T, C2, F2).transpose(1, 0, 3, 2)                # [T,B,F2,C2]
        x = np.ascontiguousarray(x)
        if taps is not None:
            taps["rf_pre"] = x.copy()
        h_out = []
        for k in range(c.rf_blocks):
            p = f"rf_block.{k}."
            if c.dpt:
                # time attention per sub-band (dptransformer/model.py:378-389), batch index b*F2+f
                xs = x.transpose(1, 2, 0, 3).reshape(B * F2, T, C2)
                hk, hv = (None, None) if h_list is None else (h_list[2 * k].astype(self.dtype), h_list[2 * k + 1].astype(self.dtype))
                ys, hk, hv = causal_time_attention(xs, w[p + "time_attn.qkv.weight"], w["time_pe"], c.rf_heads, c.lookbehind, hk, hv)
                ys = ys.reshape(B, F2, T, C2).transpose(2, 0, 1, 3).reshape(T, B * F2, C2)
                h_out += [hk, hv]
            elif c.noncausal:
                # bidirectional GRU over time, zero initial states (noncausal/model.py:186, 266-272): output = cat(forward, reverse)
                assert h_list is None, "the noncausal model has no caches"
                xs = x.reshape(T, B * F2, C2)
                ys = np.empty((T, B * F2, 2 * C2), self.dtype)
                for d, sfx in enumerate(("", "_reverse")):
                    h = np.zeros((B * F2, C2), self.dtype)
                    for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                        h = gru_step(xs[t], h, w[p + "rnn.weight_ih_l0" + sfx], w[p + "rnn.weight_hh_l0" + sfx],
                                     w[p + "rnn.bias_ih_l0" + sfx], w[p + "rnn.bias_hh_l0" + sfx])
                        ys[t, :, d * C2:(d + 1) * C2] = h
            else:
                h = np.zeros((B * F2, C2), self.dtype) if h_list is None else h_list[k][0].astype(self.dtype).copy()
                # GRU over time (model.py:266-277), batch index b*F2+f
                xs = x.reshape(T, B * F2, C2)
                ys = np.empty_like(xs)
                for t in range(T):
                    h = gru_step(xs[t], h, w[p + "rnn.weight_ih_l0"], w[p + "rnn.weight_hh_l0"],
                                 w[p + "rnn.bias_ih_l0"], w[p + "rnn.bias_hh_l0"])
                    ys[t] = h
                h_out.append(h[None].copy())
            y = ys @ w[p + "rnn_fc.weight"].T + w[p + "rnn_fc.bias"]
            x = y.reshape(T, B, F2, C2) + x
            if (p + "pe") in w:                                          # model.py:279-280
                x = x + w[p + "pe"]
            if taps is not None:
                taps[f"rf_block.{k}.rnn"] = x.copy()
            if c.dprnn:
                a = bigru_over_rows(x.reshape(T * B, F2, C2), w, p + "frnn.")
                a = a @ w[p + "frnn_fc.weight"].T + w[p + "frnn_fc.bias"]
            else:
                a = mhsa(x.reshape(T * B, F2, C2), w[p + "attn.qkv.weight"], c.rf_heads)
                a = a @ w[p + "attn_fc.weight"].T + w[p + "attn_fc.bias"]
            x = a.reshape(T, B, F2, C2) + x
            if taps is not None:
                taps[f"rf_block.{k}"] = x.copy()
        x = x.transpose(1, 0, 3, 2).reshape(B * T, C2, F2)
        x = x @ w["rf_post.0.weight"].T                                  # [BT,C2,F1]
        x = conv1d(x, w1("rf_post.1.weight"), w["rf_post.1.bias"], 0)    # [BT,C1,F1]
        if taps is not None:
            taps["rf_post"] = x.copy()
        for i in range(c.n_layers):
            x = np.concatenate([x, enc_outs.pop(-1)], axis=1)
            x = silu(conv1d(x, w1(f"decoder.{i}.0.weight"), w[f"decoder.{i}.0.bias"], 0))
            x, co = k3(x, f"decoder.{i}.2", None if dec_c is None else dec_c[i])
            x = silu(x)
            dec_c_out.append(co)
            if taps is not None:
                taps[f"decoder.{i}"] = x.copy()
        x = np.concatenate([x, enc_outs.pop(-1)], axis=1)
        x = silu(conv1d(x, w["dec_post.0.weight"], w["dec_post.0.bias"], 0))
        x = conv_transpose1d(x, w["dec_post.2.weight"], w["dec_post.2.bias"], c.stride, pad0)
        mask = x.reshape(B, T, 2, F0).transpose(0, 3, 1, 2)              # [B,F0,T,2]
        if taps is not None:
            taps["mask"] = mask.copy()
        if tk:
            h_out = enc_c_out + h_out + dec_c_out
        return np.ascontiguousarray(mask), h_out

    # ---- a4, a17: ONNXModel.forward (model.py:677-710)
    def spec_forward(self, spec: Array, h_list: Optional[List[Array]], taps: Optional[dict] = None
                     ) -> Tuple[Array, List[Array]]:
        c = self.cfg
        x = spec[:, :-1].astype(self.dtype)
        mag = np.maximum(np.sqrt(x[..., 0:1] ** 2 + x[..., 1:2] ** 2), self.dtype(1e-5))
        x = x * mag ** self.dtype(c.input_compression - 1.0)
        if taps is not None:
            taps["compressed"] = x.copy()
        mask, h_out = self.model_forward(x, h_list, taps)
        y = np.stack([x[..., 0] * mask[..., 0] - x[..., 1] * mask[..., 1],
                      x[..., 0] * mask[..., 1] + x[..., 1] * mask[..., 0]], axis=3)
        mag2 = np.sqrt(y[..., 0:1] ** 2 + y[..., 1:2] ** 2)
        y = y * mag2 ** self.dtype(1.0 / c.input_compression - 1.0)
        y = np.pad(y, ((0, 0), (0, 1), (0, 0), (0, 0)))
        return y.astype(self.dtype), h_out

    # ---- a19: scripts/export_onnx.py:48-58 (the wav->wav streaming step)
    def step(self, wav_in: Array, cache_stft: Array, cache_istft: Array, *cache_model: Array,
             taps: Optional[dict] = None):
        spec_in, cache_stft = self.stft_step(wav_in, cache_stft)
        if taps is not None:
            taps["spec_in"] = spec_in.copy()
        spec_out, h_out = self.spec_forward(spec_in, list(cache_model), taps)
        if taps is not None:
            taps["spec_out"] = spec_out.copy()
        wav_out, cache_istft = self.istft_step(spec_out, cache_istft)
        return (wav_out, cache_stft, cache_istft, *h_out)

    # ---- a26: driver loop of scripts/test_onnx.py:11-60
    def enhance_stream(self, wav: Array) -> Array:
        """wav [B,L] -> enhanced [B,L] (latency-compensated, clipped)."""
        c = self.cfg
        wav = np.clip(np.asarray(wav, self.dtype), -1, 1)
        B, length = wav.shape
        wav = np.pad(wav, ((0, 0), (0, c.n_fft)))
        caches = self.initialize_cache(B)
        outs = []
        for idx in range(0, length + c.n_fft - c.hop_size, c.hop_size):
            o, *caches = self.step(wav[:, idx:idx + c.hop_size], *caches)
            outs.append(o)
        out = np.concatenate(outs, axis=1)
        s = c.n_fft - c.hop_size
        return np.clip(out[:, s:s + length], -1.0, 1.0)

    # ---- a21: Model.forward (model.py:728-735) with CompressedSTFT
    #      (functional/audio_modules.py:70-164): torch.stft(center=True, reflect) / torch.istft
    def offline_forward(self, noisy: Array) -> Tuple[Array, Array]:
        c = self.cfg
        N, H = c.n_fft, c.hop_size
        x = np.asarray(noisy, self.dtype)
        if x.ndim == 3:
            x = x[:, 0]
        B, Tw = x.shape
        xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
        T = 1 + Tw // H
        frames = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * self.window  # [B,T,N]
        X = np.fft.rfft(frames, axis=2)                                                       # [B,T,N/2+1]
        spec = np.stack([X.real, X.imag], axis=-1).astype(self.dtype).transpose(0, 2, 1, 3)   # [B,F,T,2]
        spec = spec[:, :-1]
        mag = np.maximum(np.sqrt(spec[..., 0:1] ** 2 + spec[..., 1:2] ** 2), self.dtype(1e-5))
        spec = spec * mag ** self.dtype(c.input_compression - 1.0)
        mask, _ = self.model_forward(spec, None)
        y = np.stack([spec[..., 0] * mask[..., 0] - spec[..., 1] * mask[..., 1],
                      spec[..., 0] * mask[..., 1] + spec[..., 1] * mask[..., 0]], axis=3)
        spec_hat = y.astype(self.dtype)
        mag2 = np.sqrt(y[..., 0:1] ** 2 + y[..., 1:2] ** 2)
        yu = y * mag2 ** self.dtype(1.0 / c.input_compression - 1.0)
        Y = np.pad(yu[..., 0] + 1j * yu[..., 1], ((0, 0), (0, 1), (0, 0)))                    # [B,F0+1,T]
        fr = np.fft.irfft(Y.transpose(0, 2, 1), n=N, axis=2).astype(self.dtype) * self.window  # [B,T,N]
        full = np.zeros((B, (T - 1) * H + N), self.dtype)
        env = np.zeros((T - 1) * H + N, self.dtype)
        wsq = self.window ** 2
        for t in range(T):
            full[:, t * H:t * H + N] += fr[:, t]
            env[t * H:t * H + N] += wsq
        sl = slice(N // 2, N // 2 + H * (T - 1))
        wav = full[:, sl] / env[sl]
        return wav.astype(self.dtype), spec_hat


# --------------------------------------------------------------------------- metrics
def si_sdr(clean: Array, enhanced: Array, mask: Optional[Array] = None, eps: float = 1e-7) -> Array:
    """scripts/metrics_ns.py:38-52 (si_snr(s1 = enhanced, s2 = clean, mask)): SI-SDR in dB, no mean subtraction, eps 1e-7,
    fp32 like its inputs; the function does not mask the signals (its caller does, :128,134) and line 52's masked mean of
    the per-utterance constant returns that constant.  Pinned on the reference function's outputs (tests/golden/si_snr.npz)."""
    clean = np.asarray(clean, np.float32)
    enhanced = np.asarray(enhanced, np.float32)
    if mask is None:
        mask = np.ones_like(clean)
    mask = np.asarray(mask, np.float32)
    eps = np.float32(eps)
    alpha = (enhanced * clean).sum(-1, keepdims=True, dtype=np.float32) / ((clean * clean).sum(-1, keepdims=True, dtype=np.float32) + eps)
    target = alpha * clean
    noise = enhanced - target
    snr = np.log10((target * target).sum(-1, keepdims=True, dtype=np.float32) / ((noise * noise).sum(-1, keepdims=True, dtype=np.float32) + eps) + eps)
    return (np.float32(10.0) * (snr * mask).sum(1, dtype=np.float32) / mask.sum(1, dtype=np.float32)).astype(np.float32)
