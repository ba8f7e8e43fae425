"""CPU oracle for the LiSenNet baseline model (TEST INFRASTRUCTURE ONLY — same rules as oracle/fe_oracle.py).

numpy restatement of models/lisennet/model.py of the reference (streaming ``ONNXModel`` with its 9 caches for n_blocks = 2,
and offline ``Model``), each function citing the file:line it follows.  Pinned on outputs of the imported reference
(tools/gen_golden.py -> tests/golden/lisennet.npz, tests/test_oracle_golden.py).

NB the reference's two paths do not compute the same phase features: ``ONNXModel.cal_gd / cal_ifd`` (:357-378) take
``previous - current`` (padded copy minus x), ``Model.cal_gd / cal_ifd`` (:491-510) take ``torch.diff`` = ``current - previous``.
Both are restated as they are."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .fe_oracle import sigmoid, stft_windows

Array = np.ndarray


@dataclass
class LiSenNetConfig:
    """yaml model_kwargs of `model: lisennet` (configs/others/lisennet.yaml:2-8; defaults models/lisennet/model.py:313-323)."""
    num_channels: int = 16
    n_blocks: int = 2
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @staticmethod
    def from_model_kwargs(kw: dict) -> "LiSenNetConfig":
        assert kw.get("window", "hann") == "hann" and not kw.get("normalized", False)
        return LiSenNetConfig(num_channels=kw.get("num_channels", 16), n_blocks=kw.get("n_blocks", 2), n_fft=kw.get("n_fft", 512),
                              hop_size=kw.get("hop_size", 256), win_size=kw.get("win_size", 512),
                              input_compression=kw.get("input_compression", 0.3))

    @property
    def n_freqs(self) -> int:
        return self.n_fft // 2 + 1

    @property
    def hidden(self) -> int:                 # DPR hidden_dim (models/lisennet/model.py:335)
        return self.num_channels // 2 * 3

    @property
    def n_caches(self) -> int:               # ifd + 3 encoder + 2 per block + decoder (:380-396)
        return 1 + 3 + 2 * self.n_blocks + 1

    def enc_channels(self) -> Tuple[int, int, int, int]:
        C = self.num_channels
        return C // 4, C // 2, C // 4 * 3, C

    def cache_shapes(self, B: int) -> List[Tuple[int, ...]]:
        """ONNXModel.initialize_cache (:380-396) sized for B streams (the reference's are written for 1)."""
        c1, c2, c3, C = self.enc_channels()
        F = self.n_freqs
        sh = [(B, 1, F), (B, c1, 1, F), (B, c2, 1, F // 2), (B, c3, 1, F // 4)]
        for _ in range(self.n_blocks):
            sh += [(1, B * (F // 8), self.hidden), (B, 2 * C, 2, F // 8)]
        sh.append((B, c1, 1, F - 1))
        return sh

    def macs_per_frame(self) -> int:
        """models/lisennet/macs.py:8-66 with T = 1"""
        C, N, F1 = self.num_channels, self.n_blocks, self.n_freqs
        m = 3 * (C // 4) * F1
        for c_out, f in zip((C // 2, C // 4 * 3, C), (257, 128, 64)):
            f_hi = (f - f // 4 + 2 - 5) // 3 + 1
            m += (2 * 3 * (f // 4) + 2 * 5 * f_hi) * c_out * c_out
        gru = lambda i, h: (i + h) * h * 3 + h * 3
        h, f = 24, 32
        for _ in range(N):
            m += (gru(C, h // 2) * 2 + h * C + gru(C, h) + h * C) * f
            m += (C * C * 4 + C * 2 * 3 + C * 2 + C * 2 * C) * f
        c_in, f = C, 32
        for c_out in (C // 4 * 3, C // 2, C // 4):
            m += (3 * (f // 2) + 3 * 3 * (f // 2)) * c_in * 2 * c_out
            c_in, f = c_out, f * 2
        f += 1
        m += (c_out * 2 * 2 * 2 + 2 * 2 + 2 * 2) * f
        return int(m)

    def flops_per_frame(self) -> float:
        import math
        return 2.0 * self.macs_per_frame() + 2 * 2.5 * self.n_fft * math.log2(self.n_fft)


def state_dict_spec(cfg: LiSenNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the checkpoint (module order of ONNXModel.__init__; there is nothing to fold: remove_weight_reparameterizations
    is a no-op, :476-477)."""
    c1, c2, c3, C = cfg.enc_channels()
    F, H = cfg.n_freqs, cfg.hidden
    sp: Dict[str, Tuple[int, ...]] = {}

    def dsconv(p, cin, cout, nf):
        sp[p + ".low_conv.weight"] = (cout, cin, 2, 3)
        sp[p + ".low_conv.bias"] = (cout,)
        sp[p + ".high_conv.weight"] = (cout, cin, 2, 5)
        sp[p + ".high_conv.bias"] = (cout,)
        sp[p + ".norm.gamma"] = (1, 1, 1, nf // 2)
        sp[p + ".norm.beta"] = (1, 1, 1, nf // 2)
        sp[p + ".act.weight"] = (cout,)

    def gru(p, i, h, bi):
        for sfx in ("", "_reverse") if bi else ("",):
            sp[f"{p}.weight_ih_l0{sfx}"] = (3 * h, i)
            sp[f"{p}.weight_hh_l0{sfx}"] = (3 * h, h)
            sp[f"{p}.bias_ih_l0{sfx}"] = (3 * h,)
            sp[f"{p}.bias_hh_l0{sfx}"] = (3 * h,)

    sp["encoder.conv_1.0.weight"] = (c1, 3, 1, 1)
    sp["encoder.conv_1.0.bias"] = (c1,)
    sp["encoder.conv_1.1.gamma"] = (1, 1, 1, F)
    sp["encoder.conv_1.1.beta"] = (1, 1, 1, F)
    sp["encoder.conv_1.2.weight"] = (c1,)
    dsconv("encoder.conv_2", c1, c2, F)
    dsconv("encoder.conv_3", c2, c3, F // 2)
    dsconv("encoder.conv_4", c3, C, F // 4)
    nf = F // 8
    for b in range(cfg.n_blocks):
        p = f"blocks.{b}."
        sp[p + "dp_rnn_attn.intra_norm.weight"] = (nf, C)
        sp[p + "dp_rnn_attn.intra_norm.bias"] = (nf, C)
        gru(p + "dp_rnn_attn.intra_rnn_attn.rnn", C, H // 2, True)
        sp[p + "dp_rnn_attn.intra_rnn_attn.dense.weight"] = (C, H)
        sp[p + "dp_rnn_attn.intra_rnn_attn.dense.bias"] = (C,)
        sp[p + "dp_rnn_attn.inter_norm.weight"] = (nf, C)
        sp[p + "dp_rnn_attn.inter_norm.bias"] = (nf, C)
        gru(p + "dp_rnn_attn.inter_rnn_attn.rnn", C, H, False)
        sp[p + "dp_rnn_attn.inter_rnn_attn.dense.weight"] = (C, H)
        sp[p + "dp_rnn_attn.inter_rnn_attn.dense.bias"] = (C,)
        sp[p + "conv_glu.norm.gamma"] = (1, C, 1, nf)
        sp[p + "conv_glu.norm.beta"] = (1, C, 1, nf)
        sp[p + "conv_glu.fc1.weight"] = (4 * C, C, 1, 1)
        sp[p + "conv_glu.fc1.bias"] = (4 * C,)
        sp[p + "conv_glu.dwconv.weight"] = (2 * C, 1, 3, 3)
        sp[p + "conv_glu.dwconv.bias"] = (2 * C,)
        sp[p + "conv_glu.fc2.weight"] = (C, 2 * C, 1, 1)
        sp[p + "conv_glu.fc2.bias"] = (C,)
    for i, (cin, cout) in enumerate(((2 * C, c3), (2 * c3, c2), (2 * c2, c1))):
        p = f"decoder.up{i + 1}."
        sp[p + "low_conv.weight"] = (cout, cin, 1, 3)
        sp[p + "low_conv.bias"] = (cout,)
        sp[p + "high_conv.conv.weight"] = (3 * cout, cin, 1, 3)
        sp[p + "high_conv.conv.bias"] = (3 * cout,)
    sp["decoder.mask_conv.0.weight"] = (2, c1, 2, 2)
    sp["decoder.mask_conv.0.bias"] = (2,)
    sp["decoder.mask_conv.1.gamma"] = (1, 1, 1, F)
    sp["decoder.mask_conv.1.beta"] = (1, 1, 1, F)
    sp["decoder.mask_conv.2.weight"] = (2,)
    sp["decoder.mask_conv.3.weight"] = (2, 2, 1, 1)
    sp["decoder.mask_conv.3.bias"] = (2,)
    sp["decoder.lsigmoid.slope"] = (F, 1, 1)
    return sp


def gru_cell(x: Array, h: Array, w_ih: Array, w_hh: Array, b_ih: Array, b_hh: Array) -> Array:
    """one step of nn.GRU, gate order r, z, n"""
    Hh = h.shape[-1]
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = sigmoid(gi[..., :Hh] + gh[..., :Hh])
    z = sigmoid(gi[..., Hh:2 * Hh] + gh[..., Hh:2 * Hh])
    n = np.tanh(gi[..., 2 * Hh:] + r * gh[..., 2 * Hh:])
    return (1 - z) * n + z * h


def conv2d(x: Array, w: Array, b: Optional[Array], stride_f: int = 1, pad_f: int = 0, groups: int = 1) -> Array:
    """nn.Conv2d on [B, C, T, F] with kernel (kt, kf), stride (1, stride_f), padding (0, pad_f): no padding over time (the callers
    prepend the cached frames)"""
    B, C, T, F = x.shape
    O, Cg, kt, kf = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (0, 0), (pad_f, pad_f)))
    To, Fo = T - kt + 1, (F + 2 * pad_f - kf) // stride_f + 1
    y = np.zeros((B, O, To, Fo), x.dtype)
    og = O // groups
    for g in range(groups):
        xs = xp[:, g * Cg:(g + 1) * Cg]
        for dt in range(kt):
            for df in range(kf):
                patch = xs[:, :, dt:dt + To, df:df + stride_f * (Fo - 1) + 1:stride_f]          # [B, Cg, To, Fo]
                y[:, g * og:(g + 1) * og] += np.einsum("bctf,oc->botf", patch, w[g * og:(g + 1) * og, :, dt, df])
    return y if b is None else y + b[None, :, None, None]


def custom_ln(x: Array, gamma: Array, beta: Array, eps=1e-5) -> Array:
    """CustomLayerNorm with stat_dims (1, 3) (models/lisennet/model.py:27-37): statistics over (channel, freq) per (b, t)"""
    mu = x.mean(axis=(1, 3), keepdims=True)
    std = np.sqrt(x.var(axis=(1, 3), keepdims=True) + x.dtype.type(eps))
    return (x - mu) / std * gamma + beta


def prelu(x: Array, w: Array) -> Array:
    return np.where(x >= 0, x, x * w[None, :, None, None])


def layer_norm_fd(x: Array, w: Array, b: Array, eps=1e-5) -> Array:
    """nn.LayerNorm((n_freqs, emb_dim)) on [..., F, D]"""
    mu = x.mean(axis=(-2, -1), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(-2, -1), keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * w + b


def mish(x: Array) -> Array:
    return x * np.tanh(np.logaddexp(0, x))


def wrap(x: Array) -> Array:
    return np.arctan2(np.sin(x), np.cos(x))


class LiSenNetOracle:
    def __init__(self, cfg: LiSenNetConfig, sd: Dict[str, Array], dtype=np.float32):
        self.cfg, self.dtype = cfg, dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
        win, win_i = stft_windows(cfg.n_fft, cfg.hop_size, cfg.win_size, np.float32)
        self.window, self.window_istft = win.astype(dtype), win_i.astype(dtype)

    def initialize_cache(self, B: int) -> List[Array]:
        c = self.cfg
        caches = [np.zeros((B, c.n_fft - c.hop_size), self.dtype), np.zeros((B, c.n_fft - c.hop_size), self.dtype)]
        return caches + [np.zeros(s, self.dtype) for s in c.cache_shapes(B)]

    # ---- DSConv.forward (:190-208)
    def dsconv(self, p: str, x: Array, cache: Optional[Array]) -> Tuple[Array, Array]:
        w = self.w
        x = np.concatenate([np.zeros_like(x[:, :, :1]) if cache is None else cache.astype(self.dtype), x], axis=2)
        cache_out = x[:, :, -1:].copy()
        lowf = x.shape[3] // 4
        lo = conv2d(x[..., :lowf], w[p + ".low_conv.weight"], w[p + ".low_conv.bias"], 1, 1)
        hi = conv2d(x[..., lowf:], w[p + ".high_conv.weight"], w[p + ".high_conv.bias"], 3, 1)
        y = np.concatenate([lo, hi], axis=3)
        y = prelu(custom_ln(y, w[p + ".norm.gamma"], w[p + ".norm.beta"]), w[p + ".act.weight"])
        return y, cache_out

    # ---- USConv.forward (:218-226) with SPConvTranspose2d (:240-246)
    def usconv(self, p: str, x: Array) -> Array:
        w = self.w
        lowf = x.shape[3] // 2
        lo = conv2d(x[..., :lowf], w[p + "low_conv.weight"], w[p + "low_conv.bias"], 1, 1)
        hi = conv2d(x[..., lowf:], w[p + "high_conv.conv.weight"], w[p + "high_conv.conv.bias"], 1, 1)      # [B, 3*Cout, T, W]
        B, n, T, W = hi.shape
        hi = hi.reshape(B, 3, n // 3, T, W).transpose(0, 2, 3, 4, 1).reshape(B, n // 3, T, W * 3)
        return np.concatenate([lo, hi], axis=3)

    # ---- DualPathRNN.forward (:80-101): x [B, D, T, F]
    def dual_path(self, p: str, x: Array, h: Optional[Array], taps, tag) -> Tuple[Array, Array]:
        w, c = self.w, self.cfg
        B, D, T, F = x.shape
        x = x.transpose(0, 2, 3, 1)                        # [B, T, F, D]
        res = x
        y = layer_norm_fd(x, w[p + "intra_norm.weight"], w[p + "intra_norm.bias"]).reshape(B * T, F, D)
        Hh = c.hidden // 2
        outs = []
        q = p + "intra_rnn_attn.rnn."
        for sfx, order in (("", range(F)), ("_reverse", range(F - 1, -1, -1))):
            hh = np.zeros((B * T, Hh), self.dtype)
            o = np.empty((B * T, F, Hh), self.dtype)
            for f in order:
                hh = gru_cell(y[:, f], hh, w[q + "weight_ih_l0" + sfx], w[q + "weight_hh_l0" + sfx], w[q + "bias_ih_l0" + sfx], w[q + "bias_hh_l0" + sfx])
                o[:, f] = hh
            outs.append(o)
        y = np.concatenate(outs, axis=2) @ w[p + "intra_rnn_attn.dense.weight"].T + w[p + "intra_rnn_attn.dense.bias"]
        x = y.reshape(B, T, F, D) + res
        if taps is not None:
            taps[tag + ".intra"] = x.copy()
        res = x
        y = layer_norm_fd(x, w[p + "inter_norm.weight"], w[p + "inter_norm.bias"])
        y = y.transpose(0, 2, 1, 3).reshape(B * F, T, D)
        q = p + "inter_rnn_attn.rnn."
        hh = np.zeros((B * F, c.hidden), self.dtype) if h is None else h.astype(self.dtype).reshape(B * F, c.hidden).copy()
        ys = np.empty((B * F, T, c.hidden), self.dtype)
        for t in range(T):
            hh = gru_cell(y[:, t], hh, w[q + "weight_ih_l0"], w[q + "weight_hh_l0"], w[q + "bias_ih_l0"], w[q + "bias_hh_l0"])
            ys[:, t] = hh
        y = ys @ w[p + "inter_rnn_attn.dense.weight"].T + w[p + "inter_rnn_attn.dense.bias"]
        x = y.reshape(B, F, T, D).transpose(0, 2, 1, 3) + res
        if taps is not None:
            taps[tag + ".inter"] = x.copy()
        return x.transpose(0, 3, 1, 2), hh.reshape(1, B * F, c.hidden)

    # ---- ConvolutionalGLU.forward (:120-136)
    def conv_glu(self, p: str, x: Array, cache: Optional[Array]) -> Tuple[Array, Array]:
        w = self.w
        res = x
        y = custom_ln(x, w[p + "norm.gamma"], w[p + "norm.beta"])
        y = conv2d(y, w[p + "fc1.weight"], w[p + "fc1.bias"])
        hd = y.shape[1] // 2
        xx, v = y[:, :hd], y[:, hd:]
        xx = np.concatenate([np.zeros_like(xx[:, :, :1].repeat(2, axis=2)) if cache is None else cache.astype(self.dtype), xx], axis=2)
        cache_out = xx[:, :, -2:].copy()
        y = mish(conv2d(xx, w[p + "dwconv.weight"], w[p + "dwconv.bias"], 1, 1, groups=hd)) * v
        y = conv2d(y, w[p + "fc2.weight"], w[p + "fc2.bias"])
        return y + res, cache_out

    # ---- ONNXModel.model_forward (:398-432): x [B, 3, T, F]
    def model_forward(self, x: Array, caches: Optional[List[Array]], taps: Optional[dict] = None):
        c, w = self.cfg, self.w
        cin = [None] * (c.n_caches - 1) if caches is None else list(caches)
        out: List[Array] = []
        x1 = prelu(custom_ln(conv2d(x, w["encoder.conv_1.0.weight"], w["encoder.conv_1.0.bias"]), w["encoder.conv_1.1.gamma"],
                             w["encoder.conv_1.1.beta"]), w["encoder.conv_1.2.weight"])
        x2, c0 = self.dsconv("encoder.conv_2", x1, cin[0])
        x3, c1 = self.dsconv("encoder.conv_3", x2, cin[1])
        x4, c2 = self.dsconv("encoder.conv_4", x3, cin[2])
        out += [c0, c1, c2]
        if taps is not None:
            taps["encoder.conv_1"], taps["encoder.conv_2"], taps["encoder.conv_3"], taps["encoder.conv_4"] = x1.copy(), x2.copy(), x3.copy(), x4.copy()
        y = x4
        for b in range(c.n_blocks):
            y, h = self.dual_path(f"blocks.{b}.dp_rnn_attn.", y, cin[3 + 2 * b], taps, f"blocks.{b}")
            y, cc = self.conv_glu(f"blocks.{b}.conv_glu.", y, cin[4 + 2 * b])
            out += [h, cc]
            if taps is not None:
                taps[f"blocks.{b}"] = y.copy()
        # MaskDecoder.forward (:295-309)
        y = self.usconv("decoder.up1.", np.concatenate([y, x4], axis=1))
        y = self.usconv("decoder.up2.", np.concatenate([y, x3], axis=1))
        y = self.usconv("decoder.up3.", np.concatenate([y, x2], axis=1))
        if taps is not None:
            taps["decoder.up3"] = y.copy()
        cd = cin[3 + 2 * c.n_blocks]
        y = np.concatenate([np.zeros_like(y[:, :, :1]) if cd is None else cd.astype(self.dtype), y], axis=2)
        out.append(y[:, :, -1:].copy())
        y = conv2d(y, w["decoder.mask_conv.0.weight"], w["decoder.mask_conv.0.bias"], 1, 1)
        y = prelu(custom_ln(y, w["decoder.mask_conv.1.gamma"], w["decoder.mask_conv.1.beta"]), w["decoder.mask_conv.2.weight"])
        y = conv2d(y, w["decoder.mask_conv.3.weight"], w["decoder.mask_conv.3.bias"])
        slope = w["decoder.lsigmoid.slope"].reshape(1, 1, 1, -1)                       # per frequency
        mask = sigmoid(slope * y).transpose(0, 3, 2, 1)                                 # [B, F, T, 2]
        return mask.astype(self.dtype), out

    def features(self, spec: Array, pha_prev: Optional[Array], onnx: bool) -> Tuple[Array, Array]:
        """ONNXModel.forward :449-456 with cal_gd / cal_ifd (:357-378), or Model.forward :519-524 with its torch.diff variants (:491-510).
        spec [B, F, T, 2] (compressed) -> x [B, 3, T, F], last phase [B, 1, F]"""
        c = self.cfg
        x = spec.transpose(0, 2, 1, 3)
        mag = np.sqrt(x[..., 0] ** 2 + x[..., 1] ** 2)
        pha = np.arctan2(x[..., 1], x[..., 0])                                          # [B, T, F]
        B, T, F = pha.shape
        prev_f = np.concatenate([np.zeros((B, T, 1), self.dtype), pha[:, :, :-1]], axis=2)
        prev_t = np.concatenate([np.zeros((B, 1, F), self.dtype) if pha_prev is None else pha_prev.astype(self.dtype), pha[:, :-1]], axis=1)
        adv = (2 * np.pi * (c.hop_size / c.n_fft) * np.arange(F, dtype=np.float32)).astype(self.dtype)[None, None, :]
        if onnx:
            gd = wrap(prev_f - pha)
            ifd = wrap((prev_t - pha) - adv)
        else:
            gd = wrap(pha - prev_f)
            ifd = wrap((pha - prev_t) - adv)
        feat = np.stack([mag, gd / self.dtype(np.pi), ifd / self.dtype(np.pi)], axis=1).astype(self.dtype)
        return feat, pha[:, -1:].copy()

    # ---- ONNXModel.forward (:434-474)
    def spec_forward(self, spec: Array, caches: List[Array], taps: Optional[dict] = None):
        c = self.cfg
        x = spec.astype(self.dtype)
        mag = np.maximum(np.sqrt(x[..., 0:1] ** 2 + x[..., 1:2] ** 2), self.dtype(1e-5))
        x = x * mag ** self.dtype(c.input_compression - 1.0)
        if taps is not None:
            taps["compressed"] = x.copy()
        feat, pha_last = self.features(x, caches[0], True)
        if taps is not None:
            taps["features"] = feat.copy()
        mask, cache_out = self.model_forward(feat, list(caches[1:]), taps)
        if taps is not None:
            taps["mask"] = mask.copy()
        y = np.stack([x[..., 0] * mask[..., 0] - x[..., 1] * mask[..., 1], x[..., 0] * mask[..., 1] + x[..., 1] * mask[..., 0]], axis=3)
        mag2 = np.sqrt(y[..., 0:1] ** 2 + y[..., 1:2] ** 2)
        y = y * mag2 ** self.dtype(1.0 / c.input_compression - 1.0)
        return y.astype(self.dtype), [pha_last] + cache_out

    def stft_step(self, wav_in: Array, cache: Array):
        c = self.cfg
        x = np.concatenate([cache, wav_in.astype(self.dtype)], axis=1)
        cache = x[:, -(c.n_fft - c.hop_size):].copy()
        X = np.fft.rfft(x * self.window, axis=1)
        return np.stack([X.real, X.imag], axis=-1).astype(self.dtype)[:, :, None, :], cache

    def istft_step(self, spec: Array, cache: Array):
        c = self.cfg
        Y = spec[:, :, 0, 0] + 1j * spec[:, :, 0, 1]
        Y[:, 0] = Y[:, 0].real
        Y[:, -1] = Y[:, -1].real
        x = np.fft.irfft(Y, n=c.n_fft, axis=1).astype(self.dtype) * self.window_istft
        L = c.n_fft - c.hop_size
        x[:, :L] += cache
        return x[:, :c.hop_size].copy(), x[:, c.hop_size:].copy()

    def step(self, wav_in: Array, cache_stft: Array, cache_istft: Array, *cache_model: Array, taps: Optional[dict] = None):
        """the wav -> wav streaming step (scripts/export_onnx.py:48-58 with `model: lisennet`)"""
        spec_in, cache_stft = self.stft_step(wav_in, cache_stft)
        spec_out, cache_out = self.spec_forward(spec_in, list(cache_model), taps)
        if taps is not None:
            taps["spec_in"], taps["spec_out"] = spec_in.copy(), spec_out.copy()
        wav_out, cache_istft = self.istft_step(spec_out, cache_istft)
        return (wav_out, cache_stft, cache_istft, *cache_out)

    def offline_forward(self, noisy: Array, feat: Optional[Array] = None):
        """Model.forward (:512-531): CompressedSTFT keeping all 257 bins, torch.diff phase features, zero-padded convs.
        `feat`: use these input features [B, 3, T, F] instead of the ones extracted here (frame 0 of the reference's offline
        path is ill-conditioned - see tools/gen_golden.py::gen_lisennet)"""
        c = self.cfg
        N, H = c.n_fft, c.hop_size
        x = np.asarray(noisy, self.dtype)
        B, Tw = x.shape
        xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
        T = 1 + Tw // H
        frames = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * self.window
        X = np.fft.rfft(frames, axis=2)
        # (+ 0.0: numpy's rfft of an all-zero frame carries -0.0 in its upper bins, torch.stft's +0.0 - and atan2(0, -0.0) is pi)
        spec = np.stack([X.real, X.imag], axis=-1).astype(self.dtype).transpose(0, 2, 1, 3) + self.dtype(0.0)
        mag = np.maximum(np.sqrt(spec[..., 0:1] ** 2 + spec[..., 1:2] ** 2), self.dtype(1e-5))
        spec = spec * mag ** self.dtype(c.input_compression - 1.0)
        if feat is None:
            feat, _ = self.features(spec, None, False)
        mask, _ = self.model_forward(np.asarray(feat, self.dtype), None)
        spec_hat = np.stack([spec[..., 0] * mask[..., 0] - spec[..., 1] * mask[..., 1],
                             spec[..., 0] * mask[..., 1] + spec[..., 1] * mask[..., 0]], axis=3).astype(self.dtype)
        mag2 = np.sqrt(spec_hat[..., 0:1] ** 2 + spec_hat[..., 1:2] ** 2)
        yu = spec_hat * mag2 ** self.dtype(1.0 / c.input_compression - 1.0)
        Y = (yu[..., 0] + 1j * yu[..., 1]).transpose(0, 2, 1)
        fr = np.fft.irfft(Y, n=N, axis=2).astype(self.dtype) * self.window
        full = np.zeros((B, (T - 1) * H + N), self.dtype)
        env = np.zeros((T - 1) * H + N, self.dtype)
        for t in range(T):
            full[:, t * H:t * H + N] += fr[:, t]
            env[t * H:t * H + N] += self.window ** 2
        sl = slice(N // 2, N // 2 + H * (T - 1))
        return (full[:, sl] / env[sl]).astype(self.dtype), spec_hat


def make_state_dict(cfg: LiSenNetConfig, seed: int) -> Dict[str, Array]:
    """Seeded synthetic checkpoint (see oracle/weightgen.py for the rationale)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Array] = {}
    for key, shape in state_dict_spec(cfg).items():
        leaf = key.split(".")[-1]
        if leaf in ("gamma",) or key.endswith("_norm.weight"):
            v = rng.uniform(0.75, 1.25, shape)
        elif leaf == "beta" or key.endswith("_norm.bias") or "bias" in leaf:
            v = 0.1 * rng.standard_normal(shape)
        elif key.endswith("act.weight") or key.endswith("conv_1.2.weight") or key.endswith("mask_conv.2.weight"):      # PReLU slopes
            v = rng.uniform(0.1, 0.4, shape)
        elif leaf == "slope":
            v = rng.uniform(0.8, 1.6, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        sd[key] = np.asarray(v, dtype=np.float32)
    return sd
