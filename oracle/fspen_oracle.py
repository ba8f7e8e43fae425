"""CPU oracle for the FSPEN baseline model (TEST INFRASTRUCTURE ONLY — same rules as oracle/fe_oracle.py).

numpy restatement of models/fspen/model.py of the reference (streaming ``ONNXModel`` with 8 inter-GRU caches per DPE
block, and offline ``Model``), each function citing the file:line it follows.  Pinned on outputs of the imported
reference (tools/gen_golden.py -> tests/golden/fspen.npz, tests/test_oracle_golden.py)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .fe_oracle import sigmoid, stft_windows

Array = np.ndarray

# SubbandEncoder.forward (models/fspen/model.py:58-66): (first bin, bins, left pad, right pad, kernel, stride, outputs)
SUB_ENC = [(0, 17, 1, 0, 4, 2, 8), (13, 22, 0, 0, 7, 3, 6), (30, 36, 0, 0, 11, 5, 6), (61, 70, 0, 0, 20, 10, 6), (122, 135, 0, 5, 40, 20, 6)]
# SubbandDecoder.forward (models/fspen/model.py:83-95): (first row, rows, outputs per row, first kept, kept)
SUB_DEC = [(0, 8, 2, 0, 16), (8, 6, 3, 1, 16), (13, 8, 5, 4, 32), (19, 8, 10, 8, 64), (25, 8, 20, 16, 129)]


@dataclass
class FSPENConfig:
    """yaml model_kwargs of `model: fspen` (configs/others/fspen.yaml:2-16; defaults models/fspen/model.py:201-212, :191-197)."""
    channels: Tuple[int, ...] = (4, 16, 32)
    kernel_size: Tuple[int, ...] = (6, 8, 6)
    stride: Tuple[int, ...] = (2, 2, 2)
    num_blocks: int = 3
    dpe_channels: int = 16
    freq: int = 32
    groups: int = 8
    norm: str = "LayerNorm-FreqChannels"
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @staticmethod
    def from_model_kwargs(kw: dict) -> "FSPENConfig":
        assert kw.get("window", "hann") == "hann"
        assert kw.get("n_fft", 512) == 512, "Only n_fft == 512 is allowed (models/fspen/model.py:214)"
        d = kw.get("dpe_kwargs", {})
        return FSPENConfig(channels=tuple(kw.get("channels", (4, 16, 32))), kernel_size=tuple(kw.get("kernel_size", (6, 8, 6))),
                           stride=tuple(kw.get("stride", (2, 2, 2))), num_blocks=d.get("num_blocks", 3),
                           dpe_channels=d.get("channels", 16), freq=d.get("freq", 32), groups=d.get("groups", 8),
                           norm=d.get("norm", "LayerNorm-FreqChannels"), n_fft=kw.get("n_fft", 512),
                           hop_size=kw.get("hop_size", 256), win_size=kw.get("win_size", 512),
                           input_compression=kw.get("input_compression", 0.3))

    @property
    def n_caches(self) -> int:
        return self.num_blocks * self.groups

    def enc_lengths(self) -> List[int]:
        F, out = self.n_fft // 2 + 1, []
        for k, s in zip(self.kernel_size, self.stride):
            F = (F + 2 * ((k - s) // 2) - k) // s + 1
            out.append(F)
        return out

    def macs_per_frame(self) -> int:
        """models/fspen/macs.py:36-141 with T = 1 (its switches as committed: output-length conv counts, no BN / LN / bias)."""
        C1, K, S, C2 = self.channels, self.kernel_size, self.stride, self.dpe_channels
        F = self.n_fft // 2 + 1
        m = 0
        for i in range(len(K)):
            F = F // S[i]
            m += (2 if i == 0 else C1[i - 1]) * C1[i] * F * K[i]
        m += C1[-1] ** 2 * F
        m += C1[-1] * (4 * 8 + 7 * 6 + 11 * 6 + 20 * 6 + 40 * 6)
        m += C1[-1] * 64 * 32 + C1[-1] * C2 * 32
        gru = (C2 + C2) * C2 * 3 + C2 * 3
        m += self.num_blocks * (gru * 2 + 2 * C2 * C2 + C2 + gru + C2 * C2 + C2) * 32
        m += C2 * C1[-1] * 32 + C1[-1] * 32 * 64
        m += C1[-1] * (8 * 2 + 6 * 3 + 8 * 5 + 8 * 10 + 8 * 20)
        for i in range(len(K) - 1, -1, -1):
            m += C1[i] * (2 if i == 0 else C1[i - 1]) * F * K[i]
            F = F * S[i] + 1 if i == 0 else F * S[i]
        m += 257 * 8
        return m

    def flops_per_frame(self) -> float:
        import math
        return 2.0 * self.macs_per_frame() + 2 * 2.5 * self.n_fft * math.log2(self.n_fft)


def training_state_dict_spec(cfg: FSPENConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the training-form checkpoint (module order of ONNXModel.__init__, models/fspen/model.py:225-277)."""
    C1, K, C2, Fq = cfg.channels, cfg.kernel_size, cfg.dpe_channels, cfg.freq
    spec: Dict[str, Tuple[int, ...]] = {}

    def bn(p, c):
        spec[p + ".weight"] = (c,)
        spec[p + ".bias"] = (c,)
        spec[p + ".running_mean"] = (c,)
        spec[p + ".running_var"] = (c,)
        spec[p + ".num_batches_tracked"] = ()

    def gru(p, sfx=""):
        spec[f"{p}.weight_ih_l0{sfx}"] = (3 * C2, C2)
        spec[f"{p}.weight_hh_l0{sfx}"] = (3 * C2, C2)
        spec[f"{p}.bias_ih_l0{sfx}"] = (3 * C2,)
        spec[f"{p}.bias_hh_l0{sfx}"] = (3 * C2,)

    for i, (_, _, _, _, k, _, _) in enumerate(SUB_ENC):
        spec[f"subband_encoder.conv{i + 1}.0.weight"] = (C1[-1], 1, k)
        spec[f"subband_encoder.conv{i + 1}.0.bias"] = (C1[-1],)
    for i, (_, _, n, _, _) in enumerate(SUB_DEC):
        spec[f"subband_decoder.lin{i + 1}.0.weight"] = (n, 2 * C1[-1])
        spec[f"subband_decoder.lin{i + 1}.0.bias"] = (n,)
    for i in range(len(C1)):
        spec[f"fullband_encoder.{i}.0.weight"] = (C1[i], 2 if i == 0 else C1[i - 1], K[i])
        bn(f"fullband_encoder.{i}.1", C1[i])
    spec["fullband_encoder_post.weight"] = (C1[-1], C1[-1], 1)
    spec["feature_merge.0.weight"] = (Fq, 64)
    spec["feature_merge.2.weight"] = (C2, C1[-1], 1)
    spec["feature_merge.2.bias"] = (C2,)
    for b in range(cfg.num_blocks):
        p = f"dpe_blocks.{b}."
        gru(p + "intra_rnn")
        gru(p + "intra_rnn", "_reverse")
        spec[p + "intra_fc.weight"] = (C2, 2 * C2)
        spec[p + "intra_fc.bias"] = (C2,)
        spec[p + "intra_ln.weight"] = (Fq, C2)
        spec[p + "intra_ln.bias"] = (Fq, C2)
        for g in range(cfg.groups):
            gru(p + f"inter_rnn.inter_rnn.{g}")
        for g in range(cfg.groups):
            spec[p + f"inter_rnn.inter_fc.{g}.weight"] = (C2, C2)
            spec[p + f"inter_rnn.inter_fc.{g}.bias"] = (C2,)
    spec["feature_split.0.weight"] = (C1[-1], C2, 1)
    spec["feature_split.0.bias"] = (C1[-1],)
    spec["feature_split.1.weight"] = (64, Fq)
    for j, i in enumerate(range(len(C1) - 1, -1, -1)):
        cin, cout = C1[i], (2 if i == 0 else C1[i - 1])
        spec[f"fullband_decoder.{j}.0.weight"] = (cin, 2 * cin, 1)
        spec[f"fullband_decoder.{j}.1.weight"] = (cin, cout, K[i])
        if i == 0:
            spec[f"fullband_decoder.{j}.1.bias"] = (cout,)
        else:
            bn(f"fullband_decoder.{j}.2", cout)
    return spec


def fold_state_dict(sd: Dict[str, Array], cfg: FSPENConfig, eps: float = 1e-5) -> Dict[str, Array]:
    """ONNXModel.remove_weight_reparameterizations (models/fspen/model.py:299-340): BatchNorm AFTER the conv / transposed
    conv is folded into its weight (per OUTPUT channel: dim 0 of a Conv1d weight, dim 1 of a ConvTranspose1d weight) and a
    new bias; the rebuilt nn.Sequential drops the BN slot, so the activation's index moves and the keys stay `.0` / `.1`."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    if "fullband_encoder.0.1.running_var" not in sd:       # already fused
        return {k: v.astype(np.float32) for k, v in sd.items() if v.dtype.kind == "f"}
    out: Dict[str, Array] = {}
    n = len(cfg.channels)
    bn_keys = tuple(f"fullband_encoder.{i}.1." for i in range(n)) + tuple(f"fullband_decoder.{j}.2." for j in range(n - 1))
    for k, v in sd.items():
        if v.dtype.kind == "f" and not k.startswith(bn_keys):
            out[k] = v.astype(np.float32)

    def wb(p):
        std = np.sqrt(sd[p + ".running_var"].astype(np.float32) + np.float32(eps))
        return (sd[p + ".weight"] / std).astype(np.float32), (sd[p + ".bias"] - sd[p + ".running_mean"] * sd[p + ".weight"] / std).astype(np.float32)

    for i in range(len(cfg.channels)):
        w, b = wb(f"fullband_encoder.{i}.1")
        out[f"fullband_encoder.{i}.0.weight"] = sd[f"fullband_encoder.{i}.0.weight"].astype(np.float32) * w[:, None, None]
        out[f"fullband_encoder.{i}.0.bias"] = b
    for j in range(len(cfg.channels) - 1):
        w, b = wb(f"fullband_decoder.{j}.2")
        out[f"fullband_decoder.{j}.1.weight"] = sd[f"fullband_decoder.{j}.1.weight"].astype(np.float32) * w[None, :, None]
        out[f"fullband_decoder.{j}.1.bias"] = b
    return out


def elu(x: Array) -> Array:
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def gru_cell(x: Array, h: Array, w_ih: Array, w_hh: Array, b_ih: Array, b_hh: Array) -> Array:
    """one step of nn.GRU, gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn))."""
    C = h.shape[-1]
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = sigmoid(gi[..., :C] + gh[..., :C])
    z = sigmoid(gi[..., C:2 * C] + gh[..., C:2 * C])
    n = np.tanh(gi[..., 2 * C:] + r * gh[..., 2 * C:])
    return (1 - z) * n + z * h


def conv1d(x: Array, w: Array, b: Optional[Array], stride: int, pad: int) -> Array:
    """nn.Conv1d: x [M, Cin, F], w [Cout, Cin, K]"""
    M, Cin, F = x.shape
    K = w.shape[2]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    Fo = (F + 2 * pad - K) // stride + 1
    cols = np.stack([xp[:, :, k:k + stride * (Fo - 1) + 1:stride] for k in range(K)], axis=3)       # [M,Cin,Fo,K]
    y = np.einsum("mcfk,ock->mof", cols, w)
    return y if b is None else y + b[None, :, None]


def conv_transpose1d(x: Array, w: Array, b: Optional[Array], stride: int, pad: int, out_pad: int) -> Array:
    """nn.ConvTranspose1d: x [M, Cin, F], w [Cin, Cout, K]: y[o, f*stride + k - pad] += x[c, f] * w[c, o, k]"""
    M, Cin, F = x.shape
    K = w.shape[2]
    Fo = (F - 1) * stride - 2 * pad + K + out_pad
    full = np.zeros((M, w.shape[1], (F - 1) * stride + K + out_pad), x.dtype)
    for k in range(K):
        full[:, :, k:k + stride * (F - 1) + 1:stride] += np.einsum("mcf,co->mof", x, w[:, :, k])
    y = full[:, :, pad:pad + Fo]
    return y if b is None else y + b[None, :, None]


class FSPENOracle:
    def __init__(self, cfg: FSPENConfig, fused: Dict[str, Array], dtype=np.float32):
        self.cfg, self.dtype = cfg, dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in fused.items()}
        win, win_i = stft_windows(cfg.n_fft, cfg.hop_size, cfg.win_size, np.float32)
        self.window, self.window_istft = win.astype(dtype), win_i.astype(dtype)

    # ONNXSTFT.initialize_cache + ONNXModel.initialize_cache (models/fspen/model.py:293-297, :111-116) sized for B streams:
    # per DPE block and group one GRU state [1, B * freq/groups, C]
    def initialize_cache(self, B: int) -> List[Array]:
        c = self.cfg
        caches = [np.zeros((B, c.n_fft - c.hop_size), self.dtype), np.zeros((B, c.n_fft - c.hop_size), self.dtype)]
        caches += [np.zeros((1, B * (c.freq // c.groups), c.dpe_channels), self.dtype) for _ in range(c.n_caches)]
        return caches

    def subband_encoder(self, mag: Array) -> Array:
        """models/fspen/model.py:58-66: mag [M,1,257] -> [M,32,32]"""
        outs = []
        for i, (f0, n, pl, pr, k, s, _) in enumerate(SUB_ENC):
            x = np.pad(mag[:, :, f0:f0 + n], ((0, 0), (0, 0), (pl, pr)))
            outs.append(np.maximum(conv1d(x, self.w[f"subband_encoder.conv{i + 1}.0.weight"], self.w[f"subband_encoder.conv{i + 1}.0.bias"], s, 0), 0))
        return np.concatenate(outs, axis=2)

    def subband_decoder(self, x: Array) -> Array:
        """models/fspen/model.py:83-95: x [M,64,32] -> [M,257]"""
        M = x.shape[0]
        xt = x.transpose(0, 2, 1)                         # [M,32,64]
        xt = np.concatenate([xt, np.zeros((M, 1, xt.shape[2]), x.dtype)], axis=1)       # lin5's zero row (F.pad)
        outs = []
        for i, (r0, rows, n, k0, keep) in enumerate(SUB_DEC):
            y = np.maximum(xt[:, r0:r0 + rows] @ self.w[f"subband_decoder.lin{i + 1}.0.weight"].T + self.w[f"subband_decoder.lin{i + 1}.0.bias"], 0)
            outs.append(y.reshape(M, rows * n)[:, k0:k0 + keep])
        return np.concatenate(outs, axis=1)

    def dpe(self, b: int, x: Array, caches: List[Optional[Array]], taps: Optional[dict]) -> Tuple[Array, List[Array]]:
        """DPE.forward (models/fspen/model.py:172-189): x [T,B,F,C]"""
        c, w, p = self.cfg, self.w, f"dpe_blocks.{b}."
        T, B, F, C = x.shape
        xs = x.reshape(T * B, F, C)
        outs = []
        for sfx, order in (("", range(F)), ("_reverse", range(F - 1, -1, -1))):
            h = np.zeros((T * B, C), self.dtype)
            o = np.empty((T * B, F, C), self.dtype)
            for f in order:
                h = gru_cell(xs[:, f], h, w[p + "intra_rnn.weight_ih_l0" + sfx], w[p + "intra_rnn.weight_hh_l0" + sfx],
                             w[p + "intra_rnn.bias_ih_l0" + sfx], w[p + "intra_rnn.bias_hh_l0" + sfx])
                o[:, f] = h
            outs.append(o)
        y = np.concatenate(outs, axis=2) @ w[p + "intra_fc.weight"].T + w[p + "intra_fc.bias"]
        assert c.norm == "LayerNorm-FreqChannels"          # nn.LayerNorm([freq, channels]) (:150-151): biased variance, eps 1e-5
        mean = y.mean(axis=(1, 2), keepdims=True)
        var = ((y - mean) ** 2).mean(axis=(1, 2), keepdims=True)
        y = (y - mean) / np.sqrt(var + self.dtype(1e-5)) * w[p + "intra_ln.weight"] + w[p + "intra_ln.bias"]
        x = y.reshape(T, B, F, C) + x
        if taps is not None:
            taps[f"dpe.{b}.intra"] = x.copy()
        # InterRNNPathExtension.forward (:122-138): its own `x.add_(x_in)` and DPE's second one -> + 2 * x_in
        G, Fg = c.groups, F // c.groups
        new, pieces = [], []
        for g in range(G):
            q = p + f"inter_rnn.inter_rnn.{g}."
            xg = x[:, :, g * Fg:(g + 1) * Fg].reshape(T, B * Fg, C)
            h = np.zeros((B * Fg, C), self.dtype) if caches[g] is None else caches[g].astype(self.dtype).reshape(B * Fg, C).copy()
            ys = np.empty((T, B * Fg, C), self.dtype)
            for t in range(T):
                h = gru_cell(xg[t], h, w[q + "weight_ih_l0"], w[q + "weight_hh_l0"], w[q + "bias_ih_l0"], w[q + "bias_hh_l0"])
                ys[t] = h
            new.append(h.reshape(1, B * Fg, C).copy())
            yg = ys @ w[p + f"inter_rnn.inter_fc.{g}.weight"].T + w[p + f"inter_rnn.inter_fc.{g}.bias"]
            pieces.append(yg.reshape(T, B, Fg, C))
        x = np.concatenate(pieces, axis=2) + 2 * x
        if taps is not None:
            taps[f"dpe.{b}.inter"] = x.copy()
        return x, new

    def model_forward(self, spec: Array, caches: Optional[List[Array]], taps: Optional[dict] = None):
        """ONNXModel.model_forward (models/fspen/model.py:342-407): spec [B,257,T,2] (compressed)"""
        c, w = self.cfg, self.w
        B, F0, T, _ = spec.shape
        if caches is None:
            caches = [None] * c.n_caches
        x = spec.transpose(0, 2, 3, 1).reshape(B * T, 2, F0)
        mag = np.sqrt(x[:, 0:1] ** 2 + x[:, 1:2] ** 2)
        x_sub1 = self.subband_encoder(mag)
        if taps is not None:
            taps["subband_encoder"] = x_sub1.copy()
        enc_out = []
        for i in range(len(c.channels)):
            k, s = c.kernel_size[i], c.stride[i]
            x = elu(conv1d(x, w[f"fullband_encoder.{i}.0.weight"], w[f"fullband_encoder.{i}.0.bias"], s, (k - s) // 2))
            enc_out.append(x)
            if taps is not None:
                taps[f"fullband_encoder.{i}"] = x.copy()
        x = conv1d(x, w["fullband_encoder_post.weight"], None, 1, 0)
        x = np.concatenate([x, x_sub1], axis=2)                        # [M,32,64]
        x = elu(x @ w["feature_merge.0.weight"].T)                     # Linear over the last axis -> [M,32,32]
        x = conv1d(x, w["feature_merge.2.weight"], w["feature_merge.2.bias"], 1, 0)      # [M,16,32]
        if taps is not None:
            taps["feature_merge"] = x.copy()
        C, F1 = x.shape[1], x.shape[2]
        x = np.ascontiguousarray(x.reshape(B, T, C, F1).transpose(1, 0, 3, 2))           # [T,B,F1,C]
        cache_out = []
        for b in range(c.num_blocks):
            x, new = self.dpe(b, x, caches[b * c.groups:(b + 1) * c.groups], taps)
            cache_out += new
        x = x.transpose(1, 0, 3, 2).reshape(B * T, C, F1)
        x = conv1d(x, w["feature_split.0.weight"], w["feature_split.0.bias"], 1, 0)      # [M,32,32]
        x = elu(x @ w["feature_split.1.weight"].T)                     # [M,32,64]
        if taps is not None:
            taps["feature_split"] = x.copy()
        x_full, x_sub2 = x[:, :, :32], x[:, :, 32:]
        m_sub = self.subband_decoder(np.concatenate([x_sub1, x_sub2], axis=1))            # [M,257]
        mask_sub = m_sub.reshape(B, T, F0).transpose(0, 2, 1)[..., None]
        x = x_full
        n = len(c.channels)
        for j, i in enumerate(range(n - 1, -1, -1)):
            k, s = c.kernel_size[i], c.stride[i]
            x = np.concatenate([x, enc_out.pop(-1)], axis=1)
            x = conv1d(x, w[f"fullband_decoder.{j}.0.weight"], None, 1, 0)
            x = conv_transpose1d(x, w[f"fullband_decoder.{j}.1.weight"], w[f"fullband_decoder.{j}.1.bias"], s, (k - s) // 2, 1 if i == 0 else 0)
            if i != 0:
                x = elu(x)
            if taps is not None:
                taps[f"fullband_decoder.{j}"] = x.copy()
        mask_full = x.reshape(B, T, 2, F0).transpose(0, 3, 1, 2)        # [B,257,T,2]
        if taps is not None:
            taps["mask"] = np.concatenate([mask_full, mask_sub], axis=3)     # [B,257,T,3]: full (re, im), sub
        o_r = spec[..., 0] * mask_full[..., 0] - spec[..., 1] * mask_full[..., 1]
        o_i = spec[..., 0] * mask_full[..., 1] + spec[..., 1] * mask_full[..., 0]
        mfm = np.sqrt(mask_full[..., 0:1] ** 2 + mask_full[..., 1:2] ** 2)
        mask_mag = (mask_sub + mfm) * self.dtype(0.5)
        y = np.stack([o_r, o_i], axis=3) / mfm * mask_mag
        return y.astype(self.dtype), cache_out

    def spec_forward(self, spec: Array, caches: Optional[List[Array]], taps: Optional[dict] = None):
        """ONNXModel.forward (models/fspen/model.py:409-429)"""
        c = self.cfg
        x = spec.astype(self.dtype)
        mag = np.maximum(np.sqrt(x[..., 0:1] ** 2 + x[..., 1:2] ** 2), self.dtype(1e-5))
        x = x * mag ** self.dtype(c.input_compression - 1.0)
        if taps is not None:
            taps["compressed"] = x.copy()
        y, cache_out = self.model_forward(x, caches, taps)
        mag2 = np.sqrt(y[..., 0:1] ** 2 + y[..., 1:2] ** 2)
        y = y * mag2 ** self.dtype(1.0 / c.input_compression - 1.0)
        return y.astype(self.dtype), cache_out

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
        """the wav -> wav streaming step (scripts/export_onnx.py:48-58 with `model: fspen`)"""
        spec_in, cache_stft = self.stft_step(wav_in, cache_stft)
        spec_out, cache_out = self.spec_forward(spec_in, list(cache_model), taps)
        if taps is not None:
            taps["spec_in"], taps["spec_out"] = spec_in.copy(), spec_out.copy()
        wav_out, cache_istft = self.istft_step(spec_out, cache_istft)
        return (wav_out, cache_stft, cache_istft, *cache_out)

    def offline_forward(self, noisy: Array):
        """Model.forward (models/fspen/model.py:443-449): CompressedSTFT keeping all 257 bins"""
        c = self.cfg
        N, H = c.n_fft, c.hop_size
        x = np.asarray(noisy, self.dtype)
        B, Tw = x.shape
        xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
        T = 1 + Tw // H
        frames = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * self.window
        X = np.fft.rfft(frames, axis=2)
        spec = np.stack([X.real, X.imag], axis=-1).astype(self.dtype).transpose(0, 2, 1, 3)
        mag = np.maximum(np.sqrt(spec[..., 0:1] ** 2 + spec[..., 1:2] ** 2), self.dtype(1e-5))
        spec = spec * mag ** self.dtype(c.input_compression - 1.0)
        spec_hat, _ = self.model_forward(spec, None)
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


def make_training_state_dict(cfg: FSPENConfig, seed: int) -> Dict[str, Array]:
    """Seeded synthetic checkpoint (see oracle/weightgen.py for the rationale).  The last transposed conv (the complex
    mask) gets O(1) weights and a bias so that |mask_full| stays away from the 0 / 0 of `out_full / mask_full_mag`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Array] = {}
    last = f"fullband_decoder.{len(cfg.channels) - 1}.1."
    for key, shape in training_state_dict_spec(cfg).items():
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            sd[key] = np.asarray(100, dtype=np.int64)
            continue
        if leaf == "running_var":
            v = rng.uniform(0.75, 1.25, shape)
        elif leaf == "running_mean":
            v = 0.1 * rng.standard_normal(shape)
        elif key == last + "bias":
            v = np.asarray([0.6, 0.2]) + 0.05 * rng.standard_normal(shape)
        elif "bias" in leaf and len(shape) == 1:
            v = 0.1 * rng.standard_normal(shape)
        elif key.endswith("intra_ln.weight"):
            v = rng.uniform(0.75, 1.25, shape)
        elif key.endswith("intra_ln.bias"):
            v = 0.1 * rng.standard_normal(shape)
        elif len(shape) == 1:
            v = rng.uniform(0.75, 1.25, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            if ".1.weight" in key and key.startswith("fullband_decoder"):     # ConvTranspose1d [Cin, Cout, K]: ~K/stride taps x Cin per output
                fan_in = shape[0] * shape[2] // 2
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
            if key.startswith(last) or key.startswith("subband_decoder"):     # keeps both masks ~1 (out RMS ~ in RMS after the ^(1/0.3))
                v *= 0.45
        sd[key] = np.asarray(v, dtype=np.float32)
    return sd
