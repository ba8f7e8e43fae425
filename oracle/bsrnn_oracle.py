"""CPU oracle for the BSRNN baseline model (TEST INFRASTRUCTURE ONLY — same rules as oracle/fe_oracle.py).

numpy restatement of models/bsrnn/model.py of the reference (streaming ``ONNXModel`` with per-layer LSTM caches,
and offline ``Model``), each function citing the file:line it follows.  Pinned on outputs of the imported
reference (tools/gen_golden.py -> tests/golden/bsrnn_*.npz, tests/test_oracle_golden.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .fe_oracle import sigmoid, stft_windows

Array = np.ndarray

# models/bsrnn/model.py:107-111 (n_fft = 512 only): 31 bands covering all 257 bins (Nyquist kept)
SUBBANDS = [2] + [3] * 10 + [8] * 12 + [16] * 7 + [17]


@dataclass
class BSRNNConfig:
    """yaml model_kwargs of `model: bsrnn` (configs/others/bsrnn_xt.yaml:2-11; defaults models/bsrnn/model.py:261-272)."""
    num_channels: int = 16
    num_layers: int = 6
    bias: bool = True
    affine: bool = True
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @staticmethod
    def from_model_kwargs(kw: dict) -> "BSRNNConfig":
        assert kw.get("window", "hann") == "hann"
        assert kw.get("n_fft", 512) == 512, "Only n_fft=512 is supported (models/bsrnn/model.py:112-113)"
        return BSRNNConfig(num_channels=kw.get("num_channels", 16), num_layers=kw.get("num_layers", 6),
                           bias=kw.get("bias", True), affine=kw.get("affine", True), n_fft=kw.get("n_fft", 512),
                           hop_size=kw.get("hop_size", 256), win_size=kw.get("win_size", 512),
                           input_compression=kw.get("input_compression", 0.3))

    @property
    def n_bands(self) -> int:
        return len(SUBBANDS)

    @property
    def hidden(self) -> int:
        return 2 * self.num_channels

    def macs_per_frame(self) -> int:
        """models/bsrnn/macs.py:18-51 with T=1."""
        C, Hh, L = self.num_channels, self.hidden, self.num_layers
        m = sum(2 * s * C for s in SUBBANDS)
        m += (C * Hh * 4 + Hh * Hh * 4 + Hh * C + (C * Hh * 4 + Hh * Hh * 4) * 2 + 2 * Hh * C) * len(SUBBANDS) * L
        m += sum((C * C * 4 + 4 * C * 4 * s) * 2 for s in SUBBANDS)
        return m

    def flops_per_frame(self) -> float:
        import math
        return 2.0 * self.macs_per_frame() + 2 * 2.5 * self.n_fft * math.log2(self.n_fft)


def training_state_dict_spec(cfg: BSRNNConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the training-form checkpoint of `Model`, reference order."""
    C, Hh, L = cfg.num_channels, cfg.hidden, cfg.num_layers
    spec: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix, c, affine=True):
        if affine:
            spec[prefix + ".weight"] = (c,)
            spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    for b, s in enumerate(SUBBANDS):
        bn(f"band_split.norm.{b}", 2 * s, cfg.affine)
    for b, s in enumerate(SUBBANDS):
        spec[f"band_split.fc.{b}.weight"] = (C, 2 * s, 1)
        if cfg.bias:
            spec[f"band_split.fc.{b}.bias"] = (C,)
    for l in range(L):
        bn(f"norm_time.{l}", C, cfg.affine)
    for l in range(L):
        spec[f"rnn_time.{l}.weight_ih_l0"] = (4 * Hh, C)
        spec[f"rnn_time.{l}.weight_hh_l0"] = (4 * Hh, Hh)
        spec[f"rnn_time.{l}.bias_ih_l0"] = (4 * Hh,)
        spec[f"rnn_time.{l}.bias_hh_l0"] = (4 * Hh,)
    for l in range(L):
        spec[f"fc_time.{l}.weight"] = (C, Hh)
        if cfg.bias:
            spec[f"fc_time.{l}.bias"] = (C,)
    for l in range(L):
        bn(f"norm_freq.{l}", C, cfg.affine)
    for l in range(L):
        for sfx in ("", "_reverse"):
            spec[f"rnn_freq.{l}.weight_ih_l0{sfx}"] = (4 * Hh, C)
            spec[f"rnn_freq.{l}.weight_hh_l0{sfx}"] = (4 * Hh, Hh)
            spec[f"rnn_freq.{l}.bias_ih_l0{sfx}"] = (4 * Hh,)
            spec[f"rnn_freq.{l}.bias_hh_l0{sfx}"] = (4 * Hh,)
    for l in range(L):
        spec[f"fc_freq.{l}.weight"] = (C, 2 * Hh)
        if cfg.bias:
            spec[f"fc_freq.{l}.bias"] = (C,)
    for kind in ("mlp_mask", "mlp_residual"):
        for b, s in enumerate(SUBBANDS):
            p = f"mask_decoder.{kind}.{b}."
            bn(p + "0", C, cfg.bias)          # NB: the reference passes `bias` as the BN `affine` flag (model.py:323-325)
            spec[p + "1.weight"] = (4 * C, C, 1)
            spec[p + "1.bias"] = (4 * C,)
            spec[p + "3.weight"] = (4 * s, 4 * C, 1)
            spec[p + "3.bias"] = (4 * s,)
    return spec


def _bn_wb(sd, prefix, eps=1e-5):
    """w, b of "x*w + b" for an eval BatchNorm (affine optional): models/bsrnn/model.py:27-33."""
    std = np.sqrt(sd[prefix + ".running_var"].astype(np.float32) + np.float32(eps))
    w = 1.0 / std
    b = -sd[prefix + ".running_mean"] / std
    if prefix + ".weight" in sd:
        w = sd[prefix + ".weight"] * w
        b = b * sd[prefix + ".weight"] + sd[prefix + ".bias"]
    return w.astype(np.float32), b.astype(np.float32)


def fold_state_dict(sd: Dict[str, Array], cfg: BSRNNConfig) -> Dict[str, Array]:
    """ONNXModel.remove_weight_reparameterizations (models/bsrnn/model.py:348-366) with fuse_bn_conv1d (:14-42) and
    fuse_bn_rnn (:45-82): BatchNorm BEFORE the conv / LSTM is folded forward into its weight and bias.  Returns the fused
    dict with the streaming key names (`rnn_time.{l}.weight_ih` ...: load_state_dict rename, :450-460)."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    if "band_split.norm.0.running_var" not in sd:      # already fused
        return {k: v.astype(np.float32) for k, v in sd.items() if v.dtype.kind == "f"}
    out: Dict[str, Array] = {}
    C, L = cfg.num_channels, cfg.num_layers

    def conv(conv_key, bn_key, dst):
        w, b = _bn_wb(sd, bn_key)
        W = sd[conv_key + ".weight"].astype(np.float32)
        bias = (W * b.reshape(1, -1, 1)).sum(axis=(1, 2))
        if conv_key + ".bias" in sd:
            bias = bias + sd[conv_key + ".bias"]
        out[dst + ".weight"] = W * w.reshape(1, -1, 1)
        out[dst + ".bias"] = bias.astype(np.float32)

    def rnn(src, bn_key, dst_ih_w, dst_ih_b, sfx=""):
        w, b = _bn_wb(sd, bn_key)
        W = sd[src + ".weight_ih_l0" + sfx].astype(np.float32)
        out[dst_ih_w] = W * w.reshape(1, -1)
        out[dst_ih_b] = (sd[src + ".bias_ih_l0" + sfx] + W @ b).astype(np.float32)

    for bnd in range(len(SUBBANDS)):
        conv(f"band_split.fc.{bnd}", f"band_split.norm.{bnd}", f"band_split.fc.{bnd}")
    for l in range(L):
        rnn(f"rnn_time.{l}", f"norm_time.{l}", f"rnn_time.{l}.weight_ih", f"rnn_time.{l}.bias_ih")
        out[f"rnn_time.{l}.weight_hh"] = sd[f"rnn_time.{l}.weight_hh_l0"].astype(np.float32)
        out[f"rnn_time.{l}.bias_hh"] = sd[f"rnn_time.{l}.bias_hh_l0"].astype(np.float32)
        out[f"fc_time.{l}.weight"] = sd[f"fc_time.{l}.weight"].astype(np.float32)
        out[f"fc_time.{l}.bias"] = sd.get(f"fc_time.{l}.bias", np.zeros(C)).astype(np.float32)
        for sfx in ("", "_reverse"):
            rnn(f"rnn_freq.{l}", f"norm_freq.{l}", f"rnn_freq.{l}.weight_ih_l0{sfx}", f"rnn_freq.{l}.bias_ih_l0{sfx}", sfx)
            out[f"rnn_freq.{l}.weight_hh_l0{sfx}"] = sd[f"rnn_freq.{l}.weight_hh_l0{sfx}"].astype(np.float32)
            out[f"rnn_freq.{l}.bias_hh_l0{sfx}"] = sd[f"rnn_freq.{l}.bias_hh_l0{sfx}"].astype(np.float32)
        out[f"fc_freq.{l}.weight"] = sd[f"fc_freq.{l}.weight"].astype(np.float32)
        out[f"fc_freq.{l}.bias"] = sd.get(f"fc_freq.{l}.bias", np.zeros(C)).astype(np.float32)
    for kind in ("mlp_mask", "mlp_residual"):
        for bnd in range(len(SUBBANDS)):
            p = f"mask_decoder.{kind}.{bnd}."
            conv(p + "1", p + "0", p + "0")
            out[p + "2.weight"] = sd[p + "3.weight"].astype(np.float32)
            out[p + "2.bias"] = sd[p + "3.bias"].astype(np.float32)
    return out


def lstm_cell(x: Array, h: Array, c: Array, w_ih: Array, w_hh: Array, b_ih: Array, b_hh: Array) -> Tuple[Array, Array]:
    """nn.LSTMCell / one step of nn.LSTM, gate order i,f,g,o."""
    Hh = h.shape[1]
    g = x @ w_ih.T + b_ih + h @ w_hh.T + b_hh
    i, f, gg, o = sigmoid(g[:, :Hh]), sigmoid(g[:, Hh:2 * Hh]), np.tanh(g[:, 2 * Hh:3 * Hh]), sigmoid(g[:, 3 * Hh:])
    c2 = f * c + i * gg
    return o * np.tanh(c2), c2


class BSRNNOracle:
    def __init__(self, cfg: BSRNNConfig, fused: Dict[str, Array], dtype=np.float32):
        self.cfg, self.dtype = cfg, dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in fused.items()}
        win, win_i = stft_windows(cfg.n_fft, cfg.hop_size, cfg.win_size, np.float32)
        self.window, self.window_istft = win.astype(dtype), win_i.astype(dtype)

    # caches: ONNXSTFT.initialize_cache + ONNXModel.initialize_cache (models/bsrnn/model.py:409-416), sized for B streams:
    # per layer (h, c) of shape [B*31, 2C]
    def initialize_cache(self, B: int) -> List[Array]:
        c = self.cfg
        caches = [np.zeros((B, c.n_fft - c.hop_size), self.dtype), np.zeros((B, c.n_fft - c.hop_size), self.dtype)]
        caches += [np.zeros((B * c.n_bands, c.hidden), self.dtype) for _ in range(2 * c.num_layers)]
        return caches

    # ---- BandSplit.forward (models/bsrnn/model.py:136-153) on fused weights.  spec [B,F,T,2] -> [T,B,31,C]
    def band_split(self, spec: Array) -> Array:
        B, F, T, _ = spec.shape
        outs, start = [], 0
        for b, s in enumerate(SUBBANDS):
            x = spec[:, start:start + s].transpose(0, 1, 3, 2).reshape(B, 2 * s, T)      # index f*2 + ri
            W = self.w[f"band_split.fc.{b}.weight"][:, :, 0]
            outs.append(np.einsum("oc,bct->bot", W, x) + self.w[f"band_split.fc.{b}.bias"][None, :, None])
            start += s
        x = np.stack(outs, axis=1)                           # [B,31,C,T]
        return np.ascontiguousarray(x.transpose(3, 0, 1, 2))

    # ---- MaskDecoder.forward (models/bsrnn/model.py:225-246): x [B,31,C,T] -> mask, residual [B,F,T,2]
    def mask_decoder(self, x: Array) -> Tuple[Array, Array]:
        B, _, C, T = x.shape
        res = []
        for kind in ("mlp_mask", "mlp_residual"):
            parts = []
            for b, s in enumerate(SUBBANDS):
                p = f"mask_decoder.{kind}.{b}."
                xb = x[:, b]                                                          # [B,C,T]
                h = np.tanh(np.einsum("oc,bct->bot", self.w[p + "0.weight"][:, :, 0], xb) + self.w[p + "0.bias"][None, :, None])
                a = np.einsum("oc,bct->bot", self.w[p + "2.weight"][:, :, 0], h) + self.w[p + "2.bias"][None, :, None]
                a = a[:, :2 * s] * sigmoid(a[:, 2 * s:])                              # GLU(dim=1)
                parts.append(a.reshape(B, s, 2, T))
            res.append(np.concatenate(parts, axis=1).transpose(0, 1, 3, 2))           # [B,F,T,2]
        return res[0], res[1]

    # ---- ONNXModel.model_forward (models/bsrnn/model.py:367-407); caches = [h0, c0, h1, c1, ...] or None
    def model_forward(self, spec: Array, caches: Optional[List[Array]], taps: Optional[dict] = None):
        c, w = self.cfg, self.w
        x = self.band_split(spec)                          # [T,B,31,C]
        T, B, F, C = x.shape
        Hh = c.hidden
        if taps is not None:
            taps["band_split"] = x.copy()
        cache_out = []
        for l in range(c.num_layers):
            skip = x
            xs = x.reshape(T, B * F, C)
            h = np.zeros((B * F, Hh), self.dtype) if caches is None else caches[2 * l].astype(self.dtype).copy()
            cc = np.zeros((B * F, Hh), self.dtype) if caches is None else caches[2 * l + 1].astype(self.dtype).copy()
            ys = np.empty((T, B * F, Hh), self.dtype)
            for t in range(T):
                h, cc = lstm_cell(xs[t], h, cc, w[f"rnn_time.{l}.weight_ih"], w[f"rnn_time.{l}.weight_hh"],
                                  w[f"rnn_time.{l}.bias_ih"], w[f"rnn_time.{l}.bias_hh"])
                ys[t] = h
            cache_out += [h.copy(), cc.copy()]
            x = (ys @ w[f"fc_time.{l}.weight"].T + w[f"fc_time.{l}.bias"]).reshape(T, B, F, C) + skip
            if taps is not None:
                taps[f"layer.{l}.time"] = x.copy()
            skip = x
            xs = x.reshape(T * B, F, C)
            outs = []
            for sfx, order in (("", range(F)), ("_reverse", range(F - 1, -1, -1))):
                h = np.zeros((T * B, Hh), self.dtype)
                cc = np.zeros((T * B, Hh), self.dtype)
                o = np.empty((T * B, F, Hh), self.dtype)
                for f in order:
                    h, cc = lstm_cell(xs[:, f], h, cc, w[f"rnn_freq.{l}.weight_ih_l0{sfx}"], w[f"rnn_freq.{l}.weight_hh_l0{sfx}"],
                                      w[f"rnn_freq.{l}.bias_ih_l0{sfx}"], w[f"rnn_freq.{l}.bias_hh_l0{sfx}"])
                    o[:, f] = h
                outs.append(o)
            y = np.concatenate(outs, axis=2)               # [TB,F,2Hh] (forward | backward)
            x = (y @ w[f"fc_freq.{l}.weight"].T + w[f"fc_freq.{l}.bias"]).reshape(T, B, F, C) + skip
            if taps is not None:
                taps[f"layer.{l}.freq"] = x.copy()
        xd = x.transpose(1, 2, 3, 0)                       # [B,31,C,T]
        mask, res = self.mask_decoder(xd)
        if taps is not None:
            taps["mask_mlp"] = np.concatenate([mask, res], axis=3)      # [B,257,T,4]: mask (re, im), residual (re, im)
        y = np.stack([spec[..., 0] * mask[..., 0] - spec[..., 1] * mask[..., 1],
                      spec[..., 0] * mask[..., 1] + spec[..., 1] * mask[..., 0]], axis=3) + res
        return y.astype(self.dtype), cache_out

    # ---- ONNXModel.forward (models/bsrnn/model.py:418-448): compress / uncompress on all 257 bins
    def spec_forward(self, spec: Array, caches: Optional[List[Array]], taps: Optional[dict] = None):
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

    # streaming STFT / iSTFT are the ONNXSTFT of functional/audio_modules.py:243-303 (same as FastEnhancer)
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
        Y[:, -1] = Y[:, -1].real      # 2*Re(ifft(padded half spectrum)) - correction keeps only Re X[N/2]
        x = np.fft.irfft(Y, n=c.n_fft, axis=1).astype(self.dtype) * self.window_istft
        L = c.n_fft - c.hop_size
        x[:, :L] += cache
        return x[:, :c.hop_size].copy(), x[:, c.hop_size:].copy()

    # ---- the wav->wav streaming step (scripts/export_onnx.py:48-58 with `model: bsrnn`)
    def step(self, wav_in: Array, cache_stft: Array, cache_istft: Array, *cache_model: Array, taps: Optional[dict] = None):
        spec_in, cache_stft = self.stft_step(wav_in, cache_stft)
        spec_out, cache_out = self.spec_forward(spec_in, list(cache_model), taps)
        if taps is not None:
            taps["spec_in"], taps["spec_out"] = spec_in.copy(), spec_out.copy()
        wav_out, cache_istft = self.istft_step(spec_out, cache_istft)
        return (wav_out, cache_stft, cache_istft, *cache_out)

    # ---- Model.forward (models/bsrnn/model.py:476-483): CompressedSTFT keeps all bins (discard_last_freq_bin False)
    def offline_forward(self, noisy: Array):
        c = self.cfg
        N, H = c.n_fft, c.hop_size
        x = np.asarray(noisy, self.dtype)
        B, Tw = x.shape
        xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
        T = 1 + Tw // H
        frames = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * self.window
        X = np.fft.rfft(frames, axis=2)
        spec = np.stack([X.real, X.imag], axis=-1).astype(self.dtype).transpose(0, 2, 1, 3)      # [B,F,T,2]
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


def make_training_state_dict(cfg: BSRNNConfig, seed: int) -> Dict[str, Array]:
    """Seeded synthetic checkpoint (see oracle/weightgen.py for the rationale)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Array] = {}
    for key, shape in training_state_dict_spec(cfg).items():
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            sd[key] = np.asarray(100, dtype=np.int64)
            continue
        if leaf == "running_var":
            v = rng.uniform(0.75, 1.25, shape)
        elif leaf == "running_mean":
            v = 0.1 * rng.standard_normal(shape)
        elif "bias" in leaf:
            v = 0.1 * rng.standard_normal(shape)
        elif len(shape) == 1:
            v = rng.uniform(0.75, 1.25, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        sd[key] = np.asarray(v, dtype=np.float32)
    return sd
