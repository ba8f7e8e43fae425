"""model_kwargs (yaml) -> FEConfig.  Mirrors the constructor signature of
models/fastenhancer/default/model.py:384-403 and asserts the invariants that the
HIP kernels specialise on (true for every shipped yaml, SURVEY.md top)."""
from __future__ import annotations

import typing as tp
from dataclasses import dataclass
from typing import Any, Dict, Optional, Sequence, Tuple


@dataclass(frozen=True)
class FEConfig:
    channels: int = 64
    kernel_size: Tuple[int, ...] = (8, 3, 3)
    stride: int = 4
    rf_blocks: int = 3
    rf_channels: int = 32
    rf_freq: int = 32
    rf_heads: int = 4
    rf_eps: float = 1e-8
    positional_embedding: Optional[str] = "train"
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3
    weight_norm: bool = False
    normalize_final_conv: bool = False
    pre_post_init: Optional[str] = None
    # `model: fastenhancer.time_kernel` (models/fastenhancer/time_kernel/model.py): the k = 3 convs of the encoder / decoder
    # are causal Conv2d with kernel_size_time taps over time and (kernel_size_time - 1)-frame input caches; 1 = the default model
    kernel_size_time: int = 1
    final_scale_exp: bool = False        # final_scale: "exp" (the final conv's scale parameter is stored as its log)
    # `model: fastenhancer.dprnn` (models/fastenhancer/dprnn/model.py): the blocks' attention is a bidirectional GRU over the
    # sub-bands with channels_frnn hidden units per direction; 0 = the default RNNFormer block
    channels_frnn: int = 0

    @property
    def dprnn(self) -> bool:
        return self.channels_frnn > 0

    # `model: fastenhancer.dptransformer` (models/fastenhancer/dptransformer/model.py): the blocks' time GRU is a causal attention
    # over the last `lookbehind` frames (K / V caches per block) with a learned positional bias; 0 = the default block
    lookbehind: int = 0

    @property
    def dpt(self) -> bool:
        return self.lookbehind > 0

    # `model: fastenhancer.ln` (models/fastenhancer/ln/model.py): GroupNorm(1, C) after every conv and the reference's LayerNorm
    # after the blocks' fc layers instead of BatchNorms; nothing folds into the convs
    ln: bool = False
    # `model: fastenhancer.noncausal` (models/fastenhancer/noncausal/model.py:186-187): the blocks' time GRU is bidirectional and
    # rnn_fc maps 2 C2 -> C2; the reference module has the offline `Model` only (no caches, no streaming step)
    noncausal: bool = False

    @property
    def time_kernel(self) -> bool:
        return self.kernel_size_time > 1

    @property
    def F0(self) -> int:
        return self.n_fft // 2

    @property
    def F1(self) -> int:
        return self.n_fft // 2 // self.stride

    @property
    def n_layers(self) -> int:
        return len(self.kernel_size) - 1

    @property
    def cache_len(self) -> int:
        return self.n_fft - self.hop_size

    @staticmethod
    def from_model_kwargs(
        channels: int = 64,
        kernel_size: Sequence[int] = (8, 3, 3),
        stride: int = 4,
        rnnformer_kwargs: Optional[Dict[str, Any]] = None,
        activation: str = "ReLU",
        activation_kwargs: Optional[Dict[str, Any]] = None,
        n_fft: int = 512,
        hop_size: int = 256,
        win_size: int = 512,
        window: Optional[str] = "hann",
        stft_normalized: bool = False,
        mask: Optional[str] = None,
        input_compression: float = 0.3,
        weight_norm: bool = False,
        normalize_final_conv: bool = False,
        pre_post_init: Optional[str] = None,
        resnet: bool = False,
    ) -> "FEConfig":
        rk = dict(rnnformer_kwargs or {})
        # the reference raises / asserts on bad values (model.py:25-41,434-435; audio_modules.py:192-193);
        # options the HIP path does not implement are rejected here instead of being silently ignored.
        # (explicit raises of the reference's exception type: a bare `assert` disappears under `python -O`)
        def require(cond, msg=""):
            if not cond:
                raise AssertionError(msg)
        require(n_fft % 2 == 0, f"`n_fft` must be an even number, but given {n_fft}.")
        require(stft_normalized is False, "stft_normalized must be False")
        require(n_fft >= win_size, f"n_fft({n_fft}) must be bigger than win_size({win_size})")
        require(kernel_size[0] % stride == 0, "kernel_size[0] must be a multiple of stride")
        require((kernel_size[0] - stride) % 2 == 0, "kernel_size[0] - stride must be even")
        if pre_post_init not in (None, "linear", "linear_fixed"):
            raise RuntimeError(f"model_kwargs.pre_post_init={pre_post_init} is not supported by the HIP path "
                               "(shipped: linear, linear_fixed; the mel initialisations need librosa).")
        if rk.get("positional_embedding", "train") not in ("train", "fixed"):
            raise RuntimeError(f"rnnformer_kwargs.positional_embedding={rk.get('positional_embedding')} is not supported by "
                               "the HIP path (shipped: train; the kernel always adds rf_block.0.pe).")
        if mask is not None:
            raise RuntimeError(f"model_kwargs.mask={mask} is not supported by the HIP path (every shipped yaml uses null).")
        if activation != "SiLU":
            raise RuntimeError(f"model_kwargs.activation={activation} is not supported by the HIP path (shipped: SiLU).")
        if window != "hann":
            raise RuntimeError(f"model_kwargs.window={window} is not supported by the HIP path (shipped: hann).")
        if resnet:
            raise RuntimeError("model_kwargs.resnet=True is not supported by the HIP path (shipped: False).")
        for flag in ("attn_bias", "post_act", "pre_norm"):
            if rk.get(flag, False):
                raise RuntimeError(f"rnnformer_kwargs.{flag}=True is not supported by the HIP path (shipped: False).")
        if rk.get("p_dropout", 0.0) != 0.0:
            raise RuntimeError("dropout is a training-time option; inference path expects p_dropout=0")
        return FEConfig(
            channels=int(channels), kernel_size=tuple(int(k) for k in kernel_size), stride=int(stride),
            rf_blocks=int(rk.get("num_blocks", 3)), rf_channels=int(rk.get("channels", 32)),
            rf_freq=int(rk.get("freq", 32)), rf_heads=int(rk.get("num_heads", 4)),
            rf_eps=float(rk.get("eps", 1e-8)), positional_embedding=rk.get("positional_embedding", "train"),
            n_fft=int(n_fft), hop_size=int(hop_size), win_size=int(win_size),
            input_compression=float(input_compression), weight_norm=bool(weight_norm),
            normalize_final_conv=bool(normalize_final_conv), pre_post_init=pre_post_init,
        )


def time_kernel_config(channels: int = 64, kernel_size_freq: Sequence[int] = (8, 3, 3), kernel_size_time: int = 3, stride: int = 4,
                       rnnformer_kwargs: Optional[Dict[str, Any]] = None, activation: str = "ReLU",
                       activation_kwargs: Optional[Dict[str, Any]] = None, n_fft: int = 512, hop_size: int = 160,
                       win_size: int = 400, window: Optional[str] = "povey", stft_normalized: bool = False,
                       mask: Optional[str] = None, input_compression: float = 0.25, weight_norm: bool = False,
                       final_scale: Any = "exp", normalize_final_conv: bool = False,
                       pre_post_init: Optional[str] = None) -> FEConfig:
    """yaml model_kwargs of `model: fastenhancer.time_kernel` (configs/ablation/time_kernel_b.yaml:2-29; defaults of
    models/fastenhancer/time_kernel/model.py:503-524) -> FEConfig with kernel_size_time set."""
    if final_scale not in (True, False, "exp"):
        raise AssertionError(f"final_scale={final_scale}")
    if int(kernel_size_time) < 1:
        raise AssertionError(f"kernel_size_time={kernel_size_time}")
    base = FEConfig.from_model_kwargs(channels=channels, kernel_size=kernel_size_freq, stride=stride, rnnformer_kwargs=rnnformer_kwargs,
                                      activation=activation, activation_kwargs=activation_kwargs, n_fft=n_fft, hop_size=hop_size,
                                      win_size=win_size, window=window, stft_normalized=stft_normalized, mask=mask,
                                      input_compression=input_compression, weight_norm=weight_norm,
                                      normalize_final_conv=normalize_final_conv, pre_post_init=pre_post_init, resnet=False)
    import dataclasses
    return dataclasses.replace(base, kernel_size_time=int(kernel_size_time), final_scale_exp=(final_scale == "exp"))


def noncausal_config(normalize_final_conv: bool = True, **model_kwargs) -> FEConfig:
    """yaml model_kwargs of `model: fastenhancer.noncausal` (configs/fastenhancer_dns/huge_noncausal.yaml:2-30: the default model's keys;
    defaults of models/fastenhancer/noncausal/model.py:349-368 - normalize_final_conv defaults to True there) -> FEConfig with noncausal set."""
    base = FEConfig.from_model_kwargs(normalize_final_conv=normalize_final_conv, **model_kwargs)
    import dataclasses
    return dataclasses.replace(base, noncausal=True)


def dprnn_config(channels: int = 64, kernel_size: Sequence[int] = (8, 3, 3), stride: int = 4, dprnn_kwargs: Optional[Dict[str, Any]] = None,
                 activation: str = "ReLU", activation_kwargs: Optional[Dict[str, Any]] = None, n_fft: int = 512, hop_size: int = 256,
                 win_size: int = 512, window: Optional[str] = "hann", stft_normalized: bool = False, mask: Optional[str] = None,
                 input_compression: float = 0.3, weight_norm: bool = False, final_scale: Any = "exp", normalize_final_conv: bool = False,
                 pre_post_init: Optional[str] = None) -> FEConfig:
    """yaml model_kwargs of `model: fastenhancer.dprnn` (configs/ablation/dprnn_b.yaml:2-27; defaults of
    models/fastenhancer/dprnn/model.py:327-357) -> FEConfig with channels_frnn set."""
    if final_scale not in (True, False, "exp"):
        raise AssertionError(f"final_scale={final_scale}")
    dk = dict(dprnn_kwargs or {})
    if dk.get("pre_norm", False):
        raise RuntimeError("dprnn_kwargs.pre_norm=True is not supported by the HIP path (shipped: False).")
    C2, H = int(dk.get("channels", 32)), int(dk.get("channels_frnn", 16))
    if 2 * H != C2:
        raise RuntimeError(f"dprnn_kwargs.channels_frnn={H} is not supported by the HIP path: the kernels are built for "
                           f"channels_frnn = channels / 2 (every shipped dprnn yaml; channels={C2}).")
    rk = dict(num_blocks=dk.get("num_blocks", 3), channels=C2, freq=dk.get("freq", 32), num_heads=4, eps=dk.get("eps", 1e-5))
    base = FEConfig.from_model_kwargs(channels=channels, kernel_size=kernel_size, stride=stride, rnnformer_kwargs=rk,
                                      activation=activation, activation_kwargs=activation_kwargs, n_fft=n_fft, hop_size=hop_size,
                                      win_size=win_size, window=window, stft_normalized=stft_normalized, mask=mask,
                                      input_compression=input_compression, weight_norm=weight_norm,
                                      normalize_final_conv=normalize_final_conv, pre_post_init=pre_post_init, resnet=False)
    import dataclasses
    return dataclasses.replace(base, channels_frnn=H, positional_embedding=None, final_scale_exp=(final_scale == "exp"))


def ln_config(final_scale: Any = "exp", final_scale_init: str = "1/sqrt(fan_in)", **model_kwargs) -> FEConfig:
    """yaml model_kwargs of `model: fastenhancer.ln` (configs/ablation/ln_b.yaml:2-31: the default model's keys + final_scale,
    final_scale_init) -> FEConfig with ln set."""
    if final_scale not in (True, False, "exp"):
        raise AssertionError(f"final_scale={final_scale}")
    base = FEConfig.from_model_kwargs(**model_kwargs)
    import dataclasses
    return dataclasses.replace(base, ln=True, final_scale_exp=(final_scale == "exp"))


def dpt_config(channels: int = 64, kernel_size: Sequence[int] = (8, 3, 3), stride: int = 4, dpt_kwargs: Optional[Dict[str, Any]] = None,
               activation: str = "ReLU", activation_kwargs: Optional[Dict[str, Any]] = None, n_fft: int = 512, hop_size: int = 256,
               win_size: int = 512, window: Optional[str] = "hann", stft_normalized: bool = False, mask: Optional[str] = None,
               input_compression: float = 0.3, weight_norm: bool = False, final_scale: Any = "exp", normalize_final_conv: bool = False,
               final_scale_init: str = "1/sqrt(fan_in)", pre_post_init: Optional[str] = None) -> FEConfig:
    """yaml model_kwargs of `model: fastenhancer.dptransformer` (configs/ablation/dpt_b.yaml:2-31; defaults of
    models/fastenhancer/dptransformer/model.py:408-419, 520-545) -> FEConfig with lookbehind set.  (final_scale_init only
    shapes the initial value of the final conv's scale.)"""
    if final_scale not in (True, False, "exp"):
        raise AssertionError(f"final_scale={final_scale}")
    dk = dict(dpt_kwargs or {})
    if dk.get("pre_norm", True):
        raise RuntimeError("dpt_kwargs.pre_norm=True (the reference's default) is not supported by the HIP path (shipped yamls: False).")
    L = int(dk.get("lookbehind", 16))
    if L != 31:
        raise RuntimeError(f"dpt_kwargs.lookbehind={L} is not supported by the HIP path (every shipped dpt yaml uses 31).")
    rk = {k: v for k, v in dk.items() if k != "lookbehind"}
    rk.setdefault("eps", 1e-8)
    base = FEConfig.from_model_kwargs(channels=channels, kernel_size=kernel_size, stride=stride, rnnformer_kwargs=rk,
                                      activation=activation, activation_kwargs=activation_kwargs, n_fft=n_fft, hop_size=hop_size,
                                      win_size=win_size, window=window, stft_normalized=stft_normalized, mask=mask,
                                      input_compression=input_compression, weight_norm=weight_norm,
                                      normalize_final_conv=normalize_final_conv, pre_post_init=pre_post_init, resnet=False)
    import dataclasses
    return dataclasses.replace(base, lookbehind=L, final_scale_exp=(final_scale == "exp"))


BSRNN_SUBBANDS = (2,) + (3,) * 10 + (8,) * 12 + (16,) * 7 + (17,)     # models/bsrnn/model.py:107-111


@dataclass(frozen=True)
class BSRNNConfig:
    """yaml model_kwargs of `model: bsrnn` (configs/others/bsrnn_xt.yaml:2-11; models/bsrnn/model.py:261-272)."""
    num_channels: int = 16
    num_layers: int = 6
    bias: bool = True
    affine: bool = True
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @property
    def F0(self) -> int:
        return self.n_fft // 2

    @property
    def cache_len(self) -> int:
        return self.n_fft - self.hop_size

    @property
    def n_bands(self) -> int:
        return len(BSRNN_SUBBANDS)

    @property
    def hidden(self) -> int:
        return 2 * self.num_channels

    @staticmethod
    def from_model_kwargs(num_channels: int = 16, num_layers: int = 6, bias: bool = True, affine: bool = True,
                          n_fft: int = 512, hop_size: int = 256, win_size: int = 512, window: str = "hann",
                          input_compression: float = 0.3, onnx: bool = True) -> "BSRNNConfig":
        if n_fft != 512:
            raise RuntimeError(f"Only n_fft=512 is supported, but given {n_fft}")       # models/bsrnn/model.py:112-113
        if window != "hann":
            raise RuntimeError(f"model_kwargs.window={window} is not supported by the HIP path (shipped: hann).")
        if n_fft < win_size:
            raise AssertionError(f"n_fft({n_fft}) must be bigger than win_size({win_size})")
        return BSRNNConfig(int(num_channels), int(num_layers), bool(bias), bool(affine), int(n_fft), int(hop_size),
                           int(win_size), float(input_compression))


@dataclass(frozen=True)
class FSPENConfig:
    """yaml model_kwargs of `model: fspen` (configs/others/fspen.yaml:2-16; models/fspen/model.py:201-212, DPEConfig :191-197).
    One architecture is compiled (the yaml's); fe_create rejects any other."""
    channels: tuple = (4, 16, 32)
    kernel_size: tuple = (6, 8, 6)
    stride: tuple = (2, 2, 2)
    num_blocks: int = 3
    dpe_channels: int = 16
    freq: int = 32
    groups: int = 8
    norm: str = "LayerNorm-FreqChannels"
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @property
    def F0(self) -> int:
        return self.n_fft // 2

    @property
    def cache_len(self) -> int:
        return self.n_fft - self.hop_size

    @property
    def n_caches(self) -> int:
        return self.num_blocks * self.groups

    @staticmethod
    def from_model_kwargs(channels=(4, 16, 32), kernel_size=(6, 8, 6), stride=(2, 2, 2), dpe_kwargs=None, n_fft: int = 512,
                          hop_size: int = 256, win_size: int = 512, window: str = "hann",
                          input_compression: float = 0.3) -> "FSPENConfig":
        if n_fft != 512:
            raise AssertionError(f"Only n_fft == 512 is allowed, but given {n_fft}.")      # models/fspen/model.py:214
        if window != "hann":
            raise RuntimeError(f"model_kwargs.window={window} is not supported by the HIP path (shipped: hann).")
        d = dict(dpe_kwargs or {})
        norm = d.get("norm", "LayerNorm-FreqChannels")
        if norm not in ("LayerNorm-FreqChannels", "LayerNorm-Channels", "CustomLayerNorm"):
            raise RuntimeError(f"dpe_kwargs.norm {norm} is not supported")               # models/fspen/model.py:156-157
        if norm != "LayerNorm-FreqChannels":
            raise RuntimeError(f"dpe_kwargs.norm {norm} is not supported by the HIP path (shipped: LayerNorm-FreqChannels).")
        return FSPENConfig(tuple(int(c) for c in channels), tuple(int(k) for k in kernel_size), tuple(int(x) for x in stride),
                           int(d.get("num_blocks", 3)), int(d.get("channels", 16)), int(d.get("freq", 32)), int(d.get("groups", 8)),
                           norm, int(n_fft), int(hop_size), int(win_size), float(input_compression))


@dataclass(frozen=True)
class LiSenNetConfig:
    """yaml model_kwargs of `model: lisennet` (configs/others/lisennet.yaml:2-8; models/lisennet/model.py:313-323).
    One architecture is compiled (the yaml's: 16 channels, 2 blocks, n_fft 512, hop 256); fe_create rejects any other."""
    num_channels: int = 16
    n_blocks: int = 2
    n_fft: int = 512
    hop_size: int = 256
    win_size: int = 512
    input_compression: float = 0.3

    @property
    def F0(self) -> int:
        return self.n_fft // 2

    @property
    def cache_len(self) -> int:
        return self.n_fft - self.hop_size

    @property
    def hidden(self) -> int:
        return self.num_channels // 2 * 3

    @property
    def n_caches(self) -> int:
        return 1 + 3 + 2 * self.n_blocks + 1

    def cache_shapes(self, B: int):
        """ONNXModel.initialize_cache (models/lisennet/model.py:380-396), sized for B streams"""
        C, F = self.num_channels, self.n_fft // 2 + 1
        sh = [(B, 1, F), (B, C // 4, 1, F), (B, C // 2, 1, F // 2), (B, C // 4 * 3, 1, F // 4)]
        for _ in range(self.n_blocks):
            sh += [(1, B * (F // 8), self.hidden), (B, 2 * C, 2, F // 8)]
        sh.append((B, C // 4, 1, F - 1))
        return sh

    @staticmethod
    def from_model_kwargs(num_channels: int = 16, n_blocks: int = 2, n_fft: int = 512, hop_size: int = 256, win_size: int = 512,
                          window: tp.Optional[str] = "hann", input_compression: float = 0.3, normalized: bool = False) -> "LiSenNetConfig":
        if window != "hann":
            raise RuntimeError(f"model_kwargs.window={window} is not supported by the HIP path (shipped: hann).")
        if normalized:
            raise RuntimeError("model_kwargs.normalized=True is not supported by the HIP path (shipped: False).")
        if n_fft < win_size:
            raise AssertionError(f"n_fft({n_fft}) must be bigger than win_size({win_size})")
        return LiSenNetConfig(int(num_channels), int(n_blocks), int(n_fft), int(hop_size), int(win_size), float(input_compression))
