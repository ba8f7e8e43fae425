"""Shared helpers for the tests: the shipped model_kwargs of the BASELINE configs
(restated from the reference yamls, e.g. configs/fastenhancer/b.yaml:2-29 — data,
not code) and golden loading."""
import os

import numpy as np

from oracle.fe_oracle import FEConfig, FEOracle, fold_state_dict
from oracle.weightgen import make_input, make_training_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _kw(C1, ks, C2, F2, K, N, H, init, eps=1.0e-5):
    return dict(
        channels=C1, kernel_size=list(ks), stride=4,
        rnnformer_kwargs=dict(num_blocks=K, channels=C2, freq=F2, num_heads=4, eps=eps,
                              positional_embedding="train", attn_bias=False, post_act=False, pre_norm=False),
        pre_post_init=init, n_fft=N, hop_size=H, win_size=N, window="hann", stft_normalized=False,
        mask=None, activation="SiLU", activation_kwargs=dict(inplace=True), input_compression=0.3,
        normalize_final_conv=True, weight_norm=True, resnet=False)


def _kw_tk(C1, ksf, kt, C2, F2, K, N, H, init, eps=1.0e-5):
    """configs/ablation/time_kernel_b.yaml:2-29 (model: fastenhancer.time_kernel)"""
    kw = _kw(C1, ksf, C2, F2, K, N, H, init, eps)
    del kw["kernel_size"], kw["resnet"]
    kw.update(kernel_size_freq=list(ksf), kernel_size_time=kt, final_scale=True)
    return kw


def _kw_dprnn(C1, ks, C2, H2, F2, K, N, H):
    """configs/ablation/dprnn_b.yaml:2-27 (model: fastenhancer.dprnn)"""
    return dict(
        channels=C1, kernel_size=list(ks), stride=4,
        dprnn_kwargs=dict(num_blocks=K, channels=C2, channels_frnn=H2, freq=F2, eps=1.0e-5, pre_norm=False),
        pre_post_init="linear_fixed", n_fft=N, hop_size=H, win_size=N, window="hann", stft_normalized=False,
        mask=None, activation="SiLU", activation_kwargs=dict(inplace=True), input_compression=0.3, final_scale=True,
        normalize_final_conv=True, weight_norm=True)


def _kw_dpt(C1, ks, C2, F2, K, N, H):
    """configs/ablation/dpt_b.yaml:2-31 (model: fastenhancer.dptransformer)"""
    return dict(
        channels=C1, kernel_size=list(ks), stride=4,
        dpt_kwargs=dict(num_blocks=K, channels=C2, freq=F2, num_heads=4, eps=1.0e-5, positional_embedding="train", attn_bias=False,
                        post_act=False, pre_norm=False, lookbehind=31),
        pre_post_init="linear_fixed", n_fft=N, hop_size=H, win_size=N, window="hann", stft_normalized=False,
        mask=None, activation="SiLU", activation_kwargs=dict(inplace=True), input_compression=0.3, final_scale=True,
        normalize_final_conv=True, final_scale_init="one", weight_norm=True)


# name -> (model_kwargs, sampling rate, golden seed)
MODEL_KWARGS = {
    "fe_t": (_kw(24, (8, 3, 3), 20, 16, 2, 512, 256, "linear_fixed"), 16000, 101),
    "fe_b": (_kw(48, (8, 3, 3), 36, 24, 3, 512, 256, "linear_fixed"), 16000, 102),
    "fe_s": (_kw(64, (8, 3, 3, 3), 48, 36, 3, 512, 256, "linear_fixed"), 16000, 105),
    "fe_m": (_kw(96, (8, 3, 3, 3), 72, 48, 4, 512, 160, "linear_fixed"), 16000, 106),
    "fe_l": (_kw(128, (8, 3, 3, 3, 3), 96, 64, 5, 512, 100, "linear_fixed"), 16000, 103),
    "fe48_b": (_kw(48, (8, 3, 3), 36, 36, 3, 1024, 512, "linear"), 48000, 104),
    "fe48_t": (_kw(24, (8, 3, 3), 20, 24, 2, 1024, 512, "linear"), 48000, 107),
    "fe48_s": (_kw(64, (8, 3, 3, 3), 48, 48, 3, 1024, 512, "linear"), 48000, 108),
    "fe48_m": (_kw(96, (8, 3, 3, 3), 72, 72, 4, 1024, 320, "linear"), 48000, 109),
    "fe48_l": (_kw(128, (8, 3, 3, 3, 3), 96, 96, 5, 1024, 200, "linear"), 48000, 110),
    "fe48_b_h480": (_kw(48, (8, 3, 3), 36, 36, 3, 1024, 480, "linear"), 48000, 111),   # BASELINE config 4's "hop=480"
    "fe_tk_b": (_kw_tk(48, (8, 3, 3), 3, 36, 24, 3, 512, 256, "linear_fixed"), 16000, 120),   # configs/ablation/time_kernel_b.yaml
    # configs/ablation/dprnn_{t,b,s,m,l}.yaml
    "fe_dprnn_t": (_kw_dprnn(24, (8, 3, 3), 20, 10, 16, 2, 512, 256), 16000, 130),
    "fe_dprnn_b": (_kw_dprnn(48, (8, 3, 3), 36, 18, 24, 3, 512, 256), 16000, 131),
    "fe_dprnn_s": (_kw_dprnn(64, (8, 3, 3, 3), 48, 24, 36, 3, 512, 256), 16000, 133),
    "fe_dprnn_m": (_kw_dprnn(96, (8, 3, 3, 3), 72, 36, 48, 4, 512, 160), 16000, 134),
    "fe_dprnn_l": (_kw_dprnn(128, (8, 3, 3, 3, 3), 96, 48, 64, 5, 512, 100), 16000, 132),
    # configs/ablation/ln_b.yaml (model: fastenhancer.ln - the default model's kwargs + final_scale / final_scale_init)
    "fe_ln_b": ({**{k: v for k, v in _kw(48, (8, 3, 3), 36, 24, 3, 512, 256, "linear_fixed").items()}, "final_scale": True, "final_scale_init": "one"}, 16000, 150),
    # configs/ablation/dpt_{t,b,s,m}.yaml
    "fe_dpt_t": (_kw_dpt(24, (8, 3, 3), 20, 16, 2, 512, 256), 16000, 140),
    "fe_dpt_b": (_kw_dpt(48, (8, 3, 3), 36, 24, 3, 512, 256), 16000, 141),
    "fe_dpt_s": (_kw_dpt(64, (8, 3, 3, 3), 48, 36, 3, 512, 256), 16000, 143),
    "fe_dpt_m": (_kw_dpt(96, (8, 3, 3, 3), 72, 48, 4, 512, 160), 16000, 142),
}
# which module of the reference a name belongs to (the yaml's `model:` key)
MODEL_MODULE = {name: ("fastenhancer.dprnn" if "dprnn" in name else "fastenhancer.dptransformer" if "dpt" in name else
                       "fastenhancer.ln" if name == "fe_ln_b" else "fastenhancer.default")
                for name in MODEL_KWARGS}
MODEL_MODULE["fe_tk_b"] = "fastenhancer.time_kernel"


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def build_oracle(name, dtype=np.float32):
    kw, sr, seed = MODEL_KWARGS[name]
    cfg = FEConfig.from_model_kwargs(kw, variant=MODEL_MODULE[name].split(".")[-1])
    sd = make_training_state_dict(cfg, seed)
    fused = fold_state_dict(sd, cfg)
    return cfg, sd, fused, FEOracle(cfg, fused, dtype)


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, np.float64)))))


# ---------------------------------------------------------------- BSRNN (models/bsrnn, configs/others/bsrnn_*.yaml)
BSRNN_KWARGS = {
    "bsrnn_xxt": (dict(num_channels=16, num_layers=2, bias=True, affine=True, n_fft=512, hop_size=256, win_size=512, window="hann"), 16000, 202),
    "bsrnn_xt": (dict(num_channels=16, num_layers=6, bias=True, affine=True, n_fft=512, hop_size=256, win_size=512, window="hann"), 16000, 201),
    "bsrnn_s": (dict(num_channels=64, num_layers=6, bias=True, affine=True, n_fft=512, hop_size=256, win_size=512, window="hann"), 16000, 204),
    "bsrnn_t": (dict(num_channels=32, num_layers=6, bias=True, affine=True, n_fft=512, hop_size=256, win_size=512, window="hann"), 16000, 203),
}


def build_bsrnn_oracle(name, dtype=np.float32):
    from oracle import bsrnn_oracle as bo
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg = bo.BSRNNConfig.from_model_kwargs(kw)
    sd = bo.make_training_state_dict(cfg, seed)
    fused = bo.fold_state_dict(sd, cfg)
    return cfg, sd, fused, bo.BSRNNOracle(cfg, fused, dtype)


# configs/others/fspen.yaml:2-16
FSPEN_KWARGS = (dict(channels=[4, 16, 32], kernel_size=[6, 8, 6], stride=[2, 2, 2],
                     dpe_kwargs=dict(num_blocks=3, channels=16, freq=32, groups=8, norm="LayerNorm-FreqChannels"),
                     n_fft=512, hop_size=256, win_size=512, window="hann", input_compression=0.3), 16000, 301)


def build_fspen_oracle(dtype=np.float32):
    from oracle import fspen_oracle as fo
    kw, sr, seed = FSPEN_KWARGS
    cfg = fo.FSPENConfig.from_model_kwargs(kw)
    sd = fo.make_training_state_dict(cfg, seed)
    fused = fo.fold_state_dict(sd, cfg)
    return cfg, sd, fused, fo.FSPENOracle(cfg, fused, dtype)


def product_config(name):
    """the HIP path's FEConfig for a MODEL_KWARGS entry (the time_kernel variant's yaml has its own keys)"""
    from fastenhancer_amd.config import FEConfig as PCfg, dprnn_config, dpt_config, time_kernel_config
    kw = MODEL_KWARGS[name][0]
    if MODEL_MODULE[name] == "fastenhancer.dprnn":
        return dprnn_config(**kw)
    if MODEL_MODULE[name] == "fastenhancer.dptransformer":
        return dpt_config(**kw)
    if MODEL_MODULE[name] == "fastenhancer.ln":
        from fastenhancer_amd.config import ln_config
        return ln_config(**kw)
    return time_kernel_config(**kw) if MODEL_MODULE[name] == "fastenhancer.time_kernel" else PCfg.from_model_kwargs(**kw)


# configs/others/lisennet.yaml:2-8
LISENNET_KWARGS = (dict(num_channels=16, n_blocks=2, n_fft=512, hop_size=256, win_size=512, input_compression=0.3), 16000, 401)


def build_lisennet_oracle(dtype=np.float32):
    from oracle import lisennet_oracle as lo
    kw, sr, seed = LISENNET_KWARGS
    cfg = lo.LiSenNetConfig.from_model_kwargs(kw)
    sd = lo.make_state_dict(cfg, seed)
    return cfg, sd, sd, lo.LiSenNetOracle(cfg, sd, dtype)
