"""Shared helpers for the tests: the shipped model_kwargs (read from tests/golden/yaml_kwargs.json, the
reference's yaml values dumped verbatim by tools/dump_yaml_kwargs.py - data, not code) and golden loading."""
import json
import os

import numpy as np

from oracle.fe_oracle import FEConfig, FEOracle, fold_state_dict
from oracle.weightgen import make_input, make_training_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# {yaml path -> model, model_kwargs, sampling_rate} of every shipped config of the reference, dumped by
# tools/dump_yaml_kwargs.py in the authoring container (data: the yaml values verbatim; nothing is restated by hand)
YAML_KWARGS = json.load(open(os.path.join(GOLDEN_DIR, "yaml_kwargs.json")))

# test name -> (yaml, golden seed[, model_kwargs overrides])
_YAMLS = {
    "fe_t": ("configs/fastenhancer/t.yaml", 101),
    "fe_b": ("configs/fastenhancer/b.yaml", 102),
    "fe_s": ("configs/fastenhancer/s.yaml", 105),
    "fe_m": ("configs/fastenhancer/m.yaml", 106),
    "fe_l": ("configs/fastenhancer/l.yaml", 103),
    "fe48_b": ("configs/fastenhancer_48khz/b.yaml", 104),
    "fe48_t": ("configs/fastenhancer_48khz/t.yaml", 107),
    "fe48_s": ("configs/fastenhancer_48khz/s.yaml", 108),
    "fe48_m": ("configs/fastenhancer_48khz/m.yaml", 109),
    "fe48_l": ("configs/fastenhancer_48khz/l.yaml", 110),
    "fe48_b_h480": ("configs/fastenhancer_48khz/b.yaml", 111, {"hop_size": 480}),   # BASELINE config 4's "hop=480"
    "fe_tk_b": ("configs/ablation/time_kernel_b.yaml", 120),
    "fe_dprnn_t": ("configs/ablation/dprnn_t.yaml", 130),
    "fe_dprnn_b": ("configs/ablation/dprnn_b.yaml", 131),
    "fe_dprnn_s": ("configs/ablation/dprnn_s.yaml", 133),
    "fe_dprnn_m": ("configs/ablation/dprnn_m.yaml", 134),
    "fe_dprnn_l": ("configs/ablation/dprnn_l.yaml", 132),
    "fe_ln_b": ("configs/ablation/ln_b.yaml", 150),
    "fe_dpt_t": ("configs/ablation/dpt_t.yaml", 140),
    "fe_dpt_b": ("configs/ablation/dpt_b.yaml", 141),
    "fe_dpt_s": ("configs/ablation/dpt_s.yaml", 143),
    "fe_dpt_m": ("configs/ablation/dpt_m.yaml", 142),
    # model: fastenhancer.noncausal (bidirectional GRU over time, offline Model only)
    "fe_nc": ("configs/fastenhancer_dns/huge_noncausal.yaml", 160),
    "fe_nc24": ("configs/fastenhancer_dns/huge_noncausal_24khz.yaml", 161),
    "fe48_nc": ("configs/fastenhancer_48khz/huge_noncausal.yaml", 162),
}


def _entry(name):
    path, seed = _YAMLS[name][:2]
    y = YAML_KWARGS[path]
    kw = json.loads(json.dumps(y["model_kwargs"]))      # (a private deep copy)
    if len(_YAMLS[name]) > 2:
        kw.update(_YAMLS[name][2])
    return kw, int(y["sampling_rate"]), seed


# name -> (model_kwargs, sampling rate, golden seed)
MODEL_KWARGS = {name: _entry(name) for name in _YAMLS}
# which module of the reference a name belongs to (the yaml's `model:` key)
MODEL_MODULE = {name: YAML_KWARGS[_YAMLS[name][0]]["model"] for name in _YAMLS}
NONCAUSAL = tuple(n for n in _YAMLS if MODEL_MODULE[n] == "fastenhancer.noncausal")


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
def _other(path, seed):
    y = YAML_KWARGS[path]
    return json.loads(json.dumps(y["model_kwargs"])), int(y["sampling_rate"]), seed


BSRNN_KWARGS = {
    "bsrnn_xxt": _other("configs/others/bsrnn_xxt.yaml", 202),
    "bsrnn_xt": _other("configs/others/bsrnn_xt.yaml", 201),
    "bsrnn_s": _other("configs/others/bsrnn_s.yaml", 204),
    "bsrnn_t": _other("configs/others/bsrnn_t.yaml", 203),
}


def build_bsrnn_oracle(name, dtype=np.float32):
    from oracle import bsrnn_oracle as bo
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg = bo.BSRNNConfig.from_model_kwargs(kw)
    sd = bo.make_training_state_dict(cfg, seed)
    fused = bo.fold_state_dict(sd, cfg)
    return cfg, sd, fused, bo.BSRNNOracle(cfg, fused, dtype)


# configs/others/fspen.yaml:2-16
FSPEN_KWARGS = _other("configs/others/fspen.yaml", 301)


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
    if MODEL_MODULE[name] == "fastenhancer.noncausal":
        from fastenhancer_amd.config import noncausal_config
        return noncausal_config(**kw)
    return time_kernel_config(**kw) if MODEL_MODULE[name] == "fastenhancer.time_kernel" else PCfg.from_model_kwargs(**kw)


# configs/others/lisennet.yaml:2-8
LISENNET_KWARGS = _other("configs/others/lisennet.yaml", 401)


def build_lisennet_oracle(dtype=np.float32):
    from oracle import lisennet_oracle as lo
    kw, sr, seed = LISENNET_KWARGS
    cfg = lo.LiSenNetConfig.from_model_kwargs(kw)
    sd = lo.make_state_dict(cfg, seed)
    return cfg, sd, sd, lo.LiSenNetOracle(cfg, sd, dtype)
