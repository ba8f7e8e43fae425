#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the authoring container, where the reference checkout is mounted
read-only at /root/reference.  Nothing from the reference (source, bytecode,
pickled modules) is written to the repo: the fixtures are plain arrays (inputs
are regenerated from seeds, so only expected OUTPUTS are stored).

Import recipe, zero edits to reference files:
  * ``functional/__init__.py`` pulls ``librosa.filters`` -> a stub module is
    pre-inserted in ``sys.modules``;
  * Python 3.10 rejects ``tp.Tuple[Tensor, Tensor, ...]`` used as a return
    annotation in model.py -> the file is compiled with
    ``from __future__ import annotations`` semantics into a fresh module;
  * ``utils`` / ``wrappers`` / ``scripts.export_onnx`` are never imported
    (torchaudio / onnx / librosa are absent); the 10-line wav->wav composition of
    scripts/export_onnx.py:48-58 is driven step by step below.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import os
import sys
import types
import __future__ as _future

sys.dont_write_bytecode = True

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.fe_oracle import (FEConfig, canonical_key, fold_state_dict, reference_key, linear_filterbank, linear_filterbank_tk, stft_windows,  # noqa: E402
                              training_state_dict_spec)
from oracle.weightgen import make_input, make_training_state_dict  # noqa: E402


def import_reference_model(ref: str, rel: str, name: str) -> types.ModuleType:
    if "torchaudio" not in sys.modules:      # models/fastenhancer/time_kernel/model.py:11 (only its mel initialisation calls it)
        ta = types.ModuleType("torchaudio")
        taf = types.ModuleType("torchaudio.functional")
        taf.melscale_fbanks = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchaudio stub"))
        ta.functional = taf
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.functional"] = taf
    if "librosa" not in sys.modules:
        lib = types.ModuleType("librosa")
        filt = types.ModuleType("librosa.filters")
        filt.mel = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("librosa stub"))
        lib.filters = filt
        sys.modules["librosa"] = lib
        sys.modules["librosa.filters"] = filt
    if ref not in sys.path:
        sys.path.insert(0, ref)
    path = os.path.join(ref, rel)
    src = open(path).read()
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod      # dataclasses resolve cls.__module__ through sys.modules
    code = compile(src, path, "exec", flags=_future.annotations.compiler_flag, dont_inherit=True)
    exec(code, mod.__dict__)
    return mod


CONFIGS = {
    # name: (yaml, seed, B, hops, long_hops)
    "fe_t": ("configs/fastenhancer/t.yaml", 101, 2, 12, 200),
    "fe_b": ("configs/fastenhancer/b.yaml", 102, 2, 12, 200),
    "fe_m": ("configs/fastenhancer/m.yaml", 106, 1, 8, 0),
    "fe_l": ("configs/fastenhancer/l.yaml", 103, 1, 6, 0),
    "fe48_b": ("configs/fastenhancer_48khz/b.yaml", 104, 2, 8, 0),
    "fe48_l": ("configs/fastenhancer_48khz/l.yaml", 110, 1, 5, 0),
    # BASELINE.json config 4 words the 48 kHz case as "hop=480" (every shipped 48 kHz yaml uses 512): b.yaml with hop_size overridden
    "fe48_b_h480": ("configs/fastenhancer_48khz/b.yaml", 111, 2, 8, 0, {"hop_size": 480}),
    # the time_kernel ablation (causal Conv2d with a 3-frame time kernel and (kt-1)-frame caches)
    "fe_tk_b": ("configs/ablation/time_kernel_b.yaml", 120, 2, 10, 120),
    # the dprnn ablation (models/fastenhancer/dprnn): a bidirectional GRU over the sub-bands instead of the attention
    "fe_dprnn_t": ("configs/ablation/dprnn_t.yaml", 130, 2, 10, 0),
    "fe_dprnn_b": ("configs/ablation/dprnn_b.yaml", 131, 2, 10, 120),
    "fe_dprnn_l": ("configs/ablation/dprnn_l.yaml", 132, 1, 5, 0),
    # the dptransformer ablation (models/fastenhancer/dptransformer): causal attention over the last 31 frames instead of the time GRU
    # (40 hops: the K / V caches hold 31 frames)
    "fe_dpt_t": ("configs/ablation/dpt_t.yaml", 140, 2, 40, 0),
    "fe_dpt_b": ("configs/ablation/dpt_b.yaml", 141, 2, 40, 120),
    "fe_dpt_m": ("configs/ablation/dpt_m.yaml", 142, 1, 36, 0),
    # the ln ablation (models/fastenhancer/ln): GroupNorm / LayerNorm instead of the (folded) BatchNorms
    "fe_ln_b": ("configs/ablation/ln_b.yaml", 150, 2, 10, 120),
    # r3: the shipped shapes that were pinned on the oracle only (B = 1, 6 hops keeps each fixture small)
    "fe_s": ("configs/fastenhancer/s.yaml", 105, 1, 6, 0),
    "fe48_t": ("configs/fastenhancer_48khz/t.yaml", 107, 1, 6, 0),
    "fe48_s": ("configs/fastenhancer_48khz/s.yaml", 108, 1, 6, 0),
    "fe48_m": ("configs/fastenhancer_48khz/m.yaml", 109, 1, 6, 0),
    "fe_dprnn_s": ("configs/ablation/dprnn_s.yaml", 133, 1, 6, 0),
    "fe_dprnn_m": ("configs/ablation/dprnn_m.yaml", 134, 1, 6, 0),
    "fe_dpt_s": ("configs/ablation/dpt_s.yaml", 143, 1, 6, 0),
}


def cache_checksum(a) -> np.ndarray:
    """three float64 sums of a (large) cache tensor: plain, squared, and against a fixed position-dependent ramp - what the
    goldens hold for the K / V caches whose full tensors are not stored"""
    v = np.asarray(a, np.float64).reshape(-1)
    ramp = np.cos(0.37 * np.arange(v.size, dtype=np.float64) + 0.11)
    return np.array([v.sum(), (v * v).sum(), (v * ramp).sum()], np.float64)


def to_t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def gen_fastenhancer(ref: str, name: str, out_dir: str):
    rel_yaml, seed, B, hops, long_hops = CONFIGS[name][:5]
    hps = yaml.safe_load(open(os.path.join(ref, rel_yaml)))
    kw = hps["model_kwargs"]
    if len(CONFIGS[name]) > 5:
        kw.update(CONFIGS[name][5])
    sr = hps["data"]["sampling_rate"]
    cfg = FEConfig.from_model_kwargs(kw, variant=hps["model"].split(".")[-1])
    tk = hps["model"] == "fastenhancer.time_kernel"
    assert tk == cfg.time_kernel
    mod = import_reference_model(ref, f"models/{hps['model'].replace('.', '/')}/model.py", "ref_fe_model_" + hps["model"].split(".")[-1])
    torch.manual_seed(0)
    torch.set_num_threads(1)

    model = mod.Model(**kw).eval()           # offline, training form
    ref_sd = model.state_dict()
    spec = training_state_dict_spec(cfg)
    spec = {reference_key(k, cfg): v for k, v in spec.items()}
    assert list(ref_sd.keys()) == list(spec.keys()), (
        "state_dict schema drifted", [k for k in ref_sd if k not in spec], [k for k in spec if k not in ref_sd])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    sd = make_training_state_dict(cfg, seed)
    sd_ref_names = {reference_key(k, cfg): v for k, v in sd.items()}
    model.load_state_dict(to_t(sd_ref_names), strict=True)

    onnx_model = mod.ONNXModel(**kw).eval()  # streaming
    onnx_model.load_state_dict(to_t(sd_ref_names), strict=True)
    onnx_model.remove_weight_reparameterizations()
    fused_ref = {canonical_key(k): v.detach().numpy().copy() for k, v in onnx_model.state_dict().items()}

    # ---- fold check (a20): my restatement vs the reference's fused state_dict
    fused_mine = fold_state_dict(sd, cfg)
    fused_ref.pop("dec_post.2.scale", None)      # (the time_kernel variant keeps the - then unused - scale parameter in its state_dict)
    if cfg.ln:                                   # (no module is replaced there: the final conv keeps its index 3 and its folded-in scale)
        fused_ref = {k.replace("dec_post.3.", "dec_post.2."): v for k, v in fused_ref.items() if k != "dec_post.3.scale"}
    assert set(fused_mine) == set(fused_ref), (set(fused_mine) ^ set(fused_ref))
    worst = 0.0
    for k in fused_ref:
        d = np.abs(fused_mine[k] - fused_ref[k]).max() / (np.abs(fused_ref[k]).max() + 1e-12)
        worst = max(worst, d)
    assert worst < 2e-6, worst
    # windows (a1)
    w, wi = stft_windows(cfg.n_fft, cfg.hop_size, cfg.win_size)
    dw = np.abs(w - onnx_model.stft.window.numpy()).max()
    dwi = np.abs(wi - onnx_model.stft.window_istft.numpy()).max() / np.abs(wi).max()
    assert dw < 1e-6 and dwi < 1e-6, (dw, dwi)
    # fixed filterbank formula
    if kw.get("pre_post_init", None) == "linear_fixed":
        fresh = mod.ONNXModel(**kw)
        pre, post = (linear_filterbank_tk if tk or cfg.dprnn or cfg.dpt or cfg.ln else linear_filterbank)(cfg.F1, cfg.rf_freq)
        fpre, fpost = (fresh.dprnn_pre, fresh.dprnn_post) if cfg.dprnn else ((fresh.dpt_pre, fresh.dpt_post) if cfg.dpt else (fresh.rf_pre, fresh.rf_post))
        assert np.abs(pre - fpre[0].weight.numpy()).max() < 1e-5
        assert np.abs(post - fpost[0].weight.numpy()).max() < 1e-5

    out = {"seed": np.int64(seed), "B": np.int64(B), "hops": np.int64(hops), "sr": np.int64(sr),
           "fold_worst_rel": np.float64(worst)}
    H, N = cfg.hop_size, cfg.n_fft
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr))

    # ---- streaming wav->wav (a19): scripts/export_onnx.py:48-58 composition
    with torch.no_grad():
        caches = onnx_model.stft.initialize_cache(x)

        def model_caches(nb):       # the model's cache list sized for nb streams (its own initialize_cache is written for 1)
            hs = [torch.zeros(1, nb * cfg.rf_freq, cfg.rf_channels) for _ in range(cfg.rf_blocks)]
            if cfg.dpt:
                hs = [torch.zeros(nb * cfg.rf_freq, cfg.rf_heads, cfg.lookbehind, cfg.rf_channels // cfg.rf_heads) for _ in range(2 * cfg.rf_blocks)]
            if not tk:
                return hs
            cc = lambda: [torch.zeros(nb, cfg.channels, cfg.kernel_size_time - 1, cfg.F1) for _ in range(cfg.n_layers)]
            return cc() + hs + cc()
        caches += model_caches(B)
        cache_stft, cache_istft, *h = caches
        outs, specs_in, specs_out = [], [], []
        for t in range(hops):
            wav_in = x[:, t * H:(t + 1) * H]
            spec_in, cache_stft = onnx_model.stft(wav_in, cache_stft)
            spec_out, *h = onnx_model(spec_in, *h)
            wav_out, cache_istft = onnx_model.stft.inverse(spec_out, cache_istft)
            outs.append(wav_out.numpy().copy())
            specs_in.append(spec_in.numpy().copy())
            specs_out.append(spec_out.numpy().copy())
    out["stream_wav_out"] = np.stack(outs, 0)                    # [hops,B,H]
    out["stream_cache_stft"] = cache_stft.numpy().copy()
    out["stream_cache_istft"] = cache_istft.numpy().copy()
    for k, hk in enumerate(h):
        if cfg.dpt and name != "fe_dpt_t" and 2 <= k < len(h) - 2:
            out[f"stream_h{k}_chk"] = cache_checksum(hk.numpy())
            continue          # (K / V caches are large: all blocks for dpt_t; the first and the last block's + checksums of the others for the bigger shapes)
        out[f"stream_h{k}"] = hk.numpy().copy()
    out["stream_spec_in_last"] = specs_in[-1]
    out["stream_spec_out_last"] = specs_out[-1]

    # ---- spec->spec with a T-frame chunk (a4..a17; model.py:677-710), T=4 from zero state
    with torch.no_grad():
        spec_chunk = torch.from_numpy(np.concatenate(specs_in[:4], axis=2))   # [B,F0+1,4,2]
        h0 = model_caches(B)
        # (the dptransformer variant's cached branch is written for T = 1: its chunk runs without caches, i.e. with the start masked)
        spec_hat, *h4 = onnx_model(spec_chunk) if cfg.dpt else onnx_model(spec_chunk, *h0)
    out["chunk_spec_out"] = spec_hat.numpy().copy()
    out["chunk_h_last"] = h4[-1].numpy().copy()

    # ---- offline Model.forward (a21, a27): Tw = hops*H + 37 exercises the Tw//H truncation
    xo = torch.from_numpy(make_input(B, hops * H + 37, seed + 2000, sr))
    with torch.no_grad():
        wav_hat, spec_hat = model(xo)
    out["offline_wav"] = wav_hat.numpy().copy()
    out["offline_spec"] = spec_hat.numpy().copy()

    # ---- long single-stream run through the driver loop of scripts/test_onnx.py:11-60 (a26)
    if long_hops:
        length = long_hops * H - 3 * H // 2 + 5            # deliberately not a multiple of H
        xl = make_input(1, length, seed + 3000, sr)
        wav = np.clip(xl, -1, 1)
        wav = np.pad(wav, ((0, 0), (0, N)))
        with torch.no_grad():
            cache_stft, cache_istft = onnx_model.stft.initialize_cache(torch.zeros(1, 1))
            h = model_caches(1)
            chunks = []
            for idx in range(0, length + N - H, H):
                wav_in = torch.from_numpy(wav[:, idx:idx + H])
                spec_in, cache_stft = onnx_model.stft(wav_in, cache_stft)
                spec_out, *h = onnx_model(spec_in, *h)
                wav_out, cache_istft = onnx_model.stft.inverse(spec_out, cache_istft)
                chunks.append(wav_out.numpy()[0].copy())
        full = np.concatenate(chunks, 0)
        s = N - H
        out["long_length"] = np.int64(length)
        out["long_wav_out"] = np.clip(full[s:s + length], -1.0, 1.0).astype(np.float32)

    # ---- per-stage activations: only for the smallest config
    if name == "fe_t":
        for k in fused_ref:
            out["fused." + k] = fused_ref[k]

    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    rms_in = float(np.sqrt((x.numpy() ** 2).mean()))
    rms_out = float(np.sqrt((out["stream_wav_out"][4:] ** 2).mean()))
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) fold_worst={worst:.2e} "
          f"in_rms={rms_in:.3f} out_rms={rms_out:.3f}")


NONCAUSAL_CONFIGS = {
    # name: (yaml, seed, B, hops) - model: fastenhancer.noncausal (offline Model only)
    "fe_nc": ("configs/fastenhancer_dns/huge_noncausal.yaml", 160, 2, 14),
    "fe_nc24": ("configs/fastenhancer_dns/huge_noncausal_24khz.yaml", 161, 2, 9),
    "fe48_nc": ("configs/fastenhancer_48khz/huge_noncausal.yaml", 162, 2, 10),
}


def gen_noncausal(ref: str, name: str, out_dir: str):
    """models/fastenhancer/noncausal/model.py: bidirectional GRU over time; `Model.forward` (:628-635) is the whole surface."""
    rel_yaml, seed, B, hops = NONCAUSAL_CONFIGS[name]
    hps = yaml.safe_load(open(os.path.join(ref, rel_yaml)))
    kw = hps["model_kwargs"]
    sr = hps["data"]["sampling_rate"]
    assert hps["model"] == "fastenhancer.noncausal"
    cfg = FEConfig.from_model_kwargs(kw, variant="noncausal")
    mod = import_reference_model(ref, "models/fastenhancer/noncausal/model.py", "ref_fe_model_noncausal")
    torch.manual_seed(0)
    torch.set_num_threads(4)
    model = mod.Model(**kw).eval()
    ref_sd = model.state_dict()
    spec = training_state_dict_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), (
        "state_dict schema drifted", [k for k in ref_sd if k not in spec], [k for k in spec if k not in ref_sd])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    if kw.get("pre_post_init", None) == "linear_fixed":       # the variant's own rf_pre_post_lin (:306-345, Hz axes with sr = 16 kHz)
        pre, post = linear_filterbank_tk(cfg.F1, cfg.rf_freq)
        assert np.abs(pre - model.rf_pre[0].weight.numpy()).max() < 1e-5
        assert np.abs(post - model.rf_post[0].weight.numpy()).max() < 1e-5
    sd = make_training_state_dict(cfg, seed)
    model.load_state_dict(to_t(sd), strict=True)
    H = cfg.hop_size
    xo = torch.from_numpy(make_input(B, hops * H + 37, seed + 2000, sr))
    with torch.no_grad():
        wav_hat, spec_hat = model(xo)                    # training form (weight norm, BatchNorm in eval mode)
        fused_model = mod.Model(**kw).eval()
        fused_model.load_state_dict(to_t(sd), strict=True)
        fused_model.remove_weight_reparameterizations()
        wav_f, spec_f = fused_model(xo)                  # deployment form
    fused_ref = {k: v.detach().numpy().copy() for k, v in fused_model.state_dict().items()}
    fused_ref = {k.replace("dec_post.2.scale", "dec_post.3.scale"): v for k, v in fused_ref.items()}
    fused_mine = fold_state_dict(sd, cfg)
    # (the reference keeps the final conv's `scale` and un-normalised weight in its state_dict and applies them in forward:
    #  the product of the two is what the fold emits)
    w = fused_ref.pop("dec_post.2.weight")
    scale = fused_ref.pop("dec_post.3.scale", None)
    if scale is not None:
        w = w / max(float(np.sqrt((w.astype(np.float32) ** 2).sum())), 1e-12) * scale if kw.get("normalize_final_conv", True) else w * scale
    fused_ref["dec_post.2.weight"] = w.astype(np.float32)
    assert set(fused_mine) == set(fused_ref), (set(fused_mine) ^ set(fused_ref))
    worst = max(np.abs(fused_mine[k] - fused_ref[k]).max() / (np.abs(fused_ref[k]).max() + 1e-12) for k in fused_ref)
    assert worst < 2e-6, worst
    out = {"seed": np.int64(seed), "B": np.int64(B), "hops": np.int64(hops), "sr": np.int64(sr), "fold_worst_rel": np.float64(worst),
           "offline_wav": wav_hat.numpy().copy(), "offline_spec": spec_hat.numpy().copy(),
           "offline_fused_vs_training_max": np.float64(np.abs(wav_f.numpy() - wav_hat.numpy()).max())}
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) fold_worst={worst:.2e} "
          f"in_rms={float(np.sqrt((xo.numpy() ** 2).mean())):.3f} out_rms={float(np.sqrt((out['offline_wav'] ** 2).mean())):.3f} "
          f"fused-vs-training {out['offline_fused_vs_training_max']:.2e}")


BSRNN_CONFIGS = {
    # name: (yaml, seed, B, hops)
    "bsrnn_xt": ("configs/others/bsrnn_xt.yaml", 201, 2, 10),
    "bsrnn_xxt": ("configs/others/bsrnn_xxt.yaml", 202, 2, 8),
    "bsrnn_t": ("configs/others/bsrnn_t.yaml", 203, 1, 6),
    "bsrnn_s": ("configs/others/bsrnn_s.yaml", 204, 1, 5),
}


def gen_bsrnn(ref: str, name: str, out_dir: str):
    from oracle import bsrnn_oracle as bo
    rel_yaml, seed, B, hops = BSRNN_CONFIGS[name]
    hps = yaml.safe_load(open(os.path.join(ref, rel_yaml)))
    kw = hps["model_kwargs"]
    sr = hps["data"]["sampling_rate"]
    cfg = bo.BSRNNConfig.from_model_kwargs(kw)
    mod = import_reference_model(ref, "models/bsrnn/model.py", "ref_bsrnn_model")
    torch.manual_seed(0)
    torch.set_num_threads(1)
    model = mod.Model(**kw).eval()                       # offline, training form (nn.LSTM)
    ref_sd = model.state_dict()
    spec = bo.training_state_dict_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), ([k for k in ref_sd if k not in spec], [k for k in spec if k not in ref_sd])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    sd = bo.make_training_state_dict(cfg, seed)
    model.load_state_dict(to_t(sd), strict=True)
    onnx_model = mod.ONNXModel(**kw).eval()              # streaming (LSTMCell); load_state_dict renames rnn_time.*_l0
    onnx_model.load_state_dict(to_t(sd), strict=True)
    onnx_model.remove_weight_reparameterizations()
    fused_ref = {k: v.detach().numpy().copy() for k, v in onnx_model.state_dict().items()}
    fused_mine = bo.fold_state_dict(sd, cfg)
    assert set(fused_mine) == set(fused_ref), (sorted(set(fused_mine) ^ set(fused_ref))[:10])
    worst = max(np.abs(fused_mine[k] - fused_ref[k]).max() / (np.abs(fused_ref[k]).max() + 1e-12) for k in fused_ref)
    assert worst < 3e-6, worst
    out = {"seed": np.int64(seed), "B": np.int64(B), "hops": np.int64(hops), "sr": np.int64(sr), "fold_worst_rel": np.float64(worst)}
    H = cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr))
    with torch.no_grad():
        cache_stft, cache_istft = onnx_model.stft.initialize_cache(x)
        cm = [torch.zeros(B * cfg.n_bands, cfg.hidden) for _ in range(2 * cfg.num_layers)]
        outs = []
        for t in range(hops):
            spec_in, cache_stft = onnx_model.stft(x[:, t * H:(t + 1) * H], cache_stft)
            spec_out, *cm = onnx_model(spec_in, *cm)
            wav_out, cache_istft = onnx_model.stft.inverse(spec_out, cache_istft)
            outs.append(wav_out.numpy().copy())
    out["stream_wav_out"] = np.stack(outs, 0)
    out["stream_cache_stft"] = cache_stft.numpy().copy()
    out["stream_cache_istft"] = cache_istft.numpy().copy()
    for i, t_ in enumerate(cm):
        out[f"stream_c{i}"] = t_.numpy().copy()
    out["stream_spec_out_last"] = spec_out.numpy().copy()
    xo = torch.from_numpy(make_input(B, hops * H + 37, seed + 2000, sr))
    with torch.no_grad():
        wav_hat, spec_hat = model(xo)
    out["offline_wav"] = wav_hat.numpy().copy()
    out["offline_spec"] = spec_hat.numpy().copy()
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) fold_worst={worst:.2e} "
          f"out_rms={float(np.sqrt((out['stream_wav_out'][4:] ** 2).mean())):.3f}")


def gen_fspen(ref: str, out_dir: str):
    """the FSPEN baseline (configs/others/fspen.yaml, models/fspen/model.py)."""
    from oracle import fspen_oracle as fo
    seed, B, hops = 301, 2, 10
    hps = yaml.safe_load(open(os.path.join(ref, "configs/others/fspen.yaml")))
    kw = hps["model_kwargs"]
    sr = hps["data"]["sampling_rate"]
    cfg = fo.FSPENConfig.from_model_kwargs(kw)
    mod = import_reference_model(ref, "models/fspen/model.py", "ref_fspen_model")
    torch.manual_seed(0)
    torch.set_num_threads(1)
    model = mod.Model(**kw).eval()
    ref_sd = model.state_dict()
    spec = fo.training_state_dict_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), ([k for k in ref_sd if k not in spec], [k for k in spec if k not in ref_sd])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    sd = fo.make_training_state_dict(cfg, seed)
    model.load_state_dict(to_t(sd), strict=True)
    onnx_model = mod.ONNXModel(**kw).eval()
    onnx_model.load_state_dict(to_t(sd), strict=True)
    onnx_model.remove_weight_reparameterizations()
    fused_ref = {k: v.detach().numpy().copy() for k, v in onnx_model.state_dict().items()}
    fused_mine = fo.fold_state_dict(sd, cfg)
    assert set(fused_mine) == set(fused_ref), (sorted(set(fused_mine) ^ set(fused_ref))[:10])
    worst = max(np.abs(fused_mine[k] - fused_ref[k]).max() / (np.abs(fused_ref[k]).max() + 1e-12) for k in fused_ref)
    assert worst < 3e-6, worst
    out = {"seed": np.int64(seed), "B": np.int64(B), "hops": np.int64(hops), "sr": np.int64(sr), "fold_worst_rel": np.float64(worst)}
    H = cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr))
    with torch.no_grad():
        cache_stft, cache_istft = onnx_model.stft.initialize_cache(x)
        cm = [torch.zeros(1, B * (cfg.freq // cfg.groups), cfg.dpe_channels) for _ in range(cfg.n_caches)]
        outs = []
        for t in range(hops):
            spec_in, cache_stft = onnx_model.stft(x[:, t * H:(t + 1) * H], cache_stft)
            spec_out, *cm = onnx_model(spec_in, *cm)
            wav_out, cache_istft = onnx_model.stft.inverse(spec_out, cache_istft)
            outs.append(wav_out.numpy().copy())
    out["stream_wav_out"] = np.stack(outs, 0)
    out["stream_cache_stft"] = cache_stft.numpy().copy()
    out["stream_cache_istft"] = cache_istft.numpy().copy()
    for i, t_ in enumerate(cm):
        out[f"stream_c{i}"] = t_.numpy().copy()
    out["stream_spec_out_last"] = spec_out.numpy().copy()
    xo = torch.from_numpy(make_input(B, hops * H + 37, seed + 2000, sr))
    with torch.no_grad():
        wav_hat, spec_hat = model(xo)
    out["offline_wav"] = wav_hat.numpy().copy()
    out["offline_spec"] = spec_hat.numpy().copy()
    path = os.path.join(out_dir, "fspen.npz")
    np.savez_compressed(path, **out)
    print(f"fspen: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) fold_worst={worst:.2e} "
          f"in_rms={float(np.sqrt((x.numpy() ** 2).mean())):.3f} out_rms={float(np.sqrt((out['stream_wav_out'][4:] ** 2).mean())):.3f}")


def gen_lisennet(ref: str, out_dir: str):
    """the LiSenNet baseline (configs/others/lisennet.yaml, models/lisennet/model.py)."""
    from oracle import lisennet_oracle as lo
    seed, B, hops = 401, 2, 10
    hps = yaml.safe_load(open(os.path.join(ref, "configs/others/lisennet.yaml")))
    kw = hps["model_kwargs"]
    sr = hps["data"]["sampling_rate"]
    cfg = lo.LiSenNetConfig.from_model_kwargs(kw)
    mod = import_reference_model(ref, "models/lisennet/model.py", "ref_lisennet_model")
    torch.manual_seed(0)
    torch.set_num_threads(1)
    model = mod.Model(**kw).eval()
    ref_sd = model.state_dict()
    spec = lo.state_dict_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), ([k for k in ref_sd if k not in spec], [k for k in spec if k not in ref_sd])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    sd = lo.make_state_dict(cfg, seed)
    model.load_state_dict(to_t(sd), strict=True)
    onnx_model = mod.ONNXModel(**kw).eval()
    onnx_model.load_state_dict(to_t(sd), strict=True)
    onnx_model.remove_weight_reparameterizations()         # (a no-op for this model)
    out = {"seed": np.int64(seed), "B": np.int64(B), "hops": np.int64(hops), "sr": np.int64(sr)}
    H = cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr))
    with torch.no_grad():
        cache_stft, cache_istft = onnx_model.stft.initialize_cache(x)
        cm = [torch.zeros(*s_) for s_ in cfg.cache_shapes(B)]           # the model's own initialize_cache is written for one stream
        outs = []
        for t in range(hops):
            spec_in, cache_stft = onnx_model.stft(x[:, t * H:(t + 1) * H], cache_stft)
            spec_out, *cm = onnx_model(spec_in, *cm)
            wav_out, cache_istft = onnx_model.stft.inverse(spec_out, cache_istft)
            outs.append(wav_out.numpy().copy())
    out["stream_wav_out"] = np.stack(outs, 0)
    out["stream_cache_stft"] = cache_stft.numpy().copy()
    out["stream_cache_istft"] = cache_istft.numpy().copy()
    for i, t_ in enumerate(cm):
        out[f"stream_c{i}"] = t_.numpy().copy()
    out["stream_spec_out_last"] = spec_out.numpy().copy()
    # Model.forward's phase features are ILL-CONDITIONED on frame 0 of any signal: torch.stft(center=True, pad_mode="reflect") makes
    # that frame symmetric about its centre, its spectrum real up to rounding noise, its phases 0 / pi at random, and the wrapped
    # phase differences +-pi by the last bit of atan2 (two runs of the reference on different BLAS builds would not agree).  The
    # offline vector therefore starts with 300 samples of silence: an all-zero frame 0 has phase atan2(0, 0) = 0 everywhere.
    xo_np = make_input(B, hops * H + 37, seed + 2000, sr)
    xo_np[:, :300] = 0.0
    xo = torch.from_numpy(xo_np)
    with torch.no_grad():
        wav_hat, spec_hat = model(xo)
        # ... and even the all-zero frame is not portable: torch.stft returns -0.0 in its upper bins (atan2(0, -0.0) = pi), numpy's
        # rfft and the GPU's FFT +0.0.  The reference's own features of the vector are therefore stored too: the oracle is pinned
        # on Model.forward through them (model_forward, masking, iSTFT), its feature extraction on every frame but the first.
        xs = model.stft(xo)
        xc = torch.view_as_complex(xs).transpose(1, 2)
        pha = xc.angle()
        out["offline_feat"] = torch.stack([xc.abs(), model.cal_gd(pha) / torch.pi, model.cal_ifd(pha) / torch.pi], dim=1).numpy().copy()
    out["offline_wav"] = wav_hat.numpy().copy()
    out["offline_spec"] = spec_hat.numpy().copy()
    out["offline_leading_zeros"] = np.int64(300)
    path = os.path.join(out_dir, "lisennet.npz")
    np.savez_compressed(path, **out)
    print(f"lisennet: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) "
          f"in_rms={float(np.sqrt((x.numpy() ** 2).mean())):.3f} out_rms={float(np.sqrt((out['stream_wav_out'][4:] ** 2).mean())):.3f}")


def gen_si_snr(ref: str, out_dir: str):
    """SI-SDR of the reference's evaluation script: `si_snr` / `product` are closures inside main() of
    scripts/metrics_ns.py (which imports torchaudio / pesq / pystoi at its top), so the two function definitions
    (:38-52) are compiled out of the file's text at generation time; only their inputs' seeds and their outputs are stored."""
    import textwrap
    lines = open(os.path.join(ref, "scripts", "metrics_ns.py")).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip().startswith("def product("))
    end = next(i for i, l in enumerate(lines) if i > start and l.strip().startswith("# Load the model"))
    ns = {"torch": torch}
    exec(compile(textwrap.dedent("\n".join(lines[start:end])), "metrics_ns.si_snr", "exec"), ns)
    ref_fn = ns["si_snr"]
    rng = np.random.default_rng(77)
    B, L = 5, 4000
    clean = (0.3 * np.sin(2 * np.pi * rng.uniform(100, 900, (B, 1)) * np.arange(L)[None] / 16000.0)
             + 0.05 * rng.standard_normal((B, L))).astype(np.float32)
    est = (clean * rng.uniform(0.5, 1.5, (B, 1)) + rng.uniform(0.01, 0.2, (B, 1)) * rng.standard_normal((B, L))).astype(np.float32)
    lens = np.array([4000, 3800, 2560, 1000, 256], np.int64)
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.float32)
    with torch.no_grad():
        full = ref_fn(torch.from_numpy(est), torch.from_numpy(clean), torch.ones(B, L)).numpy()
        # the evaluation loop's use (:125-137): both signals masked by the caller
        masked = ref_fn(torch.from_numpy(est * mask), torch.from_numpy(clean * mask), torch.from_numpy(mask)).numpy()
        # the function alone on UN-masked signals with a mask (what distinguishes masking inside from masking outside)
        fn_only = ref_fn(torch.from_numpy(est), torch.from_numpy(clean), torch.from_numpy(mask)).numpy()
    path = os.path.join(out_dir, "si_snr.npz")
    np.savez_compressed(path, clean=clean, est=est, lens=lens, full=full, masked=masked, fn_only=fn_only)
    print(f"si_snr: wrote {path}; values {full.round(3)}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for name in CONFIGS:
        if args.only and name not in args.only:
            continue
        gen_fastenhancer(args.ref, name, args.out)
    for name in NONCAUSAL_CONFIGS:
        if args.only and name not in args.only:
            continue
        gen_noncausal(args.ref, name, args.out)
    for name in BSRNN_CONFIGS:
        if args.only and name not in args.only:
            continue
        gen_bsrnn(args.ref, name, args.out)
    if not args.only or "si_snr" in args.only:
        gen_si_snr(args.ref, args.out)
    if not args.only or "fspen" in args.only:
        gen_fspen(args.ref, args.out)
    if not args.only or "lisennet" in args.only:
        gen_lisennet(args.ref, args.out)


if __name__ == "__main__":
    main()
