"""Deployment transform on the host: reference checkpoint -> fused fp32 weight blob.

Restates ONNXModel.remove_weight_reparameterizations
(models/fastenhancer/default/model.py:532-608; RNNFormerBlock :215-231;
ScaledConvTranspose1d :74-81): weight-norm removal, BatchNorm folding into the
adjacent conv / linear, and baking scale/normalisation into the final transposed
conv.  Accepts the training-form state_dict written by wrappers/ns.py:323-336
(``checkpoint["model"]``) or an already fused one.  The result is laid out into
the flat blob whose section table the C library reports (fe_weight_section), which
is also what is broadcast over RCCL in multi-GPU runs.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional

import torch
from torch import Tensor

from .config import FEConfig

BN_EPS = 1e-5  # nn.BatchNorm1d default, used by every conv BN of the reference


def _f32(t) -> Tensor:
    return torch.as_tensor(t).detach().to(device="cpu", dtype=torch.float32)


def is_fused(sd: Mapping[str, Tensor]) -> bool:
    return "enc_pre.0.bias" in sd


def _weight_norm(g: Tensor, v: Tensor) -> Tensor:
    # torch.nn.utils.parametrizations.weight_norm, dim=0: w = g * v / ||v|| (norm over all dims but 0)
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g.reshape(norm.shape) / norm)


_DPRNN_NAMES = (("dprnn_pre.", "rf_pre."), ("dprnn_post.", "rf_post."), ("dprnn_block.", "rf_block."), (".trnn_fc.", ".rnn_fc."),
                (".trnn_post_norm.", ".rnn_post_norm."), (".trnn.", ".rnn."))


def dprnn_canonical_key(k: str) -> str:
    for a, b in _DPRNN_NAMES:
        k = k.replace(a, b)
    return k


# models/fastenhancer/dptransformer/model.py:580-610 module names -> the default model's (the model-level `pe` is `time_pe`)
_DPT_NAMES = (("dpt_pre.", "rf_pre."), ("dpt_post.", "rf_post."), ("dpt_block.", "rf_block."), (".time_fc.", ".rnn_fc."),
              (".time_post_norm.", ".rnn_post_norm."), (".freq_attn.", ".attn."), (".freq_fc.", ".attn_fc."),
              (".freq_post_norm.", ".attn_post_norm."))


def dpt_canonical_key(k: str) -> str:
    if k == "pe":
        return "time_pe"
    for a, b in _DPT_NAMES:
        k = k.replace(a, b)
    return k


def fold_state_dict(sd: Mapping[str, Tensor], cfg: FEConfig) -> Dict[str, Tensor]:
    """Training-form -> fused-form state_dict (keys of SURVEY.md Appendix A.1 'Fused')."""
    sd = {k: _f32(v) for k, v in sd.items() if torch.as_tensor(v).is_floating_point()}
    if cfg.dprnn:      # the dprnn variant's module names (models/fastenhancer/dprnn/model.py:412-436) -> the default model's
        sd = {dprnn_canonical_key(k): v for k, v in sd.items()}
    if cfg.dpt:
        sd = {dpt_canonical_key(k): v for k, v in sd.items()}
    if cfg.ln:
        return _fold_state_dict_ln(sd, cfg)
    if is_fused(sd):
        sd = dict(sd)
        sd.pop("dec_post.2.scale", None)     # (the time_kernel variant's fused state_dict still lists the folded-in scale)
        return sd
    out: Dict[str, Tensor] = {}

    def conv_bn(conv: str, bn: str, dst: str):
        # model.py:548-553: W' = W * gamma/std, b' = beta - mean*gamma/std
        std = (sd[bn + ".running_var"] + BN_EPS).sqrt()
        w = sd[conv + ".weight"]           # Conv1d (Co, Ci, k) or - time_kernel variant - Conv2d (Co, Ci, kt, kf)
        out[dst + ".weight"] = w * (sd[bn + ".weight"] / std).view([-1] + [1] * (w.dim() - 1))
        out[dst + ".bias"] = sd[bn + ".bias"] - sd[bn + ".running_mean"] * sd[bn + ".weight"] / std

    conv_bn("enc_pre.0", "enc_pre.1", "enc_pre.0")
    for i in range(cfg.n_layers):
        conv_bn(f"encoder.{i}.0", f"encoder.{i}.1", f"encoder.{i}.0")
    out["rf_pre.0.weight"] = sd["rf_pre.0.weight"]
    conv_bn("rf_pre.1", "rf_pre.2", "rf_pre.1")
    if cfg.dpt:
        out["time_pe"] = sd["time_pe"]
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if p + "pe" in sd:
            out[p + "pe"] = sd[p + "pe"]
        if cfg.dpt:        # DPTBlock.remove_weight_reparameterizations, dptransformer/model.py:323-336
            g = p + "time_attn.qkv.parametrizations.weight.original0"
            out[p + "time_attn.qkv.weight"] = (_weight_norm(sd[g], sd[p + "time_attn.qkv.parametrizations.weight.original1"]) if g in sd
                                               else sd[p + "time_attn.qkv.weight"])
        else:
            for sfx in ("", "_reverse") if cfg.noncausal else ("",):      # noncausal/model.py:215-222: both directions
                for name in ("weight_ih_l0" + sfx, "weight_hh_l0" + sfx):
                    g = p + f"rnn.parametrizations.{name}.original0"
                    if g in sd:
                        out[p + "rnn." + name] = _weight_norm(sd[g], sd[p + f"rnn.parametrizations.{name}.original1"])
                    else:
                        out[p + "rnn." + name] = sd[p + "rnn." + name]
                out[p + "rnn.bias_ih_l0" + sfx] = sd[p + "rnn.bias_ih_l0" + sfx]
                out[p + "rnn.bias_hh_l0" + sfx] = sd[p + "rnn.bias_hh_l0" + sfx]
        if cfg.dprnn:      # DPRNN.remove_weight_reparameterizations, dprnn/model.py:172-192
            for name in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse"):
                g = p + f"frnn.parametrizations.{name}.original0"
                out[p + "frnn." + name] = (_weight_norm(sd[g], sd[p + f"frnn.parametrizations.{name}.original1"]) if g in sd
                                           else sd[p + "frnn." + name])
            for name in ("bias_ih_l0", "bias_hh_l0", "bias_ih_l0_reverse", "bias_hh_l0_reverse"):
                out[p + "frnn." + name] = sd[p + "frnn." + name]
        else:
            g = p + "attn.qkv.parametrizations.weight.original0"
            if g in sd:
                out[p + "attn.qkv.weight"] = _weight_norm(sd[g], sd[p + "attn.qkv.parametrizations.weight.original1"])
            else:
                out[p + "attn.qkv.weight"] = sd[p + "attn.qkv.weight"]
        for fc, norm in ((("rnn_fc", "rnn_post_norm"), ("frnn_fc", "frnn_post_norm")) if cfg.dprnn
                         else (("rnn_fc", "rnn_post_norm"), ("attn_fc", "attn_post_norm"))):   # model.py:223-229
            std = (sd[p + norm + ".running_var"] + cfg.rf_eps).sqrt()
            out[p + fc + ".weight"] = sd[p + fc + ".weight"] * (sd[p + norm + ".weight"] / std).view(-1, 1)
            out[p + fc + ".bias"] = sd[p + norm + ".bias"] - sd[p + norm + ".running_mean"] * sd[p + norm + ".weight"] / std
    out["rf_post.0.weight"] = sd["rf_post.0.weight"]
    conv_bn("rf_post.1", "rf_post.2", "rf_post.1")
    for i in range(cfg.n_layers):
        conv_bn(f"decoder.{i}.0", f"decoder.{i}.1", f"decoder.{i}.0")
        conv_bn(f"decoder.{i}.3", f"decoder.{i}.4", f"decoder.{i}.2")
    conv_bn("dec_post.0", "dec_post.1", "dec_post.0")
    w = sd["dec_post.3.weight"]
    scale = sd.get("dec_post.3.scale", torch.ones(1))
    if cfg.final_scale_exp:        # time_kernel/model.py:95 (exp_scale)
        scale = scale.exp()
    if cfg.normalize_final_conv:   # F.normalize(w, dim=(0,1,2)) * scale, model.py:74-79
        w = w / w.norm().clamp_min(1e-12)
    out["dec_post.2.weight"] = w * scale
    out["dec_post.2.bias"] = sd["dec_post.3.bias"]
    return out


def _fold_state_dict_ln(sd: Dict[str, Tensor], cfg: FEConfig) -> Dict[str, Tensor]:
    """The ln variant's deployment transform (models/fastenhancer/ln/model.py:524-533, 239-245, 116-135): the weight norms of the
    GRU / qkv matrices and the final conv's normalisation + scale go away, every norm layer stays.  Fused keys = the training
    keys, except the final conv: dec_post.3.{weight,scale,bias} -> dec_post.2.{weight,bias}."""
    if "dec_post.2.weight" in sd:
        return dict(sd)
    out: Dict[str, Tensor] = {k: v for k, v in sd.items() if ".parametrizations." not in k and not k.startswith("dec_post.3.")}
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        for mod, name in (("rnn", "weight_ih_l0"), ("rnn", "weight_hh_l0"), ("attn.qkv", "weight")):
            g = p + f"{mod}.parametrizations.{name}.original0"
            if g in sd:
                out[p + f"{mod}.{name}"] = _weight_norm(sd[g], sd[p + f"{mod}.parametrizations.{name}.original1"])
    w = sd["dec_post.3.weight"]
    scale = sd.get("dec_post.3.scale", torch.ones(1))
    if cfg.final_scale_exp:
        scale = scale.exp()
    if cfg.normalize_final_conv:
        w = w / w.norm().clamp_min(1e-12)
    out["dec_post.2.weight"] = w * scale
    out["dec_post.2.bias"] = sd["dec_post.3.bias"]
    return out


def _expected_fused_shapes_ln(cfg: FEConfig) -> Dict[str, tuple]:
    C1, C2, F1, F2 = cfg.channels, cfg.rf_channels, cfg.F1, cfg.rf_freq
    s: Dict[str, tuple] = {}

    def wb(prefix, wshape, bias=True):
        s[prefix + ".weight"] = wshape
        if bias:
            s[prefix + ".bias"] = (wshape[0],)

    wb("enc_pre.0", (C1, 2 * cfg.stride, cfg.kernel_size[0] // cfg.stride)); wb("enc_pre.1", (C1,))
    for i in range(cfg.n_layers):
        wb(f"encoder.{i}.0", (C1, C1, cfg.kernel_size[i + 1])); wb(f"encoder.{i}.1", (C1,))
    s["rf_pre.0.weight"] = (F2, F1)
    wb("rf_pre.1", (C2, C1, 1)); wb("rf_pre.2", (C2,))
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if k == 0:
            s[p + "pe"] = (F2, C2)
        s[p + "rnn.weight_ih_l0"] = (3 * C2, C2)
        s[p + "rnn.weight_hh_l0"] = (3 * C2, C2)
        s[p + "rnn.bias_ih_l0"] = (3 * C2,)
        s[p + "rnn.bias_hh_l0"] = (3 * C2,)
        s[p + "rnn_fc.weight"] = (C2, C2)
        wb(p + "rnn_post_norm", (C2,))
        s[p + "attn.qkv.weight"] = (3 * C2, C2)
        s[p + "attn_fc.weight"] = (C2, C2)
        wb(p + "attn_post_norm", (C2,))
    s["rf_post.0.weight"] = (F1, F2)
    wb("rf_post.1", (C1, C2, 1)); wb("rf_post.2", (C1,))
    for i in range(cfg.n_layers):
        wb(f"decoder.{i}.0", (C1, 2 * C1, 1)); wb(f"decoder.{i}.1", (C1,))
        wb(f"decoder.{i}.3", (C1, C1, cfg.kernel_size[cfg.n_layers - i]), bias=False); wb(f"decoder.{i}.4", (C1,))
    wb("dec_post.0", (C1, 2 * C1, 1), bias=False); wb("dec_post.1", (C1,))
    s["dec_post.2.weight"] = (C1, 2, cfg.kernel_size[0])
    s["dec_post.2.bias"] = (2,)
    return s


def expected_fused_shapes(cfg: FEConfig) -> Dict[str, tuple]:
    if cfg.ln:
        return _expected_fused_shapes_ln(cfg)
    C1, C2, F1, F2 = cfg.channels, cfg.rf_channels, cfg.F1, cfg.rf_freq
    tk, kt = cfg.time_kernel, cfg.kernel_size_time
    one = (1, 1) if tk else (1,)       # the time_kernel variant's 1x1 convs are Conv2d
    s: Dict[str, tuple] = {"enc_pre.0.weight": (C1, 2 * cfg.stride, cfg.kernel_size[0] // cfg.stride), "enc_pre.0.bias": (C1,)}
    for i in range(cfg.n_layers):
        s[f"encoder.{i}.0.weight"] = (C1, C1, kt, cfg.kernel_size[i + 1]) if tk else (C1, C1, cfg.kernel_size[i + 1])
        s[f"encoder.{i}.0.bias"] = (C1,)
    s["rf_pre.0.weight"] = (F2, F1)
    s["rf_pre.1.weight"] = (C2, C1) + one
    s["rf_pre.1.bias"] = (C2,)
    if cfg.dpt:
        s["time_pe"] = (cfg.rf_heads, cfg.lookbehind + 1)
    for k in range(cfg.rf_blocks):
        p = f"rf_block.{k}."
        if k == 0 and not cfg.dprnn:
            s[p + "pe"] = (F2, C2)
        if cfg.dpt:
            s[p + "time_attn.qkv.weight"] = (3 * C2, C2)
        else:
            for sfx in ("", "_reverse") if cfg.noncausal else ("",):
                s[p + "rnn.weight_ih_l0" + sfx] = (3 * C2, C2)
                s[p + "rnn.weight_hh_l0" + sfx] = (3 * C2, C2)
                s[p + "rnn.bias_ih_l0" + sfx] = (3 * C2,)
                s[p + "rnn.bias_hh_l0" + sfx] = (3 * C2,)
        s[p + "rnn_fc.weight"] = (C2, 2 * C2 if cfg.noncausal else C2)
        s[p + "rnn_fc.bias"] = (C2,)
        if cfg.dprnn:
            H = cfg.channels_frnn
            for sfx in ("", "_reverse"):
                s[p + "frnn.weight_ih_l0" + sfx] = (3 * H, C2)
                s[p + "frnn.weight_hh_l0" + sfx] = (3 * H, H)
                s[p + "frnn.bias_ih_l0" + sfx] = (3 * H,)
                s[p + "frnn.bias_hh_l0" + sfx] = (3 * H,)
            s[p + "frnn_fc.weight"] = (C2, 2 * H)
            s[p + "frnn_fc.bias"] = (C2,)
            continue
        s[p + "attn.qkv.weight"] = (3 * C2, C2)
        s[p + "attn_fc.weight"] = (C2, C2)
        s[p + "attn_fc.bias"] = (C2,)
    s["rf_post.0.weight"] = (F1, F2)
    s["rf_post.1.weight"] = (C1, C2) + one
    s["rf_post.1.bias"] = (C1,)
    for i in range(cfg.n_layers):
        s[f"decoder.{i}.0.weight"] = (C1, 2 * C1) + one
        s[f"decoder.{i}.0.bias"] = (C1,)
        kf = cfg.kernel_size[cfg.n_layers - i]
        s[f"decoder.{i}.2.weight"] = (C1, C1, kt, kf) if tk else (C1, C1, kf)
        s[f"decoder.{i}.2.bias"] = (C1,)
    s["dec_post.0.weight"] = (C1, 2 * C1, 1)
    s["dec_post.0.bias"] = (C1,)
    s["dec_post.2.weight"] = (C1, 2, cfg.kernel_size[0])
    s["dec_post.2.bias"] = (2,)
    return s


def check_fused(fused: Mapping[str, Tensor], cfg: FEConfig, strict: bool = True):
    """load_state_dict(strict=True) semantics (wrappers/ns.py:313): missing / unexpected keys and
    shape mismatches raise RuntimeError."""
    exp = expected_fused_shapes(cfg)
    missing = [k for k in exp if k not in fused]
    unexpected = [k for k in fused if k not in exp]
    errs = []
    if missing:
        errs.append("Missing key(s) in state_dict: " + ", ".join(missing))
    if unexpected and strict:
        errs.append("Unexpected key(s) in state_dict: " + ", ".join(unexpected))
    for k, shp in exp.items():
        if k in fused and tuple(fused[k].shape) != tuple(shp):
            errs.append(f"size mismatch for {k}: checkpoint {tuple(fused[k].shape)} vs model {tuple(shp)}")
    if errs:
        raise RuntimeError("Error(s) in loading state_dict:\n\t" + "\n\t".join(errs))


# ---------------------------------------------------------------- default initialisation
def linear_filterbank(n_freq: int, n_filter: int):
    """The 'linear' / 'linear_fixed' initialisation of rf_pre / rf_post (what rf_pre_post_lin builds,
    models/fastenhancer/default/model.py:308-380), from its definition: n_filter triangular filters whose centres are
    evenly spaced over the n_freq input bins (first on bin 0, last on bin n_freq-1), each reaching zero at its
    neighbours' centres; rows normalised to unit sum.  rf_post is the transpose, again row-normalised."""
    centres = torch.arange(n_filter, dtype=torch.float64) * ((n_freq - 1) / (n_filter - 1))
    bins = torch.arange(n_freq, dtype=torch.float64)
    width = (n_freq - 1) / (n_filter - 1)
    tri = (1.0 - (bins[None, :] - centres[:, None]).abs() / width).clamp_min(0.0)      # [n_filter, n_freq]
    pre = tri / tri.sum(dim=1, keepdim=True)
    post = pre.t() / pre.t().sum(dim=1, keepdim=True)
    return pre.float().contiguous(), post.float().contiguous()


def linear_filterbank_time_kernel(n_freq: int, n_filter: int, sr: int = 16000):
    """The time_kernel variant's 'linear' / 'linear_fixed' initialisation (models/fastenhancer/time_kernel/model.py:477-489):
    triangles on Hz axes whose edges have slope 1 / w with w = (sr/2) / n_filter, while their zero crossings sit on the
    neighbouring centres (spacing s = (sr/2) / (n_filter - 1)): filter i is (s - |f - c_i|) / w, peak s / w > 1, clamped at 0;
    the first / last filter has no rising / falling edge.  rf_post is the transpose of these UN-normalised triangles; both
    are then row-normalised."""
    half = float(sr // 2)
    centres = torch.linspace(0.0, half, n_filter, dtype=torch.float64)
    bins = torch.linspace(0.0, half, n_freq, dtype=torch.float64)
    w, s = half / n_filter, half / (n_filter - 1)
    d = bins[None, :] - centres[:, None]                                                # [n_filter, n_freq]
    rising = torch.cat([torch.ones(1, n_freq, dtype=torch.float64), (s + d[1:]) / w], dim=0)
    falling = torch.cat([(s - d[:-1]) / w, torch.ones(1, n_freq, dtype=torch.float64)], dim=0)
    tri = torch.minimum(rising, falling).clamp_min(0.0)
    pre = tri / tri.sum(dim=1, keepdim=True)
    post = tri.t() / tri.t().sum(dim=1, keepdim=True)
    return pre.float().contiguous(), post.float().contiguous()


def positional_embedding(channels: int, freq: int) -> Tensor:
    """Initial value of rf_block.0.pe (calculate_positional_embedding, model.py:98-110): for sub-band f = 1..F at angle
    pi f / F, channels//2 log-spaced multipliers from 1 to F-1; sines in the first half of the channels, cosines in the second."""
    angle = math.pi * torch.arange(1, freq + 1, dtype=torch.float32) / freq
    mult = torch.exp(torch.linspace(0.0, math.log(freq - 1), channels // 2, dtype=torch.float32))
    phase = angle[:, None] * mult[None, :]
    return torch.cat((torch.sin(phase), torch.cos(phase)), dim=1)


def default_state_dict(cfg: FEConfig, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """A fresh *fused-form* state_dict with PyTorch-style default initialisation, so that a model
    constructed from model_kwargs alone is runnable like the reference's (model.py:738-756).
    (Values differ from torch's RNG stream; structure and distributions follow nn defaults.)"""
    g = generator
    shapes = expected_fused_shapes(cfg)
    sd: Dict[str, Tensor] = {}
    fb = linear_filterbank_time_kernel if cfg.time_kernel or cfg.dprnn or cfg.dpt or cfg.ln or cfg.noncausal else linear_filterbank
    pre, post = (fb(cfg.F1, cfg.rf_freq) if (cfg.pre_post_init or "").startswith("linear") else (None, None))
    for k, shp in shapes.items():
        fan_in = 1
        for s in shp[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        if k == "rf_pre.0.weight" and pre is not None:
            sd[k] = pre
        elif k == "rf_post.0.weight" and post is not None:
            sd[k] = post
        elif k == "time_pe":       # calculate_positional_embedding(num_heads, lookbehind + 1), transposed (dptransformer/model.py:582-586)
            sd[k] = positional_embedding(cfg.rf_heads, cfg.lookbehind + 1).t().contiguous()
        elif k.endswith(".pe"):
            sd[k] = positional_embedding(cfg.rf_channels, cfg.rf_freq)
        elif cfg.ln and len(shp) == 1 and k.endswith(".weight"):       # GroupNorm / LayerNorm gains
            sd[k] = torch.ones(shp)
        elif k.endswith("bias") and ".rnn." not in k and ".frnn." not in k:
            sd[k] = torch.zeros(shp)
        elif ".rnn." in k or ".frnn." in k:
            b = 1.0 / math.sqrt(cfg.channels_frnn if ".frnn." in k else cfg.rf_channels)
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    if cfg.normalize_final_conv:
        w = sd["dec_post.2.weight"]
        sd["dec_post.2.weight"] = w / w.norm().clamp_min(1e-12)
    return sd


# ------------------------------------------------------------------------------------------------ BSRNN
def _bn_wb(sd, prefix):
    """x*w + b of an eval BatchNorm placed BEFORE a conv / LSTM (models/bsrnn/model.py:27-33)."""
    std = (sd[prefix + ".running_var"] + BN_EPS).sqrt()
    w = 1.0 / std
    b = -sd[prefix + ".running_mean"] / std
    if prefix + ".weight" in sd:
        w = sd[prefix + ".weight"] * w
        b = b * sd[prefix + ".weight"] + sd[prefix + ".bias"]
    return w, b


def bsrnn_expected_fused_shapes(cfg) -> Dict[str, tuple]:
    from .config import BSRNN_SUBBANDS
    C, Hh = cfg.num_channels, cfg.hidden
    s: Dict[str, tuple] = {}
    for b, sub in enumerate(BSRNN_SUBBANDS):
        s[f"band_split.fc.{b}.weight"] = (C, 2 * sub, 1)
        s[f"band_split.fc.{b}.bias"] = (C,)
    for l in range(cfg.num_layers):
        s[f"rnn_time.{l}.weight_ih"] = (4 * Hh, C)
        s[f"rnn_time.{l}.weight_hh"] = (4 * Hh, Hh)
        s[f"rnn_time.{l}.bias_ih"] = (4 * Hh,)
        s[f"rnn_time.{l}.bias_hh"] = (4 * Hh,)
        s[f"fc_time.{l}.weight"] = (C, Hh)
        s[f"fc_time.{l}.bias"] = (C,)
        for sfx in ("", "_reverse"):
            s[f"rnn_freq.{l}.weight_ih_l0{sfx}"] = (4 * Hh, C)
            s[f"rnn_freq.{l}.weight_hh_l0{sfx}"] = (4 * Hh, Hh)
            s[f"rnn_freq.{l}.bias_ih_l0{sfx}"] = (4 * Hh,)
            s[f"rnn_freq.{l}.bias_hh_l0{sfx}"] = (4 * Hh,)
        s[f"fc_freq.{l}.weight"] = (C, 2 * Hh)
        s[f"fc_freq.{l}.bias"] = (C,)
    for kind in ("mlp_mask", "mlp_residual"):
        for b, sub in enumerate(BSRNN_SUBBANDS):
            p = f"mask_decoder.{kind}.{b}."
            s[p + "0.weight"] = (4 * C, C, 1)
            s[p + "0.bias"] = (4 * C,)
            s[p + "2.weight"] = (4 * sub, 4 * C, 1)
            s[p + "2.bias"] = (4 * sub,)
    return s


def bsrnn_fold_state_dict(sd: Mapping[str, Tensor], cfg) -> Dict[str, Tensor]:
    """ONNXModel.remove_weight_reparameterizations of models/bsrnn/model.py:348-366 (fuse_bn_conv1d :14-42,
    fuse_bn_rnn :45-82) + the rnn_time key rename of load_state_dict (:450-460).  Training-form or fused in, fused out."""
    from .config import BSRNN_SUBBANDS
    sd = {k: _f32(v) for k, v in sd.items() if torch.as_tensor(v).is_floating_point()}
    if "band_split.norm.0.running_var" not in sd:
        out = {}
        for k, v in sd.items():     # accept nn.LSTM-style names of the offline model for the time LSTM
            out[k[:-3] if (k.startswith("rnn_time.") and k.endswith("_l0")) else k] = v
        return out
    C = cfg.num_channels
    out: Dict[str, Tensor] = {}
    # the time LSTM's keys come as nn.LSTM names (`*_l0`: a checkpoint of the offline Model) or already renamed to the
    # LSTMCell names (the state_dict of the reference's ONNXModel before remove_weight_reparameterizations, :450-460)
    missing = []

    def tkey(l, stem):
        for cand in (f"rnn_time.{l}.{stem}_l0", f"rnn_time.{l}.{stem}"):
            if cand in sd:
                return cand
        missing.append(f"rnn_time.{l}.{stem}[_l0]")
        return None

    def conv(conv_key, bn_key, dst):
        w, b = _bn_wb(sd, bn_key)
        W = sd[conv_key + ".weight"]
        bias = (W * b.view(1, -1, 1)).sum(dim=(1, 2))
        if conv_key + ".bias" in sd:
            bias = bias + sd[conv_key + ".bias"]
        out[dst + ".weight"] = W * w.view(1, -1, 1)
        out[dst + ".bias"] = bias

    def rnn(wkey, bkey, bn_key, dst_w, dst_b):
        w, b = _bn_wb(sd, bn_key)
        W = sd[wkey]
        out[dst_w] = W * w.view(1, -1)
        out[dst_b] = sd[bkey] + W @ b

    for b in range(len(BSRNN_SUBBANDS)):
        conv(f"band_split.fc.{b}", f"band_split.norm.{b}", f"band_split.fc.{b}")
    for l in range(cfg.num_layers):
        keys = [tkey(l, stem) for stem in ("weight_ih", "bias_ih", "weight_hh", "bias_hh")]
        if None in keys:
            continue
        rnn(keys[0], keys[1], f"norm_time.{l}", f"rnn_time.{l}.weight_ih", f"rnn_time.{l}.bias_ih")
        out[f"rnn_time.{l}.weight_hh"] = sd[keys[2]]
        out[f"rnn_time.{l}.bias_hh"] = sd[keys[3]]
        out[f"fc_time.{l}.weight"] = sd[f"fc_time.{l}.weight"]
        out[f"fc_time.{l}.bias"] = sd.get(f"fc_time.{l}.bias", torch.zeros(C))
        for sfx in ("", "_reverse"):
            rnn(f"rnn_freq.{l}.weight_ih_l0{sfx}", f"rnn_freq.{l}.bias_ih_l0{sfx}", f"norm_freq.{l}",
                f"rnn_freq.{l}.weight_ih_l0{sfx}", f"rnn_freq.{l}.bias_ih_l0{sfx}")
            out[f"rnn_freq.{l}.weight_hh_l0{sfx}"] = sd[f"rnn_freq.{l}.weight_hh_l0{sfx}"]
            out[f"rnn_freq.{l}.bias_hh_l0{sfx}"] = sd[f"rnn_freq.{l}.bias_hh_l0{sfx}"]
        out[f"fc_freq.{l}.weight"] = sd[f"fc_freq.{l}.weight"]
        out[f"fc_freq.{l}.bias"] = sd.get(f"fc_freq.{l}.bias", torch.zeros(C))
    for kind in ("mlp_mask", "mlp_residual"):
        for b in range(len(BSRNN_SUBBANDS)):
            p = f"mask_decoder.{kind}.{b}."
            conv(p + "1", p + "0", p + "0")
            out[p + "2.weight"] = sd[p + "3.weight"]
            out[p + "2.bias"] = sd[p + "3.bias"]
    if missing:
        raise RuntimeError("Error(s) in loading state_dict:\n\tMissing key(s) in state_dict: " + ", ".join(missing))
    return out


def bsrnn_default_state_dict(cfg, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """fresh fused-form state_dict with PyTorch-style uniform initialisation"""
    sd = {}
    for k, shp in bsrnn_expected_fused_shapes(cfg).items():
        fan_in = 1
        for s_ in shp[1:]:
            fan_in *= s_
        bound = 1.0 / math.sqrt(cfg.hidden if "rnn_" in k else max(fan_in, 1))
        sd[k] = (torch.rand(shp, generator=generator) * 2 - 1) * bound
    return sd


def check_shapes(fused: Mapping[str, Tensor], exp: Mapping[str, tuple], strict: bool = True):
    missing = [k for k in exp if k not in fused]
    unexpected = [k for k in fused if k not in exp]
    errs = []
    if missing:
        errs.append("Missing key(s) in state_dict: " + ", ".join(missing))
    if unexpected and strict:
        errs.append("Unexpected key(s) in state_dict: " + ", ".join(unexpected))
    for k, shp in exp.items():
        if k in fused and tuple(fused[k].shape) != tuple(shp):
            errs.append(f"size mismatch for {k}: checkpoint {tuple(fused[k].shape)} vs model {tuple(shp)}")
    if errs:
        raise RuntimeError("Error(s) in loading state_dict:\n\t" + "\n\t".join(errs))


# ------------------------------------------------------------------------------------------------ FSPEN
FSPEN_SUB_ENC_K = (4, 7, 11, 20, 40)      # SubbandEncoder kernels (models/fspen/model.py:41-44)
FSPEN_SUB_DEC_N = (2, 3, 5, 10, 20)       # SubbandDecoder outputs per row (:70)


def fspen_expected_fused_shapes(cfg) -> Dict[str, tuple]:
    """fused state_dict of models/fspen/model.py::ONNXModel after remove_weight_reparameterizations (:299-340)."""
    C1, K, C2, Fq = cfg.channels, cfg.kernel_size, cfg.dpe_channels, cfg.freq
    sh: Dict[str, tuple] = {}
    for i, k in enumerate(FSPEN_SUB_ENC_K):
        sh[f"subband_encoder.conv{i + 1}.0.weight"] = (C1[-1], 1, k)
        sh[f"subband_encoder.conv{i + 1}.0.bias"] = (C1[-1],)
    for i, n in enumerate(FSPEN_SUB_DEC_N):
        sh[f"subband_decoder.lin{i + 1}.0.weight"] = (n, 2 * C1[-1])
        sh[f"subband_decoder.lin{i + 1}.0.bias"] = (n,)
    for i in range(len(C1)):
        sh[f"fullband_encoder.{i}.0.weight"] = (C1[i], 2 if i == 0 else C1[i - 1], K[i])
        sh[f"fullband_encoder.{i}.0.bias"] = (C1[i],)
    sh["fullband_encoder_post.weight"] = (C1[-1], C1[-1], 1)
    sh["feature_merge.0.weight"] = (Fq, 64)
    sh["feature_merge.2.weight"] = (C2, C1[-1], 1)
    sh["feature_merge.2.bias"] = (C2,)

    def gru(p, sfx=""):
        sh[f"{p}.weight_ih_l0{sfx}"] = (3 * C2, C2)
        sh[f"{p}.weight_hh_l0{sfx}"] = (3 * C2, C2)
        sh[f"{p}.bias_ih_l0{sfx}"] = (3 * C2,)
        sh[f"{p}.bias_hh_l0{sfx}"] = (3 * C2,)

    for b in range(cfg.num_blocks):
        p = f"dpe_blocks.{b}."
        gru(p + "intra_rnn")
        gru(p + "intra_rnn", "_reverse")
        sh[p + "intra_fc.weight"] = (C2, 2 * C2)
        sh[p + "intra_fc.bias"] = (C2,)
        sh[p + "intra_ln.weight"] = (Fq, C2)
        sh[p + "intra_ln.bias"] = (Fq, C2)
        for g in range(cfg.groups):
            gru(p + f"inter_rnn.inter_rnn.{g}")
        for g in range(cfg.groups):
            sh[p + f"inter_rnn.inter_fc.{g}.weight"] = (C2, C2)
            sh[p + f"inter_rnn.inter_fc.{g}.bias"] = (C2,)
    sh["feature_split.0.weight"] = (C1[-1], C2, 1)
    sh["feature_split.0.bias"] = (C1[-1],)
    sh["feature_split.1.weight"] = (64, Fq)
    for j, i in enumerate(range(len(C1) - 1, -1, -1)):
        cin, cout = C1[i], (2 if i == 0 else C1[i - 1])
        sh[f"fullband_decoder.{j}.0.weight"] = (cin, 2 * cin, 1)
        sh[f"fullband_decoder.{j}.1.weight"] = (cin, cout, K[i])
        sh[f"fullband_decoder.{j}.1.bias"] = (cout,)
    return sh


def fspen_fold_state_dict(sd: Mapping[str, Tensor], cfg, eps: float = 1e-5) -> Dict[str, Tensor]:
    """ONNXModel.remove_weight_reparameterizations of models/fspen/model.py:299-340: the BatchNorm that FOLLOWS each full-band
    encoder conv / decoder transposed conv scales the conv's output channels (dim 0 of a Conv1d weight, dim 1 of a
    ConvTranspose1d weight) and becomes its bias.  A dict without BatchNorm statistics is taken as already fused."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    n = len(cfg.channels)
    if "fullband_encoder.0.1.running_var" not in sd:
        return {k: v.detach().to(torch.float32) for k, v in sd.items() if v.is_floating_point()}
    bn_prefixes = tuple(f"fullband_encoder.{i}.1." for i in range(n)) + tuple(f"fullband_decoder.{j}.2." for j in range(n - 1))
    out: Dict[str, Tensor] = {k: v.detach().to(torch.float32).clone() for k, v in sd.items()
                              if v.is_floating_point() and not k.startswith(bn_prefixes)}

    def scale_shift(p):
        std = torch.sqrt(sd[p + "running_var"].float() + eps)
        g = sd[p + "weight"].float() / std
        return g, sd[p + "bias"].float() - sd[p + "running_mean"].float() * g

    for i in range(n):
        g, b = scale_shift(f"fullband_encoder.{i}.1.")
        out[f"fullband_encoder.{i}.0.weight"] = sd[f"fullband_encoder.{i}.0.weight"].float() * g.view(-1, 1, 1)
        out[f"fullband_encoder.{i}.0.bias"] = b
    for j in range(n - 1):
        g, b = scale_shift(f"fullband_decoder.{j}.2.")
        out[f"fullband_decoder.{j}.1.weight"] = sd[f"fullband_decoder.{j}.1.weight"].float() * g.view(1, -1, 1)
        out[f"fullband_decoder.{j}.1.bias"] = b
    return out


def fspen_default_state_dict(cfg, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Random fused weights of the right shapes (benchmarks, smoke tests): small fan-in-scaled matrices; the last transposed
    conv gets a bias so that the complex mask stays away from the 0 / 0 of `out_full / mask_full_mag`."""
    g = generator or torch.Generator().manual_seed(0)
    sd: Dict[str, Tensor] = {}
    last = f"fullband_decoder.{len(cfg.channels) - 1}.1."
    for k, shp in fspen_expected_fused_shapes(cfg).items():
        if k == last + "bias":
            sd[k] = torch.tensor([0.6, 0.2])
        elif k.endswith("intra_ln.weight"):
            sd[k] = torch.ones(shp)
        elif len(shp) == 1 or k.endswith("intra_ln.bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.randn(shp, generator=g) * (0.5 / fan_in ** 0.5)
    return sd


# ------------------------------------------------------------------------------------------------ LiSenNet
def lisennet_expected_shapes(cfg) -> Dict[str, tuple]:
    """state_dict of models/lisennet/model.py::ONNXModel (nothing is folded: remove_weight_reparameterizations is a no-op, :476-477)."""
    C, F, Hd = cfg.num_channels, cfg.n_fft // 2 + 1, cfg.hidden
    c1, c2, c3 = C // 4, C // 2, C // 4 * 3
    nf = F // 8
    sh: Dict[str, tuple] = {}

    def dsconv(p, cin, cout, nfq):
        sh[p + ".low_conv.weight"] = (cout, cin, 2, 3)
        sh[p + ".low_conv.bias"] = (cout,)
        sh[p + ".high_conv.weight"] = (cout, cin, 2, 5)
        sh[p + ".high_conv.bias"] = (cout,)
        sh[p + ".norm.gamma"] = (1, 1, 1, nfq // 2)
        sh[p + ".norm.beta"] = (1, 1, 1, nfq // 2)
        sh[p + ".act.weight"] = (cout,)

    def gru(p, i, h, bi):
        for sfx in (("", "_reverse") if bi else ("",)):
            sh[f"{p}.weight_ih_l0{sfx}"] = (3 * h, i)
            sh[f"{p}.weight_hh_l0{sfx}"] = (3 * h, h)
            sh[f"{p}.bias_ih_l0{sfx}"] = (3 * h,)
            sh[f"{p}.bias_hh_l0{sfx}"] = (3 * h,)

    sh["encoder.conv_1.0.weight"] = (c1, 3, 1, 1)
    sh["encoder.conv_1.0.bias"] = (c1,)
    sh["encoder.conv_1.1.gamma"] = (1, 1, 1, F)
    sh["encoder.conv_1.1.beta"] = (1, 1, 1, F)
    sh["encoder.conv_1.2.weight"] = (c1,)
    dsconv("encoder.conv_2", c1, c2, F)
    dsconv("encoder.conv_3", c2, c3, F // 2)
    dsconv("encoder.conv_4", c3, C, F // 4)
    for b in range(cfg.n_blocks):
        p = f"blocks.{b}."
        sh[p + "dp_rnn_attn.intra_norm.weight"] = (nf, C)
        sh[p + "dp_rnn_attn.intra_norm.bias"] = (nf, C)
        gru(p + "dp_rnn_attn.intra_rnn_attn.rnn", C, Hd // 2, True)
        sh[p + "dp_rnn_attn.intra_rnn_attn.dense.weight"] = (C, Hd)
        sh[p + "dp_rnn_attn.intra_rnn_attn.dense.bias"] = (C,)
        sh[p + "dp_rnn_attn.inter_norm.weight"] = (nf, C)
        sh[p + "dp_rnn_attn.inter_norm.bias"] = (nf, C)
        gru(p + "dp_rnn_attn.inter_rnn_attn.rnn", C, Hd, False)
        sh[p + "dp_rnn_attn.inter_rnn_attn.dense.weight"] = (C, Hd)
        sh[p + "dp_rnn_attn.inter_rnn_attn.dense.bias"] = (C,)
        sh[p + "conv_glu.norm.gamma"] = (1, C, 1, nf)
        sh[p + "conv_glu.norm.beta"] = (1, C, 1, nf)
        sh[p + "conv_glu.fc1.weight"] = (4 * C, C, 1, 1)
        sh[p + "conv_glu.fc1.bias"] = (4 * C,)
        sh[p + "conv_glu.dwconv.weight"] = (2 * C, 1, 3, 3)
        sh[p + "conv_glu.dwconv.bias"] = (2 * C,)
        sh[p + "conv_glu.fc2.weight"] = (C, 2 * C, 1, 1)
        sh[p + "conv_glu.fc2.bias"] = (C,)
    for i, (cin, cout) in enumerate(((2 * C, c3), (2 * c3, c2), (2 * c2, c1))):
        p = f"decoder.up{i + 1}."
        sh[p + "low_conv.weight"] = (cout, cin, 1, 3)
        sh[p + "low_conv.bias"] = (cout,)
        sh[p + "high_conv.conv.weight"] = (3 * cout, cin, 1, 3)
        sh[p + "high_conv.conv.bias"] = (3 * cout,)
    sh["decoder.mask_conv.0.weight"] = (2, c1, 2, 2)
    sh["decoder.mask_conv.0.bias"] = (2,)
    sh["decoder.mask_conv.1.gamma"] = (1, 1, 1, F)
    sh["decoder.mask_conv.1.beta"] = (1, 1, 1, F)
    sh["decoder.mask_conv.2.weight"] = (2,)
    sh["decoder.mask_conv.3.weight"] = (2, 2, 1, 1)
    sh["decoder.mask_conv.3.bias"] = (2,)
    sh["decoder.lsigmoid.slope"] = (F, 1, 1)
    return sh


def lisennet_state_dict(sd: Mapping[str, Tensor], cfg) -> Dict[str, Tensor]:
    """the checkpoint's floating-point tensors as fp32 (there is no reparameterisation to remove)"""
    return {k: torch.as_tensor(v).detach().to(torch.float32) for k, v in sd.items() if torch.as_tensor(v).is_floating_point()}


def lisennet_default_state_dict(cfg, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Random weights of the right shapes (benchmarks, smoke tests): unit norms, PReLU 0.25, fan-in-scaled matrices."""
    g = generator or torch.Generator().manual_seed(0)
    sd: Dict[str, Tensor] = {}
    for k, shp in lisennet_expected_shapes(cfg).items():
        leaf = k.split(".")[-1]
        if leaf == "gamma" or k.endswith("_norm.weight") or leaf == "slope":
            sd[k] = torch.ones(shp)
        elif leaf == "beta" or k.endswith("_norm.bias") or "bias" in leaf:
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("act.weight") or k.endswith("conv_1.2.weight") or k.endswith("mask_conv.2.weight"):
            sd[k] = torch.full(shp, 0.25)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.randn(shp, generator=g) * (1.0 / fan_in ** 0.5)
    return sd
