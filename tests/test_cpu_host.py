"""CPU tests of the host logic and of the C ABI surface (no compute, no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from common import MODEL_KWARGS, build_oracle, load_golden, product_config
from fastenhancer_amd import _lib
from fastenhancer_amd.config import FEConfig
from fastenhancer_amd.engine import Engine
from fastenhancer_amd.weights import check_fused, default_state_dict, expected_fused_shapes, fold_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "fastenhancer_hip.h")).read()
    declared = set(re.findall(r"\b(fe_[a-z_]+)\s*\(", header))
    declared -= {"fe_handle", "fe_config"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.fe_version()


def test_library_carries_the_digest_of_the_sources_it_was_built_from(monkeypatch):
    """r6 (engineering): fe_build_key() is fastenhancer_amd.build.source_key() of the tree - csrc/* and the C-ABI header - and the binding refuses an
    in-tree library built from other sources (a stale object cache cannot pass for the shipped sources); a side build named by FASTENHANCER_HIP_LIB is not checked."""
    from fastenhancer_amd import build as fbuild
    lib = _lib.load()
    assert lib.fe_build_key().decode() == fbuild.source_key()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(fbuild, "source_key", lambda: "0000000000000000")
    with pytest.raises(_lib.FEError, match="built from other sources"):
        _lib.load()
    monkeypatch.setenv("FASTENHANCER_HIP_LIB", _lib.LIB_PATH)          # (a side build: loaded as it is)
    assert _lib.load().fe_build_key() == lib.fe_build_key()
    monkeypatch.setattr(_lib, "_lib", lib)


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_s", "fe_m", "fe_l", "fe48_t", "fe48_b", "fe48_s", "fe48_m", "fe48_l", "fe48_b_h480", "fe_tk_b",
                                  "fe_dprnn_t", "fe_dprnn_b", "fe_dprnn_s", "fe_dprnn_m", "fe_dprnn_l", "fe_dpt_t", "fe_dpt_b", "fe_dpt_s", "fe_dpt_m", "fe_ln_b",
                                  "fe_nc", "fe_nc24", "fe48_nc"])
def test_section_table_matches_fused_schema(name):
    cfg = product_config(name)
    eng = Engine(cfg, None)
    exp = expected_fused_shapes(cfg)
    assert [s[0] for s in eng.sections] == list(exp.keys())
    end = 0
    for (sname, off, cnt) in eng.sections:
        assert cnt == int(np.prod(exp[sname])) and off % 4 == 0 and off >= end
        end = off + cnt
    assert end == eng.weight_floats
    orc_cfg = build_oracle(name)[0]
    assert eng.flops_per_frame == pytest.approx(orc_cfg.flops_per_frame())


def test_flops_match_survey_table():
    # T 1.938 M, B 8.392 M, 48k-B 16.04 M FLOPs per frame
    for name, want in (("fe_t", 1.938e6), ("fe_b", 8.392e6), ("fe48_b", 16.04e6)):
        eng = Engine(FEConfig.from_model_kwargs(**MODEL_KWARGS[name][0]), None)
        assert eng.flops_per_frame == pytest.approx(want, rel=1e-3)


def test_unsupported_shapes_are_rejected_with_a_message():
    kw = dict(MODEL_KWARGS["fe_b"][0])
    kw["channels"] = 40
    with pytest.raises(_lib.FEError, match="no kernel compiled"):
        Engine(FEConfig.from_model_kwargs(**kw), None)
    lib = _lib.load()
    c = _lib.fe_config()
    c.arch, c.n_fft, c.hop_size, c.win_size = 0, 511, 256, 511
    h = ctypes.c_void_p()
    assert lib.fe_create(ctypes.byref(c), ctypes.byref(h)) < 0
    assert b"even" in lib.fe_last_error()


def test_every_shipped_yaml_constructs_a_mirror_and_a_kernel():
    """all 41 configs/**/*.yaml of the reference (tests/golden/yaml_kwargs.json, dumped by tools/dump_yaml_kwargs.py): the module named by
    the yaml's `model:` key exists under fastenhancer_amd.models, its class takes the yaml's model_kwargs verbatim, and fe_create
    finds a compiled kernel for the shape (the reject list is empty)."""
    import importlib
    from common import YAML_KWARGS
    assert len(YAML_KWARGS) == 41
    rejected = []
    for path, y in sorted(YAML_KWARGS.items()):
        mod = importlib.import_module(f"fastenhancer_amd.models.{y['model']}.model")
        for cls in ("Model", "ONNXModel"):
            if y["model"] == "fastenhancer.noncausal" and cls == "ONNXModel":
                assert not hasattr(mod, cls)            # the reference module defines the offline Model only
                continue
            m = getattr(mod, cls)(**y["model_kwargs"])
            try:
                Engine(m.cfg, None)                     # fe_create: a kernel is compiled for this shape (no GPU needed)
            except _lib.FEError as e:
                rejected.append((path, cls, str(e)))
    assert rejected == []


def test_noncausal_host_side_without_gpu():
    """model: fastenhancer.noncausal - sections (both GRU directions, rnn_fc over 2 C2), no model caches, FLOPs, loud failure without a GPU"""
    import importlib
    from common import MODEL_MODULE
    kw = MODEL_KWARGS["fe_nc"][0]
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE['fe_nc']}.model")
    m = mod.Model(**kw)
    cfg = m.cfg
    assert cfg.noncausal and cfg.rf_channels == 128 and cfg.rf_blocks == 6
    eng = Engine(cfg, None)
    names = [s[0] for s in eng.sections]
    assert "rf_block.5.rnn.weight_hh_l0_reverse" in names and "rf_block.0.rnn.bias_ih_l0_reverse" in names
    assert dict((s[0], s[2]) for s in eng.sections)["rf_block.0.rnn_fc.weight"] == 128 * 256
    assert eng.state_floats(3) == 3 * 2 * cfg.cache_len            # the two STFT caches only: the model has none
    cfg_o = build_oracle("fe_nc")[0]
    assert eng.flops_per_frame == pytest.approx(cfg_o.flops_per_frame())
    _, sd, _, _ = build_oracle("fe_nc")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    with pytest.raises(_lib.FEError, match="GPU"):
        m(torch.zeros(1, 4000))


def test_config_rejects_what_the_reference_rejects():
    kw = dict(MODEL_KWARGS["fe_b"][0])
    with pytest.raises(AssertionError):
        FEConfig.from_model_kwargs(**{**kw, "n_fft": 511})
    with pytest.raises(AssertionError):
        FEConfig.from_model_kwargs(**{**kw, "win_size": 1024})
    with pytest.raises(RuntimeError):
        FEConfig.from_model_kwargs(**{**kw, "mask": "softmax"})


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_l", "fe48_b", "fe_tk_b", "fe_dprnn_b", "fe_dpt_b", "fe_ln_b", "fe_nc", "fe48_nc"])
def test_host_fold_matches_oracle_fold(name):
    cfg_o, sd, fused_o, _ = build_oracle(name)
    cfg = product_config(name)
    fused = fold_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, cfg)
    assert set(fused) == set(fused_o)
    for k in fused:
        np.testing.assert_allclose(fused[k].numpy(), fused_o[k], rtol=3e-6, atol=1e-7, err_msg=k)
    check_fused(fused, cfg)
    # idempotent on fused input
    again = fold_state_dict(fused, cfg)
    for k in fused:
        assert torch.equal(again[k], fused[k])


def test_host_fold_matches_reference_fused_state_dict():
    g = load_golden("fe_t")
    cfg_o, sd, _, _ = build_oracle("fe_t")
    cfg = FEConfig.from_model_kwargs(**MODEL_KWARGS["fe_t"][0])
    fused = fold_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, cfg)
    for k in fused:
        np.testing.assert_allclose(fused[k].numpy(), g["fused." + k], rtol=3e-6, atol=1e-7, err_msg=k)


def test_strict_loading_errors():
    cfg = FEConfig.from_model_kwargs(**MODEL_KWARGS["fe_t"][0])
    sd = default_state_dict(cfg)
    check_fused(sd, cfg)
    bad = dict(sd)
    bad.pop("rf_post.1.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        check_fused(bad, cfg)
    bad = dict(sd)
    bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        check_fused(bad, cfg)
    check_fused(bad, cfg, strict=False)
    bad = dict(sd)
    bad["enc_pre.0.weight"] = torch.zeros(3, 8, 2)
    with pytest.raises(RuntimeError, match="size mismatch"):
        check_fused(bad, cfg)


def test_blob_roundtrip_layout():
    cfg = FEConfig.from_model_kwargs(**MODEL_KWARGS["fe_b"][0])
    eng = Engine(cfg, None)
    sd = default_state_dict(cfg, torch.Generator().manual_seed(3))
    blob = eng.make_blob(sd)
    assert blob.dtype == torch.float32 and blob.numel() == eng.weight_floats
    for name, off, cnt in eng.sections:
        assert torch.equal(blob[off:off + cnt], sd[name].reshape(-1))


def test_compute_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import importlib
    mod = importlib.import_module("fastenhancer_amd.models.fastenhancer.default.model")
    m = mod.ONNXModel(**MODEL_KWARGS["fe_t"][0])
    caches = m.initialize_cache(torch.zeros(2, 1))
    assert [tuple(c.shape) for c in caches] == [(1, 2 * 16, 20)] * 2
    assert [tuple(c.shape) for c in m.stft.initialize_cache(torch.zeros(2, 1))] == [(2, 256)] * 2
    with pytest.raises(_lib.FEError, match="no CPU fallback"):
        m(torch.zeros(2, 257, 1, 2))


def test_cli_host_logic(tmp_path):
    """config / checkpoint lookup and WAV I/O of the command-line callers (no GPU)."""
    import yaml
    from fastenhancer_amd.scripts.common import latest_checkpoint, load_hparams, read_wav, write_wav
    d = tmp_path / "logs" / "run1"
    d.mkdir(parents=True)
    (d / "config.yaml").write_text(yaml.safe_dump({"model": "fastenhancer.default", "model_kwargs": MODEL_KWARGS["fe_t"][0],
                                                   "data": {"sampling_rate": 16000}}))
    for e in (20, 500, 100):
        (d / f"{e:05d}.pth").write_bytes(b"")
    (d / "notes.pth").write_bytes(b"")
    assert os.path.basename(latest_checkpoint(str(d))) == "00500.pth"
    assert latest_checkpoint(str(tmp_path / "missing")) is None
    hps = load_hparams(None, "run1", log_root=str(tmp_path / "logs"))
    assert hps["model_kwargs"]["channels"] == 24
    with pytest.raises(ValueError):
        load_hparams(None, None)
    x = (0.5 * np.sin(np.arange(1000) / 10.0)).astype(np.float32)
    write_wav(str(tmp_path / "a.wav"), 16000, x)
    np.testing.assert_allclose(read_wav(str(tmp_path / "a.wav"), 16000), x)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "b.wav"), 16000, (x * 32767).astype(np.int16))
    np.testing.assert_allclose(read_wav(str(tmp_path / "b.wav"), 16000), x, atol=1e-4)
    with pytest.raises(ValueError, match="sampling rate"):
        read_wav(str(tmp_path / "a.wav"), 48000)


def test_si_sdr_is_pinned_on_the_reference_function():
    """f3: tests/golden/si_snr.npz holds outputs of the reference's own si_snr (scripts/metrics_ns.py:38-52)."""
    from fastenhancer_amd.metrics import masked_si_snr, si_snr
    from oracle.fe_oracle import si_sdr
    g = load_golden("si_snr")
    clean, est, lens = g["clean"], g["est"], g["lens"]
    mask = (np.arange(clean.shape[1])[None] < lens[:, None]).astype(np.float32)
    tc, te, tm = torch.from_numpy(clean), torch.from_numpy(est), torch.from_numpy(mask)
    for got, want in ((si_snr(te, tc), g["full"]), (si_snr(te * tm, tc * tm, tm), g["masked"]), (si_snr(te, tc, tm), g["fn_only"]),
                      (masked_si_snr(te, tc, torch.from_numpy(lens)), g["masked"])):
        np.testing.assert_allclose(got.numpy(), want, rtol=2e-6, atol=2e-5)
    for got, want in ((si_sdr(clean, est), g["full"]), (si_sdr(clean * mask, est * mask, mask), g["masked"]), (si_sdr(clean, est, mask), g["fn_only"])):
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-5)
    assert not np.allclose(g["masked"][1:], g["fn_only"][1:], atol=1e-3)      # masking inside vs outside differ: the pin can tell


def test_bsrnn_fold_accepts_both_time_lstm_key_forms_and_names_missing_keys():
    """a state_dict of the reference's ONNXModel BEFORE remove_weight_reparameterizations has BN stats and
    the time LSTM already renamed (no `_l0`); the offline Model's has `_l0`.  Both must fold to the same tensors."""
    from common import BSRNN_KWARGS
    from fastenhancer_amd.config import BSRNNConfig
    from fastenhancer_amd.weights import bsrnn_fold_state_dict
    from oracle import bsrnn_oracle as bo
    kw = BSRNN_KWARGS["bsrnn_xxt"][0]
    cfg = BSRNNConfig.from_model_kwargs(**kw)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in bo.make_training_state_dict(bo.BSRNNConfig.from_model_kwargs(kw), 5).items()}
    a = bsrnn_fold_state_dict(sd, cfg)
    renamed = {(k[:-3] if k.startswith("rnn_time.") and k.endswith("_l0") else k): v for k, v in sd.items()}
    assert any(k.startswith("rnn_time.") and not k.endswith("_l0") for k in renamed)
    b = bsrnn_fold_state_dict(renamed, cfg)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    broken = {k: v for k, v in sd.items() if not k.startswith("rnn_time.0.weight_hh")}
    with pytest.raises(RuntimeError, match="Missing key"):
        bsrnn_fold_state_dict(broken, cfg)


def test_config_rejects_unsupported_initialisations():
    kw = dict(MODEL_KWARGS["fe_b"][0])
    with pytest.raises(RuntimeError, match="pre_post_init"):
        FEConfig.from_model_kwargs(**{**kw, "pre_post_init": "mel"})
    rk = dict(kw["rnnformer_kwargs"], positional_embedding=None)
    with pytest.raises(RuntimeError, match="positional_embedding"):
        FEConfig.from_model_kwargs(**{**kw, "rnnformer_kwargs": rk})


def test_time_kernel_mirror_state_layout_and_defaults():
    """fastenhancer.time_kernel mirror: yaml kwargs -> config, cache list in the reference's order / shapes, default
    initialisation (its own filterbank formula), packing of the cache list into the C ABI state order and back."""
    import importlib
    from common import MODEL_MODULE
    from fastenhancer_amd.weights import linear_filterbank_time_kernel
    from oracle.fe_oracle import linear_filterbank_tk
    kw = MODEL_KWARGS["fe_tk_b"][0]
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE['fe_tk_b']}.model")
    m = mod.ONNXModel(**kw)
    cfg = m.cfg
    assert cfg.time_kernel and cfg.kernel_size_time == 3 and cfg.kernel_size == (8, 3, 3) and not cfg.final_scale_exp
    caches = m.initialize_cache(torch.zeros(2, 1))
    assert [tuple(c.shape) for c in caches] == [(2, 48, 2, 64)] * 2 + [(1, 48, 36)] * 3 + [(2, 48, 2, 64)] * 2
    pre, post = linear_filterbank_time_kernel(64, 24)
    pre_o, post_o = linear_filterbank_tk(64, 24)
    np.testing.assert_allclose(pre.numpy(), pre_o, atol=2e-6)
    np.testing.assert_allclose(post.numpy(), post_o, atol=2e-6)
    np.testing.assert_allclose(m.state_dict()["rf_pre.0.weight"].numpy(), pre_o, atol=2e-6)
    eng = Engine(cfg, None)
    B = 2
    assert eng.state_floats(B) == B * (2 * 256 + 3 * 24 * 36 + 4 * 2 * 64 * 48)
    rng = np.random.default_rng(0)
    full = [torch.from_numpy(rng.standard_normal(s_).astype(np.float32)) for s_ in [(B, 256)] * 2 + [tuple(c.shape) for c in caches]]
    state = eng.pack_state(full, B)
    back = eng.split_state(state, B)
    assert len(back) == len(full) and all(torch.equal(a, b) for a, b in zip(back, full))
    assert eng.flops_per_frame == pytest.approx(build_oracle("fe_tk_b")[0].flops_per_frame())


def test_dprnn_mirror_loads_the_reference_module_names():
    """fastenhancer.dprnn mirror: yaml kwargs -> config (channels_frnn, no positional embedding, its own filterbank formula);
    a checkpoint with the reference's module names (dprnn_pre / dprnn_block.k.trnn / frnn / dprnn_post, training form) folds
    to the same fused weights as the oracle's fold; channels_frnn != channels / 2 is rejected with a message."""
    import importlib
    from common import MODEL_MODULE
    from fastenhancer_amd.config import dprnn_config
    from oracle.fe_oracle import linear_filterbank_tk, reference_key
    kw = MODEL_KWARGS["fe_dprnn_b"][0]
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE['fe_dprnn_b']}.model")
    m = mod.ONNXModel(**kw)
    cfg = m.cfg
    assert cfg.dprnn and cfg.channels_frnn == 18 and cfg.positional_embedding is None and not cfg.final_scale_exp and cfg.rf_eps == 1e-5
    assert [tuple(c.shape) for c in m.initialize_cache(torch.zeros(2, 1))] == [(1, 2 * 24, 36)] * 3
    pre_o, _ = linear_filterbank_tk(64, 24)
    np.testing.assert_allclose(m.state_dict()["rf_pre.0.weight"].numpy(), pre_o, atol=2e-6)
    cfg_o, sd, fused_o, _ = build_oracle("fe_dprnn_b")
    ref_named = {reference_key(k, cfg_o): torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    assert any(k.startswith("dprnn_block.0.trnn.") for k in ref_named) and any(".frnn.parametrizations." in k for k in ref_named)
    m.load_state_dict(ref_named, strict=True)
    fused = fold_state_dict(ref_named, cfg)
    assert set(fused) == set(fused_o) and "rf_block.0.pe" not in fused
    for k in fused:
        np.testing.assert_allclose(fused[k].numpy(), fused_o[k], rtol=3e-6, atol=1e-7, err_msg=k)
    with pytest.raises(RuntimeError, match="channels_frnn"):
        dprnn_config(**{**kw, "dprnn_kwargs": dict(kw["dprnn_kwargs"], channels_frnn=16)})


def test_dptransformer_mirror_loads_the_reference_module_names():
    """fastenhancer.dptransformer mirror: config (lookbehind 31, per-block K / V caches), a checkpoint with the reference's module
    names (dpt_pre / dpt_block.k.time_attn / freq_attn / dpt_post / pe) folds to the oracle's fused weights; state size; pre_norm
    (the reference's default) and other lookbehinds are rejected with a message."""
    import importlib
    from common import MODEL_MODULE
    from fastenhancer_amd.config import dpt_config
    from oracle.fe_oracle import reference_key
    kw = MODEL_KWARGS["fe_dpt_b"][0]
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE['fe_dpt_b']}.model")
    m = mod.ONNXModel(**kw)
    cfg = m.cfg
    assert cfg.dpt and cfg.lookbehind == 31 and cfg.rf_eps == 1e-5 and not cfg.final_scale_exp
    assert [tuple(c.shape) for c in m.initialize_cache(torch.zeros(2, 1))] == [(2 * 24, 4, 31, 9)] * 6
    assert tuple(m.state_dict()["time_pe"].shape) == (4, 32)
    cfg_o, sd, fused_o, _ = build_oracle("fe_dpt_b")
    ref_named = {reference_key(k, cfg_o): torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    assert "pe" in ref_named and any(k.startswith("dpt_block.0.time_attn.qkv.parametrizations") for k in ref_named) and "dpt_block.0.freq_fc.weight" in ref_named
    m.load_state_dict(ref_named, strict=True)
    fused = fold_state_dict(ref_named, cfg)
    assert set(fused) == set(fused_o)
    for k in fused:
        np.testing.assert_allclose(fused[k].numpy(), fused_o[k], rtol=3e-6, atol=1e-7, err_msg=k)
    eng = Engine(cfg, None)
    assert eng.state_floats(3) == 3 * (2 * 256 + 6 * 24 * 36 * 31 + 1)      # + the ring head per stream
    with pytest.raises(RuntimeError, match="pre_norm"):
        dpt_config(**{**kw, "dpt_kwargs": {k: v for k, v in kw["dpt_kwargs"].items() if k != "pre_norm"}})
    with pytest.raises(RuntimeError, match="lookbehind"):
        dpt_config(**{**kw, "dpt_kwargs": dict(kw["dpt_kwargs"], lookbehind=16)})


def test_fspen_host_side_without_gpu():
    """FSPEN (models/fspen/model.py): the host fold equals the oracle's (which tools/gen_golden.py checked against the reference's
    fused state_dict), the C ABI's section table is exactly the fused schema, the blob layout round-trips, flops follow
    models/fspen/macs.py, and other architectures are rejected with a message."""
    from common import FSPEN_KWARGS
    from fastenhancer_amd import _lib
    from fastenhancer_amd.config import FSPENConfig
    from fastenhancer_amd.engine import Engine
    from fastenhancer_amd.weights import fspen_expected_fused_shapes, fspen_fold_state_dict
    from oracle import fspen_oracle as fo
    kw, sr, seed = FSPEN_KWARGS
    cfg = FSPENConfig.from_model_kwargs(**kw)
    ocfg = fo.FSPENConfig.from_model_kwargs(kw)
    sd = fo.make_training_state_dict(ocfg, seed)
    mine = fspen_fold_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, cfg)
    ref = fo.fold_state_dict(sd, ocfg)
    assert set(mine) == set(ref) == set(fspen_expected_fused_shapes(cfg))
    for k in ref:
        assert tuple(mine[k].shape) == fspen_expected_fused_shapes(cfg)[k]
        np.testing.assert_allclose(mine[k].numpy(), ref[k], rtol=2e-6, atol=1e-7, err_msg=k)
    again = fspen_fold_state_dict(mine, cfg)              # an already fused dict passes through
    assert all(torch.equal(again[k], mine[k]) for k in mine)
    eng = Engine(cfg, None)
    assert {n for n, _, _ in eng.sections} == set(ref)
    blob = eng.make_blob({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    for n, off, cnt in eng.sections:
        np.testing.assert_array_equal(blob[off:off + cnt].numpy(), mine[n].reshape(-1).numpy())
    assert abs(eng.flops_per_frame - ocfg.flops_per_frame()) < 1.0
    assert eng.state_floats(3) == 3 * (2 * 256 + 24 * 4 * 16)
    with pytest.raises(_lib.FEError, match="no FSPEN kernel compiled"):
        Engine(FSPENConfig.from_model_kwargs(**dict(kw, hop_size=128)), None)
    with pytest.raises(RuntimeError, match="is not supported"):
        FSPENConfig.from_model_kwargs(**dict(kw, dpe_kwargs=dict(kw["dpe_kwargs"], norm="BatchNorm")))
    with pytest.raises(AssertionError, match="Only n_fft == 512"):
        FSPENConfig.from_model_kwargs(**dict(kw, n_fft=1024))


def test_lisennet_host_side_without_gpu():
    """LiSenNet (models/lisennet/model.py): the C ABI's section table is the checkpoint schema, the blob layout round-trips, flops
    follow models/lisennet/macs.py, the state is the reference's cache list, other architectures are rejected with a message."""
    from common import LISENNET_KWARGS
    from fastenhancer_amd import _lib
    from fastenhancer_amd.config import LiSenNetConfig
    from fastenhancer_amd.engine import Engine
    from fastenhancer_amd.weights import lisennet_expected_shapes
    from oracle import lisennet_oracle as lo
    kw, sr, seed = LISENNET_KWARGS
    cfg = LiSenNetConfig.from_model_kwargs(**kw)
    ocfg = lo.LiSenNetConfig.from_model_kwargs(kw)
    sd = lo.make_state_dict(ocfg, seed)
    assert {k: tuple(v.shape) for k, v in sd.items()} == lisennet_expected_shapes(cfg) == lo.state_dict_spec(ocfg)
    eng = Engine(cfg, None)
    assert {n for n, _, _ in eng.sections} == set(sd)
    blob = eng.make_blob({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    for n, off, cnt in eng.sections:
        np.testing.assert_array_equal(blob[off:off + cnt].numpy(), sd[n].reshape(-1))
    assert abs(eng.flops_per_frame - ocfg.flops_per_frame()) < 1.0
    assert eng.state_floats(3) == 3 * 2 * 256 + sum(int(np.prod(s)) for s in cfg.cache_shapes(3))
    assert cfg.cache_shapes(2) == ocfg.cache_shapes(2)
    with pytest.raises(_lib.FEError, match="no LiSenNet kernel compiled"):
        Engine(LiSenNetConfig.from_model_kwargs(**dict(kw, n_blocks=3)), None)
    with pytest.raises(RuntimeError, match="not supported"):
        LiSenNetConfig.from_model_kwargs(**dict(kw, normalized=True))
