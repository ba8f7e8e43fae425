"""Pin the CPU oracle (oracle/fe_oracle.py) against golden vectors produced by the
imported reference (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from common import MODEL_KWARGS, build_oracle, load_golden, rms
from oracle.weightgen import make_input

GOLDENS = ["fe_t", "fe_b", "fe_m", "fe_l", "fe48_b", "fe48_l", "fe48_b_h480"]
ORACLE_GOLDENS = GOLDENS + ["fe_tk_b", "fe_dprnn_t", "fe_dprnn_b", "fe_dprnn_l", "fe_dpt_t", "fe_dpt_b", "fe_dpt_m", "fe_ln_b",           # (the C oracle restates the default model only)
                            "fe_s", "fe48_t", "fe48_s", "fe48_m", "fe_dprnn_s", "fe_dprnn_m", "fe_dpt_s"]      # r3: every shipped shape is pinned on the reference
NONCAUSAL_GOLDENS = ["fe_nc", "fe_nc24", "fe48_nc"]     # model: fastenhancer.noncausal - offline Model.forward only


def cache_checksum(a):
    """tools/gen_golden.py::cache_checksum"""
    v = np.asarray(a, np.float64).reshape(-1)
    ramp = np.cos(0.37 * np.arange(v.size, dtype=np.float64) + 0.11)
    return np.array([v.sum(), (v * v).sum(), (v * ramp).sum()], np.float64)
# fp32-vs-fp32 different summation orders: the reference's own fp32 noise floor is ~5e-7
# relative; allow 20x that.
REL = 1e-5


def _close(a, b, rel=REL, what=""):
    err = rms(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    ref = rms(b)
    assert err <= rel * max(ref, 1e-3), f"{what}: rms err {err:.3e} vs ref rms {ref:.3e}"


@pytest.mark.parametrize("name", ORACLE_GOLDENS)
def test_streaming_step_matches_reference(name):
    g = load_golden(name)
    cfg, sd, fused, orc = build_oracle(name)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = make_input(B, hops * H, int(g["seed"]) + 1000, int(g["sr"]))
    caches = orc.initialize_cache(B)
    outs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
    _close(np.stack(outs, 0), g["stream_wav_out"], what="wav_out")
    _close(caches[0], g["stream_cache_stft"], what="cache_stft")
    _close(caches[1], g["stream_cache_istft"], what="cache_istft")
    n_model_caches = cfg.rf_blocks + (2 * cfg.n_layers if cfg.time_kernel else 0)     # time_kernel: + the causal convs' frame caches
    if cfg.dpt:
        n_model_caches = 2 * cfg.rf_blocks                                              # dptransformer: K and V caches per block
    assert len(caches) == 2 + n_model_caches
    for k in range(n_model_caches):
        if f"stream_h{k}" in g.files:                                                   # (dpt_b / dpt_m goldens hold the first and last block's in full,
            _close(caches[2 + k], g[f"stream_h{k}"], what=f"model cache {k}")
        else:                                                                           #  three checksums of the others)
            got, want = cache_checksum(caches[2 + k]), g[f"stream_h{k}_chk"]
            scale = np.sqrt(np.asarray(caches[2 + k], np.float64).size * max(want[1], 1e-12))      # |sum| <= sqrt(n * sum of squares)
            assert abs(got[0] - want[0]) <= 1e-5 * scale and abs(got[2] - want[2]) <= 1e-5 * scale, (k, got, want)
            assert abs(got[1] - want[1]) <= 1e-5 * want[1], (k, got, want)


@pytest.mark.parametrize("name", ORACLE_GOLDENS)
def test_spec_chunk_matches_reference(name):
    g = load_golden(name)
    cfg, sd, fused, orc = build_oracle(name)
    B, H = int(g["B"]), cfg.hop_size
    x = make_input(B, int(g["hops"]) * H, int(g["seed"]) + 1000, int(g["sr"]))
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(4):
        s, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s)
    spec = np.concatenate(specs, axis=2)
    h0 = None if cfg.dpt else orc.initialize_cache(B)[2:]      # (dptransformer: the reference's chunk runs without caches, start masked)
    y, h = orc.spec_forward(spec, h0)
    _close(y, g["chunk_spec_out"], what="chunk spec")
    _close(h[-1], g["chunk_h_last"], what="chunk h")


@pytest.mark.parametrize("name", ORACLE_GOLDENS + NONCAUSAL_GOLDENS)
def test_offline_matches_reference(name):
    g = load_golden(name)
    cfg, sd, fused, orc = build_oracle(name)
    B, H = int(g["B"]), cfg.hop_size
    x = make_input(B, int(g["hops"]) * H + 37, int(g["seed"]) + 2000, int(g["sr"]))
    wav, spec = orc.offline_forward(x)
    assert wav.shape == g["offline_wav"].shape and spec.shape == g["offline_spec"].shape
    _close(wav, g["offline_wav"], what="offline wav")
    _close(spec, g["offline_spec"], what="offline spec")


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_tk_b", "fe_dprnn_b", "fe_dpt_b", "fe_ln_b"])
def test_driver_loop_matches_reference(name):
    g = load_golden(name)
    cfg, sd, fused, orc = build_oracle(name)
    length = int(g["long_length"])
    x = make_input(1, length, int(g["seed"]) + 3000, int(g["sr"]))
    y = orc.enhance_stream(x)
    assert y.shape == (1, length)
    _close(y[0], g["long_wav_out"], rel=3e-5, what="long run")


def test_fold_matches_reference_fused_state_dict():
    g = load_golden("fe_t")
    cfg, sd, fused, orc = build_oracle("fe_t")
    keys = [k[len("fused."):] for k in g.files if k.startswith("fused.")]
    assert sorted(keys) == sorted(fused.keys())
    for k in keys:
        np.testing.assert_allclose(fused[k], g["fused." + k], rtol=2e-6, atol=1e-7, err_msg=k)


def test_latency_identity_stft_istft():
    """docs/docs/onnx.md:37-72: streaming STFT->iSTFT reconstructs the input delayed by N-H."""
    for name in ("fe_b", "fe_l", "fe48_b"):
        cfg, sd, fused, orc = build_oracle(name, np.float64)
        H, N = cfg.hop_size, cfg.n_fft
        x = make_input(2, 20 * H, 7, 16000).astype(np.float64)
        c1, c2 = orc.initialize_cache(2)[:2]
        out = []
        for t in range(20):
            s, c1 = orc.stft_step(x[:, t * H:(t + 1) * H], c1)
            o, c2 = orc.istft_step(s, c2)
            out.append(o)
        y = np.concatenate(out, 1)
        d = N - H
        np.testing.assert_allclose(y[:, N:], x[:, N - d:-d] if d else x[:, N:], atol=1e-6)


@pytest.mark.parametrize("name", GOLDENS)
def test_c_oracle_matches_reference(name):
    """oracle/fe_oracle.c (the C/OpenMP restatement used as bench.py's cpu_baseline) vs the reference goldens."""
    from oracle.c_oracle import COracle
    g = load_golden(name)
    cfg, sd, fused, orc = build_oracle(name)
    co = COracle(cfg, fused, threads=2)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = make_input(B, hops * H, int(g["seed"]) + 1000, int(g["sr"]))
    cs, ci, h = co.initialize_cache(B)
    outs = [co.step(x[:, t * H:(t + 1) * H], cs, ci, h) for t in range(hops)]
    _close(np.stack(outs, 0), g["stream_wav_out"], what="wav_out")
    _close(cs, g["stream_cache_stft"], what="cache_stft")
    _close(ci, g["stream_cache_istft"], what="cache_istft")
    for k in range(cfg.rf_blocks):
        _close(h[k], g[f"stream_h{k}"][0], what=f"h{k}")


# ------------------------------------------------------------------------------------------------ BSRNN
@pytest.mark.parametrize("name", ["bsrnn_xxt", "bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_streaming_matches_reference(name):
    from common import build_bsrnn_oracle
    g = load_golden(name)
    cfg, sd, fused, orc = build_bsrnn_oracle(name)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = make_input(B, hops * H, int(g["seed"]) + 1000, int(g["sr"]))
    caches = orc.initialize_cache(B)
    outs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
    _close(np.stack(outs, 0), g["stream_wav_out"], what="wav_out")
    _close(caches[0], g["stream_cache_stft"], what="cache_stft")
    _close(caches[1], g["stream_cache_istft"], what="cache_istft")
    for i in range(2 * cfg.num_layers):
        _close(caches[2 + i], g[f"stream_c{i}"], what=f"lstm cache {i}")


@pytest.mark.parametrize("name", ["bsrnn_xxt", "bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_offline_matches_reference(name):
    from common import build_bsrnn_oracle
    g = load_golden(name)
    cfg, sd, fused, orc = build_bsrnn_oracle(name)
    x = make_input(int(g["B"]), int(g["hops"]) * cfg.hop_size + 37, int(g["seed"]) + 2000, int(g["sr"]))
    wav, spec = orc.offline_forward(x)
    _close(wav, g["offline_wav"], what="offline wav")
    _close(spec, g["offline_spec"], what="offline spec")


def test_fspen_streaming_and_offline_match_reference():
    """models/fspen/model.py - golden = the imported reference (tools/gen_golden.py::gen_fspen)"""
    from common import build_fspen_oracle
    g = load_golden("fspen")
    cfg, sd, fused, orc = build_fspen_oracle()
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = make_input(B, hops * H, int(g["seed"]) + 1000, int(g["sr"]))
    caches = orc.initialize_cache(B)
    outs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
    _close(np.stack(outs, 0), g["stream_wav_out"], what="wav_out")
    _close(caches[0], g["stream_cache_stft"], what="cache_stft")
    _close(caches[1], g["stream_cache_istft"], what="cache_istft")
    for i in range(cfg.n_caches):
        _close(caches[2 + i], g[f"stream_c{i}"], what=f"inter GRU cache {i}")
    xo = make_input(B, hops * H + 37, int(g["seed"]) + 2000, int(g["sr"]))
    wav, spec = orc.offline_forward(xo)
    _close(wav, g["offline_wav"], what="offline wav")
    _close(spec, g["offline_spec"], what="offline spec")


def test_lisennet_streaming_and_offline_match_reference():
    """models/lisennet/model.py - golden = the imported reference (tools/gen_golden.py::gen_lisennet).
    Streaming: 10 hops x 2 streams with all 9 model caches.  Offline: Model.forward's phase features are ill-conditioned on frame 0
    (a frame that reflect padding makes symmetric has a real spectrum; an all-zero frame has atan2(+-0, +-0)), so the oracle is
    pinned through the reference's own features (model_forward, mask, iSTFT) and its feature extraction on all other frames."""
    from common import build_lisennet_oracle
    g = load_golden("lisennet")
    cfg, sd, _, orc = build_lisennet_oracle()
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = make_input(B, hops * H, int(g["seed"]) + 1000, int(g["sr"]))
    caches = orc.initialize_cache(B)
    outs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
    _close(np.stack(outs, 0), g["stream_wav_out"], what="wav_out")
    _close(caches[0], g["stream_cache_stft"], what="cache_stft")
    _close(caches[1], g["stream_cache_istft"], what="cache_istft")
    for i in range(cfg.n_caches):
        _close(caches[2 + i], g[f"stream_c{i}"], what=f"model cache {i}")
    xo = make_input(B, hops * H + 37, int(g["seed"]) + 2000, int(g["sr"]))
    xo[:, :int(g["offline_leading_zeros"])] = 0.0
    wav, spec = orc.offline_forward(xo, feat=g["offline_feat"])
    _close(wav, g["offline_wav"], what="offline wav")
    _close(spec, g["offline_spec"], what="offline spec")
    # the oracle's own feature extraction (torch.diff convention): every frame but the ill-conditioned first one
    N = cfg.n_fft
    xp = np.pad(xo, ((0, 0), (N // 2, N // 2)), mode="reflect")
    T = 1 + xo.shape[1] // H
    X = np.fft.rfft(np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * orc.window, axis=2)
    sp = np.stack([X.real, X.imag], axis=-1).astype(np.float32).transpose(0, 2, 1, 3) + np.float32(0.0)
    sp = sp * np.maximum(np.sqrt(sp[..., 0:1] ** 2 + sp[..., 1:2] ** 2), np.float32(1e-5)) ** np.float32(cfg.input_compression - 1.0)
    feat, _ = orc.features(sp, None, False)
    ref = g["offline_feat"]
    assert np.abs(feat[:, 0] - ref[:, 0]).max() < 1e-4
    d = np.abs(feat[:, :, 1:] - ref[:, :, 1:])
    d[:, 2, 0] = 0.0          # ifd of frame 1 looks back at frame 0's phase
    assert d.max() < 2e-3, d.max()
