"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against (1) the golden
vectors produced by the imported reference and (2) the numpy oracle on fresh seeded inputs.

Tolerance (BASELINE.json north_star): enhanced-waveform RMS error < 1e-4 absolute; because the
synthetic checkpoints emit audio of RMS 0.1-0.5 we ALSO require 1e-4 relative to the reference
RMS, which is the tighter of the two.  Observed: ~1e-6 relative."""
import importlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from common import MODEL_KWARGS, MODEL_MODULE, build_oracle, load_golden, rms
from oracle.weightgen import make_input

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-4
REL_TOL = 1e-4
# Regression bounds, relative rms (floor 1e-3 on the reference rms), next to the north_star bound above: what exact-fp32 kernels against an
# fp32 oracle deliver, so that a fast-math exp, a dropped summation order or a bf16 staging slip - two lost digits - turns the suite red.
# Per family: ~5x the LARGEST error a full run of the suite recorded (tests/golden/parity_observed_r5.json = profiles/r5_parity_observed.json,
# written by FE_RECORD_PARITY on the MI355X: fastenhancer 3.8e-6 (a dprnn_s stage), bsrnn 1.8e-6, fspen 1.9e-6, lisennet 9.6e-7);
# per test: 5x what THAT test recorded (floor 2e-6), whichever is smaller.
TIGHT_REL = {"fastenhancer": 2e-5, "bsrnn": 1e-5, "fspen": 1e-5, "lisennet": 5e-6}
try:
    _OBSERVED_R5 = {k: v["rel_rms"] for k, v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_observed_r5.json"))).items()}
except FileNotFoundError:
    _OBSERVED_R5 = {}
GPU_SHAPES = ["fe_t", "fe_b", "fe_m", "fe_l", "fe48_b", "fe48_l", "fe48_b_h480", "fe_tk_b", "fe_dprnn_t", "fe_dprnn_b", "fe_dprnn_l",
              "fe_dpt_t", "fe_dpt_b", "fe_dpt_m", "fe_ln_b",
              "fe_s", "fe48_t", "fe48_s", "fe48_m", "fe_dprnn_s", "fe_dprnn_m", "fe_dpt_s"]          # shapes with reference goldens (r3: every shipped shape)
NONCAUSAL = ["fe_nc", "fe_nc24", "fe48_nc"]           # model: fastenhancer.noncausal (offline Model only; time-batched engine)
TB_SHAPES = ["fe_t", "fe_b", "fe_s", "fe_m", "fe_l", "fe48_t", "fe48_b", "fe48_s", "fe48_m", "fe48_l", "fe48_b_h480"]      # default model: both offline engines
ALL_SHAPES = ["fe_t", "fe_b", "fe_s", "fe_m", "fe_l", "fe48_t", "fe48_b", "fe48_s", "fe48_m", "fe48_l", "fe48_b_h480", "fe_tk_b",
              "fe_dprnn_t", "fe_dprnn_b", "fe_dprnn_s", "fe_dprnn_m", "fe_dprnn_l", "fe_dpt_t", "fe_dpt_b", "fe_dpt_s", "fe_dpt_m", "fe_ln_b"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


# test id -> largest relative rms error any of its comparisons saw; FE_RECORD_PARITY=<file> writes the table at session end
# (tests/conftest.py) - the per-family tight bounds below are ~5x what a full run of this table shows
OBSERVED = {}


def _family():
    tid = os.environ.get("PYTEST_CURRENT_TEST", "")
    for fam in ("bsrnn", "fspen", "lisennet"):
        if fam in tid:
            return fam
    return "fastenhancer"


def _assert_close(got, ref, what, tight=None):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    err, r = rms(got - ref), rms(ref)
    rel = err / max(r, 1e-3)
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    if rel > OBSERVED.get(tid, (0.0, ""))[0]:
        OBSERVED[tid] = (rel, what)
    # (1) the north_star bound: absolute 1e-4 for audio-level signals (rms <= 1), relative 1e-4 always
    assert err < ABS_TOL * max(1.0, r) and err <= REL_TOL * max(r, 1e-3), f"{what}: rms err {err:.3e}, ref rms {r:.3e} (north_star bound 1e-4)"
    # (2) the regression bound: what the kernels deliver (exact fp32 arithmetic, fp32 oracle), per family - the hard assert; the per-test
    # bound (5x what THIS test recorded in r5, keyed on the node id of one recorded run) is reported as a warning only (a benign
    # compiler / ROCm update or another rootdir must not turn the suite red)
    bound = TIGHT_REL[_family()] if tight is None else tight
    assert rel <= bound, f"{what}: relative rms err {rel:.3e} > {bound:.1e} (the kernels deliver ~{bound / 5:.0e}: a precision regression)"
    per_test = max(5.0 * _OBSERVED_R5.get(tid, 1.0), 2e-6)
    if tight is None and rel > per_test:
        warnings.warn(f"{what}: relative rms err {rel:.3e} is above 5x this test's recorded r5 value ({per_test:.1e})")
    return err / max(r, 1e-12)


def _model(name, cls="ONNXModel"):
    kw, sr, seed = MODEL_KWARGS[name]
    cfg, sd, fused, orc = build_oracle(name)
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE[name]}.model")
    m = getattr(mod, cls)(**kw).to(_dev()).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, sr, seed


@pytest.mark.parametrize("name", GPU_SHAPES)
def test_streaming_step_matches_reference_golden(name):
    from fastenhancer_amd.streaming import StreamingModel
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name)
    M = StreamingModel(m)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to(_dev())
    caches = M.initialize_cache(x)
    outs = []
    for t in range(hops):
        wav_out, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        outs.append(wav_out.cpu().numpy())
    _assert_close(np.stack(outs, 0), g["stream_wav_out"], "wav_out")
    _assert_close(caches[0].cpu().numpy(), g["stream_cache_stft"], "cache_stft")
    _assert_close(caches[1].cpu().numpy(), g["stream_cache_istft"], "cache_istft")
    for k in range(len(caches) - 2):         # the GRU states (time_kernel variant: encoder conv caches, GRU states, decoder conv caches)
        if f"stream_h{k}" in g.files:        # (dpt_b / dpt_m goldens hold the first and the last block's K / V caches only)
            _assert_close(caches[2 + k].cpu().numpy(), g[f"stream_h{k}"], f"model cache {k}")


@pytest.mark.parametrize("name", GPU_SHAPES)
def test_spec_step_matches_reference_golden(name):
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name)
    B, H = int(g["B"]), cfg.hop_size
    x = make_input(B, int(g["hops"]) * H, seed + 1000, sr)
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(4):                      # the spectra the reference was fed (oracle STFT == reference STFT, CPU-tested)
        s, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s)
    spec = torch.from_numpy(np.concatenate(specs, axis=2)).to(_dev())
    h0 = m.initialize_cache(spec)
    if cfg.dpt:       # the dptransformer reference runs a chunk without caches (frames before the start masked) and returns the 4 slots it filled
        spec_hat, *h = m(spec)
        _assert_close(spec_hat.cpu().numpy(), g["chunk_spec_out"], "spec_hat")
        _assert_close(h[-1][:, :, -4:].cpu().numpy(), g["chunk_h_last"], "h_last")
        return
    spec_hat, *h = m(spec, *h0)
    assert all(float(c.abs().max()) == 0.0 for c in h0), "input caches must not be modified"
    _assert_close(spec_hat.cpu().numpy(), g["chunk_spec_out"], "spec_hat")
    _assert_close(h[-1].cpu().numpy(), g["chunk_h_last"], "h_last")


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_dprnn_b", "fe_dpt_b", "fe_ln_b"])
def test_driver_loop_matches_reference_golden(name):
    from fastenhancer_amd.streaming import enhance_stream
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name)
    length = int(g["long_length"])
    x = torch.from_numpy(make_input(1, length, seed + 3000, sr))
    y1 = enhance_stream(m, x, frames_per_call=1).cpu().numpy()         # (default per-hop kernel: the 512-thread one where built)
    assert y1.shape == (1, length)
    _assert_close(y1[0], g["long_wav_out"], "long run T=1")
    # chunked launches run the 256-thread kernel: bit-identical to per-hop launches of the SAME kernel
    m.engine.set_step_kernel("waves4")
    y1w = enhance_stream(m, x, frames_per_call=1).cpu().numpy()
    y16 = enhance_stream(m, x, frames_per_call=16).cpu().numpy()
    _assert_close(y1w[0], g["long_wav_out"], "long run T=1, 256-thread kernel")
    assert np.array_equal(y1w, y16), "chunked launches must be bit-identical to per-hop launches"
    # (two summation orders / pre-scaled gate rows through a 198-hop recurrence: a few ulp of the output scale; both are checked against the golden above)
    assert np.abs(y1 - y1w).max() <= 5e-6 * max(1.0, np.abs(y1w).max()), "the two per-hop kernels agree to fp32 rounding"


@pytest.mark.parametrize("name", ALL_SHAPES)
def test_every_stage_matches_oracle(name):
    """Per-stage activations (fe_debug_step) vs the oracle's taps on a fresh input, B=3."""
    m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    B, hops, H = 3, 4, cfg.hop_size
    x = make_input(B, hops * H, 4242, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    for t in range(hops):
        taps = {}
        o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
        o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
        for sname, r, c, off in eng.debug_stages():
            tap = taps[sname]
            if sname in ("spec_in", "spec_out", "compressed", "mask"):
                ref = tap[:, :, 0, :]
            elif sname.startswith("rf_pre") or sname.startswith("rf_block"):
                ref = tap[0]
            else:
                ref = tap.transpose(0, 2, 1)
            _assert_close(dumps[sname].cpu().numpy(), ref, f"hop {t} stage {sname}")
        _assert_close(o_gpu.cpu().numpy(), o_ref, f"hop {t} wav_out")


@pytest.mark.parametrize("name", ALL_SHAPES)
def test_chunked_launch_is_bit_identical_to_per_hop_launches(name):
    """T hops in one launch == T launches of one hop (also exercises re-use of the LDS / global scratch across frames)."""
    m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    B, T, H = 4, 6, cfg.hop_size
    x = torch.from_numpy(make_input(B, T * H, 31, sr)).to(_dev())
    eng.set_step_kernel("waves4")       # (the 512-thread per-hop kernel agrees with it to fp32 rounding only: test_wg8_*)
    s1, s2 = eng.new_state(B), eng.new_state(B)
    y1 = eng.step(x, s1, T=T)
    y2 = torch.cat([eng.step(x[:, t * H:(t + 1) * H], s2, T=1) for t in range(T)], dim=1)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)
    # and against the oracle
    caches = orc.initialize_cache(B)
    refs = []
    for t in range(T):
        o, *caches = orc.step(x.cpu().numpy()[:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(y1.cpu().numpy(), np.concatenate(refs, 1), "chunk vs oracle")


@pytest.mark.parametrize("B", [1, 5, 256, 300])
def test_wg8_per_hop_kernel_matches_oracle_and_the_four_wave_kernel(B):
    """The 512-thread per-hop kernel (fe_frame8.hip.h; FastEnhancer_B): every output and every cache against the oracle and against the
    256-thread kernel on the same input; B = 300 runs it with persistent workgroups (more streams than CUs)."""
    m, orc, cfg, sr, seed = _model("fe_b")
    eng = m.engine
    hops, H = 6, cfg.hop_size
    x = make_input(B, hops * H, 777 + B, sr)
    xd = torch.from_numpy(x).to(_dev())
    outs, states = {}, {}
    for kern in ("waves4", "wg8_persist" if B > 256 else "wg8"):
        eng.set_step_kernel(kern)
        st = eng.new_state(B)
        outs[kern] = torch.cat([eng.step(xd[:, t * H:(t + 1) * H], st, T=1) for t in range(hops)], dim=1).cpu().numpy()
        states[kern] = st.cpu().numpy()
    k8 = "wg8_persist" if B > 256 else "wg8"
    assert np.abs(outs[k8] - outs["waves4"]).max() <= 5e-6 * max(1.0, np.abs(outs["waves4"]).max())
    assert np.abs(states[k8] - states["waves4"]).max() <= 5e-6 * max(1.0, np.abs(states["waves4"]).max())
    assert not np.array_equal(outs[k8], outs["waves4"]) or B == 0, "both switch positions ran the same kernel?"
    sel = list(range(B)) if B <= 5 else [0, 1, B // 2, B - 2, B - 1]
    caches = orc.initialize_cache(len(sel))
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[sel][:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(outs[k8][sel], np.concatenate(refs, 1), f"wg8 kernel vs oracle, B = {B}")
    # the GRU states it leaves (the tail of the state buffer: [KB][B * F2][C2]) against the oracle's caches
    hs = states[k8][2 * B * (cfg.n_fft - cfg.hop_size):].reshape(len(caches) - 2, B, -1)
    for kb, c in enumerate(caches[2:]):
        _assert_close(hs[kb][sel].reshape(-1), np.asarray(c).reshape(-1), f"h[{kb}]")


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_l", "fe48_b_h480"])
def test_export_onnx_composition_runs_line_for_line(name):
    """scripts/export_onnx.py:55-57 on the mirror: spec, cache = model.stft(wav_in, cache); spec, *c = model(spec, *c);
    wav_out, cache = model.stft.inverse(spec, cache) - against the reference's own per-hop outputs (goldens)."""
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to(_dev())
    cache_stft, cache_istft = m.stft.initialize_cache(x)
    cache_model = m.initialize_cache(x)
    outs = []
    for t in range(hops):
        wav_in = x[:, t * H:(t + 1) * H]
        keep = cache_stft.clone()
        spec_in, cache_stft_new = m.stft(wav_in, cache_stft)
        assert torch.equal(cache_stft, keep), "stft.forward must not modify its input cache"
        cache_stft = cache_stft_new
        spec_out, *cache_model = m(spec_in, *cache_model)
        wav_out, cache_istft = m.stft.inverse(spec_out, cache_istft)
        outs.append(wav_out.cpu().numpy())
    _assert_close(spec_in.cpu().numpy(), g["stream_spec_in_last"], "stft.forward spec (last hop)")
    _assert_close(spec_out.cpu().numpy(), g["stream_spec_out_last"], "model spec_out (last hop)")
    _assert_close(np.stack(outs, 0), g["stream_wav_out"], "composed wav_out")
    _assert_close(cache_stft.cpu().numpy(), g["stream_cache_stft"], "cache_stft")
    _assert_close(cache_istft.cpu().numpy(), g["stream_cache_istft"], "cache_istft")


@pytest.mark.parametrize("name", ["fe_b", "fe_m", "fe48_b"])
def test_standalone_stft_modules_match_oracle(name):
    """ONNXSTFT.forward / .inverse incl. a non-zero Nyquist bin, and CompressedSTFT.forward / .inverse (Model.stft)."""
    m, orc, cfg, sr, seed = _model(name)
    mo, *_ = _model(name, "Model")
    B, H, N = 3, cfg.hop_size, cfg.n_fft
    x = make_input(B, 5 * H + 19, 808, sr)
    cache = (0.1 * np.random.default_rng(3).standard_normal((B, N - H))).astype(np.float32)
    spec_ref, cache_ref = orc.stft_step(x[:, :H], cache)
    spec, cache_new = m.stft(torch.from_numpy(x[:, :H]).to(_dev()), torch.from_numpy(cache).to(_dev()))
    _assert_close(spec.cpu().numpy(), spec_ref, "stft.forward")
    assert np.array_equal(cache_new.cpu().numpy(), cache_ref)
    sp = (0.5 * np.random.default_rng(4).standard_normal((B, N // 2 + 1, 1, 2))).astype(np.float32)     # Im X[0], X[N/2] != 0
    wav_ref, c2_ref = orc.istft_step(sp.copy(), cache)
    wav, c2 = m.stft.inverse(torch.from_numpy(sp).to(_dev()), torch.from_numpy(cache).to(_dev()))
    _assert_close(wav.cpu().numpy(), wav_ref, "stft.inverse wav")
    _assert_close(c2.cpu().numpy(), c2_ref, "stft.inverse cache")
    # offline: spec = Model.stft(noisy) is what the oracle's offline path feeds its model_forward; inverse(stft(x)) == x
    xd = torch.from_numpy(x).to(_dev())
    cs = mo.stft(xd)
    T = 1 + x.shape[1] // H
    assert tuple(cs.shape) == (B, N // 2, T, 2)
    xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
    fr = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * orc.window
    X = np.fft.rfft(fr, axis=2)[:, :, :-1].transpose(0, 2, 1)
    mag = np.maximum(np.abs(X), 1e-5)
    Xc = X * mag ** (cfg.input_compression - 1.0)
    _assert_close(cs.cpu().numpy(), np.stack([Xc.real, Xc.imag], -1), "CompressedSTFT.forward")
    back = mo.stft.inverse(torch.view_as_complex(cs.contiguous()))
    ref = x[:, :H * (T - 1)].copy()
    # (the dropped Nyquist bin makes the round trip approximate; noise input keeps little energy there)
    Xf = np.fft.rfft(fr, axis=2)
    Xf[:, :, -1] = 0
    fr2 = np.fft.irfft(Xf, n=N, axis=2) * orc.window
    full = np.zeros((B, (T - 1) * H + N))
    env = np.zeros((T - 1) * H + N)
    for t in range(T):
        full[:, t * H:t * H + N] += fr2[:, t]
        env[t * H:t * H + N] += orc.window.astype(np.float64) ** 2
    sl = slice(N // 2, N // 2 + H * (T - 1))
    _assert_close(back.cpu().numpy(), full[:, sl] / env[sl], "CompressedSTFT.inverse")
    assert ref.shape == tuple(back.shape)


def test_streaming_model_steps_without_repacking_and_stays_functional():
    """the driver-loop form `wav_out, *caches = M(wav_in, *caches)` must not re-pack the caches, must not modify the
    tensors it was given, and must equal the packed path bit for bit"""
    from fastenhancer_amd.streaming import StreamingModel
    m, orc, cfg, sr, seed = _model("fe_b")
    M1, M2 = StreamingModel(m), StreamingModel(m)
    B, hops, H = 3, 5, cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, 12, sr)).to(_dev())
    c1 = M1.initialize_cache(x)
    c2 = [t.clone() for t in M2.initialize_cache(x)]            # foreign tensors: the packing path
    n_state = m.engine.state_floats(B)
    for t in range(hops):
        before = [t_.clone() for t_ in c1]
        assert M1._source(c1, B, n_state) is not None, "the caches handed out are not recognised as views of a state buffer"
        o1, *n1 = M1(x[:, t * H:(t + 1) * H], *c1)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(c1, before)), "input caches were modified"
        assert M2._source(c2, B, n_state) is None
        o2, *n2 = M2(x[:, t * H:(t + 1) * H], *c2)
        assert torch.equal(o1, o2) and all(torch.equal(a_, b_) for a_, b_ in zip(n1, n2))
        c1, c2 = n1, [t_.clone() for t_ in n2]


@pytest.mark.parametrize("name", ["fe_b", "fe_tk_b", "fe_dpt_t", "bsrnn_xt"])
def test_streaming_model_has_the_reference_value_semantics(name):
    """scripts/export_onnx.py:48-58: caches are passed in and returned, never mutated in place, and the caller owns them.  The caches
    returned by call n stay what they were through five more calls and are still valid input: resuming from them reproduces, bit for bit,
    what the uninterrupted run produced."""
    from fastenhancer_amd.streaming import StreamingModel
    m, orc, cfg, sr, seed = _bsrnn(name) if name.startswith("bsrnn") else _model(name)
    M = StreamingModel(m)
    B, hops, H, keep = 3, 9, cfg.hop_size, 2
    x = torch.from_numpy(make_input(B, hops * H, 31, sr)).to(_dev())
    caches = M.initialize_cache(x)
    init = caches
    outs, kept, kept_snap = [], None, None
    for t in range(hops):
        o, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
        if t == keep:
            kept, kept_snap = caches, [c.clone() for c in caches]
    assert all(float(c.abs().max()) == 0.0 for c in init), "the initial caches were written"
    assert all(torch.equal(a_, b_) for a_, b_ in zip(kept, kept_snap)), "caches returned by call n changed during the later calls"
    caches = kept                       # roll back to after hop `keep` and run on
    for t in range(keep + 1, hops):
        o, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        assert torch.equal(o, outs[t]), f"resumed run differs at hop {t}"
    assert all(torch.equal(a_, b_) for a_, b_ in zip(kept, kept_snap))


def test_dptransformer_caches_edited_in_place_are_honoured():
    """the dptransformer mirror hands out rotated COPIES of its K / V rings and recognises them by identity on the way back; a
    caller who edits such a tensor in place (zeroing one stream's cache - what resetting a stream between utterances looks like with the
    reference's Model.forward, scripts/export_onnx.py:48-58) must get the edit, not the untouched internal buffer.  The edited tensors give the
    bits that clones with the same edit give."""
    from fastenhancer_amd.streaming import StreamingModel
    m, orc, cfg, sr, seed = _model("fe_dpt_t")
    M = StreamingModel(m)
    B, H = 3, cfg.hop_size
    x = torch.from_numpy(make_input(B, 4 * H, 57, sr)).to(_dev())
    caches = M.initialize_cache(x)
    for t in range(3):
        o, *caches = M(x[:, t * H:(t + 1) * H], *caches)
    clones = [c.clone() for c in caches]
    o_plain, *_ = M(x[:, 3 * H:4 * H], *caches)               # unmodified: the fast path (identity + version counters)
    for c in (caches, clones):                                # reset stream 1 in every model cache, in place
        for tns in c[2:]:
            n = tns.shape[0] // B                                # [B * F2, NH, L, hd]: stream b = rows b * F2 .. (b + 1) * F2 - 1
            tns[n:2 * n].zero_()
    o_edit, *_ = M(x[:, 3 * H:4 * H], *caches)
    o_clone, *_ = M(x[:, 3 * H:4 * H], *clones)
    assert torch.equal(o_edit, o_clone), "caches edited in place were replaced by the internal state"
    assert not torch.equal(o_edit[1], o_plain[1]) and torch.equal(o_edit[0], o_plain[0]) and torch.equal(o_edit[2], o_plain[2])


def test_full_size_batch_256_matches_oracle_and_is_stream_independent():
    """BASELINE config 2: FastEnhancer_B, 256 concurrent streams."""
    m, orc, cfg, sr, seed = _model("fe_b")
    eng = m.engine
    B, hops, H = 256, 6, cfg.hop_size
    x = make_input(B, hops * H, 777, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    out = eng.step(xd, state, T=hops)
    ref_caches = orc.initialize_cache(B)
    refs = []
    for t in range(hops):
        o, *ref_caches = orc.step(x[:, t * H:(t + 1) * H], *ref_caches)
        refs.append(o)
    _assert_close(out.cpu().numpy(), np.concatenate(refs, axis=1), "B=256 wav_out")
    for a, b in zip(eng.split_state(state, B), ref_caches):
        _assert_close(a.cpu().numpy(), b, "B=256 cache")
    # streams never interact: any stream run alone gives the identical bits
    for b in (0, 17, 255):
        st1 = eng.new_state(1)
        o1 = eng.step(xd[b:b + 1].contiguous(), st1, T=hops)
        assert torch.equal(o1[0], out[b])
    # determinism
    state2 = eng.new_state(B)
    out2 = eng.step(xd, state2, T=hops)
    assert torch.equal(out, out2) and torch.equal(state, state2)


def _full_size_check(m, orc, cfg, sr, B, hops, sample, what):
    """B streams x hops: oracle parity on `sample`, and bitwise stream independence on ALL streams - the batch is run
    a second time with the streams in reversed order (every stream then sits in a different workgroup / state slot /
    scratch slot), which must give the same bits."""
    eng = m.engine
    H = cfg.hop_size
    x = make_input(B, hops * H, 4711, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    outs = [eng.step(xd[:, t * H:(t + 1) * H], state, T=1).clone() for t in range(hops)]      # the per-hop kernel
    out = torch.cat(outs, dim=1)
    xr = xd.flip(0).contiguous()
    state_r = eng.new_state(B)
    out_r = torch.cat([eng.step(xr[:, t * H:(t + 1) * H], state_r, T=1).clone() for t in range(hops)], dim=1)
    assert torch.equal(out_r.flip(0), out), f"{what}: a stream's output depends on its position in the batch"
    for a_, b_ in zip(eng.split_state(state, B), eng.split_state(state_r, B)):      # (every cache tensor is stream-major)
        assert torch.equal(b_.reshape(B, -1).flip(0), a_.reshape(B, -1)), f"{what}: state depends on the position in the batch"
    caches = orc.initialize_cache(len(sample))
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[sample, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(out[sample].cpu().numpy(), np.concatenate(refs, axis=1), f"{what} wav_out")
    for a_, b_ in zip(eng.split_state(state, B), caches):
        got = a_.reshape(B, -1)[sample].reshape(b_.shape)
        _assert_close(got.cpu().numpy(), b_, f"{what} cache")


def test_full_size_fastenhancer_l_256_streams():
    """BASELINE config 3's per-GPU share: FastEnhancer_L, 256 streams (global skip scratch indexed per stream)."""
    m, orc, cfg, sr, seed = _model("fe_l")
    _full_size_check(m, orc, cfg, sr, 256, 3, [0, 1, 17, 100, 128, 200, 254, 255], "fe_l B=256")


def test_full_size_48khz_hop480_512_streams():
    """BASELINE config 4: 48 kHz FastEnhancer_B at a 10 ms hop, 512 streams (two rounds of workgroups)."""
    m, orc, cfg, sr, seed = _model("fe48_b_h480")
    _full_size_check(m, orc, cfg, sr, 512, 4, [0, 1, 255, 256, 257, 300, 510, 511], "fe48_b_h480 B=512")


def test_dptransformer_cacheless_chunk_longer_than_the_lookbehind():
    """`ONNXModel(spec)` without caches on 40 frames (> lookbehind 31): frames before the start are masked, not attended to as
    zeros (dptransformer/model.py:216-218); then the returned caches continue the stream like the oracle's, and a second
    chunk WITH those caches equals the oracle's continuation."""
    m, orc, cfg, sr, seed = _model("fe_dpt_t")
    B, H, T = 2, cfg.hop_size, 40
    x = make_input(B, (T + 6) * H, 99, sr)
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(T + 6):
        s_, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s_)
    spec = np.concatenate(specs, axis=2)
    y_ref, h_ref = orc.spec_forward(spec[:, :, :T], None)
    spec_d = torch.from_numpy(spec).to(_dev())
    y, *h = m(spec_d[:, :, :T].contiguous())
    _assert_close(y.cpu().numpy(), y_ref, "cache-less chunk of 40 frames")
    zero = orc.spec_forward(spec[:, :, :T], orc.initialize_cache(B)[2:])[0]
    assert rms(zero - y_ref) > 1e-3 * rms(y_ref), "masked and zero-cache starts must differ (else this test checks nothing)"
    for a_, b_ in zip(h, h_ref):
        _assert_close(a_.cpu().numpy(), b_, "caches after the cache-less chunk")
    y2_ref, _ = orc.spec_forward(spec[:, :, T:], h_ref)
    y2, *_ = m(spec_d[:, :, T:].contiguous(), *h)
    _assert_close(y2.cpu().numpy(), y2_ref, "continuation with the returned caches")


@pytest.mark.parametrize("name,B,T", [("fe_dpt_t", 2, 40), ("fe_dpt_b", 3, 70), ("fe_dpt_t", 5, 33)])
def test_dptransformer_spec_chunks_run_on_the_time_pipeline(name, B, T):
    """fe_spec_step with T >> 1 for the dptransformer variant (r4v): the frames of a chunk run on co-resident workgroups and hand their
    k / v on through rings of lookbehind + P slots in the handle; the caller's caches go in before the launch (whatever their ring head)
    and come back in the reference's order.  Checked: (1) a cache-less chunk, (2) a chunk WITH caches whose rings have a non-zero head
    (left by per-hop launches), (3) a third chunk that continues from the returned caches - each against the oracle AND against the
    serial walk of the same call (fe_set_time_pipeline(0)), outputs and caches."""
    m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    H = cfg.hop_size
    hops1 = 5                                   # per-hop launches first: the rings' heads advance to 5
    n_frames = hops1 + 3 * T
    x = make_input(B, n_frames * H, 4242, sr)
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(n_frames):
        s_, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s_)
    spec = np.concatenate(specs, axis=2)
    spec_d = torch.from_numpy(spec).to(_dev())

    def run(width):
        eng.set_time_pipeline(width)
        outs = []
        y, *h = m(spec_d[:, :, :T].contiguous())                       # (1) cache-less chunk
        outs.append((y, h))
        for t in range(T, T + hops1):                                   # per-hop launches: ring heads move
            y, *h = m(spec_d[:, :, t:t + 1].contiguous(), *h)
        y, *h = m(spec_d[:, :, T + hops1:2 * T + hops1].contiguous(), *h)            # (2) chunk with caches, head != 0
        outs.append((y, h))
        y, *h = m(spec_d[:, :, 2 * T + hops1:3 * T + hops1].contiguous(), *h)        # (3) continuation
        outs.append((y, h))
        eng.set_time_pipeline(-1)
        return outs

    piped, serial = run(-1), run(0)
    # oracle
    y_ref, h_ref = orc.spec_forward(spec[:, :, :T], None)
    refs = [(y_ref, h_ref)]
    for t in range(T, T + hops1):
        _, h_ref = orc.spec_forward(spec[:, :, t:t + 1], h_ref)
    y_ref, h_ref = orc.spec_forward(spec[:, :, T + hops1:2 * T + hops1], h_ref)
    refs.append((y_ref, h_ref))
    y_ref, h_ref = orc.spec_forward(spec[:, :, 2 * T + hops1:3 * T + hops1], h_ref)
    refs.append((y_ref, h_ref))
    for i, ((yp, hp), (ys, hs), (yr, hr)) in enumerate(zip(piped, serial, refs)):
        _assert_close(yp.cpu().numpy(), yr, f"pipelined chunk {i} vs oracle")
        _assert_close(ys.cpu().numpy(), yr, f"serial chunk {i} vs oracle")
        scale = float(np.abs(yr).max())
        assert float((yp - ys).abs().max()) <= 3e-5 * scale, f"pipelined vs serial chunk {i}"
        if (i + 1) * T + (hops1 if i else 0) >= 31:      # (before 31 frames have been seen the slots not yet filled carry the +inf marks; the reference returns fewer slots)
            for a_, b_, c_ in zip(hp, hs, hr):
                _assert_close(a_.cpu().numpy(), c_, f"caches after pipelined chunk {i}")
                _assert_close(b_.cpu().numpy(), c_, f"caches after serial chunk {i}")


@pytest.mark.parametrize("B,T", [(1, 40), (3, 17)])
def test_time_kernel_spec_chunks_run_on_the_time_pipeline(B, T):
    """fe_spec_step with T >> 1 for the time_kernel variant (r4v): the frames of a chunk run on co-resident workgroups; every causal conv's
    input goes from frame to frame through a ring of P + kt - 1 slots in the handle, filled from the caller's caches before the launch and
    read back (the chunk's last kt - 1 frames) after it; the GRU states travel as in the default model.  Three chunks with per-hop launches
    in between, against the oracle and against the serial walk of the same calls - outputs and all caches."""
    m, orc, cfg, sr, seed = _model("fe_tk_b")
    eng = m.engine
    H = cfg.hop_size
    hops1 = 3
    n_frames = hops1 + 3 * T
    x = make_input(B, n_frames * H, 777, sr)
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(n_frames):
        s_, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s_)
    spec = np.concatenate(specs, axis=2)
    spec_d = torch.from_numpy(spec).to(_dev())
    cuts = [(0, T), (T + hops1, 2 * T + hops1), (2 * T + hops1, 3 * T + hops1)]

    def run(width):
        eng.set_time_pipeline(width)
        outs = []
        y, *h = m(spec_d[:, :, :T].contiguous(), *m.initialize_cache(spec_d))
        outs.append((y, h))
        for t in range(T, T + hops1):
            y, *h = m(spec_d[:, :, t:t + 1].contiguous(), *h)
        for lo, hi in cuts[1:]:
            y, *h = m(spec_d[:, :, lo:hi].contiguous(), *h)
            outs.append((y, h))
        eng.set_time_pipeline(-1)
        return outs

    piped, serial = run(-1), run(0)
    h_ref = orc.initialize_cache(B)[2:]
    refs = []
    y_ref, h_ref = orc.spec_forward(spec[:, :, :T], h_ref)
    refs.append((y_ref, h_ref))
    for t in range(T, T + hops1):
        _, h_ref = orc.spec_forward(spec[:, :, t:t + 1], h_ref)
    for lo, hi in cuts[1:]:
        y_ref, h_ref = orc.spec_forward(spec[:, :, lo:hi], h_ref)
        refs.append((y_ref, h_ref))
    for i, ((yp, hp), (ys, hs), (yr, hr)) in enumerate(zip(piped, serial, refs)):
        _assert_close(yp.cpu().numpy(), yr, f"pipelined chunk {i} vs oracle")
        _assert_close(ys.cpu().numpy(), yr, f"serial chunk {i} vs oracle")
        scale = float(np.abs(yr).max())
        assert float((yp - ys).abs().max()) <= 3e-5 * scale, f"pipelined vs serial chunk {i}"
        for a_, b_, c_ in zip(hp, hs, hr):
            _assert_close(a_.cpu().numpy(), c_, f"caches after pipelined chunk {i}")
            _assert_close(b_.cpu().numpy(), c_, f"caches after serial chunk {i}")


@pytest.mark.parametrize("name,B,hops", [("fe_ln_b", 300, 3), ("fe_dprnn_b", 256, 3), ("fe_dprnn_l", 300, 2), ("fe_dpt_b", 256, 35), ("fe_dpt_t", 700, 4), ("fe_dpt_m", 260, 2),
                                            ("fe_dpt_t", 800, 35), ("fe_dpt_b", 1100, 34), ("fe_dprnn_b", 600, 3), ("fe_dprnn_t", 1100, 3), ("fe_dpt_b", 400, 3)])
def test_full_size_block_variants(name, B, hops):
    """the dprnn / dptransformer variants at full batch sizes (one workgroup per stream, and persistent workgroups above #CUs):
    oracle parity on sample streams, bitwise position independence on all; 35 hops of dpt_b take its K / V rings (31 slots)
    through a wrap."""
    m, orc, cfg, sr, seed = _model(name)
    _full_size_check(m, orc, cfg, sr, B, hops, [0, 1, 17, B // 2, B - 2, B - 1], f"{name} B={B}")
    # r6: the T- and B-sized dprnn / dptransformer shapes have low-LDS companions (fe_shapes.def DPTLOW ... DTBLOW); dpt_b's takes over beyond two rounds only
    want_low = {("fe_dpt_t", 700): True, ("fe_dpt_t", 800): True, ("fe_dpt_b", 1100): True, ("fe_dprnn_b", 600): True, ("fe_dprnn_t", 1100): True, ("fe_dpt_b", 400): False}.get((name, B))
    if want_low is not None:
        assert ("LOW=" in m.engine.last_step_kernel()) == want_low, m.engine.last_step_kernel()


@pytest.mark.parametrize("name,B", [("fe_b", 300), ("fe_b", 1100), ("fe_s", 520), ("fe48_t", 700), ("fe48_t", 1100), ("fe_t", 700), ("fe_t", 1100), ("fe48_b", 600), ("fe48_b_h480", 1030)])
def test_low_lds_companion_above_cus(name, B):
    """above #CUs streams fe_step switches to the shape's low-LDS companion (two workgroups per CU, weights streamed from
    L2, fe_shapes.def LOW): up to 2 x #CUs streams one workgroup per stream, beyond that persistent workgroups that walk
    several streams.  Same packed weights, same state; oracle parity and position independence as everywhere else."""
    m, orc, cfg, sr, seed = _model(name)
    _full_size_check(m, orc, cfg, sr, B, 3, [0, 1, 255, 256, 257, B // 2, B - 2, B - 1], f"{name} B={B} (low-LDS companion)")
    # r6: the T shapes' companions run three workgroups per CU; T's own plan fits twice, so its companion takes over above 2 x #CUs streams
    if name in ("fe_t", "fe48_t"):
        assert "LOW=" in m.engine.last_step_kernel(), m.engine.last_step_kernel()


def test_full_size_bsrnn_xt_256_streams():
    """BASELINE config 5: BSRNN-xt, 256 streams."""
    m, orc, cfg, sr, seed = _bsrnn("bsrnn_xt")
    _full_size_check(m, orc, cfg, sr, 256, 4, [0, 1, 17, 128, 254, 255], "bsrnn_xt B=256")


@pytest.mark.parametrize("name,B", [("bsrnn_xt", 256), ("bsrnn_xt", 5), ("bsrnn_xxt", 64)])
def test_bsrnn_role_split_part1_agrees_with_the_phase_by_phase_kernel(name, B):
    """r5: the per-hop step's PART 1 for num_channels = 16 at up to one stream per CU runs on the role-split kernel (bsrnn_ov_kernels.hip.h:
    the scans alone on two waves, the layers' matrix-core chains on the other two, LDS counters between them);
    fe_set_step_kernel(FE_STEP_KERNEL_WAVES4) selects the phase-by-phase kernel.  Both against the oracle on sampled streams, and against
    each other on every stream and every cache - twelve hops, so that the time-LSTM state both kernels write back is fed forward."""
    m, orc, cfg, sr, seed = _bsrnn(name)
    eng = m.engine
    hops, H = 12, cfg.hop_size
    x = make_input(B, hops * H, seed + 77, sr)
    xd = torch.from_numpy(x).to(_dev())
    res = {}
    for kern in ("wg8", "waves4"):
        eng.set_step_kernel(kern)
        state = eng.new_state(B)
        outs = [eng.step(xd[:, t * H:(t + 1) * H].contiguous(), state, T=1).cpu().numpy() for t in range(hops)]
        res[kern] = (np.concatenate(outs, 1), [c.cpu().numpy() for c in eng.split_state(state, B)])
    eng.set_step_kernel("wg8")
    # r6: the same step as ONE cooperative launch (fe_set_option("bsrnn_fused_step", 1): barriers over the workgroups of each sixteen-stream
    # tile instead of two kernel boundaries; measured slower and off by default) - bit for bit the three launches, short last tile included
    eng.set_option("bsrnn_fused_step", 1)
    state = eng.new_state(B)
    outs = [eng.step(xd[:, t * H:(t + 1) * H].contiguous(), state, T=1).cpu().numpy() for t in range(hops)]
    assert eng.last_step_kernel().startswith("bsrnn_ov_kernel<fused step>"), eng.last_step_kernel()
    eng.set_option("bsrnn_fused_step", 0)
    assert np.array_equal(np.concatenate(outs, 1), res["wg8"][0]), "fused step differs from the three launches"
    for a_, b_ in zip([c.cpu().numpy() for c in eng.split_state(state, B)], res["wg8"][1]):
        assert np.array_equal(a_, b_)
    sel = sorted(set([0, 1, B // 2, B - 1]))
    caches = orc.initialize_cache(len(sel))
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[sel][:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    ref = np.concatenate(refs, 1)
    for kern in ("wg8", "waves4"):
        _assert_close(res[kern][0][sel], ref, f"{name} B={B} {kern} wav_out")
        for got, want in zip(res[kern][1], caches):
            _assert_close(got[sel] if got.shape[0] == B else got.reshape(B, -1)[sel].reshape(want.shape), want, f"{name} B={B} {kern} cache")
    scale = float(np.sqrt(np.mean(ref ** 2)))
    d = float(np.sqrt(np.mean((res["wg8"][0] - res["waves4"][0]) ** 2))) / scale
    assert d < 3e-6, d
    for a_, b_ in zip(res["wg8"][1], res["waves4"][1]):
        sc = float(np.sqrt(np.mean(b_ ** 2))) + 1e-12
        assert float(np.sqrt(np.mean((a_ - b_) ** 2))) / sc < 3e-6


def test_win_size_smaller_than_n_fft():
    """ONNXSTFT pads a shorter window to n_fft (functional/audio_modules.py:213-217); no shipped yaml uses it."""
    from oracle.fe_oracle import FEConfig as OCfg, FEOracle, fold_state_dict
    from oracle.weightgen import make_training_state_dict
    kw = dict(MODEL_KWARGS["fe_b"][0])
    kw["win_size"] = 400
    ocfg = OCfg.from_model_kwargs(kw)
    sd = make_training_state_dict(ocfg, 321)
    orc = FEOracle(ocfg, fold_state_dict(sd, ocfg), np.float32)
    mod = importlib.import_module("fastenhancer_amd.models.fastenhancer.default.model")
    m = mod.ONNXModel(**kw).to(_dev()).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    B, hops, H = 3, 6, ocfg.hop_size
    x = make_input(B, hops * H, 99, 16000)
    state = m.engine.new_state(B)
    out = m.engine.step(torch.from_numpy(x).to(_dev()), state, T=hops)
    caches = orc.initialize_cache(B)
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(out.cpu().numpy(), np.concatenate(refs, 1), "win_size=400 wav_out")
    mo = mod.Model(**kw).to(_dev()).eval()
    mo.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    xo = make_input(2, 7 * H + 5, 98, 16000)
    wav_ref, spec_ref = orc.offline_forward(xo)
    wav_hat, spec_hat = mo(torch.from_numpy(xo).to(_dev()))
    _assert_close(wav_hat.cpu().numpy(), wav_ref, "win_size=400 offline wav")


def test_more_streams_than_cus_and_strided_buffers():
    """600 streams (more workgroups than the 256 CUs, BASELINE config 4 uses 512) through row-strided in / out buffers."""
    m, orc, cfg, sr, seed = _model("fe_t")
    eng = m.engine
    B, hops, H = 600, 3, cfg.hop_size
    x = make_input(B, hops * H, 4242, sr)
    big_in = torch.zeros(B, hops * H + 64, device=_dev())
    big_in[:, :hops * H] = torch.from_numpy(x).to(_dev())
    big_out = torch.full((B, hops * H + 32), 7.0, device=_dev())
    state = eng.new_state(B)
    for t in range(hops):       # per-hop launches on views: in stride T*H+64, out stride T*H+32
        eng.step(big_in[:, t * H:(t + 1) * H], state, wav_out=big_out[:, t * H:(t + 1) * H], T=1)
    assert float(big_out[:, hops * H:].min()) == 7.0            # nothing written past the rows
    sel = [0, 1, 255, 256, 257, 511, 599]
    caches = orc.initialize_cache(len(sel))
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[sel, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(big_out[sel, :hops * H].cpu().numpy(), np.concatenate(refs, axis=1), "600 streams, strided")


def test_edge_inputs():
    m, orc, cfg, sr, seed = _model("fe_b")
    eng = m.engine
    H = cfg.hop_size
    # silence in -> silence out (|X|=0 compresses to 0 and the mask multiplies it)
    z = torch.zeros(2, 4 * H, device=_dev())
    st = eng.new_state(2)
    out = eng.step(z, st, T=4)
    assert float(out.abs().max()) == 0.0
    # full-scale square wave (clipped input range of the driver loop)
    sq = torch.ones(1, 8 * H, device=_dev())
    sq[:, ::3] = -1.0
    st = eng.new_state(1)
    out = eng.step(sq, st, T=8)
    caches = orc.initialize_cache(1)
    refs = []
    for t in range(8):
        o, *caches = orc.step(sq.cpu().numpy()[:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    # A +-1 wave of period 3 is exactly periodic: most bins of its spectrum are rounding noise (~1e-5) around an exact zero, and the
    # compression gain max(|X|, 1e-5)^(c - 1) is ~3000 there - the reference's own output moves in the fifth digit with the FFT's summation
    # order (observed 8.1e-5 between the matrix-core DFT and numpy's FFT).  This input keeps the north_star bound only.
    _assert_close(out.cpu().numpy(), np.concatenate(refs, 1), "full-scale input", tight=REL_TOL)


@pytest.mark.parametrize("kern", ["wg8", "waves4"])
def test_bsrnn_edge_inputs(kern):
    """Silence (BSRNN's mask has a residual branch: the output is small, not zero) and a burst forty times full scale that drives the LSTM gates into
    saturation (exp2 overflows to inf, rcp(inf) = 0: no NaN) - both PART 1 kernels of the per-hop step against the oracle, finite everywhere."""
    m, orc, cfg, sr, seed = _bsrnn("bsrnn_xt")
    eng = m.engine
    eng.set_step_kernel(kern)
    B, hops, H = 3, 6, cfg.hop_size
    x = make_input(B, hops * H, seed + 5, sr)
    x[0] = 0.0
    x[1, 2 * H:3 * H] *= 40.0
    xd = torch.from_numpy(x).to(_dev())
    st = eng.new_state(B)
    outs = [eng.step(xd[:, t * H:(t + 1) * H].contiguous(), st, T=1).cpu().numpy() for t in range(hops)]
    eng.set_step_kernel("wg8")
    caches = orc.initialize_cache(B)
    refs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    got, ref = np.concatenate(outs, 1), np.concatenate(refs, 1)
    assert np.isfinite(got).all()
    _assert_close(got[1:], ref[1:], f"bsrnn_xt edge inputs ({kern})")
    assert float(np.abs(got[0] - ref[0]).max()) < 1e-6                      # (the silent stream: an absolute bound, its scale is ~0)
    for a_, b_ in zip(eng.split_state(st, B), caches):
        assert np.isfinite(a_.cpu().numpy()).all()
        _assert_close(a_.cpu().numpy()[B - 1 if a_.shape[0] == B else slice((B - 1) * 31, B * 31)], np.asarray(b_)[B - 1 if a_.shape[0] == B else slice((B - 1) * 31, B * 31)], f"cache ({kern})")


def test_error_behaviour():
    from fastenhancer_amd import _lib
    from fastenhancer_amd.config import FEConfig
    from fastenhancer_amd.engine import Engine
    kw, sr, seed = MODEL_KWARGS["fe_b"]
    eng = Engine(FEConfig.from_model_kwargs(**kw), _dev())
    x = torch.zeros(1, 256, device=_dev())
    st = eng.new_state(1)
    with pytest.raises(_lib.FEError, match="fe_load_weights"):
        eng.step(x, st)
    m, *_ = _model("fe_b")
    with pytest.raises(AssertionError):
        m.engine.step(torch.zeros(1, 100, device=_dev()), st)      # not T*H samples


def test_native_library_is_what_ran():
    """The GPU path must be the in-tree HIP library (no eager/CPU fallback exists)."""
    import ctypes
    from fastenhancer_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL)
    with open("/proc/self/maps") as f:
        assert "libfastenhancer_hip.so" in f.read()


@pytest.mark.parametrize("name", GPU_SHAPES)
def test_offline_model_forward_matches_reference_golden(name):
    """a21 / a27: Model.forward(noisy) -> (wav_hat, spec_hat) vs the reference's own output."""
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name, "Model")
    B, H = int(g["B"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, int(g["hops"]) * H + 37, seed + 2000, sr)).to(_dev())
    wav_hat, spec_hat = m(x)
    assert tuple(wav_hat.shape) == g["offline_wav"].shape and tuple(spec_hat.shape) == g["offline_spec"].shape
    _assert_close(wav_hat.cpu().numpy(), g["offline_wav"], "offline wav")
    _assert_close(spec_hat.cpu().numpy(), g["offline_spec"], "offline spec")
    wav3, _ = m(x.unsqueeze(1))                       # [B, 1, Tw] input form
    assert torch.equal(wav3, wav_hat)


@pytest.mark.parametrize("name", ["fe_dpt_t", "fe_dpt_b"])
def test_dptransformer_time_pipelined_offline_agrees_with_the_serial_walk(name):
    """The dptransformer variant's offline Model.forward over co-resident workgroups: the K / V caches become per-frame rings of
    L + pipe slots in the work buffer, a frame publishes its k / v of a block before it attends.  Utterances longer than the lookbehind
    (the window slides over ring wrap-arounds), several widths, twice each; against the serial walk and the oracle."""
    m, orc, cfg, sr, seed = _model(name, "Model")
    eng = m.engine
    x = make_input(2, 83 * cfg.hop_size + 9, 778, sr)
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in m(xd)]
    for width in (16, 5, -1):
        eng.set_time_pipeline(width)
        for rep in range(2):
            w, s_ = m(xd)
            assert float((w - w_ser).abs().max()) <= 2e-5 * max(1.0, float(w_ser.abs().max())), (width, rep, float((w - w_ser).abs().max()))
            assert float((s_ - s_ser).abs().max()) <= 2e-5 * max(1.0, float(s_ser.abs().max())), (width, rep)
    wav_ref, spec_ref = orc.offline_forward(x)
    _assert_close(w.cpu().numpy(), wav_ref, "dptransformer pipelined offline wav vs oracle")
    _assert_close(s_.cpu().numpy(), spec_ref, "dptransformer pipelined offline spec vs oracle")
    eng.set_time_pipeline(-1)


def test_time_kernel_time_pipelined_offline_agrees_with_the_serial_walk():
    """The time_kernel variant's offline Model.forward with the frames of an utterance over co-resident workgroups: besides the GRU
    states, every causal time conv hands its INPUT of frame t to frames t + 1, t + 2 through a ring of slots and a counter per
    (conv, stream).  Against the serial walk and the oracle, at several widths (the ring has pipe + 2 slots), twice each."""
    m, orc, cfg, sr, seed = _model("fe_tk_b", "Model")
    eng = m.engine
    x = make_input(2, 70 * cfg.hop_size + 3, 777, sr)
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in m(xd)]
    for width in (16, 3, -1):
        eng.set_time_pipeline(width)
        for rep in range(2):
            w, s_ = m(xd)
            assert float((w - w_ser).abs().max()) <= 2e-5 * max(1.0, float(w_ser.abs().max())), (width, rep, float((w - w_ser).abs().max()))
            assert float((s_ - s_ser).abs().max()) <= 2e-5 * max(1.0, float(s_ser.abs().max())), (width, rep)
    wav_ref, spec_ref = orc.offline_forward(x)
    _assert_close(w.cpu().numpy(), wav_ref, "time_kernel pipelined offline wav vs oracle")
    _assert_close(s_.cpu().numpy(), spec_ref, "time_kernel pipelined offline spec vs oracle")
    eng.set_time_pipeline(-1)


@pytest.mark.parametrize("name", ["fe_t", "fe_b", "fe_m", "fe48_b", "fe_dprnn_b", "fe_ln_b"])
def test_time_pipelined_offline_and_spec_agree_with_the_serial_walk(name):
    """fe_offline / fe_spec_step with T >= 4 spread a stream's frames over co-resident workgroups that hand the GRU state
    from frame to frame (fe_set_time_pipeline); one workgroup walking the frames serially must give the same result, and
    both must match the oracle on a long utterance (many state hand-offs, every workgroup several frames)."""
    m, orc, cfg, sr, seed = _model(name, "Model")
    eng = m.engine
    if name in TB_SHAPES:
        eng.set_offline_engine("frame_walk")          # (the default offline engine of these shapes is the time-batched one)
    H = cfg.hop_size
    x = make_input(2, 150 * H + 29, 606, sr)
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(16)
    wav_p, spec_p = m(xd)
    wav_p2, spec_p2 = m(xd)
    assert torch.equal(wav_p, wav_p2) and torch.equal(spec_p, spec_p2), "pipelined launch is not deterministic"
    eng.set_time_pipeline(3)
    wav_p3, spec_p3 = m(xd)
    eng.set_time_pipeline(0)
    wav_s, spec_s = m(xd)
    eng.set_time_pipeline(-1)
    wav_ref, spec_ref = orc.offline_forward(x)
    for got_w, got_s, what in ((wav_p, spec_p, "pipelined x16"), (wav_p3, spec_p3, "pipelined x3"), (wav_s, spec_s, "serial")):
        _assert_close(got_w.cpu().numpy(), wav_ref, f"{what} offline wav")
        _assert_close(got_s.cpu().numpy(), spec_ref, f"{what} offline spec")
    assert float((wav_p - wav_s).abs().max()) <= 2e-5 * max(1.0, float(wav_s.abs().max()))
    # spec -> spec with carried state: two chunks of 20 frames, pipelined vs serial, caches included
    mo, *_ = _model(name)
    cache = orc.initialize_cache(2)[0]
    specs = []
    for t in range(40):
        s_, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s_)
    spec = torch.from_numpy(np.concatenate(specs, axis=2)).to(_dev())
    outs = {}
    for width in (16, 0):
        mo.engine.set_time_pipeline(width)
        h = mo.initialize_cache(spec)
        o1, *h = mo(spec[:, :, :20].contiguous(), *h)
        o2, *h = mo(spec[:, :, 20:].contiguous(), *h)
        outs[width] = (torch.cat([o1, o2], dim=2), h)
    mo.engine.set_time_pipeline(-1)
    ref, href = orc.spec_forward(np.concatenate(specs, axis=2), orc.initialize_cache(2)[2:])
    for width in (16, 0):
        _assert_close(outs[width][0].cpu().numpy(), ref, f"spec chunks, pipeline {width}")
        for a_, b_ in zip(outs[width][1], href):
            _assert_close(a_.cpu().numpy(), b_, f"spec chunk caches, pipeline {width}")


@pytest.mark.parametrize("name", TB_SHAPES)
def test_time_batched_offline_matches_oracle_and_the_frame_walk(name):
    """fe_offline's two engines on the default model: the time-batched (layer-by-layer) one - encoder pass over all frames, per
    block a scan over time + an attention pass, decoder pass - and the per-hop kernel walking the frames.  Both against the oracle
    on utterances long enough for many scan steps, with a frame count (B * T = 3 * 62) that leaves a partial tile of frames."""
    m, orc, cfg, sr, seed = _model(name, "Model")
    eng = m.engine
    H = cfg.hop_size
    big = cfg.channels >= 96
    B, T1 = (2, 23) if big else (3, 61)
    x = make_input(B, T1 * H + 17, 707, sr)
    xd = torch.from_numpy(x).to(_dev())
    wav_ref, spec_ref = orc.offline_forward(x)
    eng.set_offline_engine("time_batched")
    wav_tb, spec_tb = m(xd)
    wav_tb2, spec_tb2 = m(xd)
    assert torch.equal(wav_tb, wav_tb2) and torch.equal(spec_tb, spec_tb2), "time-batched launch is not deterministic"
    eng.set_offline_engine("frame_walk")
    wav_fw, spec_fw = m(xd)
    eng.set_offline_engine("auto")
    _assert_close(wav_tb.cpu().numpy(), wav_ref, "time-batched offline wav")
    _assert_close(spec_tb.cpu().numpy(), spec_ref, "time-batched offline spec")
    _assert_close(wav_fw.cpu().numpy(), wav_ref, "frame-walk offline wav")
    assert float((wav_tb - wav_fw).abs().max()) <= 3e-5 * max(1.0, float(wav_fw.abs().max()))
    # one utterance alone == the same utterance inside the batch (frames of different utterances share tiles and scan workgroups)
    w1, s1 = m(xd[1:2].contiguous())
    assert float((w1 - wav_tb[1:2]).abs().max()) == 0.0 and float((s1 - spec_tb[1:2]).abs().max()) == 0.0


@pytest.mark.parametrize("name,T", [("fe_b", 1), ("fe_b", 3), ("bsrnn_xt", 1)])
def test_step_with_host_buffers_matches_device_stepping(name, T):
    """fe_step_host (audio in host memory, the copies of neighbouring hop blocks under each kernel) against fe_step on device
    buffers: the same kernel on the same data - bit for bit, pinned or pageable host memory, strided rows; and enhance_stream on a
    CPU tensor against enhance_stream on the device."""
    from fastenhancer_amd.streaming import enhance_stream
    m, orc, cfg, sr, seed = (_bsrnn(name) if name.startswith("bsrnn") else _model(name))
    eng = m.engine
    B, H, n = 5, cfg.hop_size, 7
    x = torch.from_numpy(make_input(B, n * T * H + 40, 515, sr))
    state_d, state_h, state_p = eng.new_state(B), eng.new_state(B), eng.new_state(B)
    xd = x.to(_dev())
    ref = torch.cat([eng.step(xd[:, c * T * H:(c + 1) * T * H], state_d, T=T).clone() for c in range(n)], dim=1)
    xh = x.pin_memory()
    yh = eng.step_host(xh[:, :n * T * H], state_h, T=T)           # (a strided view: row stride n*T*H + 40)
    yp = torch.empty(B, n * T * H)
    eng.step_host(x[:, :n * T * H], state_p, yp, T=T)               # pageable memory on both sides
    torch.cuda.synchronize()
    assert torch.equal(yh, ref.cpu()) and torch.equal(yp, ref.cpu())
    assert torch.equal(state_h, state_d) and torch.equal(state_p, state_d)
    if not name.startswith("bsrnn"):
        w = torch.from_numpy(make_input(2, 9 * H + 11, 516, sr))
        a = enhance_stream(m, w, frames_per_call=T)
        b = enhance_stream(m, w.to(_dev()), frames_per_call=T)
        assert not a.is_cuda and torch.equal(a, b.cpu())


@pytest.mark.parametrize("name", ["fe_t", "fe_b"])
def test_spec_chunks_on_the_time_batched_engine_carry_the_caches(name):
    """fe_spec_step (ONNXModel.forward(spec, *caches), model.py:677-710) on the time-batched engine: two consecutive chunks with the GRU
    caches carried from one to the next, against the frame walk (same kernels as the per-hop step) and against the oracle; the scan
    starts from the caller's state and leaves the new one in its place."""
    mo, orc, cfg, sr, seed = _model(name)
    eng = mo.engine
    B, H, T1, T2 = 3, cfg.hop_size, 21, 34
    x = make_input(B, (T1 + T2) * H, 4242, sr)
    cache = orc.initialize_cache(B)[0]
    specs = []
    for t in range(T1 + T2):
        s_, cache = orc.stft_step(x[:, t * H:(t + 1) * H], cache)
        specs.append(s_)
    spec = np.concatenate(specs, axis=2).astype(np.float32)             # [B, F, T, 2]
    sd = torch.from_numpy(spec).to(_dev())
    outs = {}
    for engine in ("frame_walk", "time_batched"):
        eng.set_offline_engine(engine)
        caches = mo.initialize_cache(sd[:, :, :1])
        y1, *caches = mo(sd[:, :, :T1].contiguous(), *caches)
        y2, *caches = mo(sd[:, :, T1:].contiguous(), *caches)
        outs[engine] = (torch.cat([y1, y2], dim=2), [c.clone() for c in caches])
    eng.set_offline_engine("auto")
    (ya, ca), (yb, cb) = outs["frame_walk"], outs["time_batched"]
    assert float((ya - yb).abs().max()) <= 3e-5 * max(1.0, float(ya.abs().max())), float((ya - yb).abs().max())
    for p_, q_ in zip(ca, cb):
        assert float((p_ - q_).abs().max()) <= 3e-5 * max(1.0, float(p_.abs().max()))
    y_ref, co = orc.spec_forward(spec, orc.initialize_cache(B)[2:])
    _assert_close(yb.cpu().numpy(), y_ref, "time-batched spec chunks vs oracle")
    for got, want in zip(cb, co):
        _assert_close(got.cpu().numpy(), want, "GRU cache after the chunks")


def test_time_batched_optional_schedules_agree_with_the_default():
    """The time-batched engine's opt-in schedules (measured slower, DESIGN 3c - but they must stay correct): the call cut into nodes
    (time chunks / utterance groups over the handle's streams, FE_TB_NC / FE_TB_G / FE_TB_STREAMS) and the fused block stage (scan and
    tile pass of a block in one cooperative launch behind progress counters, FE_TB_FUSE=1), on a batch large enough for the 16-row
    scan (the fused stage's regime)."""
    import os
    m, orc, cfg, sr, seed = _model("fe_b", "Model")
    eng = m.engine
    eng.set_offline_engine("time_batched")
    B, T1 = 56, 37
    xd = torch.from_numpy(make_input(B, T1 * cfg.hop_size + 5, 909, sr)).to(_dev())
    keys = ("FE_TB_NC", "FE_TB_G", "FE_TB_STREAMS", "FE_TB_FUSE")
    saved = {k: os.environ.pop(k, None) for k in keys}
    try:
        w0, s0 = [t.clone() for t in m(xd)]
        os.environ.update(FE_TB_NC="3", FE_TB_G="1", FE_TB_STREAMS="2")          # time chunks: the same kernels on the same rows
        w1, s1 = [t.clone() for t in m(xd)]
        assert torch.equal(w1, w0) and torch.equal(s1, s0), "time-chunked nodes changed the result"
        os.environ.update(FE_TB_NC="2", FE_TB_G="2", FE_TB_STREAMS="3")          # utterance groups: a group of 28 scans four rows per workgroup
        w2, s2 = [t.clone() for t in m(xd)]
        assert float((w2 - w0).abs().max()) <= 2e-5 * max(1.0, float(w0.abs().max()))
        assert float((s2 - s0).abs().max()) <= 2e-5 * max(1.0, float(s0.abs().max()))
        for k in keys[:3]:
            os.environ.pop(k, None)
        os.environ["FE_TB_FUSE"] = "1"
        w3, s3 = [t.clone() for t in m(xd)]
        w3b, _ = m(xd)
        assert torch.equal(w3, w3b), "fused stage is not deterministic"
        assert float((w3 - w0).abs().max()) <= 2e-5 * max(1.0, float(w0.abs().max()))
        assert float((s3 - s0).abs().max()) <= 2e-5 * max(1.0, float(s0.abs().max()))
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    eng.set_offline_engine("auto")


@pytest.mark.parametrize("name", NONCAUSAL)
def test_noncausal_model_forward_matches_reference_golden(name):
    """SURVEY.md §8(f) rank 4, models/fastenhancer/noncausal/model.py:628-635: Model.forward(noisy) of the three huge_noncausal yamls
    (bidirectional GRU over time) vs the reference's own output."""
    g = load_golden(name)
    m, orc, cfg, sr, seed = _model(name, "Model")
    B, H = int(g["B"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, int(g["hops"]) * H + 37, seed + 2000, sr)).to(_dev())
    wav_hat, spec_hat = m(x)
    assert tuple(wav_hat.shape) == g["offline_wav"].shape and tuple(spec_hat.shape) == g["offline_spec"].shape
    _assert_close(wav_hat.cpu().numpy(), g["offline_wav"], "offline wav")
    _assert_close(spec_hat.cpu().numpy(), g["offline_spec"], "offline spec")
    wav3, _ = m(x.unsqueeze(1))                       # [B, 1, Tw] input form
    assert torch.equal(wav3, wav_hat)


@pytest.mark.parametrize("name", ["fe_nc", "fe48_nc"])
def test_noncausal_longer_batch_matches_oracle(name):
    """more frames than any tile or scan chunk, three utterances (rows of different utterances share scan workgroups), and the
    reverse direction really matters: truncating the input changes the EARLIER output"""
    m, orc, cfg, sr, seed = _model(name, "Model")
    H = cfg.hop_size
    x = make_input(3, 45 * H + 5, 808, sr)
    wav_ref, spec_ref = orc.offline_forward(x)
    xd = torch.from_numpy(x).to(_dev())
    wav_hat, spec_hat = m(xd)
    _assert_close(wav_hat.cpu().numpy(), wav_ref, "noncausal offline wav")
    _assert_close(spec_hat.cpu().numpy(), spec_ref, "noncausal offline spec")
    w1, s1 = m(xd[2:3].contiguous())
    assert float((w1 - wav_hat[2:3]).abs().max()) == 0.0
    w_short, _ = m(xd[:, :30 * H + 5].contiguous())
    assert float((w_short[:, :10 * H] - wav_hat[:, :10 * H]).abs().max()) > 1e-4, "the future does not reach the past: not bidirectional"


def test_noncausal_surface_and_errors():
    """the reference module has the offline Model only: no ONNXModel, no caches, no streaming / spec step"""
    import importlib
    from fastenhancer_amd import _lib
    mod = importlib.import_module("fastenhancer_amd.models.fastenhancer.noncausal.model")
    assert hasattr(mod, "Model") and not hasattr(mod, "ONNXModel")
    m, orc, cfg, sr, seed = _model("fe_nc", "Model")
    eng = m.engine
    with pytest.raises(AttributeError):
        m.initialize_cache(torch.zeros(1, 100))
    with pytest.raises(_lib.FEError, match="noncausal"):
        eng.step(torch.zeros(1, cfg.hop_size, device=_dev()), eng.new_state(1))
    with pytest.raises(_lib.FEError, match="time-batched engine only"):
        eng.set_offline_engine("frame_walk")
    with pytest.raises(_lib.FEError, match="reflect padding"):
        m(torch.zeros(1, 100, device=_dev()))


def test_streaming_model_on_an_index_less_cuda_device():
    """`.cuda()` / 'cuda' store torch.device('cuda') (no index) while tensors report cuda:0 - the state buffers must
    not be taken for foreign tensors (and, dptransformer, the K / V rings not be lost) on every call"""
    from fastenhancer_amd.streaming import StreamingModel
    for name in ("fe_b", "fe_dpt_t"):
        kw, sr, seed = MODEL_KWARGS[name]
        cfg, sd, fused, orc = build_oracle(name)
        mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE[name]}.model")
        m = mod.ONNXModel(**kw).cuda().eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        M = StreamingModel(m)
        B, hops, H = 2, 8, cfg.hop_size
        x = make_input(B, hops * H, 909, sr)
        xd = torch.from_numpy(x).to("cuda")
        caches = M.initialize_cache(xd)
        ref_c = orc.initialize_cache(B)
        n_state = m.engine.state_floats(B)
        for t in range(hops):
            assert M._source(caches, B, n_state) is not None, "caches on 'cuda' are packed like foreign tensors on every call"
            wav_out, *caches = M(xd[:, t * H:(t + 1) * H], *caches)
            ref_o, *ref_c = orc.step(x[:, t * H:(t + 1) * H], *ref_c)
            _assert_close(wav_out.cpu().numpy(), ref_o, f"{name} hop {t}")
        # a second initialize_cache hands out FRESH buffers: the first session's caches keep their values
        snap = [c.clone() for c in caches]
        caches2 = M.initialize_cache(xd)
        assert all(float(c.abs().max()) == 0.0 for c in caches2)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(snap, caches))


@pytest.mark.parametrize("name", ["fe_dprnn_s", "fe_dprnn_m", "fe_dpt_s"])
def test_offline_matches_oracle(name):
    m, orc, cfg, sr, seed = _model(name, "Model")
    H = cfg.hop_size
    x = make_input(2, (38 if cfg.dpt else 9) * H + 11, 555, sr)        # (dptransformer: more frames than its lookbehind)
    wav_ref, spec_ref = orc.offline_forward(x)
    wav_hat, spec_hat = m(torch.from_numpy(x).to(_dev()))
    _assert_close(wav_hat.cpu().numpy(), wav_ref, "offline wav")
    _assert_close(spec_hat.cpu().numpy(), spec_ref, "offline spec")


def test_command_line_callers(tmp_path, monkeypatch):
    """a26 / a27: the test_onnx.py- and test_pytorch.py-style callers on a reference-format checkpoint."""
    import yaml
    from scipy.io import wavfile
    from fastenhancer_amd.scripts import test_offline, test_streaming
    name = "fe_t"
    kw, sr, seed = MODEL_KWARGS[name]
    cfg, sd, fused, orc = build_oracle(name)
    logs = tmp_path / "logs" / "run"
    logs.mkdir(parents=True)
    (logs / "config.yaml").write_text(yaml.safe_dump({"model": "fastenhancer.default", "model_kwargs": kw, "data": {"sampling_rate": sr}}))
    torch.save({"model": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "epoch": 7}, str(logs / "00007.pth"))
    noisy_dir = tmp_path / "noisy"
    noisy_dir.mkdir()
    x = make_input(1, 5000, 91, sr)[0]
    wavfile.write(str(noisy_dir / "a.wav"), sr, x)
    monkeypatch.chdir(tmp_path)
    test_offline.main(["-n", "run", "-i", str(noisy_dir), "-o", str(tmp_path / "out")])
    rate, y = wavfile.read(str(tmp_path / "out" / "a.wav"))
    wav_ref, _ = orc.offline_forward(x[None])
    assert rate == sr
    _assert_close(y, wav_ref[0], "test_offline CLI")
    test_streaming.main(["-n", "run", "--audio-path", str(noisy_dir / "a.wav"), "--save-output", "--output-path",
                         str(tmp_path / "s.wav"), "--n-fft", "512", "--hop-size", "256", "--sr", "16000"])
    rate, ys = wavfile.read(str(tmp_path / "s.wav"))
    _assert_close(ys, orc.enhance_stream(x[None])[0], "test_streaming CLI")


@pytest.mark.parametrize("name,n_files,engine", [("fe_b", 40, "time_batched"), ("fe_t", 9, "time_batched"), ("fe_nc", 6, None), ("fe_tk_b", 3, None), ("bsrnn_xt", 3, None),
                                                 ("fe_l", 9, "auto"), ("fe_m", 8, "auto"), ("fe_l", 9, "frame_walk"), ("fe_b", 5, "frame_walk")])
def test_ragged_offline_batch_is_bit_identical_to_one_call_per_utterance(name, n_files, engine):
    """fe_offline_ragged (Model.forward on a LIST of utterances - what scripts/test_pytorch.py:28-37 does file by file): n files of
    different lengths in one batched call give, bit for bit, what each file's own call gives.  fe_b / fe_t / fe_nc run ONE batched pass of
    the time-batched engine (per-utterance reflect padding, reverse-scan start and overlap-add); fe_tk_b / bsrnn_xt have no batched form
    and are walked one after the other inside the call.  A sample of the files is also checked against the oracle.
    r5: the big shapes under AUTO with 8+ files (AUTO walks an equal-length batch of that size; a ragged one takes the batched pass) and the
    explicit frame walk (one by one inside the call, scratch sized for it)."""
    if name.startswith("bsrnn"):
        m, orc, cfg, sr, seed = _bsrnn(name, "Model")
    else:
        m, orc, cfg, sr, seed = _model(name, "Model")
    eng = m.engine
    if engine:
        eng.set_offline_engine(engine)
    rng = np.random.default_rng(5)
    H = cfg.hop_size
    lens = [int(v) for v in rng.integers(cfg.n_fft // 2 + 1 + 3 * H, 60 * H, size=n_files)]
    lens[0] = 23 * H                       # a whole number of hops
    lens[1] = max(lens) + 5                # the longest
    lens[2] = cfg.n_fft // 2 + 1           # the shortest a centered STFT takes
    xs = [make_input(1, n, 100 + i, sr)[0] for i, n in enumerate(lens)]
    wavs, specs = m([torch.from_numpy(x) for x in xs])
    assert len(wavs) == len(specs) == n_files
    for i, x in enumerate(xs):
        w1, s1 = m(torch.from_numpy(x)[None].to(_dev()))
        assert wavs[i].shape == w1[0].shape and specs[i].shape == s1[0].shape, (i, wavs[i].shape, w1.shape)
        assert torch.equal(wavs[i], w1[0]), (i, lens[i], float((wavs[i] - w1[0]).abs().max()))
        assert torch.equal(specs[i], s1[0]), (i, lens[i])
    for i in (0, 1, n_files - 1):
        wav_ref, spec_ref = orc.offline_forward(xs[i][None])
        _assert_close(wavs[i].cpu().numpy(), wav_ref[0], f"ragged batch, file {i} vs oracle")
        _assert_close(specs[i].cpu().numpy(), spec_ref[0], f"ragged batch, file {i} spec vs oracle")
    if engine:
        eng.set_offline_engine("auto")


def test_offline_cli_pushes_a_directory_through_in_ragged_batches(tmp_path, monkeypatch):
    """scripts/test_offline.py --batch: a directory of files of different lengths, sorted by length and enhanced in ragged batches, writes
    the same samples as file-by-file calls (--batch 1, the reference's loop)."""
    import yaml
    from scipy.io import wavfile
    from fastenhancer_amd.scripts import test_offline
    name = "fe_b"
    kw, sr, seed = MODEL_KWARGS[name]
    cfg, sd, fused, orc = build_oracle(name)
    logs = tmp_path / "logs" / "run"
    logs.mkdir(parents=True)
    (logs / "config.yaml").write_text(yaml.safe_dump({"model": "fastenhancer.default", "model_kwargs": kw, "data": {"sampling_rate": sr}}))
    torch.save({"model": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "epoch": 3}, str(logs / "00003.pth"))
    noisy_dir = tmp_path / "noisy"
    noisy_dir.mkdir()
    rng = np.random.default_rng(11)
    for i, n in enumerate(rng.integers(2000, 30000, size=11)):
        wavfile.write(str(noisy_dir / f"f{i:02d}.wav"), sr, make_input(1, int(n), 300 + i, sr)[0])
    monkeypatch.chdir(tmp_path)
    test_offline.main(["-n", "run", "-i", str(noisy_dir), "-o", str(tmp_path / "batched"), "--batch", "4"])
    test_offline.main(["-n", "run", "-i", str(noisy_dir), "-o", str(tmp_path / "single"), "--batch", "1"])
    for i in range(11):
        ra, ya = wavfile.read(str(tmp_path / "batched" / f"f{i:02d}.wav"))
        rb, yb = wavfile.read(str(tmp_path / "single" / f"f{i:02d}.wav"))
        assert ra == rb == sr and ya.shape == yb.shape
        assert np.array_equal(ya, yb), i      # (both run the time-batched engine: a ragged batch's rows are bit-identical to single calls)


# ------------------------------------------------------------------------------------------------ BSRNN (a22-a25)
def _bsrnn(name, cls="ONNXModel"):
    from common import BSRNN_KWARGS, build_bsrnn_oracle
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg, sd, fused, orc = build_bsrnn_oracle(name)
    mod = importlib.import_module("fastenhancer_amd.models.bsrnn.model")
    m = getattr(mod, cls)(**kw).to(_dev()).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, sr, seed


@pytest.mark.parametrize("name", ["bsrnn_xxt", "bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_every_stage_matches_oracle(name):
    """Per-stage activations (fe_debug_step): band split, each layer's time- / band-LSTM half, the mask MLPs."""
    m, orc, cfg, sr, seed = _bsrnn(name)
    eng = m.engine
    B, hops, H = 3, 3, cfg.hop_size
    x = make_input(B, hops * H, 515, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    names = [s_[0] for s_ in eng.debug_stages()]
    assert names == ["spec_in", "compressed", "band_split"] + [f"layer.{l}.{h}" for l in range(cfg.num_layers) for h in ("time", "freq")] + ["mask_mlp", "spec_out"]
    for t in range(hops):
        taps = {}
        o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
        o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
        for sname in names:
            tap = taps[sname]
            ref = tap[:, :, 0, :] if sname in ("spec_in", "spec_out", "compressed", "mask_mlp") else tap[0]
            _assert_close(dumps[sname].cpu().numpy(), ref, f"{name} hop {t} stage {sname}")
        _assert_close(o_gpu.cpu().numpy(), o_ref, f"{name} hop {t} wav_out")
    for a_, b_ in zip(eng.split_state(state, B), caches):
        _assert_close(a_.cpu().numpy(), b_, f"{name} cache after debug steps")


@pytest.mark.parametrize("name", ["bsrnn_xxt", "bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_streaming_matches_reference_golden(name):
    from fastenhancer_amd.streaming import StreamingModel
    g = load_golden(name)
    m, orc, cfg, sr, seed = _bsrnn(name)
    M = StreamingModel(m)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to(_dev())
    caches = M.initialize_cache(x)
    outs = []
    for t in range(hops):
        wav_out, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        outs.append(wav_out.cpu().numpy())
    _assert_close(np.stack(outs, 0), g["stream_wav_out"], "wav_out")
    _assert_close(caches[0].cpu().numpy(), g["stream_cache_stft"], "cache_stft")
    _assert_close(caches[1].cpu().numpy(), g["stream_cache_istft"], "cache_istft")
    for i in range(2 * cfg.num_layers):
        _assert_close(caches[2 + i].cpu().numpy(), g[f"stream_c{i}"], f"lstm cache {i}")


@pytest.mark.parametrize("name", ["bsrnn_xxt", "bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_offline_matches_reference_golden(name):
    g = load_golden(name)
    m, orc, cfg, sr, seed = _bsrnn(name, "Model")
    x = torch.from_numpy(make_input(int(g["B"]), int(g["hops"]) * cfg.hop_size + 37, seed + 2000, sr)).to(_dev())
    wav_hat, spec_hat = m(x)
    _assert_close(wav_hat.cpu().numpy(), g["offline_wav"], "offline wav")
    _assert_close(spec_hat.cpu().numpy(), g["offline_spec"], "offline spec")


@pytest.mark.parametrize("name,B", [("bsrnn_xt", 1), ("bsrnn_xt", 3), ("bsrnn_t", 2), ("bsrnn_s", 1)])
def test_bsrnn_time_pipelined_offline_agrees_with_the_serial_walk(name, B):
    """fe_offline of BSRNN with the frames of an utterance spread over co-resident workgroups (the time-LSTM state handed from frame
    to frame through per-layer counters) against one workgroup walking the frames, and against the oracle."""
    m, orc, cfg, sr, seed = _bsrnn(name, "Model")
    eng = m.engine
    x = make_input(B, 41 * cfg.hop_size + 19, seed + 31, sr)
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in eng.offline(xd)]
    for width in (-1, 5, 16):
        eng.set_time_pipeline(width)
        for rep in range(2):          # (twice: the counters live in the work buffer and are zeroed by every call)
            w, s = eng.offline(xd)
            assert float((w - w_ser).abs().max()) <= 2e-5 * max(1.0, float(w_ser.abs().max())), (width, rep, float((w - w_ser).abs().max()))
            assert float((s - s_ser).abs().max()) <= 2e-5 * max(1.0, float(s_ser.abs().max())), (width, rep)
    eng.set_time_pipeline(-1)
    wav_ref, spec_ref = orc.offline_forward(x)
    w, s = eng.offline(xd)
    _assert_close(w.cpu().numpy(), wav_ref, "pipelined offline wav vs oracle")
    _assert_close(s.cpu().numpy(), spec_ref, "pipelined offline spec vs oracle")


@pytest.mark.parametrize("name", ["bsrnn_xt", "bsrnn_t", "bsrnn_s"])
def test_bsrnn_batch_and_chunk_vs_oracle(name):
    m, orc, cfg, sr, seed = _bsrnn(name)
    eng = m.engine
    B, T, H = 5, 5, cfg.hop_size
    x = make_input(B, T * H, 77, sr)
    xd = torch.from_numpy(x).to(_dev())
    s1, s2 = eng.new_state(B), eng.new_state(B)
    y1 = eng.step(xd, s1, T=T)
    y2 = torch.cat([eng.step(xd[:, t * H:(t + 1) * H], s2, T=1) for t in range(T)], dim=1)
    # chunked launch (generic instantiation) vs per-hop launches (the T = 1 instantiation of the same kernel): the two
    # are compiled separately, so allow fp32 re-association noise (they are identical to ~1e-6; parity with the oracle
    # below is the real check)
    assert float((y1 - y2).abs().max()) <= 2e-6 * max(1.0, float(y1.abs().max())), float((y1 - y2).abs().max())
    assert float((s1 - s2).abs().max()) <= 2e-6 * max(1.0, float(s1.abs().max())), float((s1 - s2).abs().max())
    caches = orc.initialize_cache(B)
    refs = []
    for t in range(T):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        refs.append(o)
    _assert_close(y1.cpu().numpy(), np.concatenate(refs, 1), "bsrnn chunk vs oracle")
    for a_, b_ in zip(eng.split_state(s1, B), caches):
        _assert_close(a_.cpu().numpy(), b_, "bsrnn cache")
    # spec -> spec, T frames at once (the reference's LSTMCell path only takes T=1)
    spec = []
    c0 = orc.initialize_cache(B)[0]
    for t in range(3):
        s_, c0 = orc.stft_step(x[:, t * H:(t + 1) * H], c0)
        spec.append(s_)
    spec = np.concatenate(spec, axis=2)
    ref_spec, _ = orc.spec_forward(spec, [np.zeros((B * cfg.n_bands, cfg.hidden), np.float32) for _ in range(2 * cfg.num_layers)])
    got, *_ = m(torch.from_numpy(spec).to(_dev()), *m.initialize_cache(torch.zeros(B, 1, device=_dev())))
    _assert_close(got.cpu().numpy(), ref_spec, "bsrnn spec chunk")


def test_si_sdr_of_hip_path_equals_oracle_path_to_2dp():
    """north_star: SI-SDR identical to 2 d.p. between the HIP path and the reference path (here: its pinned oracle),
    measured against a synthetic clean target."""
    from fastenhancer_amd.metrics import si_snr
    from fastenhancer_amd.streaming import enhance_stream
    m, orc, cfg, sr, seed = _model("fe_b")
    rng = np.random.default_rng(5)
    t = np.arange(3 * sr // 4) / sr
    clean = (0.3 * np.sin(2 * np.pi * 220 * t)[None] * np.ones((2, 1))).astype(np.float32)
    noisy = np.clip(clean + 0.1 * rng.standard_normal(clean.shape).astype(np.float32), -1, 1)
    y_gpu = enhance_stream(m, torch.from_numpy(noisy)).cpu()
    y_ref = torch.from_numpy(orc.enhance_stream(noisy))
    s_gpu, s_ref = si_snr(y_gpu, torch.from_numpy(clean)), si_snr(y_ref, torch.from_numpy(clean))
    assert torch.all((s_gpu - s_ref).abs() < 5e-3), (s_gpu, s_ref)
    assert [round(float(v), 2) for v in s_gpu] == [round(float(v), 2) for v in s_ref] or torch.all((s_gpu - s_ref).abs() < 1e-3)


def test_shape_outside_the_shipped_yamls_builds_and_matches_the_oracle(tmp_path):
    """SURVEY's hop-generic / shape-generic note: a model_kwargs no yaml ships (40 channels, 28 x 20 RNNFormer, 2 blocks,
    hop 128) is rejected with the build command in the message, `--add-shape` compiles its kernel (side build: the in-tree
    library is untouched), and the result matches the oracle."""
    import os
    import subprocess
    import sys
    from fastenhancer_amd import _lib
    from fastenhancer_amd.config import FEConfig
    from fastenhancer_amd.engine import Engine
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kw = dict(MODEL_KWARGS["fe_b"][0])
    kw.update(channels=40, hop_size=128, rnnformer_kwargs=dict(kw["rnnformer_kwargs"], num_blocks=2, channels=28, freq=20))
    with pytest.raises(_lib.FEError, match="--add-shape 40,2,28,20,2,512,128,1"):
        Engine(FEConfig.from_model_kwargs(**kw), _dev())
    env = dict(os.environ, FE_BUILD_TAG="addshape", FE_LOCAL_DEF=str(tmp_path / "local.def"))
    r = subprocess.run([sys.executable, "-m", "fastenhancer_amd.build", "--add-shape", "40,2,28,20,2,512,128"], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    child = """
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from common import MODEL_KWARGS, rms
from oracle.fe_oracle import FEConfig as OCfg, FEOracle, fold_state_dict
from oracle.weightgen import make_input, make_training_state_dict
import importlib
kw = dict(MODEL_KWARGS['fe_b'][0]); kw.update(channels=40, hop_size=128, rnnformer_kwargs=dict(kw['rnnformer_kwargs'], num_blocks=2, channels=28, freq=20))
ocfg = OCfg.from_model_kwargs(kw); sd = make_training_state_dict(ocfg, 77); orc = FEOracle(ocfg, fold_state_dict(sd, ocfg), np.float32)
m = importlib.import_module('fastenhancer_amd.models.fastenhancer.default.model').ONNXModel(**kw).to('cuda:0').eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
B, hops, H = 5, 8, 128
x = make_input(B, hops * H, 3, 16000)
st = m.engine.new_state(B)
out = m.engine.step(torch.from_numpy(x).cuda(), st, T=hops).cpu().numpy()
c = orc.initialize_cache(B); refs = []
for t in range(hops):
    o, *c = orc.step(x[:, t * H:(t + 1) * H], *c); refs.append(o)
ref = np.concatenate(refs, 1)
err, r = rms(out - ref), rms(ref)
assert err < 1e-4 * r and err < 1e-4, (err, r)
print('added-shape parity ok', err, r)
""" % (repo, repo)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, FASTENHANCER_HIP_LIB=os.path.join(repo, "ab", "lib_addshape.so")),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "added-shape parity ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_bsrnn_more_streams_than_cus():
    """persistent BSRNN workgroups: 300 streams on 256 CUs (workgroups 0..43 walk two streams each)"""
    m, orc, cfg, sr, seed = _bsrnn("bsrnn_xxt")
    _full_size_check(m, orc, cfg, sr, 300, 3, [0, 1, 43, 44, 255, 256, 257, 299], "bsrnn_xxt B=300")


@pytest.mark.parametrize("name,B", [("bsrnn_xt", 2605), ("bsrnn_xxt", 4096), ("bsrnn_t", 2093), ("bsrnn_s", 2829)])
def test_bsrnn_stream_batched_layers_above_2048_streams(name, B):
    """From 2048 streams the per-hop step runs its LSTM layers batched over the streams on the matrix cores (bsrnn_sb_kernels.hip.h:
    front per stream, sixteen streams per workgroup through the layers, batched mask-decoder MLP, tail per stream).  2605 streams = a
    last tile of 13; r6: bsrnn_t (num_channels = 32: the time LSTM's gate tiles split over the waves, bands in groups of eight), 2093 streams = a
    last tile of 13, and bsrnn_s (num_channels = 64, bsrnn_sb64_layers_kernel: band features in global memory, one direction at a time, x fragments
    streamed; from 2816 streams - the threshold counts 11 / 8 there), 2829 = a last tile of 13.  Oracle parity (outputs and every time-LSTM cache) on a sample that covers first / last tiles and columns, and
    bitwise position independence on all streams (_full_size_check runs the batch again in reversed order: every stream then sits
    in another tile and another column)."""
    m, orc, cfg, sr, seed = _bsrnn(name)
    _full_size_check(m, orc, cfg, sr, B, 3 if name != "bsrnn_s" else 2, [0, 1, 15, 16, 17, 1000, 2047, B - 14, B - 13, B - 2, B - 1], f"{name} B={B}")
    assert "bsrnn_sb" in m.engine.last_step_kernel(), m.engine.last_step_kernel()


def test_bsrnn_split_step_with_ragged_stream_tiles():
    """the per-hop step in three launches (frame kernel head, mask-decoder MLPs batched over the streams, tail): 1030 streams = two
    persistent workgroups per CU in the head / tail and a last MLP stream tile of 6 rows (16-stream tiles, 64-stream workgroups);
    7 streams = one partial tile.  Oracle parity on a sample and position independence (_full_size_check)."""
    m, orc, cfg, sr, seed = _bsrnn("bsrnn_xt")
    _full_size_check(m, orc, cfg, sr, 1030, 2, [0, 15, 16, 63, 64, 511, 512, 1023, 1024, 1029], "bsrnn_xt B=1030")
    _full_size_check(m, orc, cfg, sr, 7, 3, [0, 3, 6], "bsrnn_xt B=7")


def test_steps_can_be_captured_into_a_hip_graph():
    """no allocation, free or synchronisation inside fe_step: a burst of per-hop launches captured into a HIP graph
    (torch.cuda.CUDAGraph on the capture stream) replays to the same bits as the eager launches, for the LDS-skip (B),
    global-skip (M: per-workgroup scratch allocated at load time) and BSRNN kernels"""
    for name, loader in (("fe_b", _model), ("fe_m", _model), ("bsrnn_xt", _bsrnn)):
        m, orc, cfg, sr, seed = loader(name)
        eng = m.engine
        B, hops, H = 7, 5, cfg.hop_size
        x = torch.from_numpy(make_input(B, hops * H, 21, sr)).to(_dev())
        s_eager = eng.new_state(B)
        y_eager = torch.cat([eng.step(x[:, t * H:(t + 1) * H], s_eager, T=1) for t in range(hops)], dim=1)
        s_graph = eng.new_state(B)
        y_graph = torch.empty(B, hops * H, device=_dev())
        eng.step(x[:, :H], eng.new_state(B), T=1)           # (first launch outside the capture: sets the function attribute)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(hops):
                eng.step(x[:, t * H:(t + 1) * H], s_graph, wav_out=y_graph[:, t * H:(t + 1) * H], T=1)
        s_graph.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_graph, y_eager) and torch.equal(s_graph, s_eager), name
        g.replay()                                           # a second replay continues the streams (state carried)
        torch.cuda.synchronize()
        y2 = torch.cat([eng.step(x[:, t * H:(t + 1) * H], s_eager, T=1) for t in range(hops)], dim=1)
        assert torch.equal(y_graph, y2) and torch.equal(s_graph, s_eager), name


# ------------------------------------------------------------------------------------------------ FSPEN
def _fspen(cls="ONNXModel"):
    from common import FSPEN_KWARGS, build_fspen_oracle
    kw, sr, seed = FSPEN_KWARGS
    cfg, sd, fused, orc = build_fspen_oracle()
    mod = importlib.import_module("fastenhancer_amd.models.fspen.model")
    m = getattr(mod, cls)(**kw).to(_dev()).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, sr, seed


def test_fspen_every_stage_matches_oracle():
    """fe_debug_step taps of the FSPEN kernel against the oracle's (models/fspen/model.py:342-407), three hops with state."""
    m, orc, cfg, sr, seed = _fspen()
    eng = m.engine
    B, hops, H = 3, 3, cfg.hop_size
    x = make_input(B, hops * H, 616, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    names = [s_[0] for s_ in eng.debug_stages()]
    assert names == (["spec_in", "compressed", "subband_encoder", "fullband_encoder.2", "feature_merge"]
                     + [f"dpe.{b}.{h}" for b in range(cfg.num_blocks) for h in ("intra", "inter")]
                     + ["feature_split", "fullband_decoder.0", "fullband_decoder.1", "mask", "spec_out"])
    for t in range(hops):
        taps = {}
        o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
        o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
        for sname in names:
            tap = taps[sname]
            if sname in ("spec_in", "spec_out", "compressed", "mask"):
                ref = tap[:, :, 0, :]                      # [B, 257, T = 1, c]
            elif sname.startswith("dpe."):
                ref = tap[0]                               # [T = 1, B, F, C]
            else:
                ref = tap                                  # [B * T, C, F]
            _assert_close(dumps[sname].cpu().numpy(), ref, f"fspen hop {t} stage {sname}")
        _assert_close(o_gpu.cpu().numpy(), o_ref, f"fspen hop {t} wav_out")
    for a_, b_ in zip(eng.split_state(state, B), caches):
        _assert_close(a_.cpu().numpy(), b_, "fspen cache after debug steps")


def test_fspen_streaming_matches_reference_golden():
    """scripts/export_onnx.py:48-58 composition with `model: fspen`: 10 hops x 2 streams, all 24 inter-GRU caches."""
    from fastenhancer_amd.streaming import StreamingModel
    g = load_golden("fspen")
    m, orc, cfg, sr, seed = _fspen()
    M = StreamingModel(m)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to(_dev())
    caches = M.initialize_cache(x)
    outs = []
    for t in range(hops):
        wav_out, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        outs.append(wav_out.cpu().numpy())
    _assert_close(np.stack(outs, 0), g["stream_wav_out"], "wav_out")
    _assert_close(caches[0].cpu().numpy(), g["stream_cache_stft"], "cache_stft")
    _assert_close(caches[1].cpu().numpy(), g["stream_cache_istft"], "cache_istft")
    for i in range(cfg.n_caches):
        _assert_close(caches[2 + i].cpu().numpy(), g[f"stream_c{i}"], f"inter GRU cache {i}")
    # the model mirror's own spec -> spec call (ONNXModel.forward) on the last hop's input
    m2, *_ = _fspen()
    c0 = orc.initialize_cache(B)
    spec_in, _ = orc.stft_step(make_input(B, hops * H, seed + 1000, sr)[:, :H], c0[0])
    ref, ref_c = orc.spec_forward(spec_in, c0[2:])
    got, *got_c = m2(torch.from_numpy(spec_in).to(_dev()), *m2.initialize_cache(torch.zeros(B, 1, device=_dev())))
    _assert_close(got.cpu().numpy(), ref, "fspen spec -> spec")
    for a_, b_ in zip(got_c, ref_c):
        _assert_close(a_.cpu().numpy(), b_, "fspen spec -> spec cache")


def test_fspen_offline_matches_reference_golden():
    g = load_golden("fspen")
    m, orc, cfg, sr, seed = _fspen("Model")
    x = torch.from_numpy(make_input(int(g["B"]), int(g["hops"]) * cfg.hop_size + 37, seed + 2000, sr)).to(_dev())
    wav_hat, spec_hat = m(x)
    _assert_close(wav_hat.cpu().numpy(), g["offline_wav"], "offline wav")
    _assert_close(spec_hat.cpu().numpy(), g["offline_spec"], "offline spec")


@pytest.mark.parametrize("B", [1, 4])
def test_fspen_time_pipelined_offline_agrees_with_the_serial_walk(B):
    """FSPEN's offline Model.forward with the frames of an utterance over co-resident workgroups (the inter-GRU states of each DPE
    block handed from frame to frame) against one workgroup walking the frames, and against the oracle."""
    m, orc, cfg, sr, seed = _fspen("Model")
    eng = m.engine
    x = make_input(B, 57 * cfg.hop_size + 21, seed + 77, sr)
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in m(xd)]
    for width in (-1, 5):
        eng.set_time_pipeline(width)
        for rep in range(2):
            w, s_ = m(xd)
            assert float((w - w_ser).abs().max()) <= 2e-5 * max(1.0, float(w_ser.abs().max())), (width, rep, float((w - w_ser).abs().max()))
            assert float((s_ - s_ser).abs().max()) <= 2e-5 * max(1.0, float(s_ser.abs().max())), (width, rep)
    eng.set_time_pipeline(-1)
    wav_ref, spec_ref = orc.offline_forward(x)
    _assert_close(w.cpu().numpy(), wav_ref, "fspen pipelined offline wav vs oracle")
    _assert_close(s_.cpu().numpy(), spec_ref, "fspen pipelined offline spec vs oracle")


@pytest.mark.parametrize("B", [256, 600, 1000])
def test_fspen_full_size(B):
    """256 streams (one workgroup per CU), 600 (three per CU) and 1000 (persistent): oracle parity on a sample, bitwise
    position independence on all streams; chunked launch == per-hop launches"""
    m, orc, cfg, sr, seed = _fspen()
    _full_size_check(m, orc, cfg, sr, B, 3, [0, 1, 17, 255, B // 2, B - 2, B - 1], f"fspen B={B}")
    eng = m.engine
    H = cfg.hop_size
    xd = torch.from_numpy(make_input(5, 4 * H, 78, sr)).to(_dev())
    s1, s2 = eng.new_state(5), eng.new_state(5)
    y1 = eng.step(xd, s1, T=4)
    y2 = torch.cat([eng.step(xd[:, t * H:(t + 1) * H], s2, T=1) for t in range(4)], dim=1)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)


@pytest.mark.parametrize("B", [2057, 4096])
def test_fspen_stream_batched_middle_above_1536_streams(B):
    """From 1536 streams the per-hop step runs the middle of the network - fullband_encoder_post, feature merge, the three DPE blocks,
    feature split, fullband_decoder.0 - batched over the streams on the matrix cores (fspen_sb_kernels.hip.h: front per stream, sixteen
    streams per workgroup, tail per stream).  2057 streams = a last tile of 9.  Oracle parity (outputs and all 24 inter-GRU caches) on a sample that covers first / last tiles and columns, bitwise
    position independence on all streams (_full_size_check runs the batch again in reversed order)."""
    m, orc, cfg, sr, seed = _fspen()
    _full_size_check(m, orc, cfg, sr, B, 3, [0, 1, 15, 16, 17, 511, 1000, B - 10, B - 9, B - 2, B - 1], f"fspen B={B}")


# ------------------------------------------------------------------------------------------------ LiSenNet
def _lisennet(cls="ONNXModel"):
    from common import LISENNET_KWARGS, build_lisennet_oracle
    kw, sr, seed = LISENNET_KWARGS
    cfg, sd, _, orc = build_lisennet_oracle()
    mod = importlib.import_module("fastenhancer_amd.models.lisennet.model")
    m = getattr(mod, cls)(**kw).to(_dev()).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, sr, seed


@pytest.mark.parametrize("sb,B", [(False, 3), (True, 3), (True, 21)])
def test_lisennet_every_stage_matches_oracle(sb, B):
    """fe_debug_step taps of the LiSenNet kernel against the oracle's (models/lisennet/model.py:398-474), three hops with state.  sb (r6): the same taps
    and caches through the three-launch step whose middle - conv_1 .. the mask head - runs batched over sixteen streams per workgroup on the matrix cores
    (lisennet_sb_kernels.hip.h; fe_set_option("lisennet_stream_batch_min", 1)): 3 streams = one partly filled tile, 21 = a last tile of 5."""
    m, orc, cfg, sr, seed = _lisennet()
    eng = m.engine
    eng.set_option("lisennet_stream_batch_min", 1 if sb else 0)
    hops, H = 3, cfg.hop_size
    x = make_input(B, hops * H, 717, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    names = [s_[0] for s_ in eng.debug_stages()]
    assert names == (["spec_in", "compressed", "features", "encoder.conv_1", "encoder.conv_2", "encoder.conv_3", "encoder.conv_4"]
                     + [f"blocks.{b}{h}" for b in range(cfg.n_blocks) for h in (".intra", ".inter", "")] + ["decoder.up3", "mask", "spec_out"])
    for t in range(hops):
        taps = {}
        o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
        o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
        for sname in names:
            tap = taps[sname]
            if sname in ("spec_in", "spec_out", "compressed", "mask"):
                ref = tap[:, :, 0, :]                      # [B, 257, T = 1, 2]
            elif sname.endswith(".intra") or sname.endswith(".inter"):
                ref = tap[:, 0]                            # [B, T = 1, F, D]
            else:
                ref = tap[:, :, 0, :]                      # [B, C, T = 1, F]
            _assert_close(dumps[sname].cpu().numpy(), ref, f"lisennet hop {t} stage {sname}")
        _assert_close(o_gpu.cpu().numpy(), o_ref, f"lisennet hop {t} wav_out")
        assert ("lisennet_sb_kernel" in eng.last_step_kernel()) == sb, eng.last_step_kernel()
    for a_, b_ in zip(eng.split_state(state, B), caches):
        _assert_close(a_.cpu().numpy(), b_, "lisennet cache after debug steps")


def test_lisennet_streaming_matches_reference_golden():
    """scripts/export_onnx.py:48-58 composition with `model: lisennet`: 10 hops x 2 streams, all 9 model caches."""
    from fastenhancer_amd.streaming import StreamingModel
    g = load_golden("lisennet")
    m, orc, cfg, sr, seed = _lisennet()
    M = StreamingModel(m)
    B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to(_dev())
    caches = M.initialize_cache(x)
    outs = []
    for t in range(hops):
        wav_out, *caches = M(x[:, t * H:(t + 1) * H], *caches)
        outs.append(wav_out.cpu().numpy())
    _assert_close(np.stack(outs, 0), g["stream_wav_out"], "wav_out")
    _assert_close(caches[0].cpu().numpy(), g["stream_cache_stft"], "cache_stft")
    _assert_close(caches[1].cpu().numpy(), g["stream_cache_istft"], "cache_istft")
    for i in range(cfg.n_caches):
        _assert_close(caches[2 + i].cpu().numpy(), g[f"stream_c{i}"], f"model cache {i}")
    # the mirror's own spec -> spec call (ONNXModel.forward) on the first hop
    m2, *_ = _lisennet()
    c0 = orc.initialize_cache(B)
    spec_in, _ = orc.stft_step(make_input(B, hops * H, seed + 1000, sr)[:, :H], c0[0])
    ref, ref_c = orc.spec_forward(spec_in, c0[2:])
    got, *got_c = m2(torch.from_numpy(spec_in).to(_dev()), *m2.initialize_cache(torch.zeros(B, 1, device=_dev())))
    _assert_close(got.cpu().numpy(), ref, "lisennet spec -> spec")
    for a_, b_ in zip(got_c, ref_c):
        _assert_close(a_.cpu().numpy(), b_, "lisennet spec -> spec cache")


def test_lisennet_offline_matches_oracle():
    """Model.forward (models/lisennet/model.py:512-531, torch.diff phase features).  Frame 0 of the reference's offline path is
    ill-conditioned (tools/gen_golden.py::gen_lisennet): the vector starts with silence, where the oracle (pinned on the reference
    through its own features) and the kernel both see phase 0; the result is also compared with the reference's golden output on
    every frame that the first one cannot reach... which is none (GRU state), so the golden comparison is the oracle's."""
    g = load_golden("lisennet")
    m, orc, cfg, sr, seed = _lisennet("Model")
    xo = make_input(int(g["B"]), int(g["hops"]) * cfg.hop_size + 37, seed + 2000, sr)
    xo[:, :int(g["offline_leading_zeros"])] = 0.0
    wav_ref, spec_ref = orc.offline_forward(xo)
    wav_hat, spec_hat = m(torch.from_numpy(xo).to(_dev()))
    _assert_close(wav_hat.cpu().numpy(), wav_ref, "offline wav")
    _assert_close(spec_hat.cpu().numpy(), spec_ref, "offline spec")


@pytest.mark.parametrize("B", [1, 3])
def test_lisennet_time_pipelined_offline_agrees_with_the_serial_walk(B):
    """LiSenNet's offline Model.forward with the frames of an utterance over co-resident workgroups: its nine caches (previous phase,
    encoder frames, per block GRU state + ConvGLU frames, decoder frame) go through a ring of per-frame slots with a counter per cache.
    Against the serial walk and the oracle (leading silence: frame 0 of the offline path is ill-conditioned, see the test above)."""
    g = load_golden("lisennet")
    m, orc, cfg, sr, seed = _lisennet("Model")
    eng = m.engine
    x = make_input(B, 61 * cfg.hop_size + 9, seed + 88, sr)
    x[:, :int(g["offline_leading_zeros"])] = 0.0
    xd = torch.from_numpy(x).to(_dev())
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in m(xd)]
    for width in (-1, 5):
        eng.set_time_pipeline(width)
        for rep in range(2):
            w, s_ = m(xd)
            assert float((w - w_ser).abs().max()) <= 3e-5 * max(1.0, float(w_ser.abs().max())), (width, rep, float((w - w_ser).abs().max()))
            assert float((s_ - s_ser).abs().max()) <= 3e-5 * max(1.0, float(s_ser.abs().max())), (width, rep)
    eng.set_time_pipeline(-1)
    wav_ref, spec_ref = orc.offline_forward(x)
    _assert_close(w.cpu().numpy(), wav_ref, "lisennet pipelined offline wav vs oracle")
    _assert_close(s_.cpu().numpy(), spec_ref, "lisennet pipelined offline spec vs oracle")


@pytest.mark.parametrize("which", ["fe_tk_b", "fe_dpt_b", "lisennet"])
def test_time_pipeline_width_above_the_ring_size_is_clamped(which):
    """fe_set_time_pipeline(128) with one long utterance: the per-frame rings in work_dev (time_kernel inputs, dptransformer K / V,
    LiSenNet caches) hold 64 frames in flight - a wider request is clamped (it used to index past the rings); results equal the serial walk."""
    if which == "lisennet":
        g = load_golden("lisennet")
        m, orc, cfg, sr, seed = _lisennet("Model")
    else:
        m, orc, cfg, sr, seed = _model(which, "Model")
    eng = m.engine
    x = make_input(1, 150 * cfg.hop_size + 5, 4711, sr)
    if which == "lisennet":
        x[:, :int(g["offline_leading_zeros"])] = 0.0
    xd = torch.from_numpy(x).to(_dev())
    guard = torch.full((1 << 20,), 7.0, device=_dev())          # (allocated right after the work buffers of the calls below)
    eng.set_time_pipeline(0)
    w_ser, s_ser = [t.clone() for t in m(xd)]
    eng.set_time_pipeline(128)
    for rep in range(2):
        w, s_ = m(xd)
        assert float((w - w_ser).abs().max()) <= 3e-5 * max(1.0, float(w_ser.abs().max())), (rep, float((w - w_ser).abs().max()))
        assert float((s_ - s_ser).abs().max()) <= 3e-5 * max(1.0, float(s_ser.abs().max())), rep
    assert bool((guard == 7.0).all())
    eng.set_time_pipeline(-1)


@pytest.mark.parametrize("B", [601, 4096])
def test_lisennet_stream_batched_middle_above_512_streams(B):
    """From 513 streams the per-hop step runs encoder.conv_1 .. the mask head batched over the streams on the matrix cores (lisennet_sb_kernels.hip.h:
    STFT + features per stream, sixteen streams per workgroup, mask + iSTFT per stream).  601 streams = a last tile of 9.  Oracle parity (outputs and all nine model caches) on
    a sample that covers first / last tiles and columns, bitwise position independence on all streams (_full_size_check runs the batch again in reversed
    order); the step equals the per-stream kernel's to fp32 rounding; and a batch below the threshold keeps the per-stream kernel."""
    m, orc, cfg, sr, seed = _lisennet()
    eng = m.engine
    _full_size_check(m, orc, cfg, sr, B, 3, [0, 1, 15, 16, 17, 511, min(1000, B - 11), B - 10, B - 9, B - 2, B - 1], f"lisennet B={B}")
    assert "lisennet_sb_kernel" in eng.last_step_kernel(), eng.last_step_kernel()
    if B == 601:
        H = cfg.hop_size
        x = torch.from_numpy(make_input(B, 2 * H, 91, sr)).to(_dev())
        outs = []
        for sb_min in (513, 0):
            eng.set_option("lisennet_stream_batch_min", sb_min)
            st = eng.new_state(B)
            y = torch.cat([eng.step(x[:, t * H:(t + 1) * H].contiguous(), st, T=1) for t in range(2)], dim=1)
            outs.append((y.clone(), st.clone()))
        assert "lisennet_sb_kernel" not in eng.last_step_kernel()
        eng.set_option("lisennet_stream_batch_min", 513)
        _assert_close(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), "lisennet stream-batched step vs per-stream step")
        _assert_close(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), "lisennet stream-batched step vs per-stream step: state")
        eng.step(x[:5, :H].contiguous(), eng.new_state(5), T=1)
        assert "lisennet_sb_kernel" not in eng.last_step_kernel(), "a small batch keeps the per-stream kernel"


@pytest.mark.parametrize("B", [256, 700])
def test_lisennet_full_size(B):
    """256 streams (one workgroup per CU) and 700 (two per CU, then persistent): oracle parity on a sample, bitwise position
    independence on all streams; chunked launch == per-hop launches"""
    m, orc, cfg, sr, seed = _lisennet()
    _full_size_check(m, orc, cfg, sr, B, 3, [0, 1, 17, 255, B // 2, B - 2, B - 1], f"lisennet B={B}")
    eng = m.engine
    H = cfg.hop_size
    xd = torch.from_numpy(make_input(5, 4 * H, 79, sr)).to(_dev())
    s1, s2 = eng.new_state(5), eng.new_state(5)
    y1 = eng.step(xd, s1, T=4)
    y2 = torch.cat([eng.step(xd[:, t * H:(t + 1) * H], s2, T=1) for t in range(4)], dim=1)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)


@pytest.mark.parametrize("name,hops,T", [("fe_dpt_t", 400, 1), ("fe_dpt_t", 400, 7), ("fe_dprnn_t", 300, 1), ("fe_tk_b", 150, 3)])
def test_block_variants_long_run_has_no_state_drift(name, hops, T):
    """hundreds of hops of two streams (per-hop launches, or chunks of T hops) against the oracle: the dptransformer variant's K / V
    rings go round a dozen times (and chunks of 7 do not divide 31), the GRU / conv-cache variants carry their states"""
    m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    B, H = 2, cfg.hop_size
    hops -= hops % T
    x = make_input(B, hops * H, 1234, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    got, ref = [], []
    for t in range(0, hops, T):
        got.append(eng.step(xd[:, t * H:(t + T) * H], state, T=T).cpu().numpy())
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        ref.append(o)
    _assert_close(np.concatenate(got, axis=1)[:, -40 * H:], np.concatenate(ref[-40:], axis=1), f"{name} last 40 hops of {hops}")
    for a_, b_ in zip(eng.split_state(state, B), caches):
        _assert_close(a_.cpu().numpy(), b_, f"{name} cache after {hops} hops")


@pytest.mark.parametrize("which", ["fspen", "lisennet"])
def test_baseline_models_long_run_has_no_state_drift(which):
    """60 hops (0.96 s) of two streams through the per-hop kernel against the oracle: the GRU states and the causal-conv frame caches are
    carried through the stream state the whole way (the goldens stop at 10 hops)"""
    m, orc, cfg, sr, seed = _fspen() if which == "fspen" else _lisennet()
    eng = m.engine
    B, hops, H = 2, 60, cfg.hop_size
    x = make_input(B, hops * H, 4321, sr)
    xd = torch.from_numpy(x).to(_dev())
    state = eng.new_state(B)
    caches = orc.initialize_cache(B)
    got, ref = [], []
    for t in range(hops):
        got.append(eng.step(xd[:, t * H:(t + 1) * H], state, T=1).cpu().numpy())
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        ref.append(o)
    _assert_close(np.concatenate(got[-20:], axis=1), np.concatenate(ref[-20:], axis=1), f"{which} hops 40..59")
    for a_, b_ in zip(eng.split_state(state, B), caches):
        _assert_close(a_.cpu().numpy(), b_, f"{which} cache after 60 hops")


@pytest.mark.parametrize("name,B,T", [("fe_dpt_b", 5, 97), ("fe_tk_b", 7, 61)])
def test_pipelined_spec_chunks_with_ring_caches_are_bit_reproducible(name, B, T):
    """the ring hand-over of fe_spec_step's pipelined chunks (r4v: K / V caches, conv input caches) under the same stress as the offline launches: ten
    repetitions of one chunk from the same caches must reproduce the first bit for bit - outputs and returned caches - and agree with the
    serial walk; the caches that go in have been through per-hop launches (non-zero ring heads)."""
    m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    spec = torch.from_numpy(make_input(B * 257 * (T + 4) * 2, 1, 55, sr).reshape(B, 257, T + 4, 2).astype(np.float32) * 0.3).to(_dev())
    h = m.initialize_cache(spec)
    for t in range(4):
        _, *h = m(spec[:, :, t:t + 1].contiguous(), *h)
    h = [t.clone() for t in h]
    chunk = spec[:, :, 4:].contiguous()
    eng.set_time_pipeline(0)
    y0, *h0 = m(chunk, *[t.clone() for t in h])
    eng.set_time_pipeline(-1)
    first = None
    for rep in range(10):
        y, *hn = m(chunk, *[t.clone() for t in h])
        if first is None:
            first = (y.clone(), [t.clone() for t in hn])
            scale = float(y0.abs().max())
            assert float((y - y0).abs().max()) <= 3e-5 * scale, "pipelined vs serial"
            for a_, b_ in zip(hn, h0):
                assert float((a_ - b_).abs().max()) <= 3e-5 * max(float(b_.abs().max()), 1e-3), "caches: pipelined vs serial"
        else:
            assert torch.equal(y, first[0]), f"repetition {rep} differs"
            for a_, b_ in zip(hn, first[1]):
                assert torch.equal(a_, b_), f"caches of repetition {rep} differ"


@pytest.mark.parametrize("name,B", [("fe_b", 7), ("fe_tk_b", 3), ("fe_dpt_b", 1), ("fe_ln_b", 3), ("bsrnn_xt", 7), ("fspen", 7), ("lisennet", 3)])
def test_time_pipeline_stress_is_bit_reproducible_and_equals_the_serial_walk(name, B):
    """A reduced tools/gpu_pipeline_stress.py in the suite, so that EVERY box the tests run on exercises the hand-rolled hand-off ordering
    of the time-pipelined launches (relaxed agent-scope counters + s_waitcnt vmcnt(0) + barrier, fe_kernels.hip.h): per family one batch
    size the other tests do not use (7 streams: ragged against every pipeline width), 10 repetitions that must reproduce the first bit for
    bit - the hand-off order does not change the arithmetic - and agree with one workgroup walking the frames serially."""
    if name == "fspen":
        m, orc, cfg, sr, seed = _fspen("Model")
    elif name == "lisennet":
        m, orc, cfg, sr, seed = _lisennet("Model")
    elif name.startswith("bsrnn"):
        m, orc, cfg, sr, seed = _bsrnn(name, "Model")
    else:
        m, orc, cfg, sr, seed = _model(name, "Model")
    eng = m.engine
    if name in TB_SHAPES:
        eng.set_offline_engine("frame_walk")
    xn = make_input(B, 97 * cfg.hop_size + 13, 31 + B, sr)
    if name == "lisennet":
        xn[:, :1024] = 0.0           # (frame 0 of its offline path is ill-conditioned on a non-silent start, see test_lisennet_offline_matches_oracle)
    x = torch.from_numpy(xn).to(_dev())
    eng.set_time_pipeline(0)
    w_ser = m(x)[0].clone()
    tol = (6e-5 if name == "fspen" else 2e-5) * max(1.0, float(w_ser.abs().max()))      # (FSPEN at 7 streams: 4.4e-5 absolute, profiles/r3t_pipeline_stress.txt)
    for width in (-1, 6):
        eng.set_time_pipeline(width)
        w0 = m(x)[0].clone()
        assert bool(torch.isfinite(w0).all())
        assert float((w0 - w_ser).abs().max()) <= tol, (width, float((w0 - w_ser).abs().max()))
        for rep in range(10):
            assert torch.equal(m(x)[0], w0), (width, rep)
    eng.set_time_pipeline(-1)


_POISON_CASES = [("fe_t", 5, None), ("fe_t", 600, None), ("fe48_t", 600, None), ("fe_b", 5, "wg8"), ("fe_b", 5, "waves4"), ("fe_b", 300, None), ("fe_s", 3, None), ("fe_m", 3, None), ("fe_l", 3, None),
                 ("fe48_t", 3, None), ("fe48_b", 3, None), ("fe48_b_h480", 300, None), ("fe48_l", 2, None), ("fe_tk_b", 3, None), ("fe_dprnn_b", 3, None),
                 ("fe_dpt_b", 3, None), ("fe_ln_b", 3, None), ("bsrnn_xt", 5, "wg8"), ("bsrnn_xt", 5, "waves4"), ("bsrnn_xt", 300, None), ("bsrnn_xxt", 3, None),
                 ("bsrnn_t", 3, None), ("bsrnn_s", 2, None), ("fspen", 3, None), ("lisennet", 3, None)]


@pytest.mark.parametrize("name,B,kern", _POISON_CASES)
def test_lds_leftovers_of_other_kernels_do_not_matter(name, B, kern):
    """r5: the kernels read padded operands in places - zero weights against words past the end of a tensor in LDS (BSRNN's band split reads the last
    band's row padded to 36 floats: two words past the spectrum).  What an earlier kernel left there must not matter: with every CU's LDS filled with
    NaN before each launch (fe_debug_poison_lds) a per-hop run gives the bits of the plain run.  (Found by a 400-hop run that came back non-finite
    once, right after process start.)"""
    if name.startswith("bsrnn"):
        m, orc, cfg, sr, seed = _bsrnn(name)
    elif name == "fspen":
        m, orc, cfg, sr, seed = _fspen()
    elif name == "lisennet":
        m, orc, cfg, sr, seed = _lisennet()
    else:
        m, orc, cfg, sr, seed = _model(name)
    eng = m.engine
    if kern is not None:
        eng.set_step_kernel(kern)
    hops, H = 4, cfg.hop_size
    x = torch.from_numpy(make_input(B, hops * H, seed + 99, sr)).to(_dev())
    res = []
    for poison in (False, True):
        st = eng.new_state(B)
        outs = []
        for t in range(hops):
            if poison:
                eng.poison_lds()
            outs.append(eng.step(x[:, t * H:(t + 1) * H].contiguous(), st, T=1).clone())
        res.append((torch.cat(outs, 1), st.clone()))
    if kern is not None:
        eng.set_step_kernel("wg8")
    assert bool(torch.isfinite(res[1][0]).all()) and bool(torch.isfinite(res[1][1]).all()), "non-finite output after NaN leftovers in LDS"
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "the result depends on what was in LDS before the launch"


@pytest.mark.parametrize("name,B", [("fe_b", 3), ("fe_t", 2), ("fe_l", 2), ("fe48_b", 2), ("fe_nc", 2), ("fe_tk_b", 2), ("fe_dpt_b", 2), ("bsrnn_xt", 3), ("bsrnn_t", 2),
                                    ("fspen", 2), ("lisennet", 2), ("bsrnn_xt", 2100), ("bsrnn_t", 2064), ("fspen", 1600), ("lisennet", 530)])
def test_lds_leftovers_do_not_matter_chunked_offline_and_stream_batched(name, B):
    """The same for the other launch shapes: a chunked step (T = 3), offline Model.forward (time-batched engine / time-pipelined walk), and the
    stream-batched steps of the large batches (BSRNN from 2048 streams - r6: num_channels = 32 too -, FSPEN from 1536, r6: LiSenNet from 513)."""
    cls = "Model" if name == "fe_nc" else "ONNXModel"
    if name.startswith("bsrnn"):
        m, orc, cfg, sr, seed = _bsrnn(name)
        mo = _bsrnn(name, "Model")[0] if B < 100 else None
    elif name == "fspen":
        m, orc, cfg, sr, seed = _fspen()
        mo = _fspen("Model")[0] if B < 100 else None
    elif name == "lisennet":
        m, orc, cfg, sr, seed = _lisennet()
        mo = _lisennet("Model")[0] if B < 100 else None
    else:
        m, orc, cfg, sr, seed = _model(name, cls)
        mo = m if cls == "Model" else _model(name, "Model")[0]
    H = cfg.hop_size
    if cls == "ONNXModel":
        eng = m.engine
        T = 1 if B > 100 else 3
        x = torch.from_numpy(make_input(B, 2 * T * H, seed + 7, sr)).to(_dev())
        res = []
        for poison in (False, True):
            st = eng.new_state(B)
            outs = []
            for c in range(2):
                if poison:
                    eng.poison_lds()
                outs.append(eng.step(x[:, c * T * H:(c + 1) * T * H].contiguous(), st, T=T).clone())
            res.append((torch.cat(outs, 1), st.clone()))
        assert bool(torch.isfinite(res[1][0]).all())
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "chunked / stream-batched step depends on LDS leftovers"
    if mo is not None:
        xo = torch.from_numpy(make_input(2, 9 * H + 11, seed + 8, sr)).to(_dev())
        w0, s0 = [t.clone() for t in mo(xo)]
        mo.engine.poison_lds()
        w1, s1 = [t.clone() for t in mo(xo)]
        assert bool(torch.isfinite(w1).all()) and torch.equal(w0, w1) and torch.equal(s0, s1), "offline forward depends on LDS leftovers"


@pytest.mark.parametrize("name", ["fe_b", "fe_t", "fe_l", "fe48_b", "fe_nc", "fe_tk_b", "fe_dpt_b", "fe_ln_b", "bsrnn_xt", "bsrnn_t", "fspen", "lisennet"])
def test_uninitialised_work_buffers_do_not_matter(name, monkeypatch):
    """The caller-owned work / output buffers of fe_offline (the mirror gets them from torch.empty): every word the engines read they must have written
    first.  With torch.empty handing out NaN-filled buffers, Model.forward gives the bits of the plain run - on the time-batched engine and on the frame walk."""
    if name.startswith("bsrnn"):
        mo = _bsrnn(name, "Model")[0]
        cfg, sr, seed = mo.engine.cfg, 16000, 11
    elif name == "fspen":
        mo = _fspen("Model")[0]
        cfg, sr, seed = mo.engine.cfg, 16000, 12
    elif name == "lisennet":
        mo = _lisennet("Model")[0]
        cfg, sr, seed = mo.engine.cfg, 16000, 13
    else:
        mo, orc, cfg, sr, seed = _model(name, "Model")
    H = cfg.hop_size
    xo = torch.from_numpy(make_input(3, 21 * H + 5, seed + 3, sr)).to(_dev())
    engines = ["auto"] if name in ("fe_nc",) or name.startswith("bsrnn") or name in ("fspen", "lisennet") else ["auto", "frame_walk"]
    real_empty = torch.empty

    def nan_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t

    for engname in engines:
        if engname != "auto":
            mo.engine.set_offline_engine(engname)
        w0, s0 = [t.clone() for t in mo(xo)]
        monkeypatch.setattr(torch, "empty", nan_empty)
        w1, s1 = [t.clone() for t in mo(xo)]
        monkeypatch.setattr(torch, "empty", real_empty)
        assert bool(torch.isfinite(w1).all()) and bool(torch.isfinite(s1).all()), f"{name} {engname}: non-finite output from NaN-filled work buffers"
        assert torch.equal(w0, w1) and torch.equal(s0, s1), f"{name} {engname}: the result depends on the work buffers' previous content"
    if len(engines) > 1:
        mo.engine.set_offline_engine("auto")
