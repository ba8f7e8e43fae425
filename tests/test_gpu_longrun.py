"""Long runs at full batch inside -m gpu (r6).

The one real bug of round 5 - BSRNN's band split reading two LDS words past the spectrum, NaN on one run in many, in the tree since r2 -
was found by a 400-hop x 256-stream TOOL run (tools/gpu_bsrnn_longrun.py), not by the suite, whose goldens hold at most 200 hops of two
streams.  These are that run and its FastEnhancer twins as tests: the timed batch sizes of the BASELINE configs, 320 per-hop launches
with the state carried (scripts/test_onnx.py:44-50 drives the reference's step the same way), every stream's every hop checked finite on
the device, sampled streams against the golden-pinned oracle in windows of 50 hops (an error that grows with the hop count shows as a
later window failing), and the whole run TWICE - each time on a fresh handle, from a fresh state, with NaN in every CU's LDS first
(fe_debug_poison_lds) - bit for bit the same.
Reference semantics: scripts/export_onnx.py:48-58 (the step), models/bsrnn/model.py:367-390, models/fastenhancer/default/model.py:620-710."""
import importlib

import numpy as np
import pytest
import torch

from common import BSRNN_KWARGS, MODEL_KWARGS, MODEL_MODULE, build_bsrnn_oracle, build_oracle, rms
from oracle.weightgen import make_input

pytestmark = pytest.mark.gpu

HOPS = 320
WINDOW = 50
# relative rms per window: the family bounds of tests/test_gpu_parity.py (what exact-fp32 kernels deliver is ~1e-6 with no drift)
BOUND = {"fastenhancer": 2e-5, "bsrnn": 1e-5}


def _fresh_model(name):
    dev = torch.device("cuda:0")
    if name.startswith("bsrnn"):
        kw, sr, seed = BSRNN_KWARGS[name]
        cfg, sd, fused, orc = build_bsrnn_oracle(name)
        mod = importlib.import_module("fastenhancer_amd.models.bsrnn.model")
    else:
        kw, sr, seed = MODEL_KWARGS[name]
        cfg, sd, fused, orc = build_oracle(name)
        mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE[name]}.model")
    m = mod.ONNXModel(**kw).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, fused, sr, seed


def _oracle_run(name, orc, cfg, fused, x):
    """the sampled streams through the oracle, hop by hop (C port for the default FastEnhancer shapes: 300 hops of numpy cost 15 s)"""
    n, H = x.shape[0], cfg.hop_size
    hops = x.shape[1] // H
    if not name.startswith("bsrnn"):
        from oracle.c_oracle import COracle
        co = COracle(cfg, fused, threads=min(4, n))
        cs = np.zeros((n, cfg.n_fft - H), np.float32)
        ci = cs.copy()
        h = np.zeros((cfg.rf_blocks, n * cfg.rf_freq, cfg.rf_channels), np.float32)
        return np.concatenate([co.step(np.ascontiguousarray(x[:, t * H:(t + 1) * H]), cs, ci, h).copy() for t in range(hops)], 1)
    caches = orc.initialize_cache(n)
    outs = []
    for t in range(hops):
        o, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches)
        outs.append(o)
    return np.concatenate(outs, 1)


@pytest.mark.parametrize("name,B,kernel", [("bsrnn_xt", 256, "wg8"), ("fe_b", 256, "wg8"), ("fe_b", 256, "waves4"), ("fe_l", 64, "wg8")])
def test_long_run_at_full_batch_is_finite_exact_and_reproducible_from_poisoned_lds(name, B, kernel):
    dev = torch.device("cuda:0")
    runs, launched = [], None
    ref = sel = None
    for rep in range(2):
        m, orc, cfg, fused, sr, seed = _fresh_model(name)       # a fresh handle: packed weights, scratch, counters
        eng, H = m.engine, cfg.hop_size
        eng.set_step_kernel(kernel)
        x = make_input(B, HOPS * H, seed + 4242, sr)
        if ref is None:
            sel = sorted({0, B // 3, B - 1})
            ref = _oracle_run(name, orc, cfg, fused, x[sel])
        xd = torch.from_numpy(x).to(dev)
        eng.poison_lds()
        state = eng.new_state(B)
        out = torch.empty(B, HOPS * H, device=dev)
        bad = torch.zeros((), dtype=torch.int64, device=dev)       # hops with a non-finite sample in ANY stream (device side: no sync per hop)
        for t in range(HOPS):
            o = out[:, t * H:(t + 1) * H]
            eng.step(xd[:, t * H:(t + 1) * H], state, o, T=1)
            bad += (~torch.isfinite(o).all()).to(torch.int64)
            if t % WINDOW == WINDOW - 1:
                bad += (~torch.isfinite(state).all()).to(torch.int64)
        launched = eng.last_step_kernel()
        assert int(bad) == 0, f"{name} B={B} {kernel} run {rep}: {int(bad)} hops / state checks with non-finite values"
        got = out.cpu().numpy()
        for w0 in range(0, HOPS, WINDOW):
            w1 = min(HOPS, w0 + WINDOW)
            g, r = got[sel][:, w0 * H:w1 * H], ref[:, w0 * H:w1 * H]
            rel = rms(g - r) / max(rms(r), 1e-3)
            assert rel < BOUND["bsrnn" if name.startswith("bsrnn") else "fastenhancer"], f"{name} B={B} {kernel} run {rep}, hops {w0}-{w1}: rel rms {rel:.3e}"
        runs.append((got, state.cpu().numpy()))
    # the launch shape of the timed configuration, not a fallback
    want = {"bsrnn_xt": "bsrnn_ov_kernel + ", "fe_b": "fe_frame8_kernel [shape B]" if kernel == "wg8" else "fe_frame_kernel<per-hop> [shape B]", "fe_l": "fe_frame_kernel<per-hop> [shape L]"}[name]
    assert launched.startswith(want), launched
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]), f"{name} B={B} {kernel}: two runs from fresh, poisoned state differ"


def _fresh_lisennet():
    from common import LISENNET_KWARGS, build_lisennet_oracle
    kw, sr, seed = LISENNET_KWARGS
    cfg, sd, _, orc = build_lisennet_oracle()
    mod = importlib.import_module("fastenhancer_amd.models.lisennet.model")
    m = mod.ONNXModel(**kw).to(torch.device("cuda:0")).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m, orc, cfg, None, sr, seed


@pytest.mark.parametrize("name,B,hops,want", [("lisennet", 601, 240, "lisennet_sb_kernel"), ("bsrnn_t", 2093, 64, "bsrnn_sb_layers_kernel"), ("bsrnn_s", 2829, 16, "bsrnn_sb64_layers_kernel")])
def test_stream_batched_steps_long_run_from_poisoned_lds(name, B, hops, want):
    """r6: the stream-batched steps built this round (LiSenNet from conv_1 to the mask head - all of its caches but the phase travel through LDS staging and a carry in
    global memory; the BSRNN layers for num_channels = 32 / 64) over many hops with the state carried: every hop finite, sampled streams (first tile, a middle tile,
    the partly filled last tile) against the oracle in windows, the final state's sampled rows against the oracle's caches, and the whole run twice from a fresh handle with
    NaN in every CU's LDS first - bit for bit the same.  Reference: scripts/export_onnx.py:48-58, models/lisennet/model.py:398-474, models/bsrnn/model.py:367-390."""
    dev = torch.device("cuda:0")
    runs = []
    ref = sel = ref_caches = None
    window = max(4, hops // 4)
    for rep in range(2):
        m, orc, cfg, fused, sr, seed = _fresh_lisennet() if name == "lisennet" else _fresh_model(name)
        eng, H = m.engine, cfg.hop_size
        x = make_input(B, hops * H, seed + 777, sr)
        if ref is None:
            sel = sorted({0, 17, B // 2, B - 14, B - 1})
            caches = orc.initialize_cache(len(sel))
            outs = []
            for t in range(hops):
                o, *caches = orc.step(x[sel, t * H:(t + 1) * H], *caches)
                outs.append(o)
            ref, ref_caches = np.concatenate(outs, 1), caches
        xd = torch.from_numpy(x).to(dev)
        eng.poison_lds()
        state = eng.new_state(B)
        out = torch.empty(B, hops * H, device=dev)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for t in range(hops):
            o = out[:, t * H:(t + 1) * H]
            eng.step(xd[:, t * H:(t + 1) * H], state, o, T=1)
            bad += (~torch.isfinite(o).all()).to(torch.int64)
        bad += (~torch.isfinite(state).all()).to(torch.int64)
        assert want in eng.last_step_kernel(), eng.last_step_kernel()
        assert int(bad) == 0, f"{name} B={B} run {rep}: {int(bad)} hops / state checks with non-finite values"
        got = out.cpu().numpy()
        bound = 1e-5 if name.startswith("bsrnn") else 5e-6 * 4        # (LiSenNet: the family bound of tests/test_gpu_parity.py x the drift allowance of a 240-hop run: observed 4e-7)
        for w0 in range(0, hops, window):
            w1 = min(hops, w0 + window)
            g, r = got[sel][:, w0 * H:w1 * H], ref[:, w0 * H:w1 * H]
            rel = rms(g - r) / max(rms(r), 1e-3)
            assert rel < bound, f"{name} B={B} run {rep}, hops {w0}-{w1}: rel rms {rel:.3e}"
        for a_, b_ in zip(eng.split_state(state, B), ref_caches):
            g = a_.reshape(B, -1)[sel].reshape(b_.shape).cpu().numpy()
            rel = rms(g - b_) / max(rms(b_), 1e-3)
            assert rel < 10 * bound, f"{name} B={B} run {rep}: final cache {tuple(b_.shape)} rel rms {rel:.3e}"
        runs.append((got, state.cpu().numpy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]), f"{name} B={B}: two runs from fresh, poisoned state differ"
