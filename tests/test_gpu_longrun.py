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
