#!/usr/bin/env python3
"""Stress of the time-pipelined offline launches (counters / rings / agent-scope hand-offs): every family, several batch sizes, N
repetitions each - every repetition must reproduce the first bit for bit (the hand-off order does not change the arithmetic) and agree
with the serial walk.  usage: tools/gpu_pipeline_stress.py [repetitions]"""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import (BSRNN_KWARGS, FSPEN_KWARGS, LISENNET_KWARGS, MODEL_KWARGS, MODEL_MODULE, build_bsrnn_oracle, build_fspen_oracle,  # noqa: E402
                    build_lisennet_oracle, build_oracle)
from oracle.weightgen import make_input  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
bad = 0
for name in ("fe_b", "fe_t", "fe_ln_b", "fe_tk_b", "fe_dpt_b", "fe_dprnn_b", "bsrnn_xt", "bsrnn_t", "fspen", "lisennet"):
    if name == "fspen":
        kw, sr, seed = FSPEN_KWARGS
        cfg, sd, _, _ = build_fspen_oracle()
        mod = importlib.import_module("fastenhancer_amd.models.fspen.model")
    elif name == "lisennet":
        kw, sr, seed = LISENNET_KWARGS
        cfg, sd, _, _ = build_lisennet_oracle()
        mod = importlib.import_module("fastenhancer_amd.models.lisennet.model")
    elif name.startswith("bsrnn"):
        kw, sr, seed = BSRNN_KWARGS[name]
        cfg, sd, _, _ = build_bsrnn_oracle(name)
        mod = importlib.import_module("fastenhancer_amd.models.bsrnn.model")
    else:
        kw, sr, seed = MODEL_KWARGS[name]
        cfg, sd, _, _ = build_oracle(name)
        mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE[name]}.model")
    m = mod.Model(**kw).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    eng = m.engine
    if name in ("fe_b", "fe_t"):
        eng.set_offline_engine("frame_walk")
    for B in (1, 3, 7):
        xn = make_input(B, 97 * cfg.hop_size + 13, 31 + B, sr)
        if name == "lisennet":
            xn[:, :1024] = 0.0           # (frame 0 of its offline path is ill-conditioned on a non-silent start: tests/test_gpu_parity.py)
        x = torch.from_numpy(xn).to(dev)
        eng.set_time_pipeline(0)
        w_ser = m(x)[0].clone()
        for width in (-1, 6):
            eng.set_time_pipeline(width)
            w0 = m(x)[0].clone()
            err = float((w0 - w_ser).abs().max())
            same = all(torch.equal(m(x)[0], w0) for _ in range(N))
            ok = same and err <= 2e-5 * max(1.0, float(w_ser.abs().max())) and bool(torch.isfinite(w0).all())
            bad += not ok
            print(f"{name:11s} B={B} width {width:2d}: {N} repetitions identical: {same}   max |diff| to the serial walk {err:.1e}   {'ok' if ok else 'FAILED'}", flush=True)
print("ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
