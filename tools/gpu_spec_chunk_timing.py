#!/usr/bin/env python3
"""fe_spec_step on chunks of T frames (ONNXModel.forward(spec, *caches)): the frame walk / time pipeline against the time-batched engine.
usage: tools/gpu_spec_chunk_timing.py [shape] [streams] [frames per chunk]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import product_config  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
cfg = product_config(name)
eng = Engine(cfg, dev)
eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
spec = 0.1 * torch.randn(B, cfg.F0 + 1, T, 2, device=dev)
for engine in ("frame_walk", "time_batched", "auto"):
    eng.set_offline_engine(engine)
    h = torch.zeros(eng.model_state_floats(B), device=dev)
    for _ in range(3):
        eng.spec_step(spec, h)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        eng.spec_step(spec, h)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name} {B} streams x {T} frames per chunk, {engine:12s}: {dt * 1e3:8.3f} ms  {B * T / dt / 1e6:7.3f} M frames/s  "
          f"{eng.flops_per_frame * B * T / dt / 157.3e12 * 100:5.1f} % of fp32 peak")
