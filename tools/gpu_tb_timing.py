#!/usr/bin/env python3
"""fe_offline timing: the time-batched (layer-by-layer) engine vs the per-hop kernel walking / pipelining the frames.
usage: tools/gpu_tb_timing.py [shape] [seconds of audio] [utterances] [--only-tb]
Prints frames/s and the fraction of the fp32 matrix peak (157.3 TFLOP/s) the whole call reaches."""
import importlib
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, MODEL_MODULE, product_config  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

PEAK = 157.3e12


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if len(args) > 0 else "fe_b"
    secs = float(args[1]) if len(args) > 1 else 4.0
    B = int(args[2]) if len(args) > 2 else 64
    kw, sr, _ = MODEL_KWARGS[name]
    dev = torch.device("cuda:0")
    cfg = product_config(name)
    eng = Engine(cfg, dev)
    eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    x = 0.1 * torch.randn(B, int(secs * sr), device=dev)
    T = 1 + x.shape[1] // cfg.hop_size
    engines = ["time_batched"] if (cfg.noncausal or "--only-tb" in sys.argv) else ["time_batched", "frame_walk"]
    for e in engines:
        if not cfg.noncausal:
            eng.set_offline_engine(e)
        for _ in range(3):
            eng.offline(x)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            eng.offline(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fl = eng.flops_per_frame * B * T
        print(f"{name} B={B} {secs:.1f} s ({T} frames each, {B * T} frames) {e:12s}: {dt * 1e3:9.3f} ms  {B * T / dt / 1e6:7.3f} M frames/s  "
              f"{fl / dt / 1e12:6.1f} TFLOP/s = {fl / dt / PEAK * 100:5.1f} % of fp32 peak  RTF {dt / secs / B:.6f}")


if __name__ == "__main__":
    main()
