#!/usr/bin/env python3
"""Offline enhancement of a 'directory' of mixed-length files: one Model.forward per file (the reference's loop, scripts/test_pytorch.py:28-37)
against ONE ragged batch (fe_offline_ragged) against an equal-length batch of the same size.  usage: tools/gpu_ragged_timing.py [shape] [files]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS  # noqa: E402
from fastenhancer_amd.config import FEConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    kw, sr, _ = MODEL_KWARGS[name]
    cfg = FEConfig.from_model_kwargs(**kw)
    eng = Engine(cfg, dev)
    eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    rng = np.random.default_rng(0)
    lens = sorted(int(v) for v in rng.integers(2 * sr, 4 * sr + 1, size=n))
    xs = [(0.1 * torch.randn(v, device=dev)) for v in lens]
    frames = sum(1 + v // cfg.hop_size for v in lens)
    t_one = timed(lambda: [eng.offline(x[None]) for x in xs], reps=3)
    t_rag = timed(lambda: eng.offline_ragged(xs))
    xe = 0.1 * torch.randn(n, 4 * sr, device=dev)
    t_eq = timed(lambda: eng.offline(xe))
    fl = eng.flops_per_frame
    print(f"{name}: {n} files of 2 .. 4 s ({frames} frames, {sum(lens) / sr:.0f} s of audio), sorted by length")
    print(f"  one Model.forward per file          {t_one:8.2f} ms   ({frames * fl / t_one / 1e9 / 157.3 * 100:5.1f} % of the fp32 matrix peak)")
    print(f"  ONE ragged batch (host tensors built, copies in and out included) {t_rag:8.2f} ms   ({frames * fl / t_rag / 1e9 / 157.3 * 100:5.1f} %)")
    print(f"  equal-length batch {n} x 4 s          {t_eq:8.2f} ms   ({n * (1 + 4 * sr // cfg.hop_size) * fl / t_eq / 1e9 / 157.3 * 100:5.1f} %)  -> ragged / equal-length = {t_rag / t_eq:.2f}")


if __name__ == "__main__":
    main()
