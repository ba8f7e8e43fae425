#!/usr/bin/env python3
"""Observed HIP-vs-reference errors on the golden vectors (streaming wav->wav), one line per shape."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_parity as T  # noqa: E402
from common import load_golden, make_input  # noqa: E402
from fastenhancer_amd.streaming import StreamingModel  # noqa: E402


def main():
    for name in ("fe_t", "fe_b", "fe_m", "fe_l", "fe48_b"):
        g = load_golden(name)
        m, orc, cfg, sr, seed = T._model(name)
        M = StreamingModel(m)
        B, hops, H = int(g["B"]), int(g["hops"]), cfg.hop_size
        x = torch.from_numpy(make_input(B, hops * H, seed + 1000, sr)).to("cuda:0")
        caches = M.initialize_cache(x)
        outs = []
        for t in range(hops):
            wav_out, *caches = M(x[:, t * H:(t + 1) * H], *caches)
            outs.append(wav_out.cpu().numpy())
        got, ref = np.stack(outs, 0), g["stream_wav_out"]
        err = float(np.sqrt(np.mean((got - ref) ** 2)))
        r = float(np.sqrt(np.mean(ref ** 2)))
        print(f"{name:7s} streaming wav_out vs reference golden: rms err {err:.3e}  (ref rms {r:.3e}, relative {err / r:.2e})")


if __name__ == "__main__":
    main()
