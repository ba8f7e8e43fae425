#!/usr/bin/env python3
"""Offline Model.forward (fe_offline) timing: one workgroup walking a stream's frames vs the time-pipelined launch.
usage: tools/gpu_offline_timing.py [shape] [seconds of audio] [streams]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, product_config  # noqa: E402
from fastenhancer_amd.config import FEConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda:0")
    if name.startswith("bsrnn"):       # (BSRNN: the mirror with the oracle's seeded checkpoint, as the tests build it)
        import importlib
        import numpy as np
        from common import BSRNN_KWARGS, build_bsrnn_oracle
        kw, sr, _ = BSRNN_KWARGS[name]
        cfg, sd, _, _ = build_bsrnn_oracle(name)
        m = importlib.import_module("fastenhancer_amd.models.bsrnn.model").Model(**kw).to(dev).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        eng = m.engine
    elif name == "lisennet":
        import importlib
        import numpy as np
        from common import LISENNET_KWARGS, build_lisennet_oracle
        kw, sr, _ = LISENNET_KWARGS
        cfg, sd, _, _ = build_lisennet_oracle()
        m = importlib.import_module("fastenhancer_amd.models.lisennet.model").Model(**kw).to(dev).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        eng = m.engine
    elif name == "fspen":
        import importlib
        import numpy as np
        from common import FSPEN_KWARGS, build_fspen_oracle
        kw, sr, _ = FSPEN_KWARGS
        cfg, sd, _, _ = build_fspen_oracle()
        m = importlib.import_module("fastenhancer_amd.models.fspen.model").Model(**kw).to(dev).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        eng = m.engine
    else:
        kw, sr, _ = MODEL_KWARGS[name]
        cfg = product_config(name)
        eng = Engine(cfg, dev)
        if not cfg.noncausal:
            eng.set_offline_engine("frame_walk")       # (this tool times the frame walk and its time pipeline; tools/gpu_tb_timing.py the time-batched engine)
        eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    x = 0.1 * torch.randn(B, int(secs * sr), device=dev)
    T = 1 + x.shape[1] // cfg.hop_size
    res = {}
    ref = None
    for width in (0, 4, 8, 12, 16, 24, 32, 48, 64, -1):
        eng.set_time_pipeline(width)
        w0 = eng.offline(x)[0]
        if ref is None:
            ref = w0.clone()
        err = float((w0 - ref).abs().max())
        for _ in range(2):
            eng.offline(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.offline(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[width] = dt
        print(f"{name} B={B} {secs:.1f} s ({T} frames): frames in flight {width:2d}: {dt * 1e3:8.3f} ms  "
              f"({B * T / dt / 1e3:8.1f} k frames/s, RTF {dt / secs / B:.5f}, x{res[0] / dt:5.2f} vs serial, max |diff| to serial {err:.1e})")


if __name__ == "__main__":
    main()
