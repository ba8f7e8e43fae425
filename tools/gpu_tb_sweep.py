#!/usr/bin/env python3
"""Time-batched engine: sweep of the node plan (time chunks x utterance groups x streams) for one shape / batch.
usage: tools/gpu_tb_sweep.py <shape> <seconds> <utterances> [NC,G,NS ...]      (0 = the engine's default)
Every plan's output is compared with the single-node run (the node cut must not change a bit)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, product_config  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

PEAK = 157.3e12


def set_plan(nc, g, ns):
    for k, v in (("FE_TB_NC", nc), ("FE_TB_G", g), ("FE_TB_STREAMS", ns)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)


def main():
    name, secs, B = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    plans = [tuple(int(v) for v in p.split(",")) for p in sys.argv[4:]] or [(1, 1, 1), (0, 0, 0)]
    kw, sr, _ = MODEL_KWARGS[name]
    dev = torch.device("cuda:0")
    cfg = product_config(name)
    eng = Engine(cfg, dev)
    eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    if not cfg.noncausal:
        eng.set_offline_engine("time_batched")
    x = 0.1 * torch.randn(B, int(secs * sr), device=dev)
    T = 1 + x.shape[1] // cfg.hop_size
    set_plan(1, 1, 1)
    ref_w, ref_s = [t.clone() for t in eng.offline(x)]
    for nc, g, ns in plans:
        set_plan(nc, g, ns)
        w, s = eng.offline(x)
        same = bool((w == ref_w).all()) and bool((s == ref_s).all())
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
        print(f"{name} B={B} T={T} plan NC={nc} G={g} NS={ns}: {dt * 1e3:9.3f} ms  {B * T / dt / 1e6:7.3f} M frames/s  "
              f"{fl / dt / PEAK * 100:5.1f} % of fp32 peak  bit-identical to one node: {same}", flush=True)


if __name__ == "__main__":
    main()
