#!/usr/bin/env python3
"""Timeline of the role-split BSRNN PART 1 kernel (csrc/bsrnn_ov_kernels.hip.h): cycle probes of workgroup 0, per wave.
   FE_BSRNN_OV_PROF=1 python tools/gpu_phases_bsrnn_ov.py [bsrnn_xt] [256]"""
import os
import sys

os.environ["FE_BSRNN_OV_PROF"] = "1"
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import BSRNN_KWARGS  # noqa: E402
from fastenhancer_amd.config import BSRNNConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import bsrnn_default_state_dict  # noqa: E402

SCAN = ["start", "front done (STFT, compress, band split)", "layer 0 scan starts (projections ready)", "layer 0: 23 steps done", "layer 0 scan done",
        "layer 1 scan starts", "layer 1 scan done", "last layer scan starts", "last layer scan done", None, "all waves joined", "hand-over stored"]
HELP = ["start", "front done (+ h half of layer 0's gates)", "layer 0 time part done (tiles A, B)", "l=0: fc_freq / next time weights loaded, h half done",
        "l=0: both scans past step 22", "l=0: tile A chain done", "l=0: both scans done", "l=0: tile B chain done (projections of layer 1 ready)",
        "last layer: both scans done", "last layer: tile B fc_freq done", "all waves joined", "hand-over stored"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bsrnn_xt"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg = BSRNNConfig.from_model_kwargs(**kw)
    eng = Engine(cfg, dev)
    eng.load_state_dict(bsrnn_default_state_dict(cfg, torch.Generator().manual_seed(1)))
    x = (0.1 * torch.randn(B, cfg.hop_size, device=dev)).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=1)
    torch.cuda.synchronize()
    c = clk.cpu().numpy().astype("int64")
    t0 = min(int(c[16 * w]) for w in range(4))
    print(f"{name} B={B}: role-split PART 1, workgroup 0, cycles since the first wave's start; total {int(c[11]) - t0}")
    for w in range(4):
        names = SCAN if w < 2 else HELP
        print(f" wave {w} ({'scan ' + ('fwd' if w == 0 else 'bwd') if w < 2 else 'matrix-core helper ' + str(w - 2)})")
        prev = None
        for i, nm in enumerate(names):
            if nm is None or c[16 * w + i] == 0:
                continue
            t = int(c[16 * w + i]) - t0
            print(f"   {t:8d}  {'+' + str(t - prev) if prev is not None else '':>8s}  {nm}")
            prev = t
    print(" front (wave 0)")
    prev = 0
    for i, nm in ((12, "frame loaded, windowed"), (13, "FFT done"), (14, "compressed"), (15, "band split done"), (1, "h half of layer 0's gates done, barrier")):
        t = int(c[i]) - t0
        print(f"   {t:8d}  {'+' + str(t - prev):>8s}  {nm}")
        prev = t
    names = ["chain starts", "fc_freq done, x stored", "gate GEMM (x half) done", "gate math done, new h stored", "rendezvous passed", "state stored, fc_time done",
             "projection GEMM done", "projections stored"]
    print(" inside tile B's chain under layer 0's scans (wave 2)")
    prev = None
    for j, nm in enumerate(names):
        t = int(c[(44 if j < 4 else 56) + j]) - t0
        print(f"   {t:8d}  {'+' + str(t - prev) if prev is not None else '':>8s}  {nm}")
        prev = t


if __name__ == "__main__":
    main()
