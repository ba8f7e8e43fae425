#!/usr/bin/env python3
"""Phase breakdown of one LiSenNet frame (shader cycles of workgroup 0) via fe_profile_step: tools/gpu_phases_lisennet.py [streams]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import LISENNET_KWARGS  # noqa: E402
from fastenhancer_amd.config import LiSenNetConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import lisennet_default_state_dict  # noqa: E402

NAMES = ["stft + compress + phase features", "encoder (conv_1, 3 x DSConv)", "2 x DPR", "decoder + mask conv", "mask + istft"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    cfg = LiSenNetConfig.from_model_kwargs(**LISENNET_KWARGS[0])
    eng = Engine(cfg, dev)
    eng.load_state_dict(lisennet_default_state_dict(cfg, torch.Generator().manual_seed(1)))
    H = cfg.hop_size
    x = (0.1 * torch.randn(B, H, device=dev)).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=1)
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    tot = c[5] - c[0]
    print(f"lisennet B={B}: frame = {tot} cycles")
    for i in range(5):
        d = c[i + 1] - c[i]
        print(f"  {NAMES[i]:32s} {d:8d} cyc  {100.0 * d / tot:5.1f}%")
    print(f"  DPR block 0: intra norm + input projections {c[6] - c[2]}, recurrence (32 steps x 2 directions) {c[7] - c[6]}, dense {c[8] - c[7]}, "
          f"inter norm + GRU + dense {c[9] - c[8]}, conv_glu norm + fc1 {c[10] - c[9]}, dwconv + fc2 {c[11] - c[10]}")


if __name__ == "__main__":
    main()
