#!/usr/bin/env python3
"""Phase breakdown of one frame (shader cycles of workgroup 0) via fe_profile_step."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, product_config  # noqa: E402
from fastenhancer_amd.config import FEConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

# interval i = probe i -> probe i + 1
NAMES = ["frame load + window", "forward DFT", "compress", "enc_pre", "encoder", "rf_pre", "RNNFormer blocks", "rf_post",
         "decoder", "dec_post + transposed conv", "mask + un-compress", "inverse DFT", "overlap-add + store"]
BLK = ["-", "gru+gates", "(none)", "fc1", "qkv", "attention", "fc2"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    cfg = product_config(name)     # (the variants' yamls have their own keys)
    eng = Engine(cfg, dev)
    eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    H = cfg.hop_size
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 4     # 1: the per-hop kernel (needs a build with FE_EXTRA_DEFS=-DFE_PROBE_HOT)
    x = (0.1 * torch.randn(B, T * H, device=dev)).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=T)
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    tot = c[13] - c[0]
    if c[62] and c[63]:
        print(f"kernel entry -> first frame {c[0] - c[62]} cycles, last probe -> kernel end {c[63] - c[13]} cycles, entry -> end {c[63] - c[62]}")
    print(f"{name} B={B}: frame = {tot} cycles")
    for i in range(13):
        d = c[i + 1] - c[i]
        print(f"  {NAMES[i]:28s} {d:8d} cyc  {100.0 * d / tot:5.1f}%")
    print("  enc layer 0: gemm %d, epilogue %d, barrier %d" % (c[41]-c[40], c[42]-c[41], c[43]-c[42]))
    print("  block 0 GRU: setup %d, gemm %d, epilogue %d, barrier %d" % (c[45]-c[20], c[46]-c[45], c[47]-c[46], c[21]-c[47]))
    if c[48] and c[49]:
        print("  block 0 GRU (flat gates): store %d, barrier %d, gate math %d" % (c[48]-c[46], c[49]-c[48], c[47]-c[49]))
    if c[50] and c[52]:
        print("  block 0 qkv: setup %d, gemm %d, store %d, barrier %d" % (c[50]-c[23], c[51]-c[50], c[52]-c[51], c[24]-c[52]))
    print("  block 0 detail:")
    prev = c[20]
    for i, nm in enumerate(BLK[1:], start=21):
        print(f"    {nm:12s} {c[i] - prev:8d} cyc")
        prev = c[i]


if __name__ == "__main__":
    main()
