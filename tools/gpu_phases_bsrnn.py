#!/usr/bin/env python3
"""Phase breakdown of one BSRNN frame (shader cycles of workgroup 0) via fe_profile_step."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import BSRNN_KWARGS  # noqa: E402
from fastenhancer_amd.config import BSRNNConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import bsrnn_default_state_dict  # noqa: E402

NAMES = ["stft+compress", "band split", "L0 time-LSTM gates", "L0 fc_time", "L0 band-LSTM input proj", "L0 band recurrence",
         "L0 fc_freq", "layers 1..", "mask MLP layer 1", "mask MLP layer 2 + mask", "istft"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bsrnn_xt"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg = BSRNNConfig.from_model_kwargs(**kw)
    eng = Engine(cfg, dev)
    eng.load_state_dict(bsrnn_default_state_dict(cfg, torch.Generator().manual_seed(1)))
    H = cfg.hop_size
    x = (0.1 * torch.randn(B, H, device=dev)).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=1)
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    tot = c[11] - c[0]
    print(f"{name} B={B}: frame = {tot} cycles")
    for i in range(11):
        d = c[i + 1] - c[i]
        print(f"  {NAMES[i]:28s} {d:8d} cyc  {100.0 * d / tot:5.1f}%")


if __name__ == "__main__":
    main()
