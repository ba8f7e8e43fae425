#!/usr/bin/env python3
"""Phase breakdown of one FSPEN frame (shader cycles of workgroup 0) via fe_profile_step: tools/gpu_phases_fspen.py [streams]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import FSPEN_KWARGS  # noqa: E402
from fastenhancer_amd.config import FSPENConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import fspen_default_state_dict  # noqa: E402

NAMES = ["stft + compress", "sub-band / full-band encoders", "feature merge", "3 x DPE", "feature split", "sub-band / full-band decoders",
         "masks + istft"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    cfg = FSPENConfig.from_model_kwargs(**FSPEN_KWARGS[0])
    eng = Engine(cfg, dev)
    eng.load_state_dict(fspen_default_state_dict(cfg, torch.Generator().manual_seed(1)))
    H = cfg.hop_size
    x = (0.1 * torch.randn(B, H, device=dev)).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=1)
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    if c[32] != 0:      # the stream-batched DPE kernel ran (fspen_sb_kernels.hip.h): its counters
        q = c[32:]
        print(f"fspen B={B}: stream-batched kernel = {q[14] - q[0]} cycles; post + feature merge {q[1] - q[0]}")
        print(f"  enc_out[1] -> LDS {q[15] - q[0]}, fullband_encoder.2 {q[16] - q[15]}, post {q[17] - q[16]}, merge Linear {q[18] - q[17]}, merge 1x1 {q[1] - q[18]}")
        for b in range(3):
            o = 2 + 4 * b
            print(f"  block {b}: weight / state requests {q[o] - (q[1] if b == 0 else q[o - 1])}, recurrence {q[o + 1] - q[o]}, "
                  f"intra_fc + LayerNorm {q[o + 2] - q[o + 1]}, inter GRUs + fc {q[o + 3] - q[o + 2]}")
        print(f"  feature split + decoder 1x1 {q[14] - q[13]}")
        return
    tot = c[7] - c[0]
    print(f"fspen B={B}: frame = {tot} cycles")
    for i in range(7):
        d = c[i + 1] - c[i]
        print(f"  {NAMES[i]:32s} {d:8d} cyc  {100.0 * d / tot:5.1f}%")
    print(f"  DPE block 0: input projections {c[8] - c[3]}, recurrence (32 steps x 2 directions) {c[9] - c[8]}, "
          f"intra_fc + LayerNorm {c[10] - c[9]}, inter GRUs + fc {c[11] - c[10]}")


if __name__ == "__main__":
    main()
