#!/usr/bin/env python3
"""Phase breakdown of the stream-batched LiSenNet middle (lisennet_sb_kernel, shader cycles of workgroup 0) via fe_profile_step, and the
three launches' times from HIP events: tools/gpu_phases_lisennet_sb.py [streams]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("FE_LISENNET_SB", "1")
from test_gpu_parity import _lisennet, _dev  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    m, orc, cfg, sr, seed = _lisennet()
    eng = m.engine
    H = cfg.hop_size
    x = (0.1 * torch.randn(B, H, device=_dev())).contiguous()
    st = eng.new_state(B)
    for _ in range(3):
        clk = eng.profile_step(x, st, T=1)
    torch.cuda.synchronize()
    q = clk.cpu().numpy()[32:]
    print(f"lisennet B={B}: lisennet_sb_kernel, workgroup 0 = {q[25] - q[0]} cycles")
    print(f"  halos + regrouping + conv_1 {q[1] - q[0]}, conv_2 {q[24] - q[1]}, conv_3 {q[2] - q[24]}, conv_4 {q[3] - q[2]}")
    print(f"    prologue: halos + four tensors' loads + features / cached x1 -> LDS {q[26] - q[0]}, X1P out + conv_1 + LayerNorm + x1 out {q[27] - q[26]}, cached x2 -> LDS {q[28] - q[27]}, "
          f"x1 cache out + X2P out {q[29] - q[28]}, cached x3 / u3 -> LDS {q[30] - q[29]}, X3P / U3P out {q[1] - q[30]}")
    for b in range(2):
        o = 4 + 6 * b
        prev = q[3] if b == 0 else q[o - 3]
        print(f"  block {b}: intra_norm {q[o] - prev}, intra GRU (32 steps x 2 directions) {q[o + 1] - q[o]}, dense + inter_norm + inter GRU + dense {q[o + 2] - q[o + 1]}, "
              f"conv_glu {q[o + 3] - q[o + 2]}")
    print(f"  block 0 conv_glu: statistics + affine {q[20] - q[6]}, fc1 edges + cached frames + edge columns + barrier {q[21] - q[20]}, pass 0 {q[22] - q[21]}, pass 1 {q[23] - q[22]}, fc2 + barrier {q[7] - q[23]}")
    print(f"  block output -> carry {q[16] - q[13]}, up1 {q[17] - q[16]}, up2 {q[18] - q[17]}, up3 {q[19] - q[18]}, up3 cache out + mask_conv {q[25] - q[19]}")
    # wall time of the step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(5):
        eng.step(x, st, T=1)
    ev[0].record()
    for _ in range(20):
        eng.step(x, st, T=1)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"  step: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us  [{eng.last_step_kernel() if hasattr(eng, 'last_step_kernel') else ''}]")


if __name__ == "__main__":
    main()
