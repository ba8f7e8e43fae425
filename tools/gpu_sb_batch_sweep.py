#!/usr/bin/env python3
"""Stream-batched steps at many batch sizes around the sixteen-stream tile edges (thresholds forced to 1): the first streams of every batch must give the bits of the
smallest batch's, every output finite, and the same input row at the first and the last position of a batch the same bits.
   FE_LISENNET_SB=1 FE_BSRNN_SB=1 FE_FSPEN_SB=1 python tools/gpu_sb_batch_sweep.py [lisennet bsrnn_t bsrnn_s bsrnn_xt fspen]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
for k in ("FE_LISENNET_SB", "FE_BSRNN_SB", "FE_FSPEN_SB"):
    os.environ.setdefault(k, "1")
from test_gpu_parity import _lisennet, _bsrnn, _fspen, _dev
from oracle.weightgen import make_input

names = sys.argv[1:] or ["lisennet", "bsrnn_t", "bsrnn_s", "bsrnn_xt", "fspen"]
for name in names:
    m, orc, cfg, sr, seed = _lisennet() if name == "lisennet" else _fspen() if name == "fspen" else _bsrnn(name)
    eng = m.engine
    H, hops = cfg.hop_size, 3
    sizes = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 255, 256, 257, 513, 528, 529, 700]
    xall = torch.from_numpy(make_input(max(sizes), hops * H, seed + 23, sr)).to(_dev())
    ref, bad, kern = None, [], set()
    for B in sizes:
        st = eng.new_state(B)
        o = torch.cat([eng.step(xall[:B, t * H:(t + 1) * H].contiguous(), st, T=1).clone() for t in range(hops)], 1)
        kern.add(eng.last_step_kernel())
        if not bool(torch.isfinite(o).all()) or not bool(torch.isfinite(st).all()):
            bad.append((B, "non-finite"))
        if ref is None:
            ref = o
        n = min(B, ref.shape[0])
        if not torch.equal(o[:n], ref[:n]):
            bad.append((B, "differs from the smallest batch", float((o[:n] - ref[:n]).abs().max())))
        if B > 1:
            x2 = xall[:B].clone(); x2[B - 1] = x2[0]
            st2 = eng.new_state(B)
            o2 = torch.cat([eng.step(x2[:, t * H:(t + 1) * H].contiguous(), st2, T=1).clone() for t in range(hops)], 1)
            if not torch.equal(o2[0], o2[B - 1]):
                bad.append((B, "position dependence"))
    print(name, "->", "OK" if not bad else bad, "|", sorted(kern)[-1][:110])
