import os, sys, numpy as np, torch
REPO = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import BSRNN_KWARGS, make_input
from fastenhancer_amd.config import BSRNNConfig
from fastenhancer_amd.engine import Engine
from fastenhancer_amd.weights import bsrnn_default_state_dict
dev = torch.device("cuda:0")
kw, sr, seed = BSRNN_KWARGS["bsrnn_xt"]
cfg = BSRNNConfig.from_model_kwargs(**kw)
eng = Engine(cfg, dev)
eng.load_state_dict(bsrnn_default_state_dict(cfg, torch.Generator().manual_seed(1)))
H = cfg.hop_size
for B in (3, 64, 256):
    x = torch.from_numpy(make_input(B, 8 * H, 31, sr)).to(dev)
    runs = []
    for rep in range(4):
        st = eng.new_state(B)
        outs = []
        for t in range(8):
            outs.append(eng.step(x[:, t * H:(t + 1) * H].contiguous(), st, T=1).clone())
        torch.cuda.synchronize()
        runs.append((torch.stack(outs, 0).cpu().numpy(), [c.cpu().numpy().copy() for c in eng.split_state(st, B)]))
    for rep in range(1, 4):
        dw = np.abs(runs[rep][0] - runs[0][0])
        msg = f"B={B} rep {rep}: wav max diff {dw.max():.3e}"
        if dw.max() > 0:
            hop = int(np.argmax(dw.reshape(8, -1).max(1) > 0)); msg += f" first hop {hop} streams {sorted(set(np.nonzero(dw[hop].reshape(B,-1).max(1))[0].tolist()))[:8]}"
        for i, (a_, b_) in enumerate(zip(runs[rep][1], runs[0][1])):
            d = np.abs(a_ - b_)
            if d.max() > 0:
                if i >= 2:
                    dd = d.reshape(B, 31, -1)
                    bands = sorted(set(np.nonzero(dd.max(axis=(0, 2)))[0].tolist()))
                    units = sorted(set(np.nonzero(dd.max(axis=(0, 1)))[0].tolist()))
                    msg += f"\n   cache {i} (layer {(i-2)//2} {'h' if (i-2)%2==0 else 'c'}) max {d.max():.2e} bands {bands} units {units[:6]}..{units[-3:]}"
                else:
                    msg += f"\n   cache {i} max {d.max():.2e}"
        print(msg)
