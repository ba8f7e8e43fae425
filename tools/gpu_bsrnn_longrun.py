import os, sys, numpy as np, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import BSRNN_KWARGS, make_input, build_bsrnn_oracle
import importlib
dev = torch.device("cuda:0")
for name in ("bsrnn_xt", "bsrnn_xxt"):
    kw, sr, seed = BSRNN_KWARGS[name]
    cfg, sd, fused, orc = build_bsrnn_oracle(name)
    mod = importlib.import_module("fastenhancer_amd.models.bsrnn.model")
    m = mod.ONNXModel(**kw).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    eng = m.engine
    B, hops, H = 256, 400, cfg.hop_size
    x = make_input(B, hops * H, 4242, sr)
    xd = torch.from_numpy(x).to(dev)
    st = eng.new_state(B)
    outs = []
    for t in range(hops):
        outs.append(eng.step(xd[:, t * H:(t + 1) * H].contiguous(), st, T=1))
    got = torch.cat(outs, 1).cpu().numpy()
    sel = [0, 100, 255]
    caches = orc.initialize_cache(len(sel)); refs = []
    for t in range(hops):
        o, *caches = orc.step(x[sel][:, t * H:(t + 1) * H], *caches); refs.append(o)
    ref = np.concatenate(refs, 1)
    for lo, hi in ((0, 50), (150, 200), (350, 400)):
        d = got[sel][:, lo * H:hi * H] - ref[:, lo * H:hi * H]
        print(name, f"hops {lo}-{hi}: rms rel err {np.sqrt(np.mean(d**2)) / np.sqrt(np.mean(ref[:, lo*H:hi*H]**2)):.3e}  finite={np.isfinite(got).all()}")
