import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, importlib
from common import FSPEN_KWARGS, build_fspen_oracle
from oracle.weightgen import make_input
kw, sr, seed = FSPEN_KWARGS
cfg, sd, fused, orc = build_fspen_oracle()
mod = importlib.import_module("fastenhancer_amd.models.fspen.model")
m = mod.ONNXModel(**kw).to("cuda:0").eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
eng = m.engine
B, H = 3, 256
x = make_input(B, 2 * H, 616, sr); xd = torch.from_numpy(x).cuda()
state = eng.new_state(B); caches = orc.initialize_cache(B)
for t in range(2):
    taps = {}
    o_ref, *caches = orc.step(x[:, t*H:(t+1)*H], *caches, taps=taps)
    o_gpu, dumps = eng.debug_step(xd[:, t*H:(t+1)*H], state)
    for name in dumps:
        tap = taps[name]
        ref = tap[:, :, 0, :] if name in ("spec_in","spec_out","compressed","mask") else (tap[0] if name.startswith("dpe.") else tap)
        g = dumps[name].cpu().numpy()
        print(t, name, "err", float(np.sqrt(((g-ref)**2).mean())), "ref", float(np.sqrt((ref**2).mean())))
    print(t, "wav", float(np.abs(o_gpu.cpu().numpy()-o_ref).max()))
