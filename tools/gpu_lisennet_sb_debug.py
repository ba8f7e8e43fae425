"""Stage-by-stage error map of the stream-batched LiSenNet step (lisennet_sb_kernels.hip.h) against the oracle: FE_LISENNET_SB=1 python tools/gpu_lisennet_sb_debug.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("FE_LISENNET_SB", "1")
from test_gpu_parity import _lisennet, _dev
from oracle.weightgen import make_input

m, orc, cfg, sr, seed = _lisennet()
eng = m.engine
B, hops, H = int(os.environ.get("B", 3)), 3, cfg.hop_size
x = make_input(B, hops * H, 717, sr)
xd = torch.from_numpy(x).to(_dev())
state = eng.new_state(B)
caches = orc.initialize_cache(B)
names = [s_[0] for s_ in eng.debug_stages()]
for t in range(hops):
    taps = {}
    o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
    o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
    print("kernels:", eng.last_step_kernel() if hasattr(eng, "last_step_kernel") else "?")
    for sname in names:
        tap = taps[sname]
        ref = tap[:, 0] if (sname.endswith(".intra") or sname.endswith(".inter")) else tap[:, :, 0, :]
        got = dumps[sname].cpu().numpy()
        err = np.abs(got - ref)
        print(f"hop {t} {sname:16s} shape {got.shape} max err {err.max():.3e} ref rms {np.sqrt((ref**2).mean()):.3e}")
        if err.max() > 1e-4 and os.environ.get("MAP", "1") == "1":
            e = err.max(axis=0)
            np.set_printoptions(linewidth=250, precision=2, suppress=False)
            print("  per-row max:", e.max(axis=1))
            print("  per-col max:", e.max(axis=0))
    print(f"hop {t} wav_out max err {np.abs(o_gpu.cpu().numpy() - o_ref).max():.3e}")
st = eng.split_state(state, B)
for i, (a_, b_) in enumerate(zip(st, caches)):
    print(f"cache {i} {tuple(a_.shape)} max err {np.abs(a_.cpu().numpy() - b_).max():.3e}")
