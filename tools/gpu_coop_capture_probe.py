import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from common import product_config
from fastenhancer_amd.engine import Engine
from fastenhancer_amd.weights import default_state_dict
dev = torch.device("cuda:0")
cfg = product_config("fe_tk_b")      # a variant whose offline call is the time-pipelined (cooperative) frame walk
eng = Engine(cfg, dev); eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
x = 0.1 * torch.randn(2, 16000, device=dev)
w, s = eng.offline(x); torch.cuda.synchronize(); print("eager:", eng.last_step_kernel())
g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(dev)
try:
    with torch.cuda.graph(g, stream=st):
        w2, s2 = eng.offline(x)
    g.replay(); torch.cuda.synchronize()
    print("capture of a cooperative launch: OK, equal:", torch.equal(w, w2))
except Exception as e:
    print("capture of a cooperative launch FAILED:", type(e).__name__, str(e)[:300])
