#!/usr/bin/env python3
"""Phase clocks of the time-batched engine's segment kernels (workgroup 0, s_memtime), on a probe side build:
  FE_BUILD_TAG=probe FE_SHAPES_DEF=tools/dev_shapes.def FE_EXTRA_DEFS=-DFE_TB_PROBE python -m fastenhancer_amd.build
  FASTENHANCER_HIP_LIB=ab/lib_probe.so python tools/gpu_tb_phases.py <shape> <seconds> <utterances>"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, product_config  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

NAMES = {
    0: {0: "prologue", 1: "STFT + compress", 2: "enc_pre", 12: "rf_pre filterbank", 13: "rf_pre 1x1", 14: "gx block 0", **{3 + l: f"encoder {l}" for l in range(8)}},
    2: {0: "load x, hs", 1: "rnn_fc", 2: "qkv", 3: "attention", 4: "attn_fc", 5: "gx next block"},
    3: {0: "prologue", 1: "load x", 2: "rf_post filterbank", 20: "dec_post 1x1", 21: "transposed conv", 22: "mask + iDFT",
        **{3 + 2 * l: f"decoder {l} 1x1" for l in range(8)}, **{4 + 2 * l: f"decoder {l} k3" for l in range(8)}},
}


def main():
    name, secs, B = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    os.environ["FE_TB_NC"] = os.environ["FE_TB_G"] = os.environ["FE_TB_STREAMS"] = "1"
    kw, sr, _ = MODEL_KWARGS[name]
    dev = torch.device("cuda:0")
    cfg = product_config(name)
    eng = Engine(cfg, dev)
    eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(1)))
    if not cfg.noncausal:
        eng.set_offline_engine("time_batched")
    x = 0.1 * torch.randn(B, int(secs * sr), device=dev)
    for _ in range(3):
        eng.offline(x)
    buf = (ctypes.c_ulonglong * 128)()
    rd = eng.lib.fe_tb_probe_read
    rd.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    rd(eng._h, buf)
    n = 5
    for _ in range(n):
        eng.offline(x)
    rd(eng._h, buf)
    T = 1 + x.shape[1] // cfg.hop_size
    print(f"# {name} B={B} T={T}: cycles of workgroup 0 per call (mean of {n} calls), by phase")
    for st, sname in ((0, "tb_enc_kernel"), (2, "tb_blk_kernel (all blocks)"), (3, "tb_dec_kernel")):
        vals = [buf[st * 32 + i] / n for i in range(32)]
        tot = sum(vals)
        print(f"## {sname}: {tot:12.0f} cycles")
        for i, v in enumerate(vals):
            if v > 0:
                print(f"   {NAMES[st].get(i, str(i)):24s} {v:12.0f}  {v / tot * 100:5.1f} %")


if __name__ == "__main__":
    main()
