#!/usr/bin/env python3
"""Per-stage parity table: HIP debug dumps vs the numpy oracle (runs on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from common import MODEL_KWARGS, build_oracle, rms  # noqa: E402
from oracle.weightgen import make_input  # noqa: E402
from fastenhancer_amd.config import FEConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402


def oracle_tap_as_dump(name, tap, B):
    """oracle tap -> [B, rows, cols] in the dump layout"""
    if name in ("spec_in", "spec_out"):
        return tap[:, :, 0, :]
    if name in ("compressed", "mask"):
        return tap[:, :, 0, :]
    if name.startswith("rf_pre") or name.startswith("rf_block"):
        return tap[0]                       # [T=1,B,F2,C2]
    return tap.transpose(0, 2, 1)           # [B,C,F] -> [B,F,C]


def main():
    names = sys.argv[1:] or ["fe_t", "fe_b"]
    dev = torch.device("cuda:0")
    for name in names:
        kw, sr, seed = MODEL_KWARGS[name]
        cfg, sd, fused, orc = build_oracle(name)
        eng = Engine(FEConfig.from_model_kwargs(**kw), dev)
        eng.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        B, hops, H = 3, 6, cfg.hop_size
        x = make_input(B, hops * H, seed + 1000, sr)
        xd = torch.from_numpy(x).to(dev)
        state = eng.new_state(B)
        caches = orc.initialize_cache(B)
        print(f"==== {name}: B={B} hops={hops}")
        for t in range(hops):
            taps = {}
            o_ref, *caches = orc.step(x[:, t * H:(t + 1) * H], *caches, taps=taps)
            o_gpu, dumps = eng.debug_step(xd[:, t * H:(t + 1) * H], state)
            torch.cuda.synchronize()
            if t in (0, 1, hops - 1):
                for sname, r, c, off in eng.debug_stages():
                    if sname not in taps:
                        continue
                    ref = oracle_tap_as_dump(sname, taps[sname], B)
                    got = dumps[sname].cpu().numpy()
                    e = rms(got - ref) / max(rms(ref), 1e-12)
                    flag = "" if e < 1e-4 else "   <<<<<<"
                    print(f"  hop {t} {sname:18s} rel_rms_err {e:.3e}  ref_rms {rms(ref):.3e}{flag}")
            e = rms(o_gpu.cpu().numpy() - o_ref) / max(rms(o_ref), 1e-12)
            print(f"  hop {t} wav_out rel err {e:.3e} (ref rms {rms(o_ref):.3e})")
        st = eng.split_state(state, B)
        for i, (a, b_) in enumerate(zip(st, caches)):
            print(f"  cache {i} rel err {rms(a.cpu().numpy() - b_) / max(rms(b_), 1e-12):.3e}")
        # chunk invariance and timing
        T = 4
        state2 = eng.new_state(B)
        out2 = eng.step(xd[:, :T * H], state2, T=T)
        state3 = eng.new_state(B)
        out3 = torch.cat([eng.step(xd[:, t * H:(t + 1) * H], state3, T=1) for t in range(T)], dim=1)
        print("  chunk T=4 vs 4xT=1 max diff", float((out2 - out3).abs().max()), float((state2 - state3).abs().max()))
        Bb = 256
        xb = torch.from_numpy(make_input(Bb, 64 * H, 5, sr)).to(dev)
        stb = eng.new_state(Bb)
        outb = torch.empty_like(xb)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(64):
                eng.step(xb[:, t * H:(t + 1) * H], stb, outb[:, t * H:(t + 1) * H], T=1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"  B=256 T=1: {dt / 64 * 1e6:.1f} us/step  {Bb * 64 / dt / 1e6:.3f} Mframes/s  "
              f"{Bb * 64 / dt * eng.flops_per_frame / 1e12:.2f} TFLOP/s")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(xb, stb, outb, T=64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"  B=256 T=64 chunk: {dt / 64 * 1e6:.1f} us/frame-step  {Bb * 64 / dt / 1e6:.3f} Mframes/s")


if __name__ == "__main__":
    main()
