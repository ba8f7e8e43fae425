#!/usr/bin/env python3
"""Stage-by-stage check of the time-batched engine (csrc/tb_kernels.hip.h) against the oracle's taps, on the GPU box:
    python tools/gpu_tb_check.py fe_b [B] [frames]
Runs fe_offline with FE_TB_STAGES = 1, 2, ... and compares the intermediate buffers in the work buffer (token stream x, gate
pre-activations gx, GRU outputs hs) with the oracle's activations of the same input."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from common import MODEL_KWARGS, MODEL_MODULE, build_oracle, rms  # noqa: E402
from oracle.fe_oracle import gru_step  # noqa: E402
from oracle.weightgen import make_input  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    kw, sr, seed = MODEL_KWARGS[name]
    cfg, sd, fused, orc = build_oracle(name)
    mod = importlib.import_module(f"fastenhancer_amd.models.{MODEL_MODULE[name]}.model")
    m = mod.Model(**kw).to("cuda:0").eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    eng = m.engine
    if not cfg.noncausal:
        eng.set_offline_engine("time_batched")
    H, N = cfg.hop_size, cfg.n_fft
    x = make_input(B, frames * H + 17, 4242, sr)
    T = 1 + x.shape[1] // H
    NF = B * T
    # oracle activations of the offline forward
    xp = np.pad(x, ((0, 0), (N // 2, N // 2)), mode="reflect")
    fr = np.stack([xp[:, t * H:t * H + N] for t in range(T)], axis=1) * orc.window
    X = np.fft.rfft(fr, axis=2)
    spec = np.stack([X.real, X.imag], axis=-1).astype(np.float32).transpose(0, 2, 1, 3)[:, :-1]
    mag = np.maximum(np.sqrt(spec[..., 0:1] ** 2 + spec[..., 1:2] ** 2), np.float32(1e-5))
    spec = spec * mag ** np.float32(cfg.input_compression - 1.0)
    taps = {}
    orc.model_forward(spec, None, taps)
    wav_ref, spec_ref = orc.offline_forward(x)
    C2, F2, KB, nd = cfg.rf_channels, cfg.rf_freq, cfg.rf_blocks, 2 if cfg.noncausal else 1
    sz = [NF * 2 * cfg.F0, NF * (cfg.n_layers + 1) * cfg.F1 * cfg.channels, NF * F2 * C2, nd * NF * F2 * 3 * C2, NF * F2 * nd * C2, NF * N]
    off = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sz])])
    xd = torch.from_numpy(x).to("cuda:0")

    def run(n):
        os.environ["FE_TB_STAGES"] = str(n)
        wav, sp = m(xd)
        torch.cuda.synchronize()
        w = eng._last_work.cpu().numpy()
        return wav.cpu().numpy(), sp.cpu().numpy(), [w[off[i]:off[i] + sz[i]] for i in range(6)]

    def rep(what, got, ref):
        e, r = rms(got - ref), rms(ref)
        print(f"  {what:28s} rms err {e:.3e}  ref rms {r:.3e}  rel {e / max(r, 1e-12):.2e}  {'OK' if e <= 1e-4 * max(r, 1e-3) else 'MISMATCH'}")

    def tok(a):     # oracle [T, B, F2, C] -> [NF = b * T + t][F2][C]
        return np.ascontiguousarray(a.transpose(1, 0, 2, 3)).reshape(NF, F2, -1)

    print(f"{name}: B = {B}, T = {T}, NF = {NF}")
    _, _, bufs = run(1)
    print("after the encoder segment")
    xc = bufs[0].reshape(NF, 2, cfg.F0)
    rep("compressed spectrum", xc, spec.transpose(0, 2, 3, 1).reshape(NF, 2, cfg.F0))
    rep("x = rf_pre", bufs[2].reshape(NF, C2, F2).transpose(0, 2, 1), tok(taps["rf_pre"]))      # (work buffers: [frame][channel][sub-band])
    w = orc.w
    for d, sfx in enumerate(("", "_reverse")[:nd]):
        p = "rf_block.0.rnn."
        xs = tok(taps["rf_pre"])
        gxr = xs @ w[p + "weight_ih_l0" + sfx].T + w[p + "bias_ih_l0" + sfx]
        gxr[..., :2 * C2] += w[p + "bias_hh_l0" + sfx][:2 * C2]
        rep(f"gx block 0 dir {d}", bufs[3].reshape(nd, NF, 3 * C2, F2)[d].transpose(0, 2, 1), gxr)
    stage = 1
    xin = taps["rf_pre"]
    for k in range(KB):
        stage += 1
        _, _, bufs = run(stage)
        print(f"after the scan of block {k}")
        p = f"rf_block.{k}.rnn."
        xs = xin.reshape(T, B * F2, C2)
        ys = np.zeros((T, B * F2, nd * C2), np.float32)
        for d, sfx in enumerate(("", "_reverse")[:nd]):
            h = np.zeros((B * F2, C2), np.float32)
            for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                h = gru_step(xs[t], h, w[p + "weight_ih_l0" + sfx], w[p + "weight_hh_l0" + sfx], w[p + "bias_ih_l0" + sfx], w[p + "bias_hh_l0" + sfx])
                ys[t, :, d * C2:(d + 1) * C2] = h
        rep("hs", bufs[4].reshape(NF, nd * C2, F2).transpose(0, 2, 1), tok(ys.reshape(T, B, F2, nd * C2)))
        stage += 1
        _, _, bufs = run(stage)
        print(f"after the attention pass of block {k}")
        rep("x", bufs[2].reshape(NF, C2, F2).transpose(0, 2, 1), tok(taps[f"rf_block.{k}"]))
        xin = taps[f"rf_block.{k}"]
    os.environ.pop("FE_TB_STAGES")
    wav, sp, _ = run(1 << 20)
    print("end to end")
    rep("spec_hat", sp, spec_ref)
    rep("wav_hat", wav, wav_ref)


if __name__ == "__main__":
    main()
