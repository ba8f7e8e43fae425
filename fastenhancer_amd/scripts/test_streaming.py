#!/usr/bin/env python3
"""Streaming enhancement of one file, hop by hop, on the HIP path — the counterpart of the reference's
scripts/test_onnx.py (same flags: --audio-path --save-output --n-fft --hop-size --sr) with the ONNX session replaced
by the native step; the model comes from -c <yaml> / -n <name> (+ checkpoint) instead of --onnx-path.

    python -m fastenhancer_amd.scripts.test_streaming -c configs/fastenhancer/b.yaml --checkpoint logs/b/00500.pth \\
        --audio-path noisy.wav --save-output
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from ..streaming import StreamingModel
from .common import build_model, latest_checkpoint, load_hparams, read_wav, write_wav


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-n", "--name", type=str, help="checkpoint directory logs/{name}")
    p.add_argument("-c", "--config", type=str, help="path to the config yaml (default logs/{name}/config.yaml)")
    p.add_argument("--checkpoint", type=str, help="explicit .pth (default: latest in logs/{name})")
    p.add_argument("--audio-path", type=str, required=True)
    p.add_argument("--save-output", action="store_true")
    p.add_argument("--output-path", type=str, default="enhanced_streaming.wav")
    p.add_argument("--n-fft", type=int, default=None, help="checked against the config")
    p.add_argument("--hop-size", type=int, default=None, help="checked against the config")
    p.add_argument("--sr", type=int, default=None, help="checked against the config")
    p.add_argument("--device", type=str, default="cuda:0")
    args = p.parse_args(argv)

    hps = load_hparams(args.config, args.name)
    kw = hps["model_kwargs"]
    sr = hps["data"]["sampling_rate"]
    n_fft, hop = kw["n_fft"], kw["hop_size"]
    for flag, want in (("n_fft", n_fft), ("hop_size", hop), ("sr", sr)):
        got = getattr(args, flag)
        assert got is None or got == want, f"--{flag.replace('_', '-')}={got} does not match the config ({want})"
    ckpt = args.checkpoint or (latest_checkpoint(os.path.join("logs", args.name)) if args.name else None)
    model = build_model(hps, args.device, offline=False, checkpoint=ckpt)
    M = StreamingModel(model)

    print("Preparing input...", end=" ")
    wav = np.clip(read_wav(args.audio_path, sr).reshape(1, -1), -1, 1)
    length = wav.shape[-1]
    wav = np.pad(wav, ((0, 0), (0, n_fft)))                               # pad right (scripts/test_onnx.py:18)
    x = torch.from_numpy(wav).to(args.device)
    cache = M.initialize_cache(x)
    print("ok\nInferencing...")
    out = []
    torch.cuda.synchronize()
    tic = time.perf_counter()
    idx = 0
    for idx in range(0, length + n_fft - hop, hop):                       # scripts/test_onnx.py:44-50
        wav_out, *cache = M(x[:, idx:idx + hop], *cache)
        out.append(wav_out)
    torch.cuda.synchronize()
    toc = time.perf_counter()
    print(f">>> RTF: {(toc - tic) * sr / (idx + hop)}")
    if args.save_output:
        y = torch.cat(out, dim=1)[0].cpu().numpy()
        s = n_fft - hop
        y = np.clip(y[s:s + length], -1.0, 1.0)
        write_wav(args.output_path, sr, y)
        print(f"saved {args.output_path}")


if __name__ == "__main__":
    main()
