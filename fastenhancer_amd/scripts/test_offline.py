#!/usr/bin/env python3
"""Offline enhancement of a directory of wavs on the HIP path — the counterpart of the reference's
scripts/test_pytorch.py (same flags -n / -i / -o): the latest checkpoint of logs/{name} is loaded, every *.wav of the
input directory is enhanced with Model.forward and written under the output directory.

    python -m fastenhancer_amd.scripts.test_offline -n fastenhancer_b -i noisy_dir -o enhanced/dns
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import torch

from .common import build_model, latest_checkpoint, load_hparams, read_wav, write_wav


def main(argv=None):
    p = argparse.ArgumentParser("test model")
    p.add_argument("-n", "--name", type=str, required=True, help="The latest checkpoint in logs/{name} will be loaded.")
    p.add_argument("-c", "--config", type=str, default=None, help="default: logs/{name}/config.yaml")
    p.add_argument("--checkpoint", type=str, default=None)
    p.add_argument("-i", "--input-dir", type=str, required=True)
    p.add_argument("-o", "--output-dir", type=str, default="enhanced/dns")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--batch", type=int, default=64,
                   help="files per Model.forward call: the directory is sorted by length and pushed through in ragged batches of this many "
                        "(fe_offline_ragged; on the default / noncausal FastEnhancer models every file's samples are bit-identical to its own call's); "
                        "1 = file by file like the reference")
    args = p.parse_args(argv)

    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    hps = load_hparams(args.config, args.name)
    sr = hps["data"]["sampling_rate"]
    ckpt = args.checkpoint or latest_checkpoint(os.path.join("logs", args.name))
    if ckpt is None:
        raise FileNotFoundError(f"no checkpoint [0-9]{{5,}}.pth under logs/{args.name}")
    model = build_model(hps, args.device, offline=True, checkpoint=ckpt)
    files = sorted(Path(args.input_dir).glob("*.wav"))
    if args.batch <= 1:
        for path in files:
            noisy = torch.from_numpy(read_wav(str(path), sr)).float().to(args.device).unsqueeze(0)
            with torch.no_grad():
                enhanced, _ = model(noisy)                                      # return: wav, spec
            write_wav(str(out_dir / path.name), sr, enhanced.squeeze().cpu().numpy())
    else:
        # sorted by file size (a proxy for the length that needs no decoding), so that a batch is laid out for little more than its own
        # frames; a batch's files are read when its turn comes - host memory holds one batch, not the directory
        files_by_size = sorted(files, key=lambda path: (path.stat().st_size, path.name))
        for i in range(0, len(files_by_size), args.batch):
            chunk = files_by_size[i:i + args.batch]
            with torch.no_grad():
                enhanced, _ = model([torch.from_numpy(read_wav(str(path), sr)).float() for path in chunk])
            for path, e in zip(chunk, enhanced):
                write_wav(str(out_dir / path.name), sr, e.cpu().numpy())
    print(f"enhanced {len(files)} file(s) -> {out_dir}")


if __name__ == "__main__":
    main()
