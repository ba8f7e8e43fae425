#!/usr/bin/env python3
"""Times the REFERENCE itself (imported from /root/reference, torch CPU eager) on the wav->wav streaming step
(scripts/export_onnx.py:48-58 composition) for the five BASELINE.json configs, B in {1, 256}, threads in {1, all}, and
writes fixtures/ref_cpu_timing.json (numbers + lscpu summary; no reference code).  Authoring container only - the
reference cannot travel to the GPU box.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/time_reference_cpu.py [--ref /root/reference] [--budget-s 4]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

sys.dont_write_bytecode = True

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

from gen_golden import import_reference_model, to_t  # noqa: E402
from oracle.weightgen import make_input  # noqa: E402

CASES = [
    # (name, model module, yaml, overrides)
    ("FastEnhancer_T 16kHz (config 1)", "fastenhancer", "configs/fastenhancer/t.yaml", {}),
    ("FastEnhancer_B 16kHz (config 2)", "fastenhancer", "configs/fastenhancer/b.yaml", {}),
    ("FastEnhancer_L 16kHz (config 3)", "fastenhancer", "configs/fastenhancer/l.yaml", {}),
    ("FastEnhancer_B 48kHz hop 480 (config 4)", "fastenhancer", "configs/fastenhancer_48khz/b.yaml", {"hop_size": 480}),
    ("BSRNN-xt 16kHz (config 5)", "bsrnn", "configs/others/bsrnn_xt.yaml", {}),
]


def build(ref, kind, rel_yaml, over, seed):
    hps = yaml.safe_load(open(os.path.join(ref, rel_yaml)))
    kw = dict(hps["model_kwargs"])
    kw.update(over)
    sr = hps["data"]["sampling_rate"]
    if kind == "fastenhancer":
        from oracle.fe_oracle import FEConfig
        from oracle.weightgen import make_training_state_dict
        mod = import_reference_model(ref, "models/fastenhancer/default/model.py", "ref_fe_model")
        cfg = FEConfig.from_model_kwargs(kw)
        sd = make_training_state_dict(cfg, seed)
        caches = lambda B: [torch.zeros(1, B * cfg.rf_freq, cfg.rf_channels) for _ in range(cfg.rf_blocks)]
    else:
        from oracle import bsrnn_oracle as bo
        mod = import_reference_model(ref, "models/bsrnn/model.py", "ref_bsrnn_model")
        cfg = bo.BSRNNConfig.from_model_kwargs(kw)
        sd = bo.make_training_state_dict(cfg, seed)
        caches = lambda B: [torch.zeros(B * cfg.n_bands, cfg.hidden) for _ in range(2 * cfg.num_layers)]
    m = mod.ONNXModel(**kw).eval()
    m.load_state_dict(to_t(sd), strict=True)
    m.remove_weight_reparameterizations()
    return m, cfg, sr, caches


def time_case(m, cfg, sr, caches, B, threads, budget_s):
    torch.set_num_threads(threads)
    H = cfg.hop_size
    x = torch.from_numpy(make_input(B, 16 * H, 1236, sr))
    with torch.no_grad():
        cs, ci = m.stft.initialize_cache(x)
        cm = caches(B)

        def step(t):
            nonlocal cs, ci, cm
            spec, cs = m.stft(x[:, (t % 16) * H:((t % 16) + 1) * H], cs)
            spec, *cm = m(spec, *cm)
            out, ci = m.stft.inverse(spec, ci)
            return out

        for t in range(2):
            step(t)
        best, hops, t_all = None, 0, time.perf_counter()
        while time.perf_counter() - t_all < budget_s or hops < 3:
            t0 = time.perf_counter()
            step(hops)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            hops += 1
        mean = (time.perf_counter() - t_all) / hops
    return {"B": B, "threads": threads, "hops_timed": hops, "ms_per_step_min": best * 1e3, "ms_per_step_mean": mean * 1e3,
            "frames_per_s": B / mean, "rtf_per_stream": mean * sr / (H * B)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--budget-s", type=float, default=4.0)
    ap.add_argument("--out", default=os.path.join(REPO, "fixtures", "ref_cpu_timing.json"))
    args = ap.parse_args()
    ncpu = len(os.sched_getaffinity(0))
    lscpu = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
    keep = ("Model name", "CPU(s):", "Thread(s) per core", "Core(s) per socket", "Socket(s)", "L2 cache", "L3 cache", "Hypervisor")
    host = {l.split(":")[0].strip(): l.split(":", 1)[1].strip() for l in lscpu.splitlines() if l.split(":")[0].strip() in [k.rstrip(":") for k in keep]}
    res = {"what": "reference ONNXModel (fused), wav->wav streaming step stft -> model -> stft.inverse, torch CPU eager, "
                   "seeded random weights, synthetic input; mean over the timed hops (min also given)",
           "torch": torch.__version__, "host": host, "logical_cpus_available": ncpu, "cases": []}
    for i, (name, kind, rel, over) in enumerate(CASES):
        m, cfg, sr, caches = build(args.ref, kind, rel, over, 300 + i)
        for B in (1, 256):
            for threads in sorted({1, ncpu}):
                r = time_case(m, cfg, sr, caches, B, threads, args.budget_s)
                r["config"] = name
                res["cases"].append(r)
                print(f"{name:42s} B={B:3d} thr={threads}: {r['ms_per_step_mean']:9.2f} ms/step  {r['frames_per_s']:10.0f} frames/s", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
