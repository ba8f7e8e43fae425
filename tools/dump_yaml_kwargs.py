#!/usr/bin/env python3
"""Dump {yaml path -> model, model_kwargs, sampling_rate} of every shipped config of the reference to
tests/golden/yaml_kwargs.json (DATA: the yaml values, no code).  Runs only in the authoring container, where the
reference checkout is mounted at /root/reference; the tests read model_kwargs from the JSON instead of restating them.

usage: python tools/dump_yaml_kwargs.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import glob
import json
import os

import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "yaml_kwargs.json"))
    args = ap.parse_args()
    out = {}
    for path in sorted(glob.glob(os.path.join(args.ref, "configs", "**", "*.yaml"), recursive=True)):
        hps = yaml.safe_load(open(path))
        if not isinstance(hps, dict) or "model" not in hps:
            continue
        rel = os.path.relpath(path, args.ref)
        out[rel] = {"model": hps["model"], "model_kwargs": hps.get("model_kwargs", {}),
                    "sampling_rate": hps.get("data", {}).get("sampling_rate")}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"wrote {args.out}: {len(out)} yamls")


if __name__ == "__main__":
    main()
