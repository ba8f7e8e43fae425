#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    print(f"# rocprofv3 summary of {os.path.basename(out)} (bench.py --no-cpu-baseline --steps 200 --warmup 20)")
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("\n## kernel stats (--kernel-trace --stats)")
        for row in csv.DictReader(open(f)):
            name = row["Name"]
            if "frame_kernel" in name or float(row["Percentage"]) > 0.5:
                print(f"{name[:90]:90s} calls={row['Calls']} avg_ns={float(row['AverageNs']):.0f} "
                      f"min_ns={row['MinNs']} max_ns={row['MaxNs']} pct={row['Percentage']}")
    for f in sorted(glob.glob(os.path.join(out, "*.json"))):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            print(f"\n## bench line under {os.path.basename(f)[:-5]}: value={d['value']:.0f} {d['unit']} "
                  f"ms_per_step={d['ms_per_step']:.5f} kernel_ms={d['roofline']['kernel_ms']:.5f} frac={d['roofline']['frac']:.4f}")
        except Exception as e:  # noqa
            pass
    print("\n## PMC counters of the step's kernels (fe:: namespace): per-dispatch mean (sum over XCDs/SEs as reported); a step that runs as"
          "\n## several launches (BSRNN: frame kernel PART 1, bsrnn_mlp_kernel, PART 2) also gets the per-step sum")
    for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "fe::" not in kn or "ola" in kn:
                continue
            short = kn.split("(")[0].replace("void ", "")
            short = short[:40] + ".." + short[-24:] if len(short) > 70 else short
            acc[row["Counter_Name"]][short].append(float(row["Counter_Value"]))
        for k, per in sorted(acc.items()):
            tot = 0.0
            for kn, v in sorted(per.items()):
                v = v[len(v) // 10:]  # drop warm-up dispatches
                m = sum(v) / max(len(v), 1)
                tot += m
                if len(per) > 1:
                    print(f"{k:36s} {kn:70s} n={len(v):4d} mean={m:.4g}")
            print(f"{k:36s} {'per step' if len(per) > 1 else '':70s} mean={tot:.4g}")
    print("\nnotes: FETCH_SIZE/WRITE_SIZE are in KiB-equivalents as rocprofv3 reports them (x1024 = bytes; on gfx950 a wide"
          " coalesced read stream is under-reported 2x, MI355X_MICROARCH.md §HBM).")


if __name__ == "__main__":
    main()
