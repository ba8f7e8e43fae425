#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box (through gpurun, from the repo root):  tools/hbm_calib.sh <tag>
#   -> gpurun_out/hbm_calib_<tag>/{fetch,write}/... + summary.txt (bytes per counter unit for each access pattern of tools/micro/hbm_counter_calib.hip)
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_calib_$TAG
mkdir -p "$OUT"
BIN=$ROOT/ab/hbm_counter_calib
[ -x "$BIN" ] || hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/micro/hbm_counter_calib.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- "$BIN" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- "$BIN" > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
BIG, SMALL = 1 << 30, 8 << 20
print("# rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts (tools/micro/hbm_counter_calib.hip), MI355X; counter values as reported")
print("# (the tool's unit: KiB).  factor = known bytes / (counter x 1024): what a reading has to be multiplied by.")
print(f"# {'kernel':14s} {'counter':11s} {'1 GiB pass: reading':>20s} {'factor':>7s}   {'8 MiB passes (cache-resident): reading':>40s} {'factor':>7s}")
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            if r["Counter_Name"] != cname:
                continue
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        big, small = v[:3], v[3:]
        small = small[2:] if len(small) > 4 else small
        kb = ((BIG // 144) * 144) if "piece" in k else BIG
        ks = ((SMALL // 144) * 144) if "piece" in k else SMALL
        mb = sum(big) / max(len(big), 1)
        ms = sum(small) / max(len(small), 1)
        fb = kb / (mb * 1024) if mb > 0 else float("nan")
        fs = ks / (ms * 1024) if ms > 0 else float("nan")
        print(f"  {k:14s} {cname:11s} {mb:20.1f} {fb:7.3f}   {ms:40.1f} {fs:7.3f}")
PY
