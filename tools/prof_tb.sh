#!/bin/bash
# rocprofv3 --kernel-trace --stats of fe_offline on the time-batched engine, through gpurun from the repo root:
#   tools/prof_tb.sh <tag> <shape> <seconds> <utterances>   -> gpurun_out/<tag>_kernel_stats.csv (+ a short summary on stdout)
set -u
TAG=$1; SHAPE=${2:-fe_b}; SECS=${3:-4}; UTT=${4:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python $ROOT/tools/gpu_tb_timing.py $SHAPE $SECS $UTT --only-tb > "$OUT/run.txt" 2> "$OUT/run.err"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cp "$F" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
grep -v amdgpu.ids "$OUT/run.txt"
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:10.1f} pct={float(r['Percentage']):6.2f}")
PY
