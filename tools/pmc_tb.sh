#!/bin/bash
# rocprofv3 PMC passes (each its own run, only --kernel-trace next to --pmc) of fe_offline on the time-batched engine:
#   tools/pmc_tb.sh <tag> <shape> <seconds> <utterances>   -> gpurun_out/<tag>_pmc.txt  (per-kernel, per-dispatch means)
set -u
TAG=$1; SHAPE=${2:-fe_b}; SECS=${3:-4}; UTT=${4:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu_tb_timing.py $SHAPE $SECS $UTT --only-tb"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- $CMD > "$OUT/p$i.txt" 2> "$OUT/p$i.err" || echo "pmc group failed: $grp"
done
python - "$OUT" > "$ROOT/gpurun_out/${TAG}_pmc.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        short = k.split("<")[0].split("::")[-1].split("(")[0]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(f"## {k}")
    for c, v in sorted(acc[k].items()):
        print(f"  {c:34s} n={len(v):4d} mean={sum(v) / len(v):.5g}")
PY
cat "$ROOT/gpurun_out/${TAG}_pmc.txt"
