#!/bin/bash
# kernel timeline of fe_offline on the time-batched engine (rocprofv3 --kernel-trace): which launches overlap
#   tools/trace_tb.sh <tag> <shape> <seconds> <utterances>   (plan through FE_TB_NC / FE_TB_G / FE_TB_STREAMS)  -> gpurun_out/<tag>_trace.csv
set -u
TAG=$1; SHAPE=${2:-fe_b}; SECS=${3:-4}; UTT=${4:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/gpu_tb_timing.py $SHAPE $SECS $UTT --only-tb > "$OUT/run.txt" 2> "$OUT/run.err"
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_tb_timeline.py "$F" > "$ROOT/gpurun_out/${TAG}_timeline.txt"
grep -v amdgpu.ids "$OUT/run.txt"
tail -90 "$ROOT/gpurun_out/${TAG}_timeline.txt"
rm -rf "$OUT"
