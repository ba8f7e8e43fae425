#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [workload]    -> gpurun_out/prof_<tag>/{stats,pmc_*}/...
# Kernel-trace stats and each PMC group are separate passes (never combined with other trace domains).
set -u
TAG=${1:-r1}
WORKLOAD=${2:-fe_b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 20 --workload $WORKLOAD ${BENCH_ARGS:-}"      # BENCH_ARGS: e.g. "--streams 2048"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo "$grp" | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o p -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc group failed: $grp"
done
python $ROOT/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
