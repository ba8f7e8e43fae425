#!/bin/bash
# r6 evidence pass (one gpurun call): rocprofv3 --kernel-trace --stats + the PMC groups for BASELINE configs 2-5 (tools/profile.sh), trimmed to the
# summaries (the raw traces of 6 passes x 4 workloads exceed what gpurun merges back).  tools/evidence_pass.sh [full]: + GPU suite, default bench, matrix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ev; mkdir -p $O
cd $R
if [ "${1:-}" = full ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -30) > $O/pytest_gpu.log
  python bench.py > $O/bench_default.log 2>&1
  tools/bench_matrix.sh > $O/bench_matrix.txt 2>&1
fi
prof() {   # tag workload streams
  BENCH_ARGS="--streams $3" tools/profile.sh $1 $2 > $O/prof_$1.log 2>&1
  cd $R
  P=$R/gpurun_out/prof_$1
  mkdir -p $O/$1
  cp $P/summary.txt $O/$1/summary.txt
  find $P -name "*kernel_stats.csv" -exec cp {} $O/$1/kernel_stats.csv \;
  cp $P/stats_bench.json $O/$1/bench_line_under_tracer.json 2>/dev/null
  rm -rf $P
}
prof r6_fe_b fe_b 256
prof r6_fe_l fe_l 256
prof r6_fe48_b_h480 fe48_b_h480 512
prof r6_bsrnn_xt bsrnn_xt 256
du -sh $R/gpurun_out; for t in r6_fe_b r6_fe_l r6_fe48_b_h480 r6_bsrnn_xt; do echo "=== $t"; cat $O/$t/summary.txt | cut -c1-200; done
