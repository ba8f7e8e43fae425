#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ev2; mkdir -p $O; cd $R
prof() { BENCH_ARGS="--streams $3" tools/profile.sh $1 $2 > $O/prof_$1.log 2>&1; cd $R; P=$R/gpurun_out/prof_$1; mkdir -p $O/$1; cp $P/summary.txt $O/$1/summary.txt; find $P -name "*kernel_stats.csv" -exec cp {} $O/$1/kernel_stats.csv \; ; cp $P/stats_bench.json $O/$1/bench_line_under_tracer.json 2>/dev/null; rm -rf $P; }
prof r5b_fe_l fe_l 256
prof r5b_bsrnn_xt bsrnn_xt 256
python bench.py > $O/bench_default.log 2>&1
python bench.py --workload fe_l --streams 256 > $O/bench_fe_l.log 2>&1
python bench.py --workload bsrnn_xt --streams 256 > $O/bench_bsrnn_xt.log 2>&1
python bench.py --workload fe48_b_h480 --streams 512 > $O/bench_fe48.log 2>&1
grep -h "kernel stats" -A3 $O/*/summary.txt | cut -c1-200; tail -c 300 $O/bench_*.log
