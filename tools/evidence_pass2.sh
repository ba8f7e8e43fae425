#!/bin/bash
# r5 late evidence pass: rocprofv3 stats + PMC of the workloads whose kernels changed after tools/evidence_pass.sh ran, bench lines with cpu_baseline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ev3; mkdir -p $O; cd $R
prof() { BENCH_ARGS="--streams $3" tools/profile.sh $1 $2 > $O/prof_$1.log 2>&1; cd $R; P=$R/gpurun_out/prof_$1; mkdir -p $O/$1; cp $P/summary.txt $O/$1/summary.txt; find $P -name "*kernel_stats.csv" -exec cp {} $O/$1/kernel_stats.csv \; ; cp $P/stats_bench.json $O/$1/bench_line_under_tracer.json 2>/dev/null; rm -rf $P; }
prof r5c_fe_l fe_l 256
python bench.py > $O/bench_default.log 2>&1
python bench.py --workload fe_l --streams 256 > $O/bench_fe_l.log 2>&1
tools/bench_matrix.sh > $O/bench_matrix.txt 2>&1
grep -h "kernel stats" -A2 $O/*/summary.txt | cut -c1-200; cat $O/bench_matrix.txt
