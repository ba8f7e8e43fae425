#!/bin/bash
# A/B of ab/lib_old.so vs ab/lib_new.so on ONE box through FASTENHANCER_HIP_LIB (nothing in-tree is touched):
#   tools/ab_bench2.sh <workload> [streams] [reps] [tag of the candidate: ab/lib_<tag>.so, default new]
cd ${GRAFT_REPO_ROOT:-.}
W=${1:-fe_b}; S=${2:-256}; R=${3:-3}; NEW=${4:-new}
for rep in $(seq $R); do
  for v in old $NEW; do
    FASTENHANCER_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --no-cpu-baseline --workload $W --streams $S --steps 500 --warmup 50 2>/tmp/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$W', $S, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4))"
  done
done
