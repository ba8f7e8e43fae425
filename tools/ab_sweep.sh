#!/bin/bash
# A/B of two full side builds (ab/lib_<a>.so, ab/lib_<b>.so) over a list of bench workloads on ONE box:
#   tools/ab_sweep.sh <a> <b> "<workload> <streams>" ...
cd ${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; shift 2
for spec in "$@"; do
  set -- $spec
  for v in $A $B $A $B; do
    FASTENHANCER_HIP_LIB=$PWD/ab/lib_$v.so python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 200 --warmup 20 2>/tmp/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$1', $2, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4))"
  done
done
