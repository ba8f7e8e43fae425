#!/bin/bash
# bench one workload over several side builds on one box: tools/ab_libs.sh <workload> "<streams...>" <tag|main>...
w=$1; ss=$2; shift 2
for s in $ss; do for tag in "$@"; do
  if [ $tag = main ]; then unset FASTENHANCER_HIP_LIB; else export FASTENHANCER_HIP_LIB=$PWD/ab/lib_$tag.so; fi
  python bench.py --no-cpu-baseline --workload $w --streams $s --steps 200 --warmup 30 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', '$tag', $s, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4))"
done; done
