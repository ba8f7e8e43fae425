#!/bin/bash
# bench one workload at several batch sizes under two settings of an environment switch, same box:
#   tools/ab_env.sh <workload> "<streams...>" <VAR> <value A> <value B>
w=$1; ss=$2; var=$3; shift 3
for s in $ss; do for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --workload $w --streams $s --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', '$var=$v', $s, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done; done
