#!/bin/bash
# A/B of the low-LDS companions on one box: bench each workload above #CUs streams with and without FE_NO_LOWLDS
# usage: tools/ab_lowlds.sh "fe_b fe_s" "512 1024"
for w in $1; do for s in $2; do for v in 0 1; do
  if [ $v = 0 ]; then export FE_NO_LOWLDS=1; else unset FE_NO_LOWLDS; fi
  python bench.py --no-cpu-baseline --workload $w --streams $s --steps 200 --warmup 30 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'lowlds=$v', $s, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4))"
done; done; done
