#!/bin/bash
# A/B of two builds on ONE box (run through gpurun): put the candidates at ab/lib_old.so / ab/lib_new.so (ab/ is git-ignored),
# the script alternates them under fastenhancer_amd/libfastenhancer_hip.so; rebuild afterwards.
W=${1:-fe_b}
for rep in 1 2 3; do
  for v in old new; do
    cp ab/lib_$v.so fastenhancer_amd/libfastenhancer_hip.so
    python bench.py --no-cpu-baseline --workload $W --steps 500 --warmup 50 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$W', round(d['value']), d['ms_per_step'])"
  done
done
