#!/bin/bash
# Same-box A/B of the 512-thread per-hop kernel (FE_WG8=1, fe_frame8.hip.h) against the four-wave kernel (FE_WG8=0), one library.
#   tools/ab_wg8.sh [lib] [workload] [extra bench args]
LIB=${1:-fastenhancer_amd/libfastenhancer_hip.so}
W=${2:-fe_b}
shift 2
export FASTENHANCER_HIP_LIB=$(realpath $LIB)
for rep in 1 2 3; do
  for v in 0 1; do
    FE_WG8=$v python bench.py --no-cpu-baseline --workload $W --steps 500 --warmup 50 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FE_WG8=$v', '$W', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
  done
done
