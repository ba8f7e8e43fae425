#!/bin/bash
# r6: low-LDS companions of the smallest shapes at three / four workgroups per CU (Lds<S>::OCC, FE_OCC_SMALL) - same-box A/B over side builds
#   tbase = OCC 2 everywhere (FE_LOWLDS=0: the shape's own kernel), tl1o3 = TLOW LOW=1 x3, tl2o3 = TLOW LOW=2 x3, tl2o4 = TLOW LOW=2 x4
run() {  # workload streams tag lowlds
  export FASTENHANCER_HIP_LIB=$PWD/ab/lib_$3.so; export FE_LOWLDS=$4
  python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 200 --warmup 30 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$3', 'lowlds=$4', $2, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4), d.get('parity_rms_rel'), d['roofline']['kernel'][:60])" || tail -3 /tmp/ab_err.txt
}
for s in 384 512 768 1024 2048 4096; do
  run fe_t $s tbase 0; run fe_t $s tbase 1; run fe_t $s tl1o3 1; run fe_t $s tl2o3 1; run fe_t $s tl2o4 1
done
for s in 512 768 1024 2048; do
  run fe48_t $s tbase 0; run fe48_t $s tbase 1; run fe48_t $s tl1o3 1
done
