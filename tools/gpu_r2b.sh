#!/bin/bash
# round-2 GPU pass B: reworked BSRNN kernel - parity, bench, phase clocks
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2b; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -k "bsrnn" 2>&1 | tail -25 ) > $O/pytest_bsrnn.txt
echo "== pytest bsrnn"; cat $O/pytest_bsrnn.txt
B="timeout 300 python bench.py --no-cpu-baseline"
for w in bsrnn_xt bsrnn_xxt bsrnn_t bsrnn_s; do $B --steps 100 --warmup 10 --workload $w > $O/bench_$w.json 2>> $O/bench.err; done
$B --steps 100 --warmup 10 --workload bsrnn_xt --streams 512 > $O/bench_bsrnn_xt_s512.json 2>> $O/bench.err
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'frac', round(d['roofline']['frac'],4))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
tail -5 $O/bench.err
for w in bsrnn_xt bsrnn_t bsrnn_s; do timeout 120 python tools/gpu_phases_bsrnn.py $w 256 > $O/phases_$w.txt 2>&1; cat $O/phases_$w.txt; done
