#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2d; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.txt; cat $O/pytest.txt
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 200 --warmup 20 --workload fe_tk_b > $O/bench_tk.json 2> $O/bench.err
$B --steps 500 --warmup 50 > $O/bench_b.json 2>> $O/bench.err
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'frac', round(d['roofline']['frac'],4))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err') if False else 'gpurun_out/r2d/bench.err').read()[-2000:])
PY
done
