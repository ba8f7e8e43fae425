#!/bin/bash
# per-kernel average durations of one bench.py configuration (run through gpurun from the repo root):
#   tools/prof_kernels.sh <out-name> <bench args...>     -> gpurun_out/<out-name>/s_kernel_stats.csv, top kernels printed
n=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$n -o s -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$n.json 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/$n/s_kernel_stats.csv")))
for r in rows[:6]:
    print(f"{float(r['AverageNs'])/1e3:9.2f} us x {r['Calls']:>6}  {float(r['Percentage']):5.1f} %  {r['Name'][:110]}")
PY
tail -1 $R/gpurun_out/$n.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"
