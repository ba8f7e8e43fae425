run() { python bench.py --no-cpu-baseline --no-parity --workload $1 --streams $2 --steps ${3:-100} --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(f\"$1 {d['config']['streams_per_gpu']:>6} streams  {d['ms_per_step']*1e3:9.2f} us/step  frac {r['frac']:.4f}\")"; }
run bsrnn_xt 4096; run bsrnn_xxt 4096; run bsrnn_xt 256 300; run bsrnn_xt 1024; run bsrnn_t 4096 30
