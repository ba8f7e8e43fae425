#!/bin/bash
# the round's table of per-hop numbers on one box: tools/bench_matrix.sh  (through gpurun; one line per configuration)
run() { python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps ${3:-200} --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(f\"$1 {d['config']['streams_per_gpu']:>6} streams  {d['value']/1e6:8.3f} M frames/s  {d['ms_per_step']*1e3:9.2f} us/step  frac {r['frac']:.4f}  [{r['kernel'][:60]}]\")"; }
run fe_b 256; run fe_b 1024; run fe_b 2048
run fe_t 256; run fe_t 2048; run fe_s 256; run fe_m 256; run fe_l 256 100; run fe_l 1024 50
run fe48_t 256; run fe48_t 2048; run fe48_b 256; run fe48_b_h480 512; run fe48_l 256 100
run fe_tk_b 256; run fe_ln_b 256; run fe_dprnn_b 256; run fe_dpt_b 256; run fe_dpt_t 256; run fe_dpt_b 1024; run fe_dprnn_b 1024
run bsrnn_xt 256; run bsrnn_xt 4096 50; run bsrnn_xxt 4096 50; run bsrnn_t 256; run bsrnn_t 4096 20; run bsrnn_s 256 100; run bsrnn_s 4096 10
run fspen 256; run fspen 4096 50; run lisennet 256; run lisennet 2048 50; run lisennet 4096 50
