run() {
  export FASTENHANCER_HIP_LIB=$PWD/ab/lib_varlow.so; export FE_LOWLDS=$3
  python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 100 --warmup 20 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'lowlds=$3', $2, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4), d.get('parity_rms_rel'), d['roofline']['kernel'][:60])" || tail -3 /tmp/ab_err.txt
}
for w in fe_dpt_b fe_dprnn_b; do for s in 384 512 1024 2048; do run $w $s 0; run $w $s 1; done; done
for w in fe_dpt_t fe_dprnn_t; do for s in 512 768 1024 2048; do run $w $s 0; run $w $s 1; done; done
