run() {
  export FASTENHANCER_HIP_LIB=$PWD/ab/lib_$3.so
  python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 100 --warmup 20 2>/tmp/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', '$3', $2, round(d['value']), round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline']['frac'],4), d.get('parity_rms_rel'), d['roofline']['kernel'][:60])" || tail -3 /tmp/ab_err.txt
}
for s in 640 1024 2048 4096; do for t in np0 np1; do run fe_b $s $t; done; done
for s in 1024 2048 4096; do for t in np0 np1; do run fe_t $s $t; done; done
for s in 640 1024 2048; do for t in np0 np1; do run fe_dprnn_b $s $t; done; done
