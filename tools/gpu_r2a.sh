#!/bin/bash
# round-2 GPU pass A: parity suite, bench lines (driver-style short run, steady state, >256 streams), phase clocks
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2a; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt
echo "== pytest"; cat $O/pytest.txt
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $O/bench_b_20.json 2> $O/bench_b_20.err
$B --steps 20 --warmup 5 --clock-ramp-ms 0 > $O/bench_b_20_noramp.json 2>> $O/bench_b_20.err
$B --steps 500 --warmup 50 > $O/bench_b_500.json 2>> $O/bench_b_20.err
for n in 512 1024 2048; do $B --steps 300 --warmup 30 --streams $n > $O/bench_b_s$n.json 2>> $O/bench_b_20.err; done
$B --steps 200 --warmup 20 --workload fe48_b_h480 --streams 512 > $O/bench_48h480_s512.json 2>> $O/bench_b_20.err
$B --steps 200 --warmup 20 --workload fe48_b_h480 --streams 256 > $O/bench_48h480_s256.json 2>> $O/bench_b_20.err
$B --steps 100 --warmup 10 --workload fe_l --streams 256 > $O/bench_l_s256.json 2>> $O/bench_b_20.err
$B --steps 200 --warmup 20 --workload fe_t --streams 256 > $O/bench_t_s256.json 2>> $O/bench_b_20.err
$B --steps 200 --warmup 20 --workload bsrnn_xt --streams 256 > $O/bench_bsrnn_xt.json 2>> $O/bench_b_20.err
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_default_cpu.json 2> $O/bench_default_cpu.err
python bench.py --gpus 2 > $O/bench_gpus2.txt 2>&1; echo "rc=$?" >> $O/bench_gpus2.txt
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'frac', round(d['roofline']['frac'],4), 'ramp', d.get('clock_ramp_steps'), d.get('cpu_baseline',{}).get('value'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cat $O/bench_gpus2.txt | tail -3
# phase clocks of the per-hop kernels (FE_PROBE_HOT side build)
export FASTENHANCER_HIP_LIB=$PWD/ab/lib_probe.so
for w in fe_b fe_t; do timeout 120 python tools/gpu_phases.py $w 256 1 > $O/phases_$w.txt 2>&1; cat $O/phases_$w.txt; done
timeout 120 python tools/gpu_phases_bsrnn.py bsrnn_xt 256 > $O/phases_bsrnn_xt.txt 2>&1; cat $O/phases_bsrnn_xt.txt
unset FASTENHANCER_HIP_LIB
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"
