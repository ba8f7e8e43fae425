run() { FE_LISENNET_SB=$3 python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(f\"$1 sb_min=$3 {d['config']['streams_per_gpu']:>6} streams  {d['value']/1e6:8.3f} M frames/s  {d['ms_per_step']*1e3:9.2f} us/step  frac {r['frac']:.4f} parity {d.get('parity_rms_rel')} [{r['kernel'][:90]}]\")"; }
(run lisennet 4096 0; run lisennet 4096 1; run lisennet 2048 0; run lisennet 2048 1; run lisennet 1024 0; run lisennet 1024 1; run lisennet 512 1; run lisennet 256 0; run lisennet 256 1; run lisennet 8192 1) 2>&1 | tee gpurun_out/t3_bench.txt
cd /tmp && export TMPDIR=/tmp && FE_LISENNET_SB=1 rocprofv3 --kernel-trace --stats -d /tmp/prof -o lsb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --workload lisennet --streams 4096 --steps 30 --warmup 5 > /dev/null 2>&1; cat /tmp/prof/*kernel_stats.csv | head -8 | tee $GRAFT_REPO_ROOT/gpurun_out/t3_stats.csv
