MAP=0 python tools/gpu_lisennet_sb_debug.py 2>&1 | grep "hop 2\|cache" | grep -v "blocks\|spec_in\|compressed"
B=37 MAP=0 python tools/gpu_lisennet_sb_debug.py 2>&1 | grep "hop 2 wav_out\|hop 2 mask\|cache 10\|cache 1 "
run() { python bench.py --no-cpu-baseline --workload $1 --streams $2 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(f\"$1 {d['config']['streams_per_gpu']:>6} streams  {d['value']/1e6:8.3f} M frames/s  {d['ms_per_step']*1e3:9.2f} us/step  frac {r['frac']:.4f} parity {d.get('parity_rms_rel')}\")"; }
run lisennet 4096; run lisennet 2048; run lisennet 1024
timeout 300 bash tools/prof_kernels.sh t14prof --workload lisennet --streams 4096 --steps 30 --warmup 5 --no-parity
python tools/gpu_phases_lisennet_sb.py 4096 2>&1 | tail -8
