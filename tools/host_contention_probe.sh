#!/bin/bash
# Host-side contention of N launch threads under the container's CPU quota, on a ONE-GPU box: N ranks (gloo), all launching on cuda:0,
# 32 streams each (N x 32 <= 256 workgroups: the GPU is never the bottleneck of the enqueue path).  NOT a scaling number - it reads
# host_enqueue_us_per_step (the launch thread's own time per fe_step call) and host_launch_overhead_ms_per_step as N grows.
# usage (GPU box): tools/host_contention_probe.sh > gpurun_out/r5_host_contention.txt
cd "$(dirname "$0")/.."
echo "# $(date -u +%FT%TZ)  nproc=$(nproc)  cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for graph in "" "--graph"; do
  for n in 1 2 4 8; do
    python bench.py --gpus $n --share-gpu --streams 32 --steps 200 --warmup 20 --blocks 9 --no-cpu-baseline --no-parity $graph 2>/dev/null \
      | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln)
        print('ranks=%d graph=%s  enqueue %.2f us/step  wall %.2f us/step  kernel(events) %.2f us  overhead %.2f us  per-rank wall %s' % (
            d['n_gpus'], '$graph' != '', d['host_enqueue_us_per_step'], d['ms_per_step']*1e3, d['kernel_ms_hip_events']*1e3,
            d['host_launch_overhead_ms_per_step']*1e3, ['%.1f' % (v*1e3) for v in d['per_rank_ms_per_step']]))
"
  done
done
