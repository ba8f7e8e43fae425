python tools/gpu_phases_lisennet_sb.py 4096 2>&1 | tail -12
python tools/gpu_phases_lisennet_sb.py 16 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp && FE_LISENNET_SB=1 rocprofv3 --kernel-trace --stats -d /tmp/prof -o lsb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --workload lisennet --streams 4096 --steps 30 --warmup 5 > /tmp/prof.log 2>&1; find /tmp/prof -name "*kernel_stats.csv" | head; cat $(find /tmp/prof -name "*kernel_stats.csv" | head -1) | head -8
