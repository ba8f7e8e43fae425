#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2c; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -k "offline or spec or pipelined" 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
for a in "fe_b 4 1" "fe_b 4 8" "fe_t 4 1" "fe_l 4 1" "fe48_b 4 1"; do timeout 300 python tools/gpu_offline_timing.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/offline_timing.txt; done
