#!/bin/bash
# phase clocks of the BSRNN-xt per-hop kernel for each ablation side build (ab/lib_abl*.so)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/abl; mkdir -p $O
W=${1:-bsrnn_xt}
timeout 100 python tools/gpu_phases_bsrnn.py $W 256 2>/dev/null | grep -E "frame =|recurrence|MLP" | tr '\n' ' ' ; echo " <- base"
for f in ab/lib_abl*.so; do
  export FASTENHANCER_HIP_LIB=$PWD/$f
  timeout 100 python tools/gpu_phases_bsrnn.py $W 256 2>/dev/null | grep -E "frame =|recurrence|MLP" | tr '\n' ' '; echo " <- $f"
done
