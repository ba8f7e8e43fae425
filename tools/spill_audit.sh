#!/bin/bash
# Spill audit of every kernel in csrc/_obj (no GPU needed): prints the kernels with more than MIN_SGPR spilled SGPRs, any spilled VGPR, or scratch.
#   tools/spill_audit.sh [MIN_SGPR=150]
# SGPR spills are v_writelane / v_readlane pairs on the datapath the fp32 MFMAs use (r4w: FastEnhancer_L's per-hop kernel lost 1022 of them
# and 3.6 % of its time with the loop-variant zero, profiles/r4u_headline_levers_and_dpt_prefetch.txt).
MIN=${1:-150}
HERE=$(cd "$(dirname "$0")/.." && pwd)
for f in $HERE/fastenhancer_amd/csrc/_obj/*.o; do
  o=$(basename $f .o); T=$(mktemp -d)
  objcopy --dump-section .hip_fatbin=$T/fat.bin $f 2>/dev/null || { rm -rf $T; continue; }
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co --unbundle 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co 2>/dev/null | grep -E "\.name:|sgpr_spill_count|\.vgpr_count|vgpr_spill|private_segment_fixed_size" | sed -e 's/^ *//' | paste -sd' ' | sed -e 's/\.name:/\n.name:/g' \
    | awk -v O=$o -v MIN=$MIN '{s=0;v=0;g=0;p=0; for(i=1;i<=NF;i++){if($i==".sgpr_spill_count:")s=$(i+1); if($i==".vgpr_spill_count:")v=$(i+1); if($i==".vgpr_count:")g=$(i+1); if($i==".private_segment_fixed_size:")p=$(i+1)} n=$2; if(n!="" && (s>MIN||v>0||p>0)) print O, substr(n,1,36) ".." substr(n,length(n)-52), "sgpr_spill="s, "vgpr="g, "vgpr_spill="v, "scratch="p}'
  rm -rf $T
done
