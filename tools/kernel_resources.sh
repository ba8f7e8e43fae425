#!/bin/bash
# Register / LDS / spill figures of every kernel in one object of csrc/_obj (no GPU needed):
#   tools/kernel_resources.sh fe_shape_B
set -e
OBJ=${1:-fe_shape_B}; OBJDIR=${2:-_obj}
HERE=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
objcopy --dump-section .hip_fatbin=$T/fat.bin $HERE/fastenhancer_amd/csrc/$OBJDIR/$OBJ.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|spill_count|group_segment_fixed|private_segment_fixed" \
  | sed -e 's/^ *//' | paste -sd' ' | sed -e 's/\.name:/\n.name:/g' | sed -e 's/_ZN2fe15fe_frame_kernelINS_5ShapeI//' | cut -c1-260
echo
rm -rf $T
