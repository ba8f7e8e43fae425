#!/bin/bash
# Timing-ablation side builds of ONE translation unit: tools/abl_build.sh <tag> <obj stem> <extra hipcc defs...>
#   e.g. tools/abl_build.sh abl1 fe_bsrnn_xt -DBS_ABL=1    ->  ab/lib_abl1.so  (all other objects come from the main build)
set -e
TAG=$1; STEM=$2; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
CS=$HERE/fastenhancer_amd/csrc
mkdir -p $HERE/ab $CS/_obj_abl
case $STEM in
  fe_bsrnn_*) NAME=${STEM#fe_bsrnn_}; TMPL=$CS/fe_bsrnn_shape.hip.in; ARGS=$(grep -E "^XB\(\s*$NAME\s*," $CS/fe_bsrnn_shapes.def | sed -e 's/^XB(\s*[A-Za-z0-9_]*\s*,//' -e 's/)\s*$//' | tr -d ' ');;
  fe_shape_*) NAME=${STEM#fe_shape_}; TMPL=$CS/fe_shape.hip.in; ARGS=$(grep -E "^X\(\s*$NAME\s*," $CS/fe_shapes.def | sed -e 's/^X(\s*[A-Za-z0-9_]*\s*,//' -e 's/)\s*$//' | tr -d ' ');;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form "$@" -DFE_SHAPE_NAME=$NAME "-DFE_SHAPE_ARGS=$ARGS" -c -x hip $TMPL -o $CS/_obj_abl/${STEM}_$TAG.o
OBJS=$(ls $CS/_obj/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/ab/lib_$TAG.so $OBJS $CS/_obj_abl/${STEM}_$TAG.o
echo built ab/lib_$TAG.so
