#!/bin/bash
# tools: another build of ONE source of the library with extra -D flags, linked with the objects of the normal build -> gpurun_variants/lib_<tag>.so
# (load it with SF_LIB_PATH).   usage: tools/build_variant.sh <tag> <source.hip> <flags...>
TAG=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpurun_variants
OBJ=$R/slotformer_amd/csrc/build
EXTRA=""
if [ "$SRC" = "layer_tok.hip" ]; then EXTRA="-fno-slp-vectorize"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA "$@" -c $R/slotformer_amd/csrc/$SRC -o /tmp/variant_$TAG.o || exit 1
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$TAG.o -o $R/gpurun_variants/lib_$TAG.so && echo built $R/gpurun_variants/lib_$TAG.so
