#!/bin/bash
# A/B builds of the GEMM: the product objects with am_gemm.hip recompiled with extra -D flags, or from another source file
# (name=@path: e.g. the previous commit's am_gemm.hip).  Output: build/variants/libam_<name>.so, selected with ACTIONMESH_AMD_LIB.
#   tools/build_gemm_variants.sh nt="-DAM_GEMM_NT=1" prev=@/tmp/am_gemm_prev.hip
set -e
cd "$(dirname "$0")/../actionmesh_amd/csrc"
make -s all
OUT=../../build/variants
mkdir -p $OUT
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -I."
OTHERS="am_elementwise.o am_attention.o am_attention64.o am_attention_fp8.o am_norm.o am_peer.o am_pointcloud.o am_model.o host/am_phase_loop.o"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"; src=am_gemm.hip
  if [[ "$flags" == @* ]]; then src="${flags#@}"; flags=""; fi
  (
    $CXX $flags -c $src -o $OUT/$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libam_$name.so $OTHERS $OUT/$name.o
    echo "built $OUT/libam_$name.so  ($flags $src)"
  ) &
done
wait
