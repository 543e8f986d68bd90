#!/bin/bash
# Profiling build of the fp8 attention (s_memtime stamps, am_attention_fp8x64_profile): the product objects with am_attention_fp8.hip
# recompiled with -DAM_ATTN_ABLATIONS plus any extra flags.  Output: build/variants/libam_<name>.so, selected with ACTIONMESH_AMD_LIB.
#   tools/build_fp8_prof.sh fp8prof="" other="-DX64_SOMETHING=1"
set -e
cd "$(dirname "$0")/../actionmesh_amd/csrc"
make -s all
OUT=../../build/variants
mkdir -p $OUT
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -mno-amdgpu-ieee -fno-honor-nans"
OTHERS="am_elementwise.o am_gemm.o am_attention.o am_attention64.o am_norm.o am_peer.o am_pointcloud.o am_model.o host/am_phase_loop.o"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  (
    $CXX -DAM_ATTN_ABLATIONS $flags -c am_attention_fp8.hip -o $OUT/$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libam_$name.so $OTHERS $OUT/$name.o
    echo "built $OUT/libam_$name.so  ($flags)"
  ) &
done
wait
