#!/bin/bash
# A/B builds of the 4x64 attention kernel: same product objects, am_attention64.hip recompiled with extra -D flags
# (each audited like the product build).  Output: build/variants/libam_<name>.so, selected with ACTIONMESH_AMD_LIB.
#   tools/build_variants.sh name1="-DAM_A64_POSTFENCE=3" name2="-DAM_ES_MOVE=20" ...
set -e
cd "$(dirname "$0")/../actionmesh_amd/csrc"
make -s all
OUT=../../build/variants
mkdir -p $OUT
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -mno-amdgpu-ieee -fno-honor-nans"
OTHERS="am_elementwise.o am_gemm.o am_attention.o am_attention_fp8.o am_norm.o am_peer.o am_pointcloud.o am_model.o host/am_phase_loop.o"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  (
    $CXX $flags --cuda-device-only -S -o $OUT/$name.s am_attention64.hip 2>/dev/null
    python3 audit_attn64.py $OUT/$name.s
    $CXX $flags -c am_attention64.hip -o $OUT/$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libam_$name.so $OTHERS $OUT/$name.o
    echo "built $OUT/libam_$name.so  ($flags)"
  ) &
done
wait
