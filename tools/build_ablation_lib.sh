#!/bin/bash
# The timing-ablation build of the bf16 attention for tools/limiter_probe.py --energy-table and tools/kernel_bench.py --ablate64:
# am_attention.hip (the dispatcher that accepts defer_log2 = 3000 + ABL), am_attention64.hip (the ablated instantiations, ISA-audited)
# and tools/variants/am_attention_variants.hip (the hook the dispatcher links against) compiled with -DAM_ATTN_ABLATIONS; everything
# else is the product objects.  Output: build/variants/libam_abl.so, selected with ACTIONMESH_AMD_LIB.
set -e
cd "$(dirname "$0")/../actionmesh_amd/csrc"
make -s all
OUT=../../build/variants
mkdir -p $OUT
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -mno-amdgpu-ieee -fno-honor-nans -DAM_ATTN_ABLATIONS"
A64FLAGS=$(grep -E '^A64FLAGS' Makefile | head -1 | cut -d= -f2-)
$CXX $A64FLAGS --cuda-device-only -S -o $OUT/abl64.s am_attention64.hip 2>/dev/null
python3 audit_attn64.py $OUT/abl64.s
$CXX $A64FLAGS -c am_attention64.hip -o $OUT/abl64.o &
$CXX -c am_attention.hip -o $OUT/abl_attn.o &
$CXX -I. -c ../../tools/variants/am_attention_variants.hip -o $OUT/abl_variants.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libam_abl.so am_elementwise.o am_gemm.o am_attention_fp8.o am_norm.o am_peer.o \
  am_pointcloud.o am_model.o host/am_phase_loop.o $OUT/abl_attn.o $OUT/abl64.o $OUT/abl_variants.o
echo "built $OUT/libam_abl.so"
