#!/bin/bash
# GEMM intra-XCD sub-skew experiment: same build, env knob, interleaved
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rnd in 0 1; do for sub in 0 5 10 20 40 80; do
  echo "=== subskew $sub (x0.1 us per step, 4 steps) round $rnd"
  ACTIONMESH_AMD_GEMM_SUBSKEW=$sub timeout 200 python tools/kernel_bench.py --only gemm --product-only --reps 20 2>&1 | grep pingpong
done; done | tee gpurun_out/r04m_gemm_subskew.txt
