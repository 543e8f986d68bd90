#!/bin/bash
# round 4: GEMM epilogue A/B (residual prefetch; non-temporal C / residual traffic) against the previous build, interleaved on one box
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for rnd in 0 1; do
  for v in prod prev nt; do
    if [ $v = prod ]; then lib=""; else lib="build/variants/libam_$v.so"; fi
    echo "=== $v round $rnd"
    ACTIONMESH_AMD_LIB=$lib timeout 200 python tools/kernel_bench.py --only gemm --product-only $( [ $rnd = 0 ] && [ $v = prod ] && echo --blas ) 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r04h_gemm_ab.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" 2>&1 | tail -3 | tee -a gpurun_out/r04h_gemm_ab.txt
