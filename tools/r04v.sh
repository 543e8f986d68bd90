#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/diag/f16_gemm_worst.py 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/r04v_f16_gemm_worst.txt
cat gpurun_out/r04v_f16_gemm_worst.txt
