#!/bin/bash
# round 4: float16 build - kernel tests, tiny + BASELINE-architecture parity; bf16 sanity after the am_common.h refactor
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_f16_gpu.py -q -s 2>&1 | grep -v "^$" | tail -25 | tee gpurun_out/r04i_f16_tests.txt
timeout 900 python -m pytest tests/test_baseline_arch_gpu.py -q -s -k "float16" 2>&1 | grep -v "^$" | tail -12 | tee -a gpurun_out/r04i_f16_tests.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_denoiser_gpu.py -q -x 2>&1 | tail -4 | tee -a gpurun_out/r04i_f16_tests.txt
