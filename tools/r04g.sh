#!/bin/bash
# round 4, second GPU call: the 4x64 fp8 kernel - parity tests, same-box A/B against the 8-wave kernel, phase stamps
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_fp8.py -q -s 2>&1 | grep -v "^$" | tail -70 > gpurun_out/r04g_fp8_tests.txt
tail -50 gpurun_out/r04g_fp8_tests.txt
timeout 300 python tools/kernel_bench.py --only attn --product-only --fp8 2>&1 | tee gpurun_out/r04g_fp8_ab.txt
ACTIONMESH_AMD_LIB=build/variants/libam_fp8prof.so timeout 120 python tools/attn_profile.py --fp8p 2>&1 | tee gpurun_out/r04g_fp8_stamps.txt | tail -20
