#!/bin/bash
# round 4, first GPU call: fp8 parity curves at the BASELINE architectures, the N > 1 bench dry run, the limiter telemetry
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
python tools/limiter_probe.py --out gpurun_out/r04a_limiter.json > gpurun_out/r04a_limiter.txt 2>&1
tail -12 gpurun_out/r04a_limiter.txt
python -m pytest tests/test_baseline_arch_gpu.py -q -k "fp8" -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r04a_fp8_parity.txt
tail -25 gpurun_out/r04a_fp8_parity.txt
python -m pytest tests/test_multi_gpu.py tests/test_actionbench.py tests/test_attention_fp8.py -q 2>&1 | tail -15 | tee gpurun_out/r04a_tests.txt
