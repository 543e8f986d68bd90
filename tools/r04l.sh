#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_fp8.py -q -s -k "fast or rebase or round4 or coverage" 2>&1 | grep -E "fp8_fast|re-base|passed|failed|Error" | tail -20 | tee gpurun_out/r04l_fast.txt
timeout 300 python tools/kernel_bench.py --only attn --product-only --fp8 --reps 5 2>&1 | grep -E "attend only" | tee -a gpurun_out/r04l_fast.txt
