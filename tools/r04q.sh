#!/bin/bash
# round 4: LN fold with canonical row statistics - tiny-model tests (bit identity of the shortcuts, producer statistics vs read-back)
mkdir -p gpurun_out
O=gpurun_out/r04q_ln_fold_tiny.txt
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_denoiser_gpu.py -q -s 2>&1 | grep -E "rel-L2|passed|failed|FAILED|Error|assert" | cut -c1-260 | head -80 > $O
cat $O
