#!/bin/bash
# round 4: kernel-level tests of the folded LayerNorm (consumer fold, producer row statistics)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ln_fold_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r04n_ln_fold_tests.txt
cat gpurun_out/r04n_ln_fold_tests.txt
