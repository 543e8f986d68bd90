#!/bin/bash
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_ln_fold_gpu.py -q -k canonical 2>&1 | tail -8 > gpurun_out/r04ae_canonical.txt
cat gpurun_out/r04ae_canonical.txt
