#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04z_split_tail.txt
timeout 300 python tools/diag/split_tail_time.py 2>&1 | grep -v amdgpu.ids | tail -6 > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_denoiser_gpu.py tests/test_f16_gpu.py -q 2>&1 | tail -4 >> $O
cat $O
