#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/diag/fold_shard.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r04r_fold_shard_diag.txt
cat gpurun_out/r04r_fold_shard_diag.txt
