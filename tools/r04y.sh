#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/diag/cross_attn_geometry.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r04y_cross_attn_geometry.txt
cat gpurun_out/r04y_cross_attn_geometry.txt
