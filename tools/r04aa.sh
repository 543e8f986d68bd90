#!/bin/bash
# round 4: the latency-built GEMM tail kernel - tests, then same-box step A/B (ACTIONMESH_AMD_GEMM_TAIL=128 = the 128x128 kernel on the tails)
mkdir -p gpurun_out
O=gpurun_out/r04aa_gemm_tail.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_ln_fold_gpu.py tests/test_denoiser_gpu.py tests/test_f16_gpu.py -q 2>&1 | tail -3 > $O
timeout 600 python -m pytest tests/test_baseline_arch_gpu.py -q -k "full_headline and bf16" 2>&1 | tail -2 >> $O
for rep in 1 2; do
  for tail in new 128; do
    echo "== GEMM_TAIL=$tail rep $rep" >> $O
    ACTIONMESH_AMD_GEMM_TAIL=$tail timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-nominal 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','ms_per_step') if k in r})" >> $O
  done
done
cat $O
