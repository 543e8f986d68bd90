#!/bin/bash
# round 4: LayerNorm folded into its consumer linears - model-level parity tests, then a same-box A/B of the step
# (ACTIONMESH_AMD_LN_FOLD=0 = the round-3 sequence: LayerNorm kernel + plain linear)
mkdir -p gpurun_out
O=gpurun_out/r04o_ln_fold_model.txt
: > $O
timeout 1500 python -m pytest tests/test_denoiser_gpu.py tests/test_baseline_arch_gpu.py tests/test_f16_gpu.py tests/test_ln_fold_gpu.py -x -q 2>&1 | tail -15 >> $O
for rep in 1 2; do
  for fold in 1 0; do
    echo "== LN_FOLD=$fold rep $rep" >> $O
    ACTIONMESH_AMD_LN_FOLD=$fold timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-nominal 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','ms_per_step') if k in r})" >> $O
  done
done
cat $O
