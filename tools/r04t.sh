#!/bin/bash
# round 4: LN fold, fold operands prefetched by LDS-DMA - tests, then same-box step A/B (fold on / off, two interleaved rounds)
mkdir -p gpurun_out
O=gpurun_out/r04t_ln_fold_ab.txt
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_denoiser_gpu.py -q -x 2>&1 | tail -4 > $O
for rep in 1 2; do
  for fold in 1 0; do
    echo "== LN_FOLD=$fold rep $rep" >> $O
    ACTIONMESH_AMD_LN_FOLD=$fold timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-nominal 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','ms_per_step') if k in r})" >> $O
  done
done
cat $O
