#!/bin/bash
# Same-box A/B of 4x64 attention kernel builds (tools/build_variants.sh): two interleaved timing rounds over all variants,
# then the attention kernel tests on the candidates, then per-phase s_memtime stamps of the *_prof builds.
#   tools/ab_attn64.sh "pf0 pf1 pf3 ..." "candidates to test" "prof builds"
OUT=gpurun_out/ab_attn64.txt
mkdir -p gpurun_out; : > $OUT
V=$PWD/build/variants
for round in 1 2; do
  for v in $1; do
    echo "=== round $round $v" >> $OUT
    ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
for v in $2; do
  echo "=== tests $v" >> $OUT
  ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x 2>&1 | tail -3 >> $OUT
done
for v in $3; do
  echo "=== stamps $v" >> $OUT
  ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/attn_profile.py --k64 2>&1 | grep -v amdgpu.ids | sed -n 1,12p >> $OUT
done
cat $OUT
