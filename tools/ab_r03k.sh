V=$PWD/build/variants
for round in 1 2; do
  for v in product pf2 pf5; do
    if [ $v = product ]; then L=""; else L=$V/libam_$v.so; fi
    echo "=== round $round $v"
    ACTIONMESH_AMD_LIB=$L python tools/kernel_bench.py --only gemm --product-only --reps 8 2>&1 | grep "gemm" | cut -c1-110
  done
done
for round in 1 2; do
  for v in product f8pk; do
    if [ $v = product ]; then L=""; else L=$V/libam_$v.so; fi
    echo "=== round $round $v"
    ACTIONMESH_AMD_LIB=$L python tools/kernel_bench.py --only attn --product-only --fp8 --reps 4 2>&1 | grep "fp8" | cut -c1-120
  done
done
ACTIONMESH_AMD_LIB=$V/libam_f8pk.so python -m pytest tests/test_attention_fp8.py -q -m gpu -k "matches_fp32 or coverage" 2>&1 | tail -2
ACTIONMESH_AMD_LIB=$V/libam_pf2.so python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -2
