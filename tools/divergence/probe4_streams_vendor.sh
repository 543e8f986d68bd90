#!/bin/bash
# Fourth probe: is it cross-process only (one process, two streams)?  does a vendor GEMM do it too?  which code shape of head_post is immune?
OUT=gpurun_out/r03d_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/build/variants
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
{
echo "=== O: one process, two streams"
timeout 300 python tools/divergence/overlap_probe.py 2>&1 | grep "overlap_probe\|Error"
export AM_PHASES=idle,gemm,gemm256,matmul,matmul_f32
for lib in product hpnoslp hpdrain; do
  echo "=== X: two processes, victim library = $lib"
  T0=$(( $(date +%s) + 25 ))
  if [ $lib = product ]; then L=""; else L=$V/libam_$lib.so; fi
  ACTIONMESH_AMD_LIB=$L timeout 200 python tools/divergence/interference_probe.py victim $T0 2>&1 | grep "victim\|Error" | cut -c1-330 &
  timeout 200 python tools/divergence/interference_probe.py aggressor $T0 2>&1 | grep "Error" &
  wait
done
} >> $OUT 2>&1
cat $OUT
