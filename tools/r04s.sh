#!/bin/bash
# round 4: per-kernel time of one headline step with the LayerNorms folded (default) and with ACTIONMESH_AMD_LN_FOLD=0
mkdir -p gpurun_out
export TMPDIR=/tmp
for fold in 1 0; do
  OUT=$PWD/gpurun_out/prof_fold$fold
  rm -rf $OUT; mkdir -p $OUT
  ACTIONMESH_AMD_LN_FOLD=$fold timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-nominal > $OUT/bench.log 2>&1
  python tools/summarize_prof.py $OUT gpurun_out/r04s_fold$fold > $OUT/summary.log 2>&1
  rm -rf $OUT/bench
done
head -30 gpurun_out/r04s_fold1_kernel_stats.csv | cut -c1-200
