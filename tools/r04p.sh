#!/bin/bash
# round 4: LN fold - printed errors of the model tests with and without the fold, and a kernel trace of one step with it
mkdir -p gpurun_out
O=gpurun_out/r04p_ln_fold_errors.txt
: > $O
for fold in 1 0; do
  echo "== LN_FOLD=$fold" >> $O
  ACTIONMESH_AMD_LN_FOLD=$fold timeout 900 python -m pytest tests/test_denoiser_gpu.py tests/test_baseline_arch_gpu.py -q -s 2>&1 | grep -E "rel|err|passed|failed|FAILED|curve|e-0" | cut -c1-220 | head -80 >> $O
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fold
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_fold -o fold -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-nominal > /tmp/prof_fold.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_fold -name "*kernel_stats.csv" | head -1)
echo "== kernel stats (fold on): $f" >> $O
head -40 "$f" | cut -c1-260 >> $O
tail -60 $O
