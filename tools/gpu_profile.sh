#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command + PMC passes of the
# dominant kernels.  Outputs under gpurun_out/prof_$TAG/ ; copy the summaries into profiles/.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1) per-kernel time of the exact bench command
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
# 2) PMC passes (counters only, no tracing domains) on the kernel micro-bench
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python tools/kernel_bench.py --only attn,gemm --reps 1 > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
