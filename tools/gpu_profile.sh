#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the exact bench command + PMC passes of the dominant kernels.
# Outputs under gpurun_out/prof_$TAG/ ; tools/summarize_prof.py turns them into the summaries committed under profiles/.
# PMC passes carry counters only (no tracing domains), FETCH_SIZE / WRITE_SIZE in their own passes (MI355X_MICROARCH.md).
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# 1) per-kernel time of the bench command
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
# 2) PMC passes on the kernel micro-bench (product kernels only)
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python tools/kernel_bench.py --only attn,gemm --product-only --fp8 --reps 1 > $OUT/pmc_$name.log 2>&1
done
python tools/summarize_prof.py $OUT gpurun_out/${TAG}_final > $OUT/summary.log 2>&1
