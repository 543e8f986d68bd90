#!/bin/bash
# PMC passes on the attention kernels only (counters, no tracing domains).  Output: gpurun_out/pmc_attn/summary.txt
OUT=$PWD/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$i -o pmc -- python tools/kernel_bench.py --only attn --reps 1 > $OUT/pmc_$i.log 2>&1
done
python tools/summarize_prof.py $OUT $OUT/attn > $OUT/summary.txt 2>&1
