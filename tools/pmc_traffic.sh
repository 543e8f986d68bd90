#!/bin/bash
# HBM traffic of the dominant kernel (inflated self-attention at the bench.py workload shape), collected exactly as
# MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass),
# counters only (no tracing domains), FETCH_SIZE doubled (gfx950 tallies 128-B requests of 16 B/lane streams at 64 B).
# Run on the GPU box through gpurun; writes gpurun_out/r02_attention_traffic.json, which names the kernel sources it was
# measured on (bench.py quotes it as roofline.traffic only while those are the sources that are built).
#   tools/pmc_traffic.sh [shape]        shape = headline (default) | nominal
SHAPE=${1:-headline}
OUT=$PWD/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$name -o pmc -- python tools/kernel_bench.py --shape $SHAPE --only attn --product-only --reps 1 > $OUT/pmc_$name.log 2>&1
done
python tools/make_traffic_json.py $OUT $SHAPE gpurun_out/r02_attention_traffic.json
