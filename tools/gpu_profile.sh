#!/bin/bash
# Run on the GPU box (via gpurun).  ONE script, ONE sources sha: for the kernel sources that are built now (bench.source_sha) it
# produces together
#   profiles-ready summaries   gpurun_out/<tag>_<sha>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the exact bench command
#                              gpurun_out/<tag>_<sha>_by_launch_shape.csv
#                              gpurun_out/<tag>_<sha>_pmc.csv                 SQ counter passes on the product kernels
#                              gpurun_out/<tag>_<sha>_attention_traffic.json  FETCH_SIZE / WRITE_SIZE passes of the attention launch
#                              gpurun_out/<tag>_<sha>_bench.json              the bench line of the traced run
# so that every number of a round's DESIGN.md row can be tied to one build.  PMC passes carry counters only
# (no tracing domains); FETCH_SIZE / WRITE_SIZE get their own passes (MI355X_MICROARCH.md "HBM").  bench.py quotes the traffic JSON as
# roofline.traffic only while its source_sha is the sha of the sources in the tree.
#   tools/gpu_profile.sh [tag]        e.g. tools/gpu_profile.sh r03
TAG=${1:-r03}
SHA=$(python -c "import bench; print(bench.source_sha())")
NAME=${TAG}_${SHA}
OUT=$PWD/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# 1) per-kernel time of the bench command
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nominal > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log | tail -1 > gpurun_out/${NAME}_bench.json
# 2) SQ counter passes on the kernel micro-bench (product kernels only)
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python tools/kernel_bench.py --only attn,gemm --product-only --fp8 --reps 1 > $OUT/pmc_$name.log 2>&1
done
# 3) HBM traffic of the attention launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
mkdir -p $OUT/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $OUT/traffic/pmc_$c -o pmc -- python tools/kernel_bench.py --shape headline --only attn --product-only --reps 1 > $OUT/traffic/pmc_$c.log 2>&1
done
python tools/summarize_prof.py $OUT gpurun_out/$NAME > $OUT/summary.log 2>&1
python tools/make_traffic_json.py $OUT/traffic headline gpurun_out/${NAME}_attention_traffic.json >> $OUT/summary.log 2>&1
tail -5 $OUT/summary.log
ls -la gpurun_out/${NAME}_*
