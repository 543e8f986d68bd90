#!/bin/bash
# round 4, after the last source changes (lazy re-fold at the forward's entry point, sub-skew experiment removed): the tests they touch,
# the profile set for the new sources sha, and the default bench line (which now finds its traffic record)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_denoiser_gpu.py tests/test_ln_fold_gpu.py tests/test_f16_gpu.py tests/test_kernels_gpu.py -q 2>&1 | tail -4 > gpurun_out/r04w_tests.txt
timeout 600 python -m pytest tests/test_attention_fp8.py -q -s -k "world or shard" 2>&1 | grep -E "world|passed|failed" | cut -c1-200 >> gpurun_out/r04w_tests.txt
cat gpurun_out/r04w_tests.txt
bash tools/gpu_profile.sh r04 2>&1 | tail -4
SHA=$(python -c "import bench; print(bench.source_sha())")
cp gpurun_out/r04_${SHA}_attention_traffic.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 1 2>gpurun_out/r04w_bench.err | tail -1 > gpurun_out/r04w_bench_headline.json
python -c "
import json; d=json.load(open('gpurun_out/r04w_bench_headline.json'))
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('nominal',{}).get('ms_per_step'))"
