#!/bin/bash
# round 4: profile set (kernel trace, PMC, traffic record) and default bench line for the FINAL sources
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_profile.sh r04 2>&1 | tail -3
SHA=$(python -c "import bench; print(bench.source_sha())")
cp gpurun_out/r04_${SHA}_attention_traffic.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 1 2>gpurun_out/r04ab_bench.err | tail -1 > gpurun_out/r04ab_bench_headline.json
python -c "
import json; d=json.load(open('gpurun_out/r04ab_bench_headline.json'))
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('nominal',{}).get('ms_per_step'), d['with_exact_shortcuts']['ms_per_step'])"
