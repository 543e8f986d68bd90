#!/bin/bash
# round 4, end-of-round evidence for the final build: tools/final_run.sh (full GPU suite, smoke, bench line, rocprofv3 trace + PMC +
# traffic passes named by the sources sha, long64 fp8 lines, e2e_synthetic) + the fp8 / fp8_fast headline lines
mkdir -p gpurun_out
bash tools/final_run.sh r04 2>&1 | tail -40
for dt in fp8 fp8_fast; do
  timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r04_bench_headline_$dt.json
  python -c "
import json; d=json.load(open('gpurun_out/r04_bench_headline_$dt.json'))
print('$dt', {k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_dtype_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'])"
done
