#!/bin/bash
# round 4: fp8 / fp8_fast headline lines on the final sources
mkdir -p gpurun_out
for dt in fp8 fp8_fast; do
  timeout 200 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r04af_bench_headline_$dt.json
  python -c "
import json; d=json.load(open('gpurun_out/r04af_bench_headline_$dt.json'))
print('$dt', {k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_dtype_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['with_exact_shortcuts']['ms_per_step'])"
done
