#!/bin/bash
# round 4 mid-round validation: the full GPU suite, smoke, the bench line, fp8 / fp8_fast headline lines
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04k_gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a gpurun_out/r04k_gputest.txt
timeout 900 python bench.py --steps 5 --warmup 1 2>gpurun_out/r04k_bench.err | tail -1 > gpurun_out/r04k_bench_headline.json
python -c "
import json; d=json.load(open('gpurun_out/r04k_bench_headline.json'))
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline'].get('effective_clock_ghz'), d['roofline'].get('pipe_busy'), d['roofline'].get('clock_telemetry'), d.get('nominal',{}).get('ms_per_step'), d['cpu_baseline']['value'], d['cpu_baseline'].get('reference_full_shape'))"
for dt in fp8 fp8_fast; do
  timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r04k_bench_headline_$dt.json
  python -c "
import json; d=json.load(open('gpurun_out/r04k_bench_headline_$dt.json'))
print('$dt', {k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_dtype_peak','attention_probabilities')}, d['roofline']['launch_ms'], d['roofline']['frac'])"
done
