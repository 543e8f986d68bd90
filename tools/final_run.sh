#!/bin/bash
# End-of-round evidence for one build: full GPU suite, the bench line (with cpu_baseline + nominal), the profile set named by the sources sha.
TAG=${1:-r03}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/${TAG}_gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee -a gpurun_out/${TAG}_gputest.txt
python bench.py --steps 5 --warmup 1 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_headline.json
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json'))
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('nominal',{}).get('ms_per_step'), d['cpu_baseline']['value'], d['cpu_baseline']['fit']['max_relative_residual'])"
tools/gpu_profile.sh $TAG 2>&1 | tail -8
# BASELINE configs[4] (64 frames x 8192 tokens, fp8 attention): the whole job on ONE device, and rank 0's share of the 8-GPU run it is meant for
python bench.py --shape long64 --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-nominal --no-roofline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_long64_fp8.json
python bench.py --shape long64 --dtype fp8 --emulate-world 8 --steps 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_emulate8_long64_fp8.json
python -c "
import json
for f in ('bench_long64_fp8', 'emulate8_long64_fp8'):
    d = json.load(open('gpurun_out/${TAG}_' + f + '.json')); print(f, {k: d[k] for k in d if k in ('ms_per_step', 'value', 'dtype', 'rank0_ms_per_step', 'modelled_link_ms_per_layer', 'step_tflops_per_gpu')})"
python tools/e2e_synthetic.py 2>/dev/null | tail -1 | tee gpurun_out/${TAG}_e2e_synthetic.json | cut -c1-400
