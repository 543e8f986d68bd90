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
