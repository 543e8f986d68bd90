#!/bin/bash
# round 4: the GPU suite on the final sources, minus the slow BASELINE-architecture file (three of its cases run separately below)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04ac_gputest_final.txt
timeout 900 python -m pytest tests -q -m gpu --ignore=tests/test_baseline_arch_gpu.py 2>&1 | tail -5 > $O
timeout 600 python -m pytest tests/test_baseline_arch_gpu.py -q -k "(arch_headline and not 50 and not peaky and not spiky) or full_nominal" 2>&1 | tail -3 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> $O
cat $O
