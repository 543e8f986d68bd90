#!/bin/bash
# round 4: the sharded forward's phase loop in C (am_forward_sharded_peer) against the Python loop, between real processes on one device
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_multi_gpu.py -q -s -k "phase_loop or copy_engine_exchange_across or fp8_shards_across" 2>&1 | grep -E "peer_selftest|passed|failed|Error|error" | cut -c1-220 | tail -30 > gpurun_out/r04ad_phase_loop.txt
cat gpurun_out/r04ad_phase_loop.txt
