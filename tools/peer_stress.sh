#!/bin/bash
# repeat the cross-process copy-engine selftest: tools/peer_stress.sh <runs> <world> [extra args]
N=${1:-6}; W=${2:-2}; shift 2
ok=0; bad=0; exact=0
for i in $(seq 1 $N); do
  out=$(HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) tools/peer_selftest.py --same-device "$@" 2>&1)
  if ! echo "$out" | grep -q "vs 0: [1-9]"; then exact=$((exact+1)); fi
  if echo "$out" | grep -q "\[peer_selftest\] ok"; then ok=$((ok+1)); else bad=$((bad+1)); echo "$out" | grep "differ, max\|Error" | head -6; fi
done
echo "peer_stress world=$W lib=${ACTIONMESH_AMD_LIB:-product}: ok=$ok bad=$bad bitwise-repeatable=$exact/$N"
