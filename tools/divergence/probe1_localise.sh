#!/bin/bash
# One GPU call that localises the same-device run-to-run divergence of the sharded forward:
#   A  every kernel of the sharded layer ALONE, repeated, in two concurrent processes (no exchange, no IPC)
#   B  the two-pass forward emulated by two engines, in two concurrent processes (no exchange, no IPC)
#   C  the cross-process copy-engine selftest with a checksum behind every kernel (which kernel's output moves first?)
#   D  the same with A/B builds of the 4x64 attention kernel (full vmcnt drain; compiler-issued LDS-DMA pieces)
# usage: tools/divergence_probe.sh [runs_per_case]
N=${1:-6}
OUT=gpurun_out/r03a_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/build/variants
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
{
echo "=== A: op_determinism x2 concurrent (product)"
for i in 0 1; do timeout 300 python tools/divergence/op_determinism.py --reps 400 --big 2>&1 | grep "op_determinism\|Error" & done; wait
echo "=== B: twopass_determinism x2 concurrent"
for i in 0 1; do timeout 300 python tools/twopass_determinism.py 2>&1 | grep "twopass\|Error" & done; wait
peer() {   # peer <tag> <runs> [selftest args]
  tag=$1; runs=$2; shift 2
  for i in $(seq 1 $runs); do
    echo "--- $tag run $i"
    timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      tools/peer_selftest.py --same-device --forwards 4 "$@" 2>&1 | grep "kernel checksums\|differ, max\|Error\|peer_selftest\] ok" | cut -c1-400
  done
}
echo "=== C: peer selftest --ktrace (product lib)"
peer product $N --ktrace
echo "=== C2: peer selftest --ktrace --defer 28 (exact attention kernel)"
peer exact 3 --ktrace --defer 28
echo "=== D1: vm0 build (vmcnt(0) in front of every tile barrier)"
ACTIONMESH_AMD_LIB=$V/libam_vm0.so peer vm0 $N --ktrace
echo "=== D2: nosaddr build (LDS-DMA pieces through the compiler builtin)"
ACTIONMESH_AMD_LIB=$V/libam_nosaddr.so peer nosaddr $N --ktrace
} >> $OUT 2>&1
tail -120 $OUT
