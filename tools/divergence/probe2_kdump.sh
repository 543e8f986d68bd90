#!/bin/bash
# Second probe (after r03a localised the divergence to head_post's Q / K outputs on bit-identical qkv inputs):
#   R  the standalone reproducer: two processes, same virtual address, different contents (tools/repro/vmid_l1_alias.hip)
#   K  the peer selftest with the local K shard dumped behind every layer_pre: which tokens moved, and do they carry the OTHER
#      rank's RoPE angle (the ranks' cos / sin tables sit at the same virtual address and differ)?
#   S  the peer selftest with the ranks' allocations shifted apart (--va-shift)
N=${1:-6}
OUT=gpurun_out/r03b_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "=== R1: reproducer, same VA"
for r in 1 2 3; do tools/repro/vmid_l1_alias 0 & tools/repro/vmid_l1_alias 1 & wait; done
echo "=== R2: reproducer, rank 1 shifted by 64 MiB"
for r in 1 2; do tools/repro/vmid_l1_alias 0 & tools/repro/vmid_l1_alias 1 shift & wait; done
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
peer() {
  tag=$1; runs=$2; shift 2
  for i in $(seq 1 $runs); do
    echo "--- $tag run $i"
    timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      tools/peer_selftest.py --same-device --forwards 4 "$@" 2>&1 | grep "kernel checksums\|K shard differs\|Error\|peer_selftest\] ok" | cut -c1-420
  done
}
echo "=== K: peer selftest --ktrace with K dumps"
peer product $N --ktrace
echo "=== S: peer selftest --ktrace --va-shift"
peer shifted $((N + 4)) --ktrace --va-shift
} >> $OUT 2>&1
tail -150 $OUT
