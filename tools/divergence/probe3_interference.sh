#!/bin/bash
# Third probe: which co-running work moves head_post's Q / K rows (interference_probe.py), with the product library and with the
# DPP-reduction build of head_post; then the peer selftest on both builds with an element-level picture of the moved K rows.
N=${1:-8}
OUT=gpurun_out/r03c_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/build/variants
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
{
echo "=== I1: interference probe, product library"
T0=$(( $(date +%s) + 30 ))
timeout 200 python tools/divergence/interference_probe.py victim $T0 2>&1 | grep "victim\|Error" &
timeout 200 python tools/divergence/interference_probe.py aggressor $T0 2>&1 | grep "aggressor\|Error" &
wait
echo "=== I2: interference probe, victim on the DPP build of head_post"
T0=$(( $(date +%s) + 30 ))
ACTIONMESH_AMD_LIB=$V/libam_hpdpp.so timeout 200 python tools/divergence/interference_probe.py victim $T0 2>&1 | grep "victim\|Error" &
timeout 200 python tools/divergence/interference_probe.py aggressor $T0 2>&1 | grep "aggressor\|Error" &
wait
peer() {
  tag=$1; runs=$2; shift 2
  bad=0
  for i in $(seq 1 $runs); do
    o=$(timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      tools/peer_selftest.py --same-device --forwards 4 "$@" 2>&1 | grep "kernel checksums differ from forward 0:\|K row\|K shard differs\|Error" | cut -c1-420)
    if [ -n "$o" ]; then bad=$((bad+1)); echo "--- $tag run $i"; echo "$o" | head -12; fi
  done
  echo "$tag: $bad of $runs runs had a divergent forward"
}
echo "=== P1: peer selftest --ktrace, product"
peer product $N --ktrace
echo "=== P2: peer selftest --ktrace, DPP build of head_post"
ACTIONMESH_AMD_LIB=$V/libam_hpdpp.so peer hpdpp $N --ktrace
} >> $OUT 2>&1
tail -100 $OUT
