#!/bin/bash
# Fifth probe: the stand-alone reproducer (no library code) in its four settings + which packed-FP32 forms are affected; then the
# peer selftest on the rebuilt library (row-wise kernels without packed-FP32 instructions).
N=${1:-12}
OUT=gpurun_out/r03e_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=tools/repro/pk_fma_cross_process
{
echo "=== R1: victim alone"
$R victim 2
echo "=== R2: two processes: aggressor (bf16 MFMA loop) beside the victim"
$R aggressor 3 & sleep 1; $R victim 3; wait
echo "=== R3: one process, two streams"
$R both 2
echo "=== R4: two processes: torch.matmul (vendor bf16 GEMM) beside the victim"
python - <<'PY' &
import torch, time
a = torch.randn(4096, 256, device="cuda").bfloat16(); b = torch.randn(256, 768, device="cuda").bfloat16()
t0 = time.time()
while time.time() - t0 < 26:
    for _ in range(200): a @ b
    torch.cuda.synchronize()
print("[matmul aggressor] done")
PY
sleep 6; $R victim 3; wait
peer() {
  tag=$1; runs=$2; shift 2
  bad=0
  for i in $(seq 1 $runs); do
    o=$(timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      tools/peer_selftest.py --same-device --forwards 4 "$@" 2>&1 | grep "kernel checksums differ from forward 0:\|K row\|Error" | cut -c1-420)
    if [ -n "$o" ]; then bad=$((bad+1)); echo "--- $tag run $i"; echo "$o" | head -8; fi
  done
  echo "$tag: $bad of $runs runs had a divergent forward"
}
echo "=== P: peer selftest --ktrace, rebuilt product library"
peer product $N --ktrace
} >> $OUT 2>&1
cat $OUT
