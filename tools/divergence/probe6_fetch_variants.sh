#!/bin/bash
# Sixth probe: the table-load reproducer (no library code); head_post variants (how the cos / sin tables are fetched) against a GEMM and
# against a whole denoiser forward running in another process; the peer selftest on the candidates.
N=${1:-8}
OUT=gpurun_out/r03f_divergence.txt
mkdir -p gpurun_out; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/build/variants
R=tools/repro/table_load_cross_process
{
echo "=== T1: table-load reproducer alone"
$R 3
echo "=== T2: table-load reproducer beside torch.matmul (bf16) of another process"
python - <<'PY' &
import torch, time
a = torch.randn(4096, 256, device="cuda").bfloat16(); b = torch.randn(256, 768, device="cuda").bfloat16()
t0 = time.time()
while time.time() - t0 < 16:
    for _ in range(200): a @ b
    torch.cuda.synchronize()
PY
sleep 7; $R 6; wait
export AM_PHASES=idle,gemm,forward
for lib in product rope1 rope2 rope4; do
  echo "=== X: two processes, victim library = $lib"
  T0=$(( $(date +%s) + 25 ))
  if [ $lib = product ]; then L=""; else L=$V/libam_$lib.so; fi
  ACTIONMESH_AMD_LIB=$L timeout 200 python tools/divergence/interference_probe.py victim $T0 2>&1 | grep "victim\|Error" | cut -c1-260 &
  timeout 200 python tools/divergence/interference_probe.py aggressor $T0 2>&1 | grep "Error" &
  wait
done
peer() {
  tag=$1; runs=$2; shift 2
  bad=0
  for i in $(seq 1 $runs); do
    o=$(timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      tools/peer_selftest.py --same-device --forwards 4 "$@" 2>&1 | grep "kernel checksums differ from forward 0:\|K row\|Error" | cut -c1-300)
    if [ -n "$o" ]; then bad=$((bad+1)); echo "--- $tag run $i"; echo "$o" | head -4; fi
  done
  echo "$tag: $bad of $runs runs had a divergent forward"
}
for lib in rope1 rope2 rope4; do
  echo "=== P: peer selftest --ktrace, $lib"
  ACTIONMESH_AMD_LIB=$V/libam_$lib.so peer $lib $N --ktrace
done
} >> $OUT 2>&1
cat $OUT
