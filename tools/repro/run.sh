#!/bin/bash
# tools/repro/run.sh : the stand-alone reproducer alone, then - three builds - beside a bf16 GEMM loop of ANOTHER PROCESS
# (torch.matmul = the vendor's kernel).  Builds: as hipcc -O3 compiles it; -fno-slp-vectorize (scalar FMAs instead of v_pk_fma_f32);
# -DDELAY (64 idle cycles between the wait for the table loads and their first use).
cd "$(dirname "$0")"
for v in "" "-fno-slp-vectorize" "-DDELAY"; do
  n=rope_rows$(echo "$v" | tr -d ' -' | tr 'A-Z' 'a-z')
  [ -x $n ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $v -o $n rope_rows_cross_process.hip 2>/dev/null
done
[ -x pk_mul ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o pk_mul pk_mul_cross_process.hip 2>/dev/null
echo "=== alone (default build)"; ./rope_rows 4; ./pk_mul 2
python - <<'PY' &
import torch, time
a = torch.randn(4096, 256, device="cuda").bfloat16(); b = torch.randn(256, 768, device="cuda").bfloat16()
t0 = time.time()
while time.time() - t0 < 50:
    for _ in range(200): a @ b
    torch.cuda.synchronize()
PY
sleep 9
echo "=== beside torch.matmul (bf16) of another process: default build"; ./rope_rows 7
echo "=== same, -fno-slp-vectorize build"; ./rope_rowsfnoslpvectorize 7
echo "=== same, -DDELAY build"; ./rope_rowsddelay 7
echo "=== the minimal ALU-only form beside the same GEMM loop"; ./pk_mul 4
wait
