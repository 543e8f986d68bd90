"""Debug helper: compares an attention variant against the fp32 reference and prints where it differs."""
import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import actionmesh_amd._lib as L
if len(sys.argv) > 2:
    L.LIB_PATH = os.path.abspath(sys.argv[2])
from actionmesh_amd import ops
from test_kernels_gpu import _layout, _sdpa_ref, _randn

dev = torch.device("cuda:0")
defer = int(sys.argv[1]) if len(sys.argv) > 1 else 68
shapes = [(1, 1, 256, 64 * n, 1) for n in (1, 2, 3, 4, 5, 6, 7, 9, 11)] + [(1, 2, 300, 300, 1), (2, 2, 520, 1040, 4)]
for (nseq, H, sq, sk, nch) in shapes:
    q = _randn((nseq, H, sq, 128), 1, dev).to(torch.bfloat16)
    k = _randn((nseq, H, sk, 128), 2, dev).to(torch.bfloat16)
    v = _randn((nseq, H, sk, 128), 3, dev).to(torch.bfloat16)
    Q, K, Vt, skc = _layout(q, k, v, nch)
    ref = _sdpa_ref(q, k, v).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    res = []
    for rep in range(6):
        out = torch.full((nseq * sq, H * 128), 777.0, dtype=torch.bfloat16, device=dev)
        ops.attention(Q, K, Vt, sq, skc, nchunks=nch, defer_log2=defer, out=out)
        torch.cuda.synchronize()
        o = out.float()
        unw = (o == 776.0) | (o == 777.0) | (o == 778.0)
        nan = torch.isnan(o) | torch.isinf(o)
        wrong = ((o - ref).abs() > 2e-2) & ~unw & ~nan
        f = lambda m: (m.any(dim=1).nonzero().flatten().tolist()[:2], int(m.any(dim=1).sum()), int(m.any(dim=0).sum()))
        res.append(("unw", f(unw), "nan", f(nan), "wrong", f(wrong)) if (unw.any() or nan.any() or wrong.any()) else "ok")
    print(f"sq={sq} sk={sk} H={H} nch={nch}:", res)
