import torch, sys, math
sys.path.insert(0, ".")
from actionmesh_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
sq, sk = 256, 64
idx = ops.perm16_index(sk)
g = torch.Generator().manual_seed(0)
q = torch.randn(1,1,sq,128,generator=g).to(torch.bfloat16); k = torch.randn(1,1,sk,128,generator=g).to(torch.bfloat16); v = torch.randn(1,1,sk,128,generator=g).to(torch.bfloat16)
f8 = lambda x: x.clamp(-448,448).to(torch.float8_e4m3fn).float()
mul = 128**-0.5 * 1.4426950408889634
def run(qq, kk, vv):
    Vt = vv.transpose(-1,-2)[..., idx][None].contiguous().to(dev)
    return ops.attention_fp8(qq.to(dev), kk[None].to(dev), Vt, sq, sk).float().cpu()
# (1) random q,k ; V one-hot per key -> weights
V1 = torch.zeros(1,1,sk,128); 
for key in range(sk): V1[0,0,key,key] = 1.0
w = run(q, k, V1.to(torch.bfloat16))[:, :64]
s8 = (f8(q.float()*mul) @ f8(k.float()).transpose(-1,-2))[0,0]
w_exp8 = torch.softmax(s8 * math.log(2), -1)
w_true = torch.softmax((q.float() @ k.float().transpose(-1,-2))[0,0] / math.sqrt(128), -1)
print("weights: sum per row (expect 1):", w.sum(-1)[:6])
print("weights vs fp8-emulated softmax rel:", float((w - w_exp8).norm()/w_exp8.norm()), " vs true:", float((w - w_true).norm()/w_true.norm()))
r = 0
print("row0 got", w[r,:8]); print("row0 emu", w_exp8[r,:8])
# implied score scale: regress log w on s8
lw = torch.log2(w.clamp_min(1e-20)); a = ((lw - lw.mean(-1,keepdim=True)) * (s8 - s8.mean(-1,keepdim=True))).sum() / ((s8 - s8.mean(-1,keepdim=True))**2).sum()
print("slope of log2(w) on emulated log2-scores (expect 1):", float(a))
# (2) structured p (q one-hot) and random V
K2 = torch.zeros(1,1,sk,128)
for key in range(sk):
    for c in range(128): K2[0,0,key,c] = ((key * 7 + c * 3) % 5) - 2
A = 2.0 * math.sqrt(128) * math.log(2)
Q2 = torch.zeros(1,1,sq,128)
for rr in range(sq): Q2[0,0,rr,rr % 128] = A
o2 = run(Q2.to(torch.bfloat16), K2.to(torch.bfloat16), v)
ref2 = F.scaled_dot_product_attention(Q2, K2, f8(v.float()))[0,0]
print("structured p, random V: rel", float((o2-ref2).norm()/ref2.norm()), o2[0,:4], ref2[0,:4])
# (3) p with mantissa: scores in log2 units = 0.5*K  -> p = 2^(0.5 k) non powers of two
Q3 = Q2 / 4
o3 = run(Q3.to(torch.bfloat16), K2.to(torch.bfloat16), V1.to(torch.bfloat16))[:, :64]
ref3 = torch.softmax((Q3 @ K2.transpose(-1,-2))[0,0] / math.sqrt(128), -1)
print("fractional-exponent p, one-hot V: rel", float((o3-ref3).norm()/ref3.norm()), "row sums", o3.sum(-1)[:4])
print(" got", o3[0,:6], "\n exp", ref3[0,:6])
