"""The oracle port timed on THIS host at the three bounded-sample sizes of bench.py cpu_baseline (see oracle/cpu_rate_build_container.json)."""
import sys, time, torch, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from oracle import denoiser_oracle as O
T, N, C, H, NL, S, Dc, Din = bench.SHAPES["headline"]
hp = dict(in_channels=Din, num_layers=NL, num_attention_heads=H, width=C, mlp_ratio=4.0, cross_attention_dim=Dc, inflated_layers=list(range(NL)))
sd = bench.random_state_dict(hp, seed=0)
cfg = O.OracleConfig(**hp)
torch.set_num_threads(os.cpu_count())
B=2
pts=[]
for n in (256, 512, 1024):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, n, Din, generator=g); ctx = torch.randn(B, T, S, Dc, generator=g); ctx[0] = 0
    fs = torch.arange(T, dtype=torch.float32).repeat(B, 1); mask = torch.zeros(B, T); mask[:, 0] = 1; t = torch.full((B,), 500.0)
    with torch.no_grad():
        t0 = time.perf_counter(); O.denoiser_forward(sd, cfg, x, ctx, fs, t, mask); dt = time.perf_counter() - t0
    fl = O.step_flops(B, T, n, cfg, S)
    pts.append({"TL": T*(n+1), "seconds": round(dt,3), "tflops": round(fl/dt/1e12,3)})
    print(pts[-1], flush=True)
print(json.dumps({"threads": torch.get_num_threads(), "points": pts}))
