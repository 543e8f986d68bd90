#!/usr/bin/env python
"""Reduce the --pmc passes of tools/gpu_profile.sh to the per-launch HBM traffic record bench.py quotes."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(src, shape, dst):
    import bench
    T, N, C, H = bench.SHAPES[shape][:4]
    per = {}      # counter -> {kernel: (sum, dispatches)}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*.db"), recursive=True)):
        c = sqlite3.connect(d)
        try:
            rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                  "group by kernel_name, counter_name"))
        except sqlite3.Error:
            continue
        for kn, cn, v, n in rows:
            if "attn" in kn:
                per.setdefault(cn, {})[kn] = (v, n)
    main_k = [k for k in per.get("FETCH_SIZE", {}) if "attn_fwd64_kernel" in k and "true>" in k.replace(" ", "")]
    main_k = main_k or [k for k in per.get("FETCH_SIZE", {}) if "attn_fwd64_kernel" in k]
    if not main_k:
        raise SystemExit(f"no attn_fwd64_kernel rows with FETCH_SIZE under {src}")
    launches = max(per["FETCH_SIZE"][k][1] for k in main_k)        # logical launches = dispatches of the main grid
    fetch_kb = sum(v for v, _ in per["FETCH_SIZE"].values()) / launches
    write_kb = sum(v for v, _ in per.get("WRITE_SIZE", {}).values()) / launches
    L = N + 1
    unique = 2 * (T * L) * H * 128 * 2 * 4          # Q, K, V read once + O written once, bf16, both CFG samples
    rec = {
        "kernel": "inflated self-attention launch = attn_fwd64_kernel (lazy) + its exact-fallback grid + split tail + combine",
        "shape": [T, N, H], "source_sha": bench.source_sha(),
        "source": "tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/kernel_bench.py --only attn",
        "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
        "gfx950_correction": "FETCH_SIZE doubled (64 B tallied per 128 B request on 16 B/lane streams, MI355X_MICROARCH.md HBM); WRITE_SIZE uncalibrated, as is",
        "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
        "algorithmic_unique_bytes_per_launch": unique, "logical_launches": launches,
        "other_counters": {cn: {k[:60]: v for k, (v, _) in d.items()} for cn, d in per.items() if cn not in ("FETCH_SIZE", "WRITE_SIZE")},
    }
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("traffic_bytes_per_launch", "algorithmic_unique_bytes_per_launch", "logical_launches", "source_sha")}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
