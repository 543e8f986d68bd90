#!/usr/bin/env python
"""Static check of the 4x64 attention kernel's hand placement: how many VALU / exp / LDS / DMA instructions hipcc left in
each MFMA gap of the main loop (the source threads softmax steps through the gaps; the machine scheduler is free to move
un-pinned instructions inside a sched_barrier region, e.g. above the region's MFMA).
    python tools/attn64_gaps.py actionmesh_amd/csrc/am_attention64.s [mangled-substring]   (default: the product lazy kernel)"""
import re
import sys


def kernel_body(text, key):
    t = text.split("\n")
    start = [i for i, l in enumerate(t) if re.match(r"_ZN.*attn_fwd64_kernel.*:", l) and key in l][0]
    end = [i for i, l in enumerate(t) if i > start and l.strip().startswith("s_endpgm")][0]
    return t[start:end + 1]


def gaps(body):
    out, cur, run = [], {}, []
    for l in body:
        c = l.split(";")[0].strip()
        if not c or c.startswith(".") or c.startswith("#"):
            if re.match(r"\.LBB\d+_\d+:", c):
                run.append(("label", c))
            continue
        op = c.split()[0]
        if op.startswith("v_mfma"):
            run.append(("gap", cur)); cur = {}
            run.append(("mfma", c))
            continue
        k = ("exp" if op.startswith("v_exp") else "acc" if "accvgpr" in op else "valu" if op.startswith("v_") else
             "ds" if op.startswith("ds_") else "dma" if op.startswith("global_load_lds") else
             "vmem" if op.startswith(("global_", "buffer_")) else "br" if op.startswith(("s_cbranch", "s_branch")) else
             "bar" if op.startswith("s_barrier") else "salu")
        cur[k] = cur.get(k, 0) + 1
    run.append(("gap", cur))
    return run


def main():
    text = open(sys.argv[1]).read()
    key = sys.argv[2] if len(sys.argv) > 2 else "ILi8ELi0ELb0ELi0ELb1E"
    run = gaps(kernel_body(text, key))
    # print the stretch between the first barrier-bearing gap of the main loop and the next 128 MFMAs
    n = 0
    started = False
    model = []
    for kind, v in run:
        if kind == "gap" and v.get("bar") and not started and n > 40:
            started = True; n0 = n
        if kind == "mfma":
            n += 1
        if started and kind == "gap":
            slots = v.get("valu", 0) + 2 * v.get("exp", 0) + v.get("acc", 0)
            # issue-cycle estimate of the gap (one wave per SIMD: 4 cycles per issue slot; v_exp_f32 two slots; a ds_read_b128
            # ~10 cycles and an LDS-DMA piece ~50 beside a busy stream) against the 32 cycles its MFMA covers
            est = 4 + 4 * (slots + v.get("salu", 0)) + 10 * v.get("ds", 0) + 50 * v.get("dma", 0)
            model.append((n - n0, est))
            print(f"after mfma {n - n0:3d}: slots {slots:2d}  est {est:3d} cyc  {v}")
        if started and n - n0 >= 130:
            break
    for name, lo in (("P.V phase (mfma 1..32)", 1), ("QK^T phase (mfma 33..64)", 33)):
        gs = [e for i, e in model if lo <= i < lo + 32 and e < 600]
        print(f"{name}: sum of max(32, est) = {sum(max(32, e) for e in gs)} cycles over {len(gs)} gaps "
              f"({sum(1 for e in gs if e > 32)} over the 32-cycle budget; MFMA time alone 1024)")


if __name__ == "__main__":
    main()
