#!/usr/bin/env python
"""Turn the rocprofv3 (rocpd sqlite) outputs of tools/gpu_profile.sh into committed summaries.
    python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>
writes <name>_kernel_stats.csv (per-kernel calls / total / avg / %), <name>_by_launch_shape.csv, <name>_dominant_kernel.csv (the dominant
kernel's FULL-GRID average apart from its mixed-grid average) and <name>_pmc.csv."""
import csv
import glob
import os
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n if len(n) < 110 else n[:107] + "..."


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    db = os.path.join(src, "bench", "bench_results.db")
    if os.path.exists(db):
        c = sqlite3.connect(db)
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        with open(dst + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
            for r in rows:
                w.writerow([short(r[0]), r[1], int(r[2]), int(r[3]), round(r[4], 3)])
        print("wrote", dst + "_kernel_stats.csv", len(rows), "kernels")
        q = ("select name, grid_x, grid_y, workgroup_x, vgpr_count, lds_size, count(*), avg(end-start), min(end-start), "
             "max(end-start) from kernels group by name, grid_x, grid_y order by sum(end-start) desc limit 24")
        with open(dst + "_by_launch_shape.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_x", "grid_y", "block", "vgprs", "lds_bytes", "launches", "avg_ms", "min_ms", "max_ms"])
            for r in c.execute(q):
                w.writerow([short(r[0]), r[1], r[2], r[3], r[4], r[5], r[6], round(r[7] / 1e6, 4), round(r[8] / 1e6, 4),
                            round(r[9] / 1e6, 4)])
        print("wrote", dst + "_by_launch_shape.csv")
        # The dominant kernel by its launch grids (VERDICT r03 next #7c): top_kernels averages every launch of a name - the headline run
        # mixes the full grid with the half grids of the exact-shortcut loop, which under-reports the launch the roofline is quoted on.
        dom = c.execute("select name from kernels group by name order by sum(end-start) desc limit 1").fetchone()
        if dom:
            grids = list(c.execute("select grid_x, grid_y, count(*), avg(end-start), sum(end-start) from kernels where name = ? "
                                   "group by grid_x, grid_y order by grid_x * grid_y desc", (dom[0],)))
            allavg = c.execute("select count(*), avg(end-start) from kernels where name = ?", (dom[0],)).fetchone()
            with open(dst + "_dominant_kernel.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "which", "grid_x", "grid_y", "launches", "avg_ms"])
                w.writerow([short(dom[0]), "FULL GRID (largest grid: the launch roofline.achieved is quoted on)", grids[0][0], grids[0][1], grids[0][2],
                            round(grids[0][3] / 1e6, 4)])
                for g in grids[1:]:
                    w.writerow([short(dom[0]), "other grid", g[0], g[1], g[2], round(g[3] / 1e6, 4)])
                w.writerow([short(dom[0]), "all launches mixed (what top_kernels / kernel_stats.csv shows)", "", "", allavg[0], round(allavg[1] / 1e6, 4)])
            print("wrote", dst + "_dominant_kernel.csv:", short(dom[0])[:60], "full grid avg", round(grids[0][3] / 1e6, 4), "ms over", grids[0][2],
                  "launches; mixed", round(allavg[1] / 1e6, 4), "ms over", allavg[0])
    out = []
    for d in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_results.db"))):
        c = sqlite3.connect(d)
        q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
             "group by kernel_name, counter_name")
        for kn, cn, v, n in c.execute(q):
            if any(t in kn for t in ("attn", "gemm", "layernorm", "head_post")):
                out.append([short(kn), cn, v, n])
    if out:
        with open(dst + "_pmc.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatches"])
            w.writerows(out)
        print("wrote", dst + "_pmc.csv", len(out), "rows")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
