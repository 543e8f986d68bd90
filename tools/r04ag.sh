#!/bin/bash
# round 4: what the captured-forward path (bench.py --graph, opt-in) is worth after this round's extra small kernels - one A/B
mkdir -p gpurun_out
O=gpurun_out/r04ag_graph_ab.txt
: > $O
for g in "" "--graph"; do
  echo "== bench.py $g" >> $O
  timeout 100 python bench.py $g --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-nominal 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','ms_per_step','hip_graph') if k in r}, r['with_exact_shortcuts']['ms_per_step'])" >> $O
done
cat $O
