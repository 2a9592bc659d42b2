#!/bin/bash
# Sweep of the paths-in-flight budget (cap_set_batch_paths) on the headline workload; run through gpurun.
for b in ${SWEEP:-8388608 16777216 33554432 67108864 134217728}; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tree-variant --batch-paths $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($b, round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stage_ms'].items()})"
done
