#!/bin/bash
# Stage split of the 262 k-triangle procedural scene (LBVH path); run through gpurun.
python bench.py --scene sponza --spp 16 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stage_ms'].items()})"
