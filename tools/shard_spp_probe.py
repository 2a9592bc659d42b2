"""Predicted parallel efficiency of the hall scene (bench.py shard_cost) at 32 and at 128 spp: python tools/shard_spp_probe.py (through gpurun)"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
import torch
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for spp in (32, 128):
        r = bench.shard_cost(lambda: bench.make_hall(0, s.cuda_stream)[0], spp, bench.DEPTH, reps=2)
        print(spp, json.dumps(r), flush=True)
