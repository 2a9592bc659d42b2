"""Paths in flight per batch on the 262 k-triangle hall at its own 128 spp: default (64 Mi), 128 Mi, 32 Mi.  Round 4: 101.2 / 99.4 / 102.8 ms."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from capsaicin_amd import capi
r, bi = bench.make_hall(0, None)
for batch in (0, 128 << 20, 32 << 20):
    r.set_batch_paths(batch)
    r.render(0, 128, 8, 0)
    dt, st = bench.timed(r, 0, 128, 8, 0, 2)
    rays = (st.rays_primary + st.rays_extension + st.rays_shadow) / 2
    print("batch %d Mi paths: %.1f ms per 128 spp, %.2f Grays/s" % (batch >> 20, dt * 1e3, rays / dt / 1e9))
r.close()
