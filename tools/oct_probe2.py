#!/usr/bin/env python3
"""Standalone quadtree on the level-0 candidates of one 3840x2160 frame, against the quota: the time between two quotas
is the cost of the extra split steps (each is two passes over the keys)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")
w, h = 3840, 2160
p = vsa.make_params(8000, 1.2, 12)
ex = vsa.Extractor(p, w, h, 1, 0)
ex.extract(vsa.synth_frame(0, w, h))
keys = ex.debug_candidates(0, 0)
print(len(keys), "keys")
for st in (1,):
    for N in (1, 2, 8, 32, 128, 400, 512, 1502, 3000):
        ts = []
        for _ in range(12):
            t = time.perf_counter()
            out = vsa.distribute_octree(keys, 16, w - 16, 16, h - 16, N, device=0, key_storage=st)
            ts.append(time.perf_counter() - t)
        print(f"storage {st} N {N}: {np.median(ts) * 1e6:.0f} us, {len(out)} selected", flush=True)
