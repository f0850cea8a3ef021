#!/usr/bin/env python3
"""Per-call time of the windowed matchers (host arrays in and out) next to the CPU oracle on the same inputs."""
import importlib, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
vsa = importlib.import_module("visual-slam_amd")
import oracle_lib as orc
w, h = 640, 480
a = vsa.synth_frame(12, w, h)
rng = np.random.default_rng(1)
b = np.clip(np.roll(a, (3, 5), axis=(0, 1)).astype(np.int32) + rng.integers(-2, 3, a.shape), 0, 255).astype(np.uint8)
oex = orc.Extractor(orc.make_params(2000))
k1, d1 = oex.extract(a); k2, d2 = oex.extract(b)
m = vsa.Matcher(device=0)
prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
bounds = (0.0, float(w), 0.0, float(h))
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("SearchForInitialization 2000x2000, window 100: gpu %.0f us, oracle %.0f us" % (
    t(lambda: m.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, True)),
    t(lambda: orc.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, True), 3)))
nodes = lambda d: ((d[:, 0].astype(np.int64) >> 3) << 5) | (d[:, 16].astype(np.int64) >> 3)
e1 = np.ones(len(k1), np.uint8)
print("SearchByBoW frame form: gpu %.0f us, oracle %.0f us" % (
    t(lambda: m.search_by_bow(d1, k1["angle"], e1, nodes(d1), d2, k2["angle"], None, nodes(d2), 0.7, True, 0)),
    t(lambda: orc.search_by_bow(d1, k1["angle"], e1, nodes(d1), d2, k2["angle"], None, nodes(d2), 0.7, True, 0), 3)))
table = vsa.synth_descriptors(42, 1 << 18); q, _ = vsa.synth_queries(43, table, 2000)
print("brute force 2000 x 262144: gpu %.0f us, oracle %.0f us" % (t(lambda: m.hamming_best2(q, table), 5), t(lambda: orc.hamming_best2(q[:100], table), 1) * 20))
