#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer batch entry point (orbx_extract_batch: upload, extract, download)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")
w, h = 640, 480
for B in (1, 8, 16, 32, 64, 128, 256, 512):
    imgs = np.stack([vsa.synth_frame(i % 16, w, h) for i in range(B)])
    ex = vsa.Extractor(vsa.make_params(), w, h, max_batch=B, device=0)
    ex.extract_batch(imgs)
    n = max(10, 200 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        ex.extract_batch(imgs)
    dt = (time.perf_counter() - t0) / n
    print(f"host batch of {B}: {dt*1e3:.3f} ms per call, {B/dt:.0f} frames/s (PCIe in and out included)")
    ex.close()
