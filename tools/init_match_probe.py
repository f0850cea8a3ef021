#!/usr/bin/env python3
"""SearchForInitialization: the per-call host form against the batched device form, on pairs taken straight from
orbx_extract_batch_device (640x480, 1000 features, 8 levels; and 2000 features, 1 level: every keypoint a query)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")
w, h = 640, 480
bounds = (0.0, float(w), 0.0, float(h))
st = torch.cuda.current_stream().cuda_stream
m = vsa.Matcher(device=0)
for nf, nl, P in ((1000, 8, 64), (1000, 8, 1024), (2000, 1, 64), (2000, 1, 512)):
    U = 16
    frames = []
    for p in range(U):
        a = vsa.synth_frame(700 + p, w, h)
        frames += [a, np.roll(a, (p % 5 - 2, p % 7 - 3), axis=(0, 1)).copy()]
    ex = vsa.Extractor(vsa.make_params(nf, 1.2, nl), w, h, max_batch=2 * U, device=0)
    cap = ex.cap
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    d_k = torch.zeros((2 * U, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((2 * U, cap, 32), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(2 * U, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(imgs.data_ptr(), w, w * h, 2 * U, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    rep = P // U
    k1 = d_k[0::2].repeat(rep, 1, 1).contiguous(); k2 = d_k[1::2].repeat(rep, 1, 1).contiguous()
    dd1 = d_d[0::2].repeat(rep, 1, 1).contiguous(); dd2 = d_d[1::2].repeat(rep, 1, 1).contiguous()
    n1 = d_c[0::2].repeat(rep).contiguous(); n2 = d_c[1::2].repeat(rep).contiguous()
    prev0 = k1[:, :, :2].contiguous(); prev = prev0.clone()
    m12 = torch.zeros((P, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    def run():
        m.search_for_initialization_batch_device(k1.data_ptr(), dd1.data_ptr(), n1.data_ptr(), k2.data_ptr(), dd2.data_ptr(), n2.data_ptr(), cap, P, bounds,
                                                 prev.data_ptr(), 100, 0.9, True, m12.data_ptr(), nm.data_ptr(), st)
    for _ in range(3):
        prev.copy_(prev0); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # host form, one pair per call
    hk1 = k1[0].cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(cap)[:int(n1[0])]; hk2 = k2[0].cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(cap)[:int(n2[0])]
    hd1 = dd1[0].cpu().numpy()[:len(hk1)]; hd2 = dd2[0].cpu().numpy()[:len(hk2)]
    hp = np.stack([hk1["x"], hk1["y"]], 1).astype(np.float32)
    for _ in range(5):
        m.search_for_initialization(hk1, hd1, hk2, hd2, bounds, hp, 100, 0.9, True)
    t = time.perf_counter()
    for _ in range(50):
        m.search_for_initialization(hk1, hd1, hk2, hd2, bounds, hp, 100, 0.9, True)
    host_us = (time.perf_counter() - t) / 50 * 1e6
    print(f"nfeatures {nf} levels {nl}: {P} pairs in {ms * 1e3:.0f} us = {P / ms * 1e3:.0f} pairs/s ({ms * 1e3 / P:.2f} us per pair; mean matches {nm.float().mean().item():.0f}); "
          f"host form {host_us:.0f} us per call = {1e6 / host_us:.0f} pairs/s", flush=True)
    ex.close()

# ---- the projection search (last-frame form): queries = the first frame's keypoints projected by the known shift
print("SearchByProjection(Frame, LastFrame) form:", flush=True)
for nf, nl, P in ((1000, 8, 64), (1000, 8, 1024)):
    U = 16
    frames = []
    for p in range(U):
        a = vsa.synth_frame(700 + p, w, h)
        frames += [a, np.roll(a, (2, 3), axis=(0, 1)).copy()]
    ex = vsa.Extractor(vsa.make_params(nf, 1.2, nl), w, h, max_batch=2 * U, device=0)
    cap = ex.cap
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    d_k = torch.zeros((2 * U, cap, 7), dtype=torch.float32, device="cuda"); d_d = torch.zeros((2 * U, cap, 32), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(2 * U, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(imgs.data_ptr(), w, w * h, 2 * U, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    rep = P // U
    k1 = d_k[0::2].repeat(rep, 1, 1).contiguous(); k2 = d_k[1::2].repeat(rep, 1, 1).contiguous()
    dd1 = d_d[0::2].repeat(rep, 1, 1).contiguous(); dd2 = d_d[1::2].repeat(rep, 1, 1).contiguous()
    n1 = d_c[0::2].repeat(rep).contiguous(); n2 = d_c[1::2].repeat(rep).contiguous()
    scale = torch.from_numpy(vsa.scale_tables(vsa.make_params(nf, 1.2, nl))["scale"]).cuda()
    qu = (k1[:, :, 0] + 3.0).contiguous(); qv = (k1[:, :, 1] + 2.0).contiguous()
    lvl = k1[:, :, 5].contiguous().view(torch.int32).contiguous()
    rad = (15.0 * scale[lvl.long().clamp(0, nl - 1)]).contiguous()
    ang = k1[:, :, 3].contiguous()
    cl0 = torch.full((P, cap), -1, dtype=torch.int32, device="cuda"); cl = cl0.clone(); nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    def run():
        m.search_by_projection_batch_device(0, qu.data_ptr(), qv.data_ptr(), rad.data_ptr(), None, lvl.data_ptr(), ang.data_ptr(), dd1.data_ptr(), None,
                                            n1.data_ptr(), cap, k2.data_ptr(), dd2.data_ptr(), None, n2.data_ptr(), cap, P, bounds, 0, 100, 0.0, True,
                                            cl.data_ptr(), nm.data_ptr(), st)
    for _ in range(3):
        cl.copy_(cl0); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(10):
        cl.copy_(cl0)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / 10
    a, b = int(n1[0]), int(n2[0])
    hk2 = k2[0].cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(cap)[:b]; hd2 = dd2[0].cpu().numpy()[:b]
    hq = [t[0, :a].cpu().numpy() for t in (qu, qv, rad)]; hl = lvl[0, :a].cpu().numpy(); ha = ang[0, :a].cpu().numpy(); hd1 = dd1[0, :a].cpu().numpy()
    c0 = np.full(b, -1, np.int32)
    for _ in range(5):
        m.search_by_projection_lastframe(hq[0], hq[1], hq[2], hq[0], hl, ha, hd1, hk2, hd2, None, bounds, 0, 100, True, c0)
    t = time.perf_counter()
    for _ in range(50):
        m.search_by_projection_lastframe(hq[0], hq[1], hq[2], hq[0], hl, ha, hd1, hk2, hd2, None, bounds, 0, 100, True, c0)
    host_us = (time.perf_counter() - t) / 50 * 1e6
    print(f"nfeatures {nf}: {P} calls of ~{a} queries x ~{b} features in {ms * 1e3:.0f} us = {P / ms * 1e3:.0f} calls/s (mean matches {nm.float().mean().item():.0f}); "
          f"host form {host_us:.0f} us per call = {1e6 / host_us:.0f} calls/s", flush=True)
    ex.close()
