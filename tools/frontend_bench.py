#!/usr/bin/env python3
"""Throughput of the whole device-resident stereo front end at the EuRoC shape: raw pair -> rectification (remap) ->
left + right extraction (two contexts / streams) -> stereo row search + sub-pixel refinement -> keypoint undistortion
(identity camera after rectification, so a copy, as in the reference) -> vocabulary descent (k = 10, L = 6).
Prints pairs/s with and without the steps around extraction."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
vsa = importlib.import_module("visual-slam_amd")
from test_remap_gpu import rectification_like_maps

w, h, B, steps = 752, 480, 128, 20
p = vsa.make_params(1200)
dev = torch.device("cuda:0")
pairs = [vsa.synth_stereo(30 + i, w, h) for i in range(16)]
L = torch.from_numpy(np.stack([q[0] for q in pairs])).to(dev).repeat(B // 16, 1, 1).contiguous()
R = torch.from_numpy(np.stack([q[1] for q in pairs])).to(dev).repeat(B // 16, 1, 1).contiguous()
Lr, Rr = torch.empty_like(L), torch.empty_like(R)
mxl, myl = rectification_like_maps(w, h, w, h, 1); mxr, myr = rectification_like_maps(w, h, w, h, 2)
rml, rmr = vsa.Remapper(mxl, myl, w, h, device=0), vsa.Remapper(mxr, myr, w, h, device=0)
exl, exr = vsa.Extractor(p, w, h, max_batch=B, device=0), vsa.Extractor(p, w, h, max_batch=B, device=0)
m = vsa.Matcher(device=0)
voc = vsa.Vocabulary(10, 6, *vsa.synth_vocabulary(3, 10, 6), device=0)
cap = exl.cap
mk = lambda: (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
              torch.zeros(B, dtype=torch.int32, device=dev))
kl, dl, nl = mk(); kr, dr, nr = mk()
kun = torch.zeros_like(kl)
bd = torch.zeros((B, cap), dtype=torch.int32, device=dev); bi = torch.zeros_like(bd); sad = torch.zeros_like(bd)
word = torch.zeros_like(bd); node = torch.zeros_like(bd)
ur = torch.zeros((B, cap), dtype=torch.float32, device=dev); dp = torch.zeros_like(ur)
scale = vsa.scale_tables(p)["scale"]
bf, fx = 47.90639384423901, 435.2046959714599
cam = vsa.Camera(fx, fx, 367.2, 252.2, 0.0, 0.0, 0.0, 0.0, 0.0)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def step(full):
    li, ri = (Lr, Rr) if full else (L, R)
    if full:
        rml.remap_batch_device(L.data_ptr(), w, w * h, B, Lr.data_ptr(), w, w * h, s0.cuda_stream)
        rmr.remap_batch_device(R.data_ptr(), w, w * h, B, Rr.data_ptr(), w, w * h, s1.cuda_stream)
    exl.extract_batch_device(li.data_ptr(), w, w * h, B, kl.data_ptr(), dl.data_ptr(), nl.data_ptr(), cap, s0.cuda_stream)
    exr.extract_batch_device(ri.data_ptr(), w, w * h, B, kr.data_ptr(), dr.data_ptr(), nr.data_ptr(), cap, s1.cuda_stream)
    s0.wait_stream(s1)
    m.stereo_match_batch_device(kl.data_ptr(), dl.data_ptr(), nl.data_ptr(), kr.data_ptr(), dr.data_ptr(), nr.data_ptr(), cap, B, scale, h,
                                0.0, fx, bd.data_ptr(), bi.data_ptr(), s0.cuda_stream)
    exl.stereo_refine_batch_device(exr, kl.data_ptr(), nl.data_ptr(), kr.data_ptr(), cap, B, bd.data_ptr(), bi.data_ptr(), bf, 0.0, fx,
                                   ur.data_ptr(), dp.data_ptr(), sad.data_ptr(), s0.cuda_stream)
    if full:
        exl.undistort_keypoints_batch_device(cam, kl.data_ptr(), nl.data_ptr(), cap, B, kun.data_ptr(), s0.cuda_stream)
        voc.descend_device(dl.data_ptr(), nl.data_ptr(), cap, B, 4, word.data_ptr(), node.data_ptr(), s0.cuda_stream)
    s1.wait_stream(s0)


for full in (False, True):
    for _ in range(3):
        step(full)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(full)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    what = "remap + extract x2 + stereo match/refine + undistort + BoW descent" if full else "extract x2 + stereo match/refine"
    print(f"{what}: {B * steps / dt:.0f} pairs/s, {1e3 * dt / steps:.3f} ms per {B} pairs; matched per pair {(ur[0] >= 0).sum().item()} of {int(nl[0])}")
