#!/usr/bin/env python3
"""Throughput of the EuRoC-shape stereo front end (BASELINE configs[1]): left + right extraction of 752x480 pairs on two
extractor contexts / streams, stereo row search and sub-pixel refinement, everything resident in HBM.  Prints pairs/s."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
vsa = importlib.import_module("visual-slam_amd")
w, h, B, steps = 752, 480, 128, 20
p = vsa.make_params()
dev = torch.device("cuda:0")
pairs = [vsa.synth_stereo(30 + i, w, h) for i in range(16)]
L = torch.from_numpy(np.stack([q[0] for q in pairs])).to(dev).repeat(B // 16, 1, 1).contiguous()
R = torch.from_numpy(np.stack([q[1] for q in pairs])).to(dev).repeat(B // 16, 1, 1).contiguous()
exl, exr = vsa.Extractor(p, w, h, max_batch=B, device=0), vsa.Extractor(p, w, h, max_batch=B, device=0)
m = vsa.Matcher(device=0)
cap = exl.cap
mk = lambda: (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
              torch.zeros(B, dtype=torch.int32, device=dev))
kl, dl, nl = mk(); kr, dr, nr = mk()
bd = torch.zeros((B, cap), dtype=torch.int32, device=dev); bi = torch.zeros_like(bd); sad = torch.zeros_like(bd)
ur = torch.zeros((B, cap), dtype=torch.float32, device=dev); dp = torch.zeros_like(ur)
scale = vsa.scale_tables(p)["scale"]
bf, fx = 47.90639384423901, 435.2046959714599
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def step():
    exl.extract_batch_device(L.data_ptr(), w, w * h, B, kl.data_ptr(), dl.data_ptr(), nl.data_ptr(), cap, s0.cuda_stream)
    exr.extract_batch_device(R.data_ptr(), w, w * h, B, kr.data_ptr(), dr.data_ptr(), nr.data_ptr(), cap, s1.cuda_stream)
    s0.wait_stream(s1)
    m.stereo_match_batch_device(kl.data_ptr(), dl.data_ptr(), nl.data_ptr(), kr.data_ptr(), dr.data_ptr(), nr.data_ptr(), cap, B, scale, h,
                                0.0, fx, bd.data_ptr(), bi.data_ptr(), s0.cuda_stream)
    exl.stereo_refine_batch_device(exr, kl.data_ptr(), nl.data_ptr(), kr.data_ptr(), cap, B, bd.data_ptr(), bi.data_ptr(), bf, 0.0, fx,
                                   ur.data_ptr(), dp.data_ptr(), sad.data_ptr(), s0.cuda_stream)
    s1.wait_stream(s0)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"stereo pairs/s {B * steps / dt:.0f}  ms/step {1e3 * dt / steps:.3f}  matched per pair {(ur[0] >= 0).sum().item()} of {int(nl[0])}")
