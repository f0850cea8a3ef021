#!/usr/bin/env python3
"""FAST stage time against the second-threshold setting: (ini, min) = (20, 7) as in the settings files, (20, 20), which
never makes the second attempt, and (7, 7).  The difference between the first two is what the cells that stay empty at the
first threshold cost."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")

B, W, H = 256, 640, 480
uniq = np.stack([vsa.synth_frame(i, W, H) for i in range(32)])
frames = torch.from_numpy(uniq).cuda().repeat(B // 32, 1, 1).contiguous()
stream = torch.cuda.current_stream().cuda_stream
for ini, mn in ((20, 7), (20, 20), (7, 7)):
    ex = vsa.Extractor(vsa.make_params(1000, 1.2, 8, ini, mn), W, H, max_batch=B, device=0)
    cap = ex.cap
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.set_profiling(True)
    acc = np.zeros(len(vsa.STAGES))
    for it in range(6):
        ex.extract_batch_device(frames.data_ptr(), W, W * H, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
        torch.cuda.synchronize()
        if it:
            acc += ex.stage_ms()
    print(f"ini {ini} min {mn}: " + ", ".join(f"{n} {v / 5 * 1e3:.0f} us" for n, v in zip(vsa.STAGES, acc)), f"mean keypoints {counts.float().mean().item():.0f}")
    ex.close()
