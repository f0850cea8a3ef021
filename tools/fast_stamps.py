#!/usr/bin/env python3
"""Per-phase wave cycles of k_fast_cells from a build with -DORBX_FAST_STAMPS (tools/ab_build.sh stamps -DORBX_FAST_STAMPS):
ORBX_LIB_PATH=build_ab/stamps.so python tools/fast_stamps.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")
lib = C.CDLL(os.environ["ORBX_LIB_PATH"])
B, W, H = 256, 640, 480
uniq = np.stack([vsa.synth_frame(i, W, H) for i in range(32)])
frames = torch.from_numpy(uniq).cuda().repeat(B // 32, 1, 1).contiguous()
ex = vsa.Extractor(vsa.make_params(1000, 1.2, 8, 20, 7), W, H, max_batch=B, device=0)
cap = ex.cap
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
counts = torch.zeros(B, dtype=torch.int32, device="cuda")
out = (C.c_ulonglong * 16)()
names = sys.argv[1:] or ["tile", "clear", "pretest", "enqueue", "score", "nms", "emit", "empty stamp"]
for it in range(3):
    ex.extract_batch_device(frames.data_ptr(), W, W * H, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out[15] = 815 * B
    lib.orbx_debug_fast_stamps(out)
    n = out[15]
    tot = sum(out[i] for i in range(len(names)))
    print(f"cells {n}: " + ", ".join(f"{nm} {out[i] / n:.2f} ({100 * out[i] / tot:.0f}%)" for i, nm in enumerate(names)) + f" | total {tot / n:.0f} cycles per cell")
