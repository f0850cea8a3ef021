"""Throughput of the rectification kernel and of colour input on device-resident frames (information only)."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
vsa = importlib.import_module("visual-slam_amd")
from test_remap_gpu import rectification_like_maps

w, h, B = 752, 480, 256
mx, my = rectification_like_maps(w, h, w, h, 5)
rm = vsa.Remapper(mx, my, w, h, device=0)
raw = torch.from_numpy(np.stack([vsa.synth_frame(i % 8, w, h) for i in range(B)])).cuda()
rect = torch.empty_like(raw)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rm.remap_batch_device(raw.data_ptr(), w, w * h, B, rect.data_ptr(), w, w * h, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rm.remap_batch_device(raw.data_ptr(), w, w * h, B, rect.data_ptr(), w, w * h, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"remap {B} x {w}x{h}: {ms*1e3:.0f} us, {B/ms*1e3:.0f} frames/s, {B*w*h*(2+6)/ms/1e6:.0f} GB/s (src + dst + 6 B/px tables)")

# extraction from grey vs from BGR frames
for fmt in ("gray", "bgr", "bgra"):
    w2, h2 = 640, 480
    ex = vsa.Extractor(vsa.make_params(), w2, h2, max_batch=B, device=0)
    ex.set_input_format(fmt)
    ch = ex.channels
    imgs = torch.from_numpy(np.stack([np.repeat(vsa.synth_frame(i % 8, w2, h2)[:, :, None], ch, axis=2) for i in range(B)])).cuda()
    cap = ex.cap
    d_k = torch.zeros(B * cap * 7, dtype=torch.float32, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ex.extract_batch_device(imgs.data_ptr(), w2 * ch, w2 * h2 * ch, B, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ex.extract_batch_device(imgs.data_ptr(), w2 * ch, w2 * h2 * ch, B, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"extract {fmt}: {ms:.3f} ms per {B} frames, {B/ms*1e3:.0f} frames/s")
    ex.close()
