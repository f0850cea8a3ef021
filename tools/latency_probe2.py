import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
vsa = importlib.import_module("visual-slam_amd")
w,h,nf=640,480,1000
img = vsa.synth_frame(3, w, h)
ex = vsa.Extractor(vsa.make_params(nf), w, h, max_batch=1, device=0)
dev=torch.device("cuda:0")
d_img = torch.from_numpy(img).to(dev)
cap=ex.cap
k = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev); d = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev); c = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(5): ex.extract_batch_device(d_img.data_ptr(), w, w*h, 1, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, s)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200):
    ex.extract_batch_device(d_img.data_ptr(), w, w*h, 1, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, s)
    torch.cuda.synchronize()
print("device-resident single frame incl sync: %.1f us" % ((time.perf_counter()-t0)/200*1e6))
t0=time.perf_counter()
for _ in range(200):
    ex.extract_batch_device(d_img.data_ptr(), w, w*h, 1, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, s)
torch.cuda.synchronize()
print("device-resident single frame back-to-back: %.1f us" % ((time.perf_counter()-t0)/200*1e6))
pin = torch.from_numpy(img).pin_memory()
t0=time.perf_counter()
for _ in range(200):
    d_img.copy_(pin, non_blocking=True); torch.cuda.synchronize()
print("pinned H2D 307KB + sync: %.1f us" % ((time.perf_counter()-t0)/200*1e6))
pg = torch.from_numpy(img)
t0=time.perf_counter()
for _ in range(200):
    d_img.copy_(pg); torch.cuda.synchronize()
print("pageable H2D 307KB + sync: %.1f us" % ((time.perf_counter()-t0)/200*1e6))
