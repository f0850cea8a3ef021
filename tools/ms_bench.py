import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
vsa = importlib.import_module("visual-slam_amd")
W,H,B=640,480,256
p = vsa.make_params()
dev = torch.device("cuda:0")
uniq = np.stack([vsa.synth_frame(i, W, H) for i in range(32)])
frames = torch.from_numpy(uniq).to(dev).repeat(8,1,1).contiguous()
for S in (1,2,4):
    b = B//S
    exs = [vsa.Extractor(p, W, H, max_batch=b, device=0) for _ in range(S)]
    cap = exs[0].cap
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream() for _ in range(S)]
    def step():
        for s in range(S):
            exs[s].extract_batch_device(frames[s*b:].data_ptr(), W, W*H, b, kps[s*b:].data_ptr(), desc[s*b:].data_ptr(), counts[s*b:].data_ptr(), cap, streams[s].cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print("streams",S,"fps",int(B*20/dt), "ms/step", 1e3*dt/20, "mean kp", float(counts.float().mean()))
    for e in exs: e.close()
