import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
vsa = importlib.import_module("visual-slam_amd")
for (w,h,nf) in ((640,480,1000),(752,480,1200),(1241,376,2000)):
    img = vsa.synth_frame(3, w, h)
    ex = vsa.Extractor(vsa.make_params(nf), w, h, max_batch=1, device=0)
    for _ in range(5): ex.extract(img)
    t0=time.perf_counter()
    for _ in range(200): k,d = ex.extract(img)
    dt=(time.perf_counter()-t0)/200
    print(w,h,nf,"single-frame host-to-host latency %.1f us (%d kp)"%(dt*1e6, len(k)))
    ex.close()
