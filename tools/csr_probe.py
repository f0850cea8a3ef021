import importlib, sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.')
vsa = importlib.import_module("visual-slam_amd")
m = vsa.Matcher(device=0)
nq, nt, per = 2000, 2000, 50
rng = np.random.default_rng(0)
q = rng.integers(0,256,(nq,32),dtype=np.uint8); t = rng.integers(0,256,(nt,32),dtype=np.uint8)
rowptr = (np.arange(nq+1)*per).astype(np.int32); cand = rng.integers(0,nt,nq*per).astype(np.int32); dist = np.zeros(nq*per,np.int32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
f = lambda: vsa.lib.orbx_hamming_csr(m.h, P(q), nq, P(t), nt, P(rowptr), P(cand), P(dist))
f()
t0=time.perf_counter()
for _ in range(50): f()
print("orbx_hamming_csr 2000 rows x 50: %.0f us" % ((time.perf_counter()-t0)/50*1e6))
