"""Throughput of the vocabulary descent (k = 10, L = 6, the ORB vocabulary's shape) and of the distinctive-descriptor
selection on device-resident data (information only)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsa = importlib.import_module("visual-slam_amd")

k, L = 10, 6
parent, leaf, desc, weight = vsa.synth_vocabulary(3, k, L)
voc = vsa.Vocabulary(k, L, parent, leaf, desc, weight, device=0)
rng = np.random.default_rng(1)
leaves = np.flatnonzero(leaf)
for n in (1000, 256 * 1000, 1 << 20):
    feats = desc[rng.choice(leaves, n)] ^ np.packbits(rng.random((n, 256)) < 0.04, axis=1)
    d = torch.from_numpy(feats).cuda()
    w = torch.empty(n, dtype=torch.int32, device="cuda"); nd = torch.empty_like(w)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        voc.descend_device(d.data_ptr(), None, n, 1, 4, w.data_ptr(), nd.data_ptr(), st)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        voc.descend_device(d.data_ptr(), None, n, 1, 4, w.data_ptr(), nd.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = n * k * L
    print(f"descend n={n}: {ms*1e3:.1f} us, {n/ms/1e3:.2f} M features/s, {pairs/ms/1e6:.1f} G node distances/s")

m = vsa.Matcher(device=0)
for n_groups, max_n in ((1000, 14), (100_000, 14), (100_000, 40)):
    sizes = rng.integers(1, max_n + 1, n_groups)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rows = int(ptr[-1])
    dd = torch.from_numpy(vsa.synth_descriptors(5, rows)).cuda(); dp = torch.from_numpy(ptr).cuda()
    best = torch.empty(n_groups, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        m.distinctive_descriptors_device(dd.data_ptr(), dp.data_ptr(), n_groups, rows, best.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m.distinctive_descriptors_device(dd.data_ptr(), dp.data_ptr(), n_groups, rows, best.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    pairs = int((sizes.astype(np.int64) ** 2).sum())
    print(f"distinctive groups={n_groups} max_n={max_n}: {ms*1e3:.1f} us, {n_groups/ms/1e3:.2f} M points/s, {pairs/ms/1e6:.2f} G matrix entries/s")
