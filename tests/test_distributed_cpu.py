"""world_size-2 gloo rehearsal of the two multi-GPU paths (CPU only):
  * frames sharded over ranks, no collective: the union of per-rank results equals one pass over all frames;
  * map sharded over ranks, all-gather of the per-rank top-2 then the product's merge: equals one pass over
    the whole map.  On the GPU box each rank's shard result comes from the HIP kernel; here the oracle stands
    in for it so the exchange + merge logic runs without a device."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    vsa = importlib.import_module("visual-slam_amd")
    import oracle_lib as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # --- map sharding: contiguous row ranges, queries replicated
    rows, nq = 6000, 128
    table = vsa.synth_descriptors(42, rows)
    table[4000:4004] = table[10:14]                       # equal rows in both shards
    q, _ = vsa.synth_queries(43, table, nq)
    lo, hi = rank * rows // world, (rank + 1) * rows // world
    part = orc.hamming_best2(q, table[lo:hi])
    part["best_idx"] = np.where(part["best_idx"] >= 0, part["best_idx"] + lo, -1)
    mine = torch.from_numpy(part.view(np.int32).reshape(nq, 4).copy())
    gathered = torch.zeros((world * nq, 4), dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, mine)
    merged = vsa.best2_merge(gathered.numpy().view(vsa.BEST2_DTYPE).reshape(world, nq))
    whole = orc.hamming_best2(q, table)
    ok_map = all(np.array_equal(merged[f], whole[f]) for f in ("best", "best_idx", "second"))
    # --- frame sharding: rank r extracts frames r, r+world, ...; only counts are reduced for the report
    frames = [vsa.synth_frame(i, 160, 120) for i in range(4)]
    oex = orc.Extractor(orc.make_params(300, 1.2, 2))
    local = {i: oex.extract(frames[i]) for i in range(rank, 4, world)}
    n_local = torch.tensor([sum(len(k) for k, _ in local.values())])
    dist.all_reduce(n_local)
    total = sum(len(oex.extract(f)[0]) for f in frames)
    ok_frames = int(n_local.item()) == total
    ret[rank] = (ok_map, ok_frames)
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: (True, True), 1: (True, True)}
