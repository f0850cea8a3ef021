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
    # --- frame sharding: the library's partition gives each rank a contiguous run of the batch; the rank fills its run
    # of the batch-layout output (blocks of `cap` rows per frame, the library's capacity for this shape, plus a count per
    # frame) -- the oracle stands in for the device kernel -- and the runs are gathered without any reduction.  The
    # gathered buffers must equal one pass over the whole batch.
    n_frames, w, h = 5, 240, 180
    p = vsa.make_params(nfeatures=300, nlevels=2)
    cap = vsa.keypoint_capacity(p, w, h)
    fb, fe = vsa.shard_range(n_frames, rank, world)
    ranges = [vsa.shard_range(n_frames, r, world) for r in range(world)]
    ok_cover = ranges[0][0] == 0 and ranges[-1][1] == n_frames and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    oex = orc.Extractor(orc.make_params(300, 1.2, 2))
    frames = [vsa.synth_frame(i, w, h) for i in range(n_frames)]

    def run(lo, hi):
        counts = np.zeros(hi - lo, np.int32)
        desc = np.zeros((hi - lo, cap, 32), np.uint8)
        for i in range(lo, hi):
            k, d = oex.extract(frames[i])
            assert len(k) <= cap
            counts[i - lo] = len(k)
            desc[i - lo, :len(k)] = d
        return counts, desc
    counts, desc = run(fb, fe)
    per_rank = max(e - b for b, e in ranges)                    # all_gather wants equal shapes: pad the shorter runs
    pad_c = np.zeros(per_rank, np.int32); pad_c[:fe - fb] = counts
    pad_d = np.zeros((per_rank, cap, 32), np.uint8); pad_d[:fe - fb] = desc
    g_c = torch.zeros(world * per_rank, dtype=torch.int32)
    g_d = torch.zeros((world * per_rank, cap, 32), dtype=torch.uint8)
    dist.all_gather_into_tensor(g_c, torch.from_numpy(pad_c))
    dist.all_gather_into_tensor(g_d, torch.from_numpy(pad_d))
    all_c = np.concatenate([g_c.numpy()[r * per_rank:r * per_rank + (e - b)] for r, (b, e) in enumerate(ranges)])
    all_d = np.concatenate([g_d.numpy()[r * per_rank:r * per_rank + (e - b)] for r, (b, e) in enumerate(ranges)])
    whole_c, whole_d = run(0, n_frames)
    ok_frames = ok_cover and np.array_equal(all_c, whole_c) and np.array_equal(all_d, whole_d) and int(whole_c.sum()) > 100
    ret[rank] = (ok_map, ok_frames)
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: (True, True), 1: (True, True)}


def _run_bench(args):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_launcher_spawns_ranks_itself():
    """`bench.py --gpus 2` with no WORLD_SIZE: the parent starts two rank processes without touching the GPU runtime,
    both ranks reach the collective, and the merge that follows the all-gather is the library's."""
    out = _run_bench(["--gpus", "2", "--dry", "--backend", "gloo"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1]
    assert out["launcher"] == {"spawned": 2, "by": "bench.py", "parent_touched_hip": False}
    assert out["merge"] == "orbx_best2_merge" and out["frames_covered"] == 4096
    assert out["frame_range_rank0"] == [0, 2048] and out["map_chunks_rank0"] == [0, 4]
    # the stand-in records of dry mode, merged independently: lowest distance, then lowest index; second = second smallest of the union
    i = np.arange(64)
    best = np.stack([(i * 7 + r * 3) % 50 for r in range(2)])
    idx = np.stack([c0 * 1000 + i for c0 in (0, 4)])
    second = np.stack([best[r] + 1 + r for r in range(2)])
    win = np.where(best[1] < best[0], 1, 0)                      # a tie goes to rank 0: its indices are lower
    mb = best[win, i]
    ms = np.minimum(np.minimum(second[0], second[1]), best[1 - win, i])
    assert out["merged_best_sum"] == int(mb.sum()) and out["merged_idx_sum"] == int(idx[win, i].sum())
    assert out["merged_second_sum"] == int(ms.sum())


def test_bench_launcher_eight_ranks():
    """The shape the scaling record is taken at: eight ranks, every rank seen by the collective, the 4096-frame batch
    covered once by contiguous ranges, one map chunk per rank."""
    out = _run_bench(["--gpus", "8", "--dry", "--backend", "gloo"])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == list(range(8)) and out["launcher"]["spawned"] == 8
    assert out["frames_covered"] == 4096
    assert out["frame_ranges"] == [[512 * r, 512 * (r + 1)] for r in range(8)]
    assert out["map_chunk_ranges"] == [[r, r + 1] for r in range(8)]


def test_bench_single_rank_needs_no_launcher():
    out = _run_bench(["--gpus", "1", "--dry"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == [0] and "launcher" not in out


def test_bench_launcher_reports_a_dead_rank():
    """A rank that dies before the rendezvous must end the launcher with a non-zero status (it stops the rank that is
    left waiting), not hang it."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_DRY_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "rank 1 exited with status 3" in r.stderr
