#!/usr/bin/env python3
"""Benchmark of the ORB front-end hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only launches: it starts N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each), relays rank 0's JSON line and exits with their status.
It imports neither torch nor the library and makes no HIP call itself.  Under `python -m torch.distributed.run` the
ranks exist already and every process is a worker.

One "step" is one pass of the hot path over one batch of synthetic input per GPU:
  * headline (`value`): ORBextractor::operator() over a batch of 640x480 frames (1000 keypoints, 8 levels); frames are
    sharded over ranks with no collective, per-GPU batch fixed (weak scaling);
  * `configs`: the other workloads of the baseline sheet, each timed the same way --
      stereo_752x480      left + right extraction, row-band Hamming search, SAD refinement (pairs/s; replicas per GPU),
      kitti_1241x376      a fixed batch of 4096 frames, 2000 features, split over the ranks (strong scaling),
      uhd_3840x2160       8000 features, 12 levels (replicas per GPU),
  * `hamming`: brute-force best-2 of 4096 queries against a 2^24-row map split over the ranks; for N > 1 followed by
    the RCCL all-gather of the per-rank top-2 and its merge (strong scaling; N = 1 holds the whole map).
Inputs are resident in HBM before every timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK = 8.0e12                          # bytes/s, MI355X HBM3E
VALU_PEAK = 256 * 64 * 2.4e9               # lane-ops/s: 256 CUs x 64 lanes x 2.4 GHz
OPS_PER_PAIR = 16                          # xor / popcount form: 8 x v_xor_b32 + 8 x v_bcnt_u32_b32
MFMA_I8_PEAK = 1024 * 2048 * 2.4e9         # int8 op/s, dense: 1024 SIMDs x 2048 op/clk (v_mfma_i32_16x16x64_i8: 32768 op in 16 clk) x 2.4 GHz
MFMA_OPS_PER_PAIR = 512                    # 256 int8 multiply-accumulates: distance = popc(q) + dot(+-1 bytes of q, 0/1 bytes of row)
MAP_CHUNKS = 8                             # the 2^24-row map is 8 chunks of 2^21 rows, chunk c drawn from PCG32(42 + c)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step (spread over the contexts: 256 per launch by default)")
    ap.add_argument("--contexts", type=int, default=4,
                    help="extractor contexts per GPU, each on its own HIP stream with batch/contexts frames "
                         "(the reference runs two extractors concurrently for stereo, src/Frame.cc:78-81); a process has "
                         "four hardware queues, so more than four streams share queues and serialise")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--map-rows", type=int, default=1 << 24, help="rows of the whole map (split over the ranks)")
    ap.add_argument("--kitti-frames", type=int, default=4096, help="frames of the strong-scaling 1241x376 batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the stereo / 1241x376 / 3840x2160 legs")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="headline and brute-force legs only: no copy-input / one-call / one-context legs and no matcher leg on extracted "
                         "descriptors (profiling runs: every extraction launch in the trace then has the headline's size)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true",
                    help="no GPU work: ranks rendezvous, exchange and merge stand-in records (launcher / collective rehearsal on CPU)")
    return ap.parse_args(argv)


def hip_mapped():
    """True when this process has a HIP runtime or the library mapped (the launching parent must not)."""
    with open("/proc/self/maps") as f:
        maps = f.read()
    return any(k in maps for k in ("libamdhip64", "liborbx", "libhsa-runtime"))


# ---------------------------------------------------------------------------------------------- launcher
def launch(a, argv):
    import socket
    import subprocess
    import threading
    assert not hip_mapped() and "torch" not in sys.modules, "the launching process must stay clear of the GPU runtime"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.1)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:                      # a rank died: the others would wait in a collective for ever
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with status {failed[1]}\n")
        sys.exit(1)
    reader.join(timeout=10)
    out = None
    for ln in lines:
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                out = json.loads(ln)
            except ValueError:
                pass
        elif ln:
            sys.stderr.write(ln + "\n")
    if out is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        sys.exit(1)
    out["launcher"] = {"spawned": a.gpus, "by": "bench.py", "parent_touched_hip": hip_mapped()}
    print(json.dumps(out))


# ---------------------------------------------------------------------------------------------- worker
class Ctx:
    """Rank plumbing: barrier + device synchronise on both sides of a timed region, MAX over ranks of its duration."""

    def __init__(self, a):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = a.dry
        self.dist = None
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if a.backend == "nccl" and not a.dry:
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device(f"cuda:{self.local}"))
            else:
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
        if not a.dry:
            torch.cuda.set_device(self.local)
            self.dev = torch.device(f"cuda:{self.local}")
        else:
            self.dev = torch.device("cpu")

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def ranks_seen(self):
        """All-gather of the rank ids: the record shows how many ranks the collective backend really connected."""
        if self.dist is None:
            return [0]
        mine = self.torch.tensor([self.rank], dtype=self.torch.int32, device=self.dev)
        got = self.torch.zeros(self.world, dtype=self.torch.int32, device=self.dev)
        self.dist.all_gather_into_tensor(got, mine)
        return [int(v) for v in got.cpu().tolist()]

    def timed(self, step, steps, warmup):
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)


def alg_bytes_per_image(vsa, p, W, H, mean_kp):
    """Algorithmic bytes of one ORBextractor call: pyramid read + write, FAST read, blur read + write over the sum of the
    level areas, and per keypoint a 749-byte orientation disc, a 37x37 descriptor window and 60 bytes of output."""
    sum_p = sum(w * h for w, h in (vsa.level_size(p, W, H, l) for l in range(p.nlevels)))
    return sum_p, 5 * sum_p + mean_kp * 2178


class ExtractJob:
    """n_frames HBM-resident frames of one shape spread over `contexts` extractor contexts / HIP streams; a context
    takes its share in launches of at most `chunk` frames."""

    def __init__(self, cx, vsa, np, p, W, H, n_frames, contexts, chunk, seed0, stereo_right=None, n_unique=32, input_view=True):
        torch = cx.torch
        self.cx, self.vsa, self.p, self.W, self.H, self.n = cx, vsa, p, W, H, n_frames
        C = max(1, min(contexts, n_frames))
        self.per = (n_frames + C - 1) // C
        self.chunk = max(1, min(chunk, self.per))
        self.exs = [vsa.Extractor(p, W, H, max_batch=self.chunk, device=cx.local) for _ in range(C)]
        self.streams = [torch.cuda.Stream(device=cx.dev) for _ in range(C)]
        self.cap = self.exs[0].cap
        n_unique = min(n_frames, n_unique)
        if stereo_right is None:
            uniq = np.stack([vsa.synth_frame(seed0 + i, W, H) for i in range(n_unique)])
        else:
            uniq = np.stack([vsa.synth_stereo(seed0 + i, W, H)[1 if stereo_right else 0] for i in range(n_unique)])
        self.uniq = uniq
        reps = (n_frames + n_unique - 1) // n_unique
        self.frames = torch.from_numpy(uniq).to(cx.dev).repeat(reps, 1, 1)[:n_frames].contiguous()
        self.kps = torch.zeros((n_frames, self.cap, 7), dtype=torch.float32, device=cx.dev)
        self.desc = torch.zeros((n_frames, self.cap, 32), dtype=torch.uint8, device=cx.dev)
        self.counts = torch.zeros(n_frames, dtype=torch.int32, device=cx.dev)
        # Zero-copy input (orbx_extractor_input_view): the frames of a launch lie where level 0 of the context's pyramid
        # lives, as a producer on the device (rectification, colour conversion, an upload) would leave them, so a launch
        # copies nothing.  The views are filled here, before any timed region, by one ordinary extraction from the plain
        # buffer (whose copy kernel writes exactly that place); a context whose share needs several launches shows every
        # launch the frames its last one left there (the synthetic frames repeat with period n_unique <= chunk anyway).
        self.input_view = input_view
        self.views = None
        if input_view:
            self.input_view = False
            self.step()
            cx.sync()
            self.input_view = True
            self.views = [ex.input_view() for ex in self.exs]

    def step(self):
        W, H = self.W, self.H
        for c, ex in enumerate(self.exs):
            f0, f1 = c * self.per, min((c + 1) * self.per, self.n)
            while f0 < f1:
                nf = min(self.chunk, f1 - f0)
                if self.input_view:
                    ptr, rs, fs = self.views[c]
                else:
                    ptr, rs, fs = self.frames[f0:].data_ptr(), W, W * H
                ex.extract_batch_device(ptr, rs, fs, nf, self.kps[f0:].data_ptr(), self.desc[f0:].data_ptr(),
                                        self.counts[f0:].data_ptr(), self.cap, self.streams[c].cuda_stream)
                f0 += nf

    def mean_kp(self):
        return float(self.counts.float().mean().item())

    def close(self):
        for e in self.exs:
            e.close()
        self.exs = []


def leg_config_extract(cx, vsa, np, name, W, H, nfeatures, nlevels, frames_total, strong, contexts, chunk, steps, warmup, note):
    """One extraction workload of the baseline sheet.  strong: frames_total is the whole job, split over the ranks
    (orbx_shard_range); otherwise every rank runs frames_total of its own."""
    p = vsa.make_params(nfeatures=nfeatures, nlevels=nlevels)
    if strong:
        b, e = vsa.shard_range(frames_total, cx.rank, cx.world)
        n_local, job_frames = e - b, frames_total
    else:
        n_local, job_frames = frames_total, frames_total * cx.world
    job = ExtractJob(cx, vsa, np, p, W, H, n_local, contexts, chunk, seed0=cx.rank * 100000 + 500, stereo_right=None, n_unique=8)
    t = cx.timed(job.step, steps, warmup)
    mean_kp = cx.sum_over_ranks(job.mean_kp() * n_local) / job_frames
    job.close()
    sum_p, a_img = alg_bytes_per_image(vsa, p, W, H, mean_kp)
    fps = job_frames * steps / t
    return {"name": name, "workload": note, "value": fps, "unit": "frames/s", "ms_per_step": 1e3 * t / steps, "steps": steps,
            "frames_per_step_all_gpus": job_frames, "scaling": "strong" if strong else "weak",
            "mean_keypoints_per_frame": round(mean_kp, 1), "alg_bytes_per_frame": int(a_img),
            "achieved_GBps_per_gpu": round(a_img * fps / cx.world / 1e9, 1), "frac_hbm": round(a_img * fps / cx.world / HBM_PEAK, 4)}


def leg_stereo(cx, vsa, np, steps, warmup, B=512, nfeatures=1200):
    """BASELINE configs[1]: 752x480 pairs, left and right extraction on two contexts / streams, then the row-band Hamming
    search (src/Frame.cc:466-549) and the sub-pixel refinement (:551-639); camera constants of Examples/Stereo/EuRoC.yaml."""
    torch = cx.torch
    W, H = 752, 480
    p = vsa.make_params(nfeatures=nfeatures, nlevels=8)
    # two independent halves of the batch, each with its own left and right extractor context, matcher context and pair
    # of streams: four streams in all (the process has four hardware queues), so one half's row search and refinement
    # run beside the other half's extraction
    G = 2 if B >= 2 else 1
    scale = vsa.scale_tables(p)["scale"]
    bf, fx = 47.90639384423901, 435.2046959714599

    class Half:
        def __init__(self, g, n):
            seed = cx.rank * 100000 + 30 + 1000 * g
            self.n = n
            self.left = ExtractJob(cx, vsa, np, p, W, H, n, 1, n, seed0=seed, stereo_right=False, n_unique=16)
            self.right = ExtractJob(cx, vsa, np, p, W, H, n, 1, n, seed0=seed, stereo_right=True, n_unique=16)
            self.m = vsa.Matcher(device=cx.local)
            self.cap = self.left.cap
            self.bd = torch.zeros((n, self.cap), dtype=torch.int32, device=cx.dev)
            self.bi, self.sad = torch.zeros_like(self.bd), torch.zeros_like(self.bd)
            self.ur = torch.zeros((n, self.cap), dtype=torch.float32, device=cx.dev)
            self.dp = torch.zeros_like(self.ur)

        def step(self):
            left, right, cap, n = self.left, self.right, self.cap, self.n
            s0, s1 = left.streams[0], right.streams[0]
            left.step()
            right.step()
            s0.wait_stream(s1)
            self.m.stereo_match_batch_device(left.kps.data_ptr(), left.desc.data_ptr(), left.counts.data_ptr(), right.kps.data_ptr(),
                                             right.desc.data_ptr(), right.counts.data_ptr(), cap, n, scale, H, 0.0, fx,
                                             self.bd.data_ptr(), self.bi.data_ptr(), s0.cuda_stream)
            left.exs[0].stereo_refine_batch_device(right.exs[0], left.kps.data_ptr(), left.counts.data_ptr(), right.kps.data_ptr(), cap, n,
                                                   self.bd.data_ptr(), self.bi.data_ptr(), bf, 0.0, fx, self.ur.data_ptr(), self.dp.data_ptr(),
                                                   self.sad.data_ptr(), s0.cuda_stream)
            s1.wait_stream(s0)

        def close(self):
            self.left.close(); self.right.close(); self.m.close()

    halves = [Half(g, B // G + (1 if g < B % G else 0)) for g in range(G)]

    def step():
        for h in halves:
            h.step()

    t = cx.timed(step, steps, warmup)
    mean_kp = sum(0.5 * (h.left.mean_kp() + h.right.mean_kp()) * h.n for h in halves) / B
    matched = sum(float((h.ur >= 0).sum().item()) for h in halves) / B
    for h in halves:
        h.close()
    _, a_img = alg_bytes_per_image(vsa, p, W, H, mean_kp)
    pps = cx.world * B * steps / t
    return {"name": "stereo_752x480", "workload": f"{B} HBM-resident 752x480 pairs per GPU per step, nFeatures={nfeatures} (EuRoC.yaml), left + right "
                                                  "ORBextractor contexts for each half of the batch (four streams), stereo row-band Hamming search + SAD refinement",
            "value": pps, "unit": "pairs/s", "ms_per_step": 1e3 * t / steps, "steps": steps, "scaling": "weak",
            "mean_keypoints_per_image": round(mean_kp, 1), "mean_stereo_matches_per_pair": round(matched, 1),
            "alg_bytes_per_pair": int(2 * a_img), "achieved_GBps_per_gpu": round(2 * a_img * pps / cx.world / 1e9, 1),
            "frac_hbm": round(2 * a_img * pps / cx.world / HBM_PEAK, 4)}


def leg_hamming(cx, vsa, np, a, steps, warmup):
    """BASELINE configs[4]: 4096 queries against a 2^24-row map.  The map is MAP_CHUNKS chunks of rows; rank r holds the
    chunks orbx_shard_range gives it, searches them in one call and, for N > 1, all-gathers its top-2 and merges."""
    torch = cx.torch
    nq, total = a.queries, a.map_rows
    chunk_rows = total // MAP_CHUNKS
    assert chunk_rows * MAP_CHUNKS == total and MAP_CHUNKS % cx.world == 0, "map rows / ranks must divide the 8 map chunks"
    c0, c1 = vsa.shard_range(MAP_CHUNKS, cx.rank, cx.world)
    rows = (c1 - c0) * chunk_rows
    table = torch.empty((rows, 32), dtype=torch.uint8, device=cx.dev)
    first = None
    for c in range(c0, c1):
        part_np = vsa.synth_descriptors(42 + c, chunk_rows)
        if c == 0:
            first = part_np
        table[(c - c0) * chunk_rows:(c - c0 + 1) * chunk_rows].copy_(torch.from_numpy(part_np))
    if first is None:
        first = vsa.synth_descriptors(42, chunk_rows)
    q_np, _ = vsa.synth_queries(43, first, nq)             # queries are noisy copies of rows of chunk 0
    del first
    q = torch.from_numpy(q_np).to(cx.dev)
    m = vsa.Matcher(device=cx.local)
    stream = torch.cuda.current_stream().cuda_stream
    part = torch.zeros((nq, 4), dtype=torch.int32, device=cx.dev)
    gathered = torch.zeros((cx.world * nq, 4), dtype=torch.int32, device=cx.dev)
    merged = torch.zeros((nq, 4), dtype=torch.int32, device=cx.dev)

    def step():
        m.hamming_best2_device(q.data_ptr(), nq, table.data_ptr(), rows, c0 * chunk_rows, part.data_ptr(), stream)
        if cx.dist is not None:
            cx.dist.all_gather_into_tensor(gathered, part)
            m.best2_merge_device(gathered.data_ptr(), cx.world, nq, merged.data_ptr(), stream)

    t = cx.timed(step, steps, warmup)
    res = (merged if cx.dist is not None else part).cpu().numpy()
    # how much of the multiplied work also walked its keys (the data-dependent part of the kernel), on this rank
    m.best2_tile_walks(reset=True)
    m.hamming_best2_device(q.data_ptr(), nq, table.data_ptr(), rows, c0 * chunk_rows, part.data_ptr(), stream)
    walked_frac = m.best2_tile_walks() / (((rows + 15) // 16) * ((nq + 15) // 16))
    # the legs of a multi-GPU step one by one (events on the stream; separate steps, so their sum need not equal a step)
    legs = None
    if cx.dist is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        reps = 5
        for _ in range(reps):
            cx.barrier()
            ev[0].record()
            m.hamming_best2_device(q.data_ptr(), nq, table.data_ptr(), rows, c0 * chunk_rows, part.data_ptr(), stream)
            ev[1].record()
            cx.dist.all_gather_into_tensor(gathered, part)
            ev[2].record()
            m.best2_merge_device(gathered.data_ptr(), cx.world, nq, merged.data_ptr(), stream)
            ev[3].record()
            torch.cuda.synchronize()
            for k in range(3):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
        legs = {"search_ms": round(cx.max_over_ranks(acc[0] / reps), 4), "allgather_ms": round(cx.max_over_ranks(acc[1] / reps), 4),
                "merge_ms": round(cx.max_over_ranks(acc[2] / reps), 4),
                "note": "max over ranks of each leg, events on the stream; the all-gather moves 16 bytes per query per rank and "
                        "includes the wait for the slowest rank's search"}
    m.close()
    pps = float(nq) * total * steps / t
    return {"metric": "Hamming pair-comparisons/s", "value": pps, "queries_per_s": nq * steps / t, "ms_per_step": 1e3 * t / steps,
            "steps": steps, "queries": nq, "map_rows_total": total, "map_rows_per_gpu": rows, "scaling": "strong",
            "collective": None if cx.dist is None else "all_gather_into_tensor of 16-byte top-2 records + k_best2_merge",
            "legs": legs, "tiles_walked_frac": round(walked_frac, 6), "data": "iid uniform 256-bit rows, queries = rows with ~8 % of the bits flipped",
            "matched_within_50": int((res[:, 0] <= 50).sum()),
            # the search runs on the matrix cores as an exact int8 contraction; popcount_frac prices the same pairs
            # against the vector pipe's xor / popcount issue peak (above 1: more than that formulation can reach)
            "roofline": {"bound": "mfma", "achieved": round(MFMA_OPS_PER_PAIR * pps / cx.world / 1e12, 1),
                         "peak": MFMA_I8_PEAK / 1e12, "unit": "TOP/s",
                         "frac": round(MFMA_OPS_PER_PAIR * pps / cx.world / MFMA_I8_PEAK, 4),
                         "popcount_frac": round(OPS_PER_PAIR * pps / cx.world / VALU_PEAK, 4)}}


def leg_hamming_extracted(cx, vsa, np, steps, warmup, n_map_frames=256, W=1241, H=376, nfeatures=2000, nq=4096):
    """The same brute-force search on descriptors the extractor itself produced: the map is every descriptor of
    n_map_frames different synthetic 1241x376 frames, the queries are descriptors of other frames.  Real descriptors are
    not uniform (bits correlate, frames repeat structure), so more tiles walk their keys than on the iid map."""
    torch = cx.torch
    p = vsa.make_params(nfeatures=nfeatures, nlevels=8)
    n_q_frames = 4
    chunk = 64
    ex = vsa.Extractor(p, W, H, max_batch=chunk, device=cx.local)
    cap = ex.cap
    stream = torch.cuda.current_stream().cuda_stream
    rows, qrows = [], []
    kps = torch.zeros((chunk, cap, 7), dtype=torch.float32, device=cx.dev)
    desc = torch.zeros((chunk, cap, 32), dtype=torch.uint8, device=cx.dev)
    counts = torch.zeros(chunk, dtype=torch.int32, device=cx.dev)
    seed0 = 700000 + cx.rank * 10000
    total_frames = n_map_frames + n_q_frames
    for f0 in range(0, total_frames, chunk):
        nf = min(chunk, total_frames - f0)
        frames = torch.from_numpy(np.stack([vsa.synth_frame(seed0 + f0 + i, W, H) for i in range(nf)])).to(cx.dev)
        ex.extract_batch_device(frames.data_ptr(), W, W * H, nf, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
        torch.cuda.synchronize()
        cn = counts.cpu().numpy()
        for i in range(nf):
            (rows if f0 + i < n_map_frames else qrows).append(desc[i, :int(cn[i])].clone())
    ex.close()
    table = torch.cat(rows).contiguous()
    q = torch.cat(qrows)[:nq].contiguous()
    nt, nq = table.shape[0], q.shape[0]
    m = vsa.Matcher(device=cx.local)
    out = torch.zeros((nq, 4), dtype=torch.int32, device=cx.dev)

    def step():
        m.hamming_best2_device(q.data_ptr(), nq, table.data_ptr(), nt, 0, out.data_ptr(), stream)

    t = cx.timed(step, steps, warmup)
    m.best2_tile_walks(reset=True)
    step()
    walked_frac = m.best2_tile_walks() / (((nt + 15) // 16) * ((nq + 15) // 16))
    res = out.cpu().numpy()
    # like for like: an iid-uniform map of the same number of rows (a map this small amortises the kernel's start-up
    # less than the 2^24-row one), queries noisy copies of its rows
    u_np = vsa.synth_descriptors(4242, nt)
    uq_np, _ = vsa.synth_queries(4343, u_np, nq)
    table.copy_(torch.from_numpy(u_np))
    q.copy_(torch.from_numpy(uq_np))
    t_uniform = cx.timed(step, steps, warmup)
    m.close()
    pps = cx.world * float(nq) * nt * steps / t
    return {"metric": "Hamming pair-comparisons/s on extracted descriptors", "value": pps, "ms_per_step": 1e3 * t / steps, "steps": steps,
            "uniform_map_of_the_same_size_value": cx.world * float(nq) * nt * steps / t_uniform,
            "queries": nq, "map_rows_per_gpu": nt, "scaling": "weak",
            "data": f"map = all descriptors of {n_map_frames} different synthetic {W}x{H} frames ({nfeatures} features), queries = descriptors of "
                    f"{n_q_frames} other frames; one map per GPU",
            "tiles_walked_frac": round(walked_frac, 6), "median_best": int(np.median(res[:, 0])), "median_second": int(np.median(res[:, 2])),
            "roofline": {"bound": "mfma", "achieved": round(MFMA_OPS_PER_PAIR * pps / cx.world / 1e12, 1), "peak": MFMA_I8_PEAK / 1e12,
                         "unit": "TOP/s", "frac": round(MFMA_OPS_PER_PAIR * pps / cx.world / MFMA_I8_PEAK, 4),
                         "popcount_frac": round(OPS_PER_PAIR * pps / cx.world / VALU_PEAK, 4)}}


def dry_worker(cx, a):
    """Rehearsal without a GPU: rendezvous, rank census, shard ranges from the library, an all-gather of stand-in top-2
    records and the library's host merge.  What the ranks would have computed on their GPUs is replaced by a fixed
    arithmetic pattern; launching, partitioning, exchange and merge are the real code."""
    import importlib
    import numpy as np
    torch = cx.torch
    sys.path.insert(0, ROOT)
    vsa = importlib.import_module("visual-slam_amd")
    seen = cx.ranks_seen()
    nq = 64
    fb, fe = vsa.shard_range(a.kitti_frames, cx.rank, cx.world)
    c0, c1 = vsa.shard_range(MAP_CHUNKS, cx.rank, cx.world)
    rec = np.zeros(nq, vsa.BEST2_DTYPE)
    i = np.arange(nq)
    rec["best"] = (i * 7 + cx.rank * 3) % 50
    rec["best_idx"] = c0 * 1000 + i
    rec["second"] = rec["best"] + 1 + cx.rank
    mine = torch.from_numpy(rec.view(np.int32).reshape(nq, 4).copy())
    gathered = torch.zeros((cx.world * nq, 4), dtype=torch.int32)
    if cx.dist is not None:
        cx.dist.all_gather_into_tensor(gathered, mine)
    else:
        gathered.copy_(mine)
    merged = vsa.best2_merge(gathered.numpy().view(vsa.BEST2_DTYPE).reshape(cx.world, nq))
    frames = cx.sum_over_ranks(fe - fb)
    ranges = torch.tensor([fb, fe, c0, c1], dtype=torch.int64)
    all_ranges = torch.zeros(cx.world * 4, dtype=torch.int64)
    if cx.dist is not None:
        cx.dist.all_gather_into_tensor(all_ranges, ranges)
    else:
        all_ranges.copy_(ranges)
    all_ranges = all_ranges.reshape(cx.world, 4).tolist()
    if cx.rank == 0:
        print(json.dumps({"dry": True, "n_gpus": cx.world, "ranks_seen": seen, "frames_covered": int(frames),
                          "frame_range_rank0": [fb, fe], "map_chunks_rank0": [c0, c1],
                          "frame_ranges": [r[:2] for r in all_ranges], "map_chunk_ranges": [r[2:] for r in all_ranges], "merge": "orbx_best2_merge",
                          "merged_best_sum": int(merged["best"].sum()), "merged_idx_sum": int(merged["best_idx"].sum()),
                          "merged_second_sum": int(merged["second"].sum()), "backend": "gloo"}))
    if cx.dist is not None:
        cx.dist.destroy_process_group()


def worker(a):
    import importlib
    import numpy as np
    if a.dry and os.environ.get("BENCH_DRY_FAIL_RANK") == os.environ.get("RANK", "0"):
        sys.exit(3)                          # rehearsal of a rank that dies before the rendezvous (launcher test)
    cx = Ctx(a)
    if a.dry:
        return dry_worker(cx, a)
    torch = cx.torch
    sys.path.insert(0, ROOT)
    vsa = importlib.import_module("visual-slam_amd")
    rank, n_gpus, dev = cx.rank, cx.world, cx.dev
    seen = cx.ranks_seen()

    # ------------------------------------------------------------------ headline: 640x480 extraction
    W, H, B = a.width, a.height, a.batch
    p = vsa.make_params(nfeatures=a.nfeatures, nlevels=a.nlevels)
    job = ExtractJob(cx, vsa, np, p, W, H, B, a.contexts, B, seed0=rank * 100000, stereo_right=None)
    C = len(job.exs)
    t_extract = cx.timed(job.step, a.steps, a.warmup)
    fps = n_gpus * B * a.steps / t_extract
    # how much the figure moves from one timed region to the next on this box (clock state, other tenants): three more
    # regions of the same length (at most 50 steps each), reported beside the headline, never instead of it
    rsteps = min(a.steps, 50)
    repeats = [n_gpus * B * rsteps / cx.timed(job.step, rsteps, 0) for _ in range(3)]
    mean_kp = job.mean_kp()
    frames, kps, desc, counts, cap, uniq = job.frames, job.kps, job.desc, job.counts, job.cap, job.uniq
    # the same step with the frames in a plain buffer of the caller's (row stride = width): every launch then starts
    # with the library's copy into the padded level 0, as ComputePyramid does (src/ORBextractor.cc:1124-1128)
    job.input_view = False
    csteps = max(2, a.steps // 4)
    fps_copy = fps_solo = None
    if not a.no_extra_legs:
        t_copy = cx.timed(job.step, csteps, min(a.warmup, 3))
        fps_copy = n_gpus * B * csteps / t_copy
    # one caller stream, one call per step: orbx_extract_batch_device_multi spreads the batch over the same contexts and
    # joins them on the caller's stream (frames in the input views, as in the headline)
    one_stream = torch.cuda.current_stream().cuda_stream

    def one_call_step():
        vsa.extract_batch_device_multi(job.exs, None, 0, 0, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, one_stream)

    def one_call_nojoin_step():
        vsa.extract_batch_device_multi(job.exs, None, 0, 0, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, one_stream, join=False)

    fps_one_call = fps_one_call_nojoin = None
    if C > 1 and job.per == job.chunk and not a.no_extra_legs:
        t_one = cx.timed(one_call_step, csteps, min(a.warmup, 3))
        fps_one_call = n_gpus * B * csteps / t_one
        t_one = cx.timed(one_call_nojoin_step, csteps, min(a.warmup, 3))     # the barrier's device synchronise is the join
        fps_one_call_nojoin = n_gpus * B * csteps / t_one
    # ... and a single context on a single stream working through the same batch in launches of the same size
    if not a.no_extra_legs:
        solo = ExtractJob(cx, vsa, np, p, W, H, job.chunk, 1, job.chunk, seed0=rank * 100000, stereo_right=None)
        t_solo = cx.timed(solo.step, csteps * 2, min(a.warmup, 3))
        fps_solo = n_gpus * job.chunk * csteps * 2 / t_solo
        solo.close()
    job.close()

    # per-stage device time: HIP events recorded by the library on the launch stream around each stage, one launch of
    # the size the timed region used (a context's share of the batch) alone on one stream (separate steps, same frames)
    PB = job.chunk
    ex = vsa.Extractor(p, W, H, max_batch=PB, device=cx.local)
    stream = torch.cuda.current_stream().cuda_stream

    src = [frames.data_ptr(), W, W * H]

    def profiled_step():
        ex.extract_batch_device(src[0], src[1], src[2], PB, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)

    profiled_step()                      # from the plain buffer: its copy kernel leaves the frames in the input view ...
    torch.cuda.synchronize()
    src[:] = ex.input_view()             # ... which the measured steps read, as the headline does
    ex.set_profiling(True)
    stage_ms = np.zeros(len(vsa.STAGES))
    prof_steps = max(3, min(a.steps, 10))
    for _ in range(prof_steps):
        profiled_step()
        stage_ms += ex.stage_ms()
    stage_ms /= prof_steps
    ex.set_profiling(False)
    ex.close()

    sum_p, a_extract = alg_bytes_per_image(vsa, p, W, H, mean_kp)
    stage_bytes = {"pyramid": 2 * sum_p, "fast": sum_p, "octree": 0, "blur": 2 * sum_p, "describe": mean_kp * 2178}
    stages = {}
    for i, name in enumerate(vsa.STAGES):
        ms = float(stage_ms[i])
        if name == "blur" and ms < 0.02:
            # no blur launch: at this many pixels per keypoint the library blurs each keypoint's window inside k_describe
            stages[name] = {"ms": round(ms, 4), "alg_GBps": None, "frac_hbm": None, "note": "no launch (windows blurred inside k_describe)"}
            continue
        gbps = stage_bytes[name] * PB / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        stages[name] = {"ms": round(ms, 4), "alg_GBps": round(gbps, 1), "frac_hbm": round(gbps * 1e9 / HBM_PEAK, 4)}
    dom = max((n for n in vsa.STAGES if stage_bytes[n] > 0), key=lambda n: stage_ms[vsa.STAGES.index(n)])
    dom_ms = float(stage_ms[vsa.STAGES.index(dom)])
    achieved = stage_bytes[dom] * PB / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch of that stage's kernels from the stored rocprofv3 --pmc passes (tools/pmc_traffic.py): only
    # valid for the kernels they were taken on, so the file's build stamp must equal the loaded library's
    traffic, traffic_src, valu_issue = None, None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and (W, H, PB, a.nfeatures, a.nlevels) == (640, 480, 256, 1000, 8):
        with open(tfile) as f:
            tj = json.load(f)
        if tj.get("build_stamp") != vsa.build_stamp():
            traffic_src = f"profiles/traffic.json is stale (taken on build {tj.get('build_stamp')}, library is {vsa.build_stamp()})"
        elif dom in tj.get("stages", {}):
            traffic, traffic_src = tj["stages"][dom], tj.get("source")
            dom_kernel = {"fast": "k_fast_cells", "octree": "k_octree", "blur": "k_blur", "describe": "k_describe"}.get(dom)
            valu_issue = tj.get("kernels_mean_per_launch", {}).get(dom_kernel, {}).get("valu_issue_frac") if dom_kernel else None
    kernel_of = {"pyramid": "k_level0_frame + k_resize per level", "fast": "k_fast_cells", "octree": "k_octree", "blur": "k_blur", "describe": "k_describe"}
    roofline = {"bound": "hbm", "kernel": kernel_of[dom], "stage": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": int(stage_bytes[dom] * PB), "launch_ms": round(dom_ms, 4),
                "valu_issue_frac": valu_issue,   # vector instructions x 4 cycles / SIMD cycles: an upper estimate of the issue pipe's load (2 to 4.5 cycles each on this chip)
                "pipeline_alg_bytes_per_frame": int(a_extract),
                "pipeline_frac": round(a_extract * (fps / n_gpus) / HBM_PEAK, 4)}
    del frames, kps, desc, counts, job

    # ------------------------------------------------------------------ the other workloads of the baseline sheet
    configs = []
    if not a.no_configs:
        ksteps = max(2, min(a.steps, 20))
        configs.append(leg_stereo(cx, vsa, np, ksteps, 2))
        torch.cuda.empty_cache()
        configs.append(leg_config_extract(cx, vsa, np, "kitti_1241x376", 1241, 376, 2000, 8, a.kitti_frames, True, 4, 256,
                                          max(2, min(a.steps, 5)), 1,
                                          f"one batch of {a.kitti_frames} HBM-resident 1241x376 frames, nFeatures=2000, split over the ranks "
                                          "(BASELINE configs[2]; one step = the whole batch)"))
        torch.cuda.empty_cache()
        configs.append(leg_config_extract(cx, vsa, np, "uhd_3840x2160", 3840, 2160, 8000, 12, 128, False, 4, 32, max(2, min(a.steps, 6)), 1,
                                          "128 HBM-resident 3840x2160 frames per GPU per step (four contexts of 32; a single context of 64 "
                                          "is within 3 % of it), nFeatures=8000, 12 levels (BASELINE configs[3]); at 3300 pixels per keypoint the library blurs "
                                          "each keypoint's window inside k_describe instead of every level"))
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ Hamming workload
    hamming = None
    if not a.no_hamming:
        hamming = leg_hamming(cx, vsa, np, a, max(2, min(a.steps, 40)), min(a.warmup, 3))
        torch.cuda.empty_cache()
        if not a.no_extra_legs:
            hamming["extracted_descriptors"] = leg_hamming_extracted(cx, vsa, np, max(2, min(a.steps, 40)), min(a.warmup, 3))

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1)
    cpu = None
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a, uniq, W, H)

    if rank == 0:
        out = {
            "metric": "ORB frames/s (1000 kp, 8 lvl, 640x480) + Hamming matches/s, 1/2/4/8 GPUs",
            "value": fps, "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * t_extract / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"ORBextractor {W}x{H} nFeatures={a.nfeatures} nLevels={a.nlevels} scale=1.2 ini=20 min=7, "
                                   f"{B} HBM-resident frames per GPU per step (BASELINE configs[0] shape, batched)",
                       "input": "frames resident in the extractors' input views (orbx_extractor_input_view: level 0 of the pyramid, "
                                "no copy at the start of a launch); copy_input_value is the same step from a plain width-strided buffer",
                       "repeat_values": [round(v) for v in repeats],   # three more timed regions of the headline step: run-to-run spread on this box
                       "copy_input_value": fps_copy,
                       "one_call_value": fps_one_call,      # the same contexts behind one call on one caller stream (orbx_extract_batch_device_multi)
                       "one_call_no_join_value": fps_one_call_nojoin,   # ... with ORBX_MULTI_NO_JOIN: calls queued back to back, joined once at the end
                       "one_context_value": fps_solo,       # contexts_per_gpu = 1: one extractor, one stream
                       "frames_per_gpu_per_step": B, "contexts_per_gpu": C, "frames_per_launch": PB, "mean_keypoints_per_frame": round(mean_kp, 1),
                       "parallelism": f"frames sharded over {n_gpus} GPU(s), no collective"},
            "ranks_seen": seen, "build_stamp": vsa.build_stamp(),
            "roofline": roofline, "stages": stages, "configs": configs, "hamming": hamming, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if cx.dist is not None:
        cx.dist.destroy_process_group()


def cpu_baseline(a, uniq, W, H):
    """The CPU oracle on the same frames, bounded to about 12 s on one thread plus 6 s on the box's CPU share."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    n_unique = len(uniq)
    oex = orc.Extractor(orc.make_params(nfeatures=a.nfeatures, nlevels=a.nlevels))
    n_cpu = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 12.0 and n_cpu < 400:
        oex.extract(uniq[n_cpu % n_unique])
        n_cpu += 1
    dt = time.perf_counter() - t0
    cpu = {"value": round(n_cpu / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n_cpu} frames of the same {W}x{H}/{a.nfeatures}kp/{a.nlevels}lvl workload, CPU oracle, 1 thread "
                     f"(host has {os.cpu_count()} logical CPUs)"}
    # the same oracle frame-parallel over the box's CPU share (one extractor per thread; the C calls release the GIL)
    import concurrent.futures as cf
    nthreads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 16))

    def one(k):
        e = orc.Extractor(orc.make_params(nfeatures=a.nfeatures, nlevels=a.nlevels))
        n, t_end = 0, time.perf_counter() + 6.0
        while time.perf_counter() < t_end and n < 200:
            e.extract(uniq[(k + n) % n_unique])
            n += 1
        return n
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(nthreads) as pool:
        done = sum(pool.map(one, range(nthreads)))
    cpu["all_cores"] = {"value": round(done / (time.perf_counter() - t0), 2), "unit": "frames/s", "cores": nthreads,
                        "sample": f"{done} frames, one oracle extractor per thread"}
    return cpu


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(a, argv)
    worker(a)


if __name__ == "__main__":
    main()
