#!/usr/bin/env python3
"""Benchmark of the ORB front-end hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path over one batch of synthetic input per GPU:
  * ORBextractor::operator() over a batch of 640x480 frames (1000 keypoints, 8 levels) -- `value`,
  * brute-force Hamming best-2 of 4096 queries against one 2^21-row map shard (the per-GPU share
    of the 2^24-row map when it is sharded over 8 GPUs) plus, for N>1, the RCCL all-gather of the
    per-rank top-2 and its merge -- reported under "hamming".
Inputs are resident in HBM before the timed region.  Frames are sharded over ranks with no
collective (weak scaling: per-GPU batch fixed); the map is sharded over ranks with one all-gather.
Rank 0 prints one JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
vsa = importlib.import_module("visual-slam_amd")

HBM_PEAK = 8.0e12                          # bytes/s, MI355X HBM3E
VALU_PEAK = 256 * 64 * 2.4e9               # lane-ops/s: 256 CUs x 64 lanes x 2.4 GHz
OPS_PER_PAIR = 16                          # xor / popcount form: 8 x v_xor_b32 + 8 x v_bcnt_u32_b32
MFMA_I8_PEAK = 1024 * 2048 * 2.4e9         # int8 op/s, dense: 1024 SIMDs x 2048 op/clk (v_mfma_i32_16x16x64_i8: 32768 op in 16 clk) x 2.4 GHz
MFMA_OPS_PER_PAIR = 512                    # 256 int8 multiply-accumulates: distance = popc(q) + dot(+-1 bytes of q, 0/1 bytes of row)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--contexts", type=int, default=8,
                    help="extractor contexts per GPU, each on its own HIP stream with batch/contexts frames "
                         "(the reference runs two extractors concurrently for stereo, src/Frame.cc:78-81)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--map-rows-per-gpu", type=int, default=1 << 21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming", action="store_true")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    n_gpus = world

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ extraction workload
    W, H, B = a.width, a.height, a.batch
    C = max(1, min(a.contexts, B))
    per = (B + C - 1) // C
    p = vsa.make_params(nfeatures=a.nfeatures, nlevels=a.nlevels)
    exs = [vsa.Extractor(p, W, H, max_batch=per, device=local) for _ in range(C)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
    cap = exs[0].cap
    # distinct synthetic frames per rank (32 generated, tiled to the batch)
    n_unique = min(B, 32)
    uniq = np.stack([vsa.synth_frame(rank * 100000 + i, W, H) for i in range(n_unique)])
    frames = torch.from_numpy(uniq).to(dev).repeat((B + n_unique - 1) // n_unique, 1, 1)[:B].contiguous()
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def extract_step():
        # one pass over the GPU's batch: context c takes frames [c*per, (c+1)*per) on its own stream
        for c in range(C):
            f0 = c * per
            nf = min(per, B - f0)
            if nf > 0:
                exs[c].extract_batch_device(frames[f0:].data_ptr(), W, W * H, nf, kps[f0:].data_ptr(), desc[f0:].data_ptr(),
                                            counts[f0:].data_ptr(), cap, streams[c].cuda_stream)

    for _ in range(a.warmup):
        extract_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        extract_step()
    barrier()
    t_extract = max_over_ranks(time.perf_counter() - t0)
    fps = n_gpus * B * a.steps / t_extract
    mean_kp = float(counts.float().mean().item())
    for e in exs:
        e.close()

    # per-stage device time: HIP events recorded by the library on the launch stream around each stage, whole batch
    # in one context on one stream (separate steps, same workload)
    ex = vsa.Extractor(p, W, H, max_batch=B, device=local)
    stream = torch.cuda.current_stream().cuda_stream

    def profiled_step():
        ex.extract_batch_device(frames.data_ptr(), W, W * H, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)

    profiled_step()
    torch.cuda.synchronize()
    ex.set_profiling(True)
    stage_ms = np.zeros(len(vsa.STAGES))
    prof_steps = max(3, min(a.steps, 10))
    for _ in range(prof_steps):
        profiled_step()
        stage_ms += ex.stage_ms()
    stage_ms /= prof_steps
    ex.set_profiling(False)

    sizes = [vsa.level_size(p, W, H, l) for l in range(a.nlevels)]
    sum_p = sum(w * h for w, h in sizes)
    # algorithmic bytes per frame: pyramid read+write, FAST read, (blur read+write + per-keypoint
    # orientation disc 749 B + descriptor window 1369 B + 60 B output) in the describe stage
    stage_bytes = {
        "pyramid": 2 * sum_p,
        "fast": sum_p,
        "octree": 0,
        "describe": 2 * sum_p + mean_kp * 2178,
    }
    a_extract = 5 * sum_p + mean_kp * 2178
    stages = {}
    for i, name in enumerate(vsa.STAGES):
        ms = float(stage_ms[i])
        gbps = stage_bytes[name] * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        stages[name] = {"ms": round(ms, 4), "alg_GBps": round(gbps, 1), "frac_hbm": round(gbps * 1e9 / HBM_PEAK, 4)}
    dom = max((n for n in vsa.STAGES if stage_bytes[n] > 0), key=lambda n: stage_ms[vsa.STAGES.index(n)])
    dom_ms = float(stage_ms[vsa.STAGES.index(dom)])
    achieved = stage_bytes[dom] * B / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch of that stage's kernels, measured with rocprofv3 --pmc on this workload (tools/pmc_traffic.py)
    traffic, traffic_src, valu_issue = None, None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and (W, H, B, a.nfeatures, a.nlevels) == (640, 480, 256, 1000, 8):
        with open(tfile) as f:
            tj = json.load(f)
        if dom in tj.get("stages", {}):
            traffic, traffic_src = tj["stages"][dom], tj.get("source")
        dom_kernel = {"fast": "k_fast_cells", "octree": "k_octree"}.get(dom)
        valu_issue = tj.get("kernels_mean_per_launch", {}).get(dom_kernel, {}).get("valu_issue_frac") if dom_kernel else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": int(stage_bytes[dom] * B), "launch_ms": round(dom_ms, 4),
                "valu_issue_frac": valu_issue,   # measured share of cycles issuing vector ALU work: what actually bounds this kernel
                "pipeline_alg_bytes_per_frame": int(a_extract),
                "pipeline_frac": round(a_extract * (fps / n_gpus) / HBM_PEAK, 4)}

    ex.close()
    stream = torch.cuda.current_stream().cuda_stream
    # ------------------------------------------------------------------ Hamming workload
    hamming = None
    if not a.no_hamming:
        nq, rows = a.queries, a.map_rows_per_gpu
        m = vsa.Matcher(device=local)
        table_np = vsa.synth_descriptors(42 + rank, rows)
        q_np, _ = vsa.synth_queries(43, vsa.synth_descriptors(42, rows), nq)   # queries come from shard 0
        table = torch.from_numpy(table_np).to(dev)
        q = torch.from_numpy(q_np).to(dev)
        part = torch.zeros((nq, 4), dtype=torch.int32, device=dev)
        gathered = torch.zeros((n_gpus * nq, 4), dtype=torch.int32, device=dev)
        merged = torch.zeros((nq, 4), dtype=torch.int32, device=dev)

        def match_step():
            m.hamming_best2_device(q.data_ptr(), nq, table.data_ptr(), rows, rank * rows, part.data_ptr(), stream)
            if dist is not None:
                dist.all_gather_into_tensor(gathered, part)
                m.best2_merge_device(gathered.data_ptr(), n_gpus, nq, merged.data_ptr(), stream)

        for _ in range(a.warmup):
            match_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            match_step()
        barrier()
        t_match = max_over_ranks(time.perf_counter() - t0)
        pairs = float(nq) * rows * n_gpus * a.steps
        pps = pairs / t_match
        hamming = {"metric": "Hamming pair-comparisons/s", "value": pps, "queries_per_s": nq * a.steps / t_match,
                   "ms_per_step": 1e3 * t_match / a.steps, "queries": nq, "map_rows_total": rows * n_gpus,
                   # the search runs on the matrix cores as an exact int8 contraction; popcount_frac prices the same pairs
                   # against the vector pipe's xor / popcount issue peak (above 1: more than that formulation can reach)
                   "roofline": {"bound": "mfma", "achieved": round(MFMA_OPS_PER_PAIR * pps / n_gpus / 1e12, 1),
                                "peak": MFMA_I8_PEAK / 1e12, "unit": "TOP/s",
                                "frac": round(MFMA_OPS_PER_PAIR * pps / n_gpus / MFMA_I8_PEAK, 4),
                                "popcount_frac": round(OPS_PER_PAIR * pps / n_gpus / VALU_PEAK, 4)}}
        m.close()

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1)
    cpu = None
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as orc
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

        def worker(k):
            e = orc.Extractor(orc.make_params(nfeatures=a.nfeatures, nlevels=a.nlevels))
            n, t_end = 0, time.perf_counter() + 6.0
            while time.perf_counter() < t_end and n < 200:
                e.extract(uniq[(k + n) % n_unique])
                n += 1
            return n
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(nthreads) as pool:
            done = sum(pool.map(worker, range(nthreads)))
        cpu["all_cores"] = {"value": round(done / (time.perf_counter() - t0), 2), "unit": "frames/s", "cores": nthreads,
                            "sample": f"{done} frames, one oracle extractor per thread"}

    if rank == 0:
        out = {
            "metric": "ORB frames/s (1000 kp, 8 lvl, 640x480) + Hamming matches/s, 1/2/4/8 GPUs",
            "value": fps, "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * t_extract / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"ORBextractor {W}x{H} nFeatures={a.nfeatures} nLevels={a.nlevels} scale=1.2 ini=20 min=7, "
                                   f"{B} HBM-resident frames per GPU per step (BASELINE configs[0] shape, batched)",
                       "frames_per_gpu_per_step": B, "contexts_per_gpu": C, "mean_keypoints_per_frame": round(mean_kp, 1),
                       "parallelism": f"frames sharded over {n_gpus} GPU(s), no collective"},
            "roofline": roofline, "stages": stages, "hamming": hamming, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
