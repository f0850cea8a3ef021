#!/usr/bin/env python3
"""Stage times of two (or more) builds of the library on the same box, alternating: `ab_probe.py a.so b.so [...]`.
Each build runs in its own process (ORBX_LIB_PATH); medians over the timed steps, three rounds per build."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    vsa = importlib.import_module("visual-slam_amd")
    W, H, NF, NL, B = (int(v) for v in sys.argv[2:7])
    uniq = np.stack([vsa.synth_frame(i, W, H) for i in range(min(32, B))])
    frames = torch.from_numpy(uniq).cuda().repeat(B // len(uniq), 1, 1).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    ex = vsa.Extractor(vsa.make_params(NF, 1.2, NL, 20, 7), W, H, max_batch=B, device=0)
    cap = ex.cap
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.set_profiling(True)
    rows = []
    for it in range(25):
        ex.extract_batch_device(frames.data_ptr(), W, W * H, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
        torch.cuda.synchronize()
        if it >= 5:
            rows.append(ex.stage_ms())
    med = np.median(np.array(rows), axis=0) * 1e3
    print(" ".join(f"{n} {v:.1f}" for n, v in zip(vsa.STAGES, med)), f"| sum {med.sum():.1f} us | kp {counts.float().mean().item():.1f}", flush=True)
    sys.exit(0)

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
shape = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["640", "480", "1000", "8", "256"]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, ORBX_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "--child"] + shape, env=env, capture_output=True, text=True, timeout=300)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-500:]}", flush=True)
        if r.returncode != 0:
            sys.exit(1)
