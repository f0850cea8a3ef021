#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle.

The reference ships no test vectors and cannot be run here (its pixel arithmetic is OpenCV, absent from
this image), so these fixtures pin the HIP path and the oracle against each other and against
regressions -- not the oracle against a real OpenCV build.  Inputs are seeded synthetic frames.
Run from the repository root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
vsa = importlib.import_module("visual-slam_amd")
import oracle_lib as orc


def extract_fixture(name, w, h, frame, nf, sf, nl, ini, mn, full):
    img = vsa.synth_frame(frame, w, h)
    ex = orc.Extractor(orc.make_params(nf, sf, nl, ini, mn))
    kps, desc = ex.extract(img)
    out = {"params": np.array([nf, nl, ini, mn], np.int32), "scale_factor": np.float32(sf), "image": img, "kps": kps, "desc": desc}
    if full:
        for l in range(nl):
            out[f"pyr{l}"] = ex.level_padded(l)
            x, y, s = ex.level_candidates(l)
            out[f"cand{l}"] = vsa.pack_keys(x, y, s)
            out[f"lvlkp{l}"] = ex.level_keypoints(l)
    np.savez_compressed(os.path.join(HERE, name), **out)
    return img, kps, desc


def main():
    extract_fixture("extract_160x120.npz", 160, 120, 6, 300, 1.2, 2, 20, 7, True)
    img, k1, d1 = extract_fixture("extract_320x240.npz", 320, 240, 7, 500, 1.2, 4, 20, 7, False)
    rng = np.random.default_rng(1)
    img2 = np.clip(np.roll(img, (2, 4), axis=(0, 1)).astype(np.int32) + rng.integers(-2, 3, img.shape), 0, 255).astype(np.uint8)
    k2, d2 = orc.Extractor(orc.make_params(500, 1.2, 4)).extract(img2)
    bounds = (0.0, 320.0, 0.0, 240.0)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    m12, n, prev_out = orc.search_for_initialization(k1, d1, k2, d2, bounds, prev, 30, 0.9, True)
    scale = orc.Extractor(orc.make_params(500, 1.2, 4)).tables()["scale"]
    lvl = k1["octave"].astype(np.int32)
    claimed0 = np.full(len(k2), -1, np.int32)
    qu, qv = (k1["x"] + 4).astype(np.float32), (k1["y"] + 2).astype(np.float32)
    radius = (15.0 * scale[lvl]).astype(np.float32)
    cl_last, n_last = orc.search_by_projection_lastframe(qu, qv, radius, qu, lvl, k1["angle"], d1, k2, d2, None, bounds, 0, 100, True, claimed0)
    qr = (4.0 * scale[lvl]).astype(np.float32)
    cl_mp, n_mp = orc.search_by_projection_mappoints(qu, qv, qr, qu, lvl, d1, k2, d2, None, bounds, 0.8, claimed0)
    sd, si = orc.stereo_match(k1, d1, k2, d2, scale, 240, 0.0, 40.0)
    table = vsa.synth_descriptors(42, 600)
    q, src = vsa.synth_queries(43, table, 64)
    best2 = orc.hamming_best2(q, table)
    np.savez_compressed(os.path.join(HERE, "match_320x240.npz"), k1=k1, d1=d1, k2=k2, d2=d2, image2=img2,
                        init_matches12=m12, init_n=np.int32(n), init_prev=prev_out,
                        last_claimed=cl_last, last_n=np.int32(n_last), mp_claimed=cl_mp, mp_n=np.int32(n_mp),
                        stereo_dist=sd, stereo_idx=si, table=table, queries=q, best2=best2)
    print("init", n, "lastframe", n_last, "mappoints", n_mp, "stereo<75", int(((sd >= 0) & (sd < 75)).sum()))


if __name__ == "__main__":
    main()
