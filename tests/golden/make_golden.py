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
    around_fixture(k1, d1, k2, d2, scale)


def around_fixture(k1, d1, k2, d2, scale):
    """The steps either side of extraction and the remaining matchers, on small inputs."""
    rng = np.random.default_rng(7)
    w, h = 96, 64
    bgr = np.stack([vsa.synth_frame(50 + c, w, h) for c in range(3)], axis=2)
    grey = orc.cvt_color_to_gray(bgr, blue_first=True)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    map_x = (xx * np.float32(0.97) + yy * np.float32(0.02) + np.float32(1.3)).astype(np.float32)
    map_y = (yy * np.float32(1.02) - xx * np.float32(0.015) - np.float32(0.8)).astype(np.float32)
    rect = orc.remap_linear_u8(grey, map_x, map_y)
    cam = np.array([517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    pts = np.stack([rng.uniform(0, 640, 200), rng.uniform(0, 480, 200)], 1).astype(np.float32)
    undist = orc.undistort_points(pts, *[float(c) for c in cam])
    voc = vsa.synth_vocabulary(11, 4, 3)
    word, node, bow_ids, bow_vals, fv_nodes, fv_ptr, fv_feats = orc.vocabulary_transform(4, 3, *voc, d1[:300], 2)
    sizes = rng.integers(1, 9, 40)
    gptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    gdesc = d1[rng.integers(0, len(d1), int(gptr[-1]))] ^ np.packbits(rng.random((int(gptr[-1]), 256)) < 0.1, axis=1)
    distinct = np.array([orc.distinctive_descriptor(gdesc[gptr[g]:gptr[g + 1]]) for g in range(40)], np.int32)
    bounds = (0.0, 320.0, 0.0, 240.0)
    lvl = k1["octave"].astype(np.int32)
    qu, qv = (k1["x"] + 4).astype(np.float32), (k1["y"] + 2).astype(np.float32)
    fuse_idx, fuse_dist = orc.fuse_candidates(qu, qv, (3.0 * scale[lvl]).astype(np.float32), lvl, d1, k2, d2, bounds)
    lvl2 = k2["octave"].astype(np.int32)
    s1 = (np.ones(len(k1), np.uint8), qu, qv, (7.5 * scale[lvl]).astype(np.float32), lvl, d1)
    s2 = (np.ones(len(k2), np.uint8), (k2["x"] - 4).astype(np.float32), (k2["y"] - 2).astype(np.float32), (7.5 * scale[lvl2]).astype(np.float32), lvl2, d2)
    sim3, nsim3 = orc.search_by_sim3(s1, k1, d1, s2, k2, d2, bounds)
    nodes1, nodes2 = (d1[:, 0] >> 4).astype(np.int64), (d2[:, 0] >> 4).astype(np.int64)
    b12, b21, nb = orc.search_by_bow(d1, k1["angle"], np.ones(len(k1), np.uint8), nodes1, d2, k2["angle"], None, nodes2, 0.7, True, 0)
    np.savez_compressed(os.path.join(HERE, "around_96x64.npz"), bgr=bgr, grey=grey, map_x=map_x, map_y=map_y, rect=rect, cam=cam, pts=pts,
                        undist=undist, voc_parent=voc[0], voc_leaf=voc[1], voc_desc=voc[2], voc_weight=voc[3], word=word, node=node,
                        bow_ids=bow_ids, bow_vals=bow_vals, fv_nodes=fv_nodes, fv_ptr=fv_ptr, fv_feats=fv_feats, gptr=gptr, gdesc=gdesc,
                        distinct=distinct, fuse_idx=fuse_idx, fuse_dist=fuse_dist, sim3=sim3, nsim3=np.int32(nsim3), bow21=b21, nbow=np.int32(nb))
    print("around: words", len(bow_ids), "fused<=50", int((fuse_dist <= 50).sum()), "sim3", nsim3, "bow", nb)


if __name__ == "__main__":
    main()
