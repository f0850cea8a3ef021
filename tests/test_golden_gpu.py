"""GPU results against the committed golden fixtures: no oracle code runs in these tests."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_extract_fixture_full(vsa):
    g = np.load(os.path.join(GOLD, "extract_160x120.npz"))
    nf, nl, ini, mn = g["params"].tolist()
    ex = vsa.Extractor(vsa.make_params(nf, float(g["scale_factor"]), nl, ini, mn), 160, 120, device=0)
    kps, desc = ex.extract(g["image"])
    for l in range(nl):
        assert np.array_equal(ex.pyramid_level(0, l), g[f"pyr{l}"])
        assert np.array_equal(ex.debug_candidates(0, l), g[f"cand{l}"])
        gx, gy, gs = vsa.unpack_keys(ex.debug_level_keypoints(0, l))
        want = g[f"lvlkp{l}"]
        assert np.array_equal(gx + 16, want["x"].astype(np.int32)) and np.array_equal(gy + 16, want["y"].astype(np.int32))
    assert np.array_equal(kps, g["kps"].astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, g["desc"])
    ex.close()


def test_extract_fixture_320(vsa):
    g = np.load(os.path.join(GOLD, "extract_320x240.npz"))
    nf, nl, ini, mn = g["params"].tolist()
    ex = vsa.Extractor(vsa.make_params(nf, float(g["scale_factor"]), nl, ini, mn), 320, 240, device=0)
    kps, desc = ex.extract(g["image"])
    assert np.array_equal(kps, g["kps"].astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, g["desc"])
    ex.close()


def test_match_fixture(vsa):
    g = np.load(os.path.join(GOLD, "match_320x240.npz"))
    k1, d1, k2, d2 = g["k1"], g["d1"], g["k2"], g["d2"]
    m = vsa.Matcher(device=0)
    bounds = (0.0, 320.0, 0.0, 240.0)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    m12, n, prev_out = m.search_for_initialization(k1, d1, k2, d2, bounds, prev, 30, 0.9, True)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_matches12"]) and np.array_equal(prev_out, g["init_prev"])
    scale = vsa.scale_tables(vsa.make_params(500, 1.2, 4))["scale"]
    lvl = k1["octave"].astype(np.int32)
    claimed0 = np.full(len(k2), -1, np.int32)
    qu, qv = (k1["x"] + 4).astype(np.float32), (k1["y"] + 2).astype(np.float32)
    cl, nn = m.search_by_projection_lastframe(qu, qv, (15.0 * scale[lvl]).astype(np.float32), qu, lvl, k1["angle"], d1, k2, d2, None,
                                              bounds, 0, 100, True, claimed0)
    assert nn == int(g["last_n"]) and np.array_equal(cl, g["last_claimed"])
    cl, nn = m.search_by_projection_mappoints(qu, qv, (4.0 * scale[lvl]).astype(np.float32), qu, lvl, d1, k2, d2, None, bounds, 0.8, claimed0)
    assert nn == int(g["mp_n"]) and np.array_equal(cl, g["mp_claimed"])
    sd, si = m.stereo_match(k1, d1, k2, d2, scale, 240, 0.0, 40.0)
    assert np.array_equal(sd, g["stereo_dist"]) and np.array_equal(si, g["stereo_idx"])
    got = m.hamming_best2(g["queries"], g["table"])
    assert all(np.array_equal(got[f], g["best2"][f]) for f in ("best", "best_idx", "second"))
    m.close()


def test_around_fixture(vsa):
    """Colour input, rectification, undistortion, vocabulary transform, distinctive descriptors and the remaining matchers
    against the committed fixture."""
    g = np.load(os.path.join(GOLD, "around_96x64.npz"))
    mt = np.load(os.path.join(GOLD, "match_320x240.npz"))
    k1, d1, k2, d2 = mt["k1"], mt["d1"], mt["k2"], mt["d2"]
    h, w = g["grey"].shape
    ex = vsa.Extractor(vsa.make_params(100, 1.2, 1), w, h, device=0)
    ex.set_input_format("bgr")
    ex.extract(g["bgr"])
    assert np.array_equal(ex.pyramid_level(0, 0)[19:-19, 19:-19], g["grey"])
    cam = vsa.Camera(*[float(c) for c in g["cam"]])
    assert np.array_equal(ex.undistort_points(cam, g["pts"]), g["undist"])
    ex.close()
    rm = vsa.Remapper(g["map_x"], g["map_y"], w, h, device=0)
    assert np.array_equal(rm.remap(g["grey"]), g["rect"])
    rm.close()
    voc = vsa.Vocabulary(4, 3, g["voc_parent"], g["voc_leaf"], g["voc_desc"], g["voc_weight"], device=0)
    ids, vals, fn, fp, ff = voc.transform(d1[:300], 2)
    assert np.array_equal(ids, g["bow_ids"]) and np.array_equal(vals, g["bow_vals"])
    assert np.array_equal(fn, g["fv_nodes"]) and np.array_equal(fp, g["fv_ptr"]) and np.array_equal(ff, g["fv_feats"])
    voc.close()
    m = vsa.Matcher(device=0)
    assert np.array_equal(m.distinctive_descriptors(g["gdesc"], g["gptr"]), g["distinct"])
    scale = vsa.scale_tables(vsa.make_params(500, 1.2, 4))["scale"]
    bounds = (0.0, 320.0, 0.0, 240.0)
    lvl = k1["octave"].astype(np.int32); lvl2 = k2["octave"].astype(np.int32)
    qu, qv = (k1["x"] + 4).astype(np.float32), (k1["y"] + 2).astype(np.float32)
    bi, bd = m.project_best1(qu, qv, (3.0 * scale[lvl]).astype(np.float32), lvl, d1, k2, d2, bounds)
    assert np.array_equal(bi, g["fuse_idx"]) and np.array_equal(bd, g["fuse_dist"])
    s1 = (np.ones(len(k1), np.uint8), qu, qv, (7.5 * scale[lvl]).astype(np.float32), lvl, d1)
    s2 = (np.ones(len(k2), np.uint8), (k2["x"] - 4).astype(np.float32), (k2["y"] - 2).astype(np.float32), (7.5 * scale[lvl2]).astype(np.float32), lvl2, d2)
    m12, n = m.search_by_sim3(s1, k1, d1, s2, k2, d2, bounds)
    assert n == int(g["nsim3"]) and np.array_equal(m12, g["sim3"])
    nodes1, nodes2 = (d1[:, 0] >> 4).astype(np.int64), (d2[:, 0] >> 4).astype(np.int64)
    r = m.search_by_bow(d1, k1["angle"], np.ones(len(k1), np.uint8), nodes1, d2, k2["angle"], None, nodes2, 0.7, True, 0)
    assert r[2] == int(g["nbow"]) and np.array_equal(r[1], g["bow21"])
    m.close()
