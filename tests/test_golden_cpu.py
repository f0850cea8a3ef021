"""The oracle reproduces the committed golden fixtures (regression pin for the checker itself)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_extract_fixture(oracle, vsa):
    g = np.load(os.path.join(GOLD, "extract_160x120.npz"))
    nf, nl, ini, mn = g["params"].tolist()
    assert np.array_equal(g["image"], vsa.synth_frame(6, 160, 120))
    ex = oracle.Extractor(oracle.make_params(nf, float(g["scale_factor"]), nl, ini, mn))
    kps, desc = ex.extract(g["image"])
    assert np.array_equal(kps, g["kps"]) and np.array_equal(desc, g["desc"])
    for l in range(nl):
        assert np.array_equal(ex.level_padded(l), g[f"pyr{l}"])
        x, y, s = ex.level_candidates(l)
        assert np.array_equal(vsa.pack_keys(x, y, s), g[f"cand{l}"])
        assert np.array_equal(ex.level_keypoints(l), g[f"lvlkp{l}"])


def test_oracle_reproduces_match_fixture(oracle):
    g = np.load(os.path.join(GOLD, "match_320x240.npz"))
    k1, d1, k2, d2 = g["k1"], g["d1"], g["k2"], g["d2"]
    bounds = (0.0, 320.0, 0.0, 240.0)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    m12, n, prev_out = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev, 30, 0.9, True)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_matches12"]) and np.array_equal(prev_out, g["init_prev"])
    assert np.array_equal(oracle.hamming_best2(g["queries"], g["table"]), g["best2"])


def test_oracle_reproduces_around_fixture(oracle):
    g = np.load(os.path.join(GOLD, "around_96x64.npz"))
    assert np.array_equal(oracle.cvt_color_to_gray(g["bgr"], blue_first=True), g["grey"])
    assert np.array_equal(oracle.remap_linear_u8(g["grey"], g["map_x"], g["map_y"]), g["rect"])
    assert np.array_equal(oracle.undistort_points(g["pts"], *[float(c) for c in g["cam"]]), g["undist"])
    m = np.load(os.path.join(GOLD, "match_320x240.npz"))
    r = oracle.vocabulary_transform(4, 3, g["voc_parent"], g["voc_leaf"], g["voc_desc"], g["voc_weight"], m["d1"][:300], 2)
    for got, name in zip(r, ("word", "node", "bow_ids", "bow_vals", "fv_nodes", "fv_ptr", "fv_feats")):
        assert np.array_equal(got, g[name]), name
    gp = g["gptr"]
    assert [oracle.distinctive_descriptor(g["gdesc"][gp[i]:gp[i + 1]]) for i in range(len(gp) - 1)] == g["distinct"].tolist()
