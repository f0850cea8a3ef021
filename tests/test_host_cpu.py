"""CPU tests of the product's host logic and of the C ABI surface (no compute call needs a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(vsa):
    header = open(os.path.join(ROOT, "include", "orbx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    lib = ctypes.CDLL(vsa.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    assert set(vsa.EXPORTED) <= declared


def test_stage_names_follow_the_header(vsa):
    """orbx_extractor_stage_ms fills ORBX_STAGE_COUNT floats in the order of the ORBX_STAGE_* enumerators: the Python view's
    stage names (and with them bench.py's per-stage accounting) must be that list."""
    header = open(os.path.join(ROOT, "include", "orbx.h")).read()
    enum = dict((name.lower(), int(value)) for name, value in re.findall(r"ORBX_STAGE_([A-Z]+) = (\d+)", header))
    count = enum.pop("count")
    assert count == len(vsa.STAGES) == len(enum)
    assert [name for name, _ in sorted(enum.items(), key=lambda kv: kv[1])] == list(vsa.STAGES)


def test_tables_match_oracle(vsa, oracle):
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (1200, 1.2, 8), (8000, 1.2, 12), (500, 1.3, 5), (1000, 2.0, 3)):
        a = vsa.scale_tables(vsa.make_params(nf, sf, nl))
        b = oracle.Extractor(oracle.make_params(nf, sf, nl)).tables()
        for k in a:
            assert np.array_equal(a[k], b[k]), (nf, sf, nl, k)


def test_level_sizes_and_capacity(vsa):
    p = vsa.make_params()
    want = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert [vsa.level_size(p, 640, 480, l) for l in range(8)] == want
    p4k = vsa.make_params(8000, 1.2, 12)
    assert vsa.level_size(p4k, 3840, 2160, 11) == (517, 291)
    assert sum(np.prod(vsa.level_size(p4k, 3840, 2160, l)) for l in range(12)) == 26804551
    cap = vsa.keypoint_capacity(p, 640, 480)
    assert 1000 + 3 * 8 <= cap <= 1000 + 5 * 8
    with pytest.raises(vsa.OrbxError) as e:
        vsa.keypoint_capacity(p, 100, 300)
    assert e.value.status == 2
    with pytest.raises(vsa.OrbxError):
        vsa.keypoint_capacity(vsa.make_params(nlevels=0), 640, 480)
    # levels too small to hold a FAST cell are accepted (they yield nothing); levels smaller than their own borders are not
    assert vsa.keypoint_capacity(vsa.make_params(500, 1.2, 8), 160, 120) >= 500
    with pytest.raises(vsa.OrbxError) as e:
        vsa.keypoint_capacity(vsa.make_params(500, 1.2, 8), 100, 80)
    assert e.value.status == 2


def test_errors_without_gpu_are_loud(vsa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vsa.OrbxError) as e:
        vsa.Extractor(vsa.make_params(), 640, 480)
    assert e.value.status == 5          # no device: the library does not fall back to a CPU path
    with pytest.raises(vsa.OrbxError) as e:
        vsa.Matcher()
    assert e.value.status == 5


def test_synthetic_generators_are_deterministic(vsa):
    a, b = vsa.synth_frame(3, 320, 240), vsa.synth_frame(3, 320, 240)
    assert np.array_equal(a, b) and not np.array_equal(a, vsa.synth_frame(4, 320, 240))
    assert a.std() > 20
    l, r = vsa.synth_stereo(3, 320, 240)
    assert l.shape == r.shape == (240, 320) and not np.array_equal(l, r)
    t = vsa.synth_descriptors(42, 1000)
    assert np.array_equal(t, vsa.synth_descriptors(42, 1000))
    bits = np.unpackbits(t).mean()
    assert 0.49 < bits < 0.51
    q, src = vsa.synth_queries(43, t, 200)
    d = np.array([vsa.descriptor_distance(q[i], t[src[i]]) for i in range(200)])
    assert 10 < d.mean() < 31 and d.max() < 60      # Binomial(256, 0.08): mean 20.5


def test_frame_grid_matches_oracle(vsa, oracle):
    rng = np.random.default_rng(0)
    n = 1500
    kps = np.zeros(n, vsa.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 752, n); kps["y"] = rng.uniform(0, 480, n); kps["octave"] = rng.integers(0, 8, n)
    kps["x"][:5] = [0.0, 751.9, 752.0, 375.5, 11.75]      # grid-edge cases of PosInGrid's round()
    bounds = (0.0, 752.0, 0.0, 480.0)
    g, o = vsa.FrameGrid(kps, bounds), oracle.FrameGrid(kps, bounds)
    for _ in range(400):
        x, y = rng.uniform(-50, 800), rng.uniform(-50, 530)
        r = float(rng.choice([1.0, 7.5, 15.0, 40.0, 100.0, 1000.0]))
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        a, b = g.features_in_area(x, y, r, lo, hi), o.features_in_area(x, y, r, lo, hi)
        assert np.array_equal(a, b)
    assert len(g.features_in_area(376.0, 240.0, 1000.0)) > 1400


def test_best2_merge_host(vsa, oracle):
    t = vsa.synth_descriptors(9, 4000)
    t[2000:2010] = t[100:110]                          # duplicates across shards: lowest global index wins
    q, _ = vsa.synth_queries(10, t, 300)
    whole = oracle.hamming_best2(q, t)
    for shards in (1, 2, 3, 8):
        bounds = np.linspace(0, 4000, shards + 1).astype(int)
        parts = []
        for s in range(shards):
            part = oracle.hamming_best2(q, t[bounds[s]:bounds[s + 1]])
            part["best_idx"] = np.where(part["best_idx"] >= 0, part["best_idx"] + bounds[s], -1)
            parts.append(part)
        merged = vsa.best2_merge(np.stack(parts))
        for f in ("best", "best_idx", "second"):
            assert np.array_equal(merged[f], whole[f]), (shards, f)


def test_mat_record_layout(vsa):
    """The cv::Mat payload of the reference's map files (include/MapPoint.h:466-501): cols, rows as int32, elemSize and
    type as 64-bit, then the bytes."""
    desc = (np.arange(3 * 32, dtype=np.uint32) * 7 % 256).astype(np.uint8).reshape(3, 32)
    rec = vsa.mat_record_write(desc)
    assert len(rec) == 24 + 96
    assert rec[:24] == (32).to_bytes(4, "little") + (3).to_bytes(4, "little") + (1).to_bytes(8, "little") + (0).to_bytes(8, "little")
    assert rec[24:] == desc.tobytes()
    back, es, et, n = vsa.mat_record_read(rec + b"trailing")
    assert n == len(rec) and es == 1 and et == 0 and np.array_equal(back, desc)
    pose = np.arange(16, dtype=np.float32).reshape(4, 4)                      # a CV_32F matrix (type code 5), e.g. a pose
    back, es, et, _ = vsa.mat_record_read(vsa.mat_record_write(pose, elem_type=5))
    assert es == 4 and et == 5 and np.array_equal(back.view(np.float32), pose)
    with pytest.raises(ValueError):
        vsa.mat_record_read(rec[:50])
    empty, _, _, n0 = vsa.mat_record_read(vsa.mat_record_write(np.zeros((0, 32), np.uint8)))
    assert n0 == 24 and empty.shape == (0, 32)
