"""CPU tests of the oracle itself: the known-answer values that can be derived by reading the reference
(no golden vectors ship with it) and the hand-computable cases of the OpenCV primitives."""
import numpy as np
import pytest


def test_quotas(oracle):
    cases = {
        (1000, 1.2, 8): [217, 181, 151, 126, 105, 87, 73, 60],
        (2000, 1.2, 8): [434, 362, 302, 251, 209, 175, 145, 122],
        (1200, 1.2, 8): [261, 217, 181, 151, 126, 105, 87, 72],
        (8000, 1.2, 12): [1502, 1251, 1043, 869, 724, 604, 503, 419, 349, 291, 243, 202],
    }
    for (nf, sf, nl), want in cases.items():
        t = oracle.Extractor(oracle.make_params(nf, sf, nl)).tables()
        assert t["quota"].tolist() == want
        assert t["quota"].sum() == nf


def test_umax_and_scale_tables(oracle):
    t = oracle.Extractor(oracle.make_params(1000, 1.2, 12)).tables()
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert 2 * sum(2 * u + 1 for u in t["umax"][1:]) + 2 * t["umax"][0] + 1 == 749
    want = [1.0, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(t["scale"][:8], want, rtol=0, atol=5e-10 * 4)
    assert [int(31 * s) for s in t["scale"]] == [31, 37, 44, 53, 64, 77, 92, 111, 133, 159, 191, 230]
    assert np.array_equal(t["inv_scale"], (np.float32(1.0) / t["scale"]).astype(np.float32))
    assert np.array_equal(t["sigma2"], t["scale"] * t["scale"])


def test_level_sizes(oracle, vsa):
    table = {
        (640, 480, 8): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
        (752, 480, 8): [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)],
        (1241, 376, 8): [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)],
    }
    for (w, h, nl), want in table.items():
        ex = oracle.Extractor(oracle.make_params(1000, 1.2, nl))
        ex.extract(np.zeros((h, w), np.uint8))
        assert [ex.level_size(l) for l in range(nl)] == want
        assert sum(a * b for a, b in want) in (950532, 1117367, 1444097)


def test_cv_round_and_atan2(oracle):
    assert [oracle.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert oracle.fast_atan2(0, 0) == 0.0
    for y, x, deg in ((0, 1, 0), (1, 0, 90), (0, -1, 180), (-1, 0, 270), (1, 1, 45), (-1, -1, 225)):
        assert abs(oracle.fast_atan2(y, x) - deg) < 0.02
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-2 ** 20, 2 ** 20, 2)
        a = oracle.fast_atan2(float(y), float(x))
        assert 0 <= a <= 360
        assert abs((a - np.degrees(np.arctan2(y, x))) % 360) < 0.02 or abs((a - np.degrees(np.arctan2(y, x))) % 360) > 359.98


def numpy_fast_atan2(y, x):
    """cv::fastAtan2 in numpy float32 arithmetic (every numpy float32 operation is one IEEE single-precision operation, so
    no fusing and no wider intermediates)."""
    f = np.float32
    rad2deg = f(180.0 / 3.1415926535897932384626433832795)
    p1, p3, p5, p7 = (f(v) * rad2deg for v in (0.9997878412794807, -0.3258083974640975, 0.1555786518463281, -0.04432655554792128))
    eps = f(2.220446049250313e-16)
    y = np.asarray(y, np.float32); x = np.asarray(x, np.float32)
    ax, ay = np.abs(x), np.abs(y)
    swap = ax < ay
    num = np.where(swap, ax, ay); den = np.where(swap, ay, ax) + eps
    c = num / den
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(swap, f(90.0) - a, a)
    a = np.where(x < 0, f(180.0) - a, a)
    a = np.where(y < 0, f(360.0) - a, a)
    return a.astype(np.float32)


def test_fast_atan2_bitwise_against_numpy_float32(oracle):
    """cv::fastAtan2 (called at src/ORBextractor.cc:102) restated a second time, in numpy float32 arithmetic: the oracle's C
    version must agree bit for bit on a million (y, x) pairs, among them integer-valued moments as IC_Angle produces."""
    f = np.float32
    np_atan2 = numpy_fast_atan2
    rng = np.random.default_rng(11)
    n = 1_000_000
    y = np.concatenate([rng.integers(-2 ** 23, 2 ** 23, n // 2).astype(np.float32), rng.standard_normal(n // 2).astype(np.float32) * f(1e3)])
    x = np.concatenate([rng.integers(-2 ** 23, 2 ** 23, n // 2).astype(np.float32), rng.standard_normal(n // 2).astype(np.float32) * f(1e3)])
    y[:2000] = 0; x[1000:3000] = 0; y[3000:4000] = x[3000:4000]; y[4000:5000] = -x[4000:5000]
    got = oracle.fast_atan2_array(y, x)
    want = np_atan2(y, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
    assert got.min() >= 0 and got.max() <= 360


def test_orientation_and_descriptor_against_numpy(oracle, vsa):
    """IC_Angle (src/ORBextractor.cc:76-103) and computeOrbDescriptor (:107-146) restated a second time in numpy, on the
    oracle's own intermediate images and selected keypoints: the moments as plain sums over the umax disc, the angle
    through the numpy fastAtan2 above, the 256 tests with float32 products and sums (unfused, as compiled C++ without
    contraction gives) and round-half-even, bits LSB first.  Angles and descriptors must agree bit for bit with the C
    oracle's (trig mode 0: cos / sin correctly rounded from double)."""
    import os, re
    f = np.float32
    here = os.path.dirname(os.path.abspath(__file__))
    txt = open(os.path.join(here, "..", "include", "orbx_pattern.inc")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    pat = np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int32).reshape(256, 4)
    assert pat.min() == -13 and pat.max() == 12
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    w, h = 320, 240
    ex = oracle.Extractor(oracle.make_params(300, 1.2, 4))
    kps, desc = ex.extract(vsa.synth_frame(77, w, h))
    assert len(kps) > 200
    factor = f(3.1415926535897932384626433832795 / 180.0)
    row = 0
    for level in range(4):
        lk = ex.level_keypoints(level)
        if len(lk) == 0:
            continue
        img = ex.level_padded(level).astype(np.int64)[19:-19, 19:-19]
        blur = ex.level_blurred(level).astype(np.int32)
        for k in lk:
            x, y = int(k["x"]), int(k["y"])
            m10 = m01 = 0
            for v in range(-15, 16):
                d = umax[abs(v)]
                seg = img[y + v, x - d:x + d + 1]
                m10 += int((seg * np.arange(-d, d + 1)).sum()); m01 += v * int(seg.sum())
            angle = numpy_fast_atan2(f(m01), f(m10))
            assert angle.view(np.uint32) == np.float32(k["angle"]).view(np.uint32), (level, x, y)
            theta = f(angle) * factor
            a = f(np.cos(np.float64(theta))); b = f(np.sin(np.float64(theta)))
            px0, py0, px1, py1 = (pat[:, i].astype(np.float32) for i in range(4))
            r0 = np.rint(px0 * b + py0 * a).astype(np.int32); c0 = np.rint(px0 * a - py0 * b).astype(np.int32)
            r1 = np.rint(px1 * b + py1 * a).astype(np.int32); c1 = np.rint(px1 * a - py1 * b).astype(np.int32)
            bits = (blur[y + r0, x + c0] < blur[y + r1, x + c1]).astype(np.uint8)
            want = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").ravel()
            assert np.array_equal(desc[row], want), (level, x, y)
            row += 1
    assert row == len(kps)
    ex.close()


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def numpy_fast9(img, threshold, nonmax=True):
    """FAST-9 straight from its definition, independent of the oracle's min/max formulation: a pixel is a corner at
    threshold t when 9 contiguous pixels of its 16-pixel ring are all brighter than v + t or all darker than v - t; its
    score is the largest t at which it still is one (found by bisection on t, per pixel); with nonmax a corner stays
    only if its score is strictly above its 8 neighbours' (non-corners and pixels outside the detection area, which
    excludes a 3-pixel margin, count as 0).  Returns x, y, score in raster order."""
    v = img.astype(np.int32)
    H, W = v.shape
    c = v[3:H - 3, 3:W - 3]
    d = np.stack([c - v[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in RING])      # v - p_k

    def corner(t):                                        # t: scalar or per-pixel array
        dark = d > t; bright = d < -t
        hit = np.zeros(c.shape, bool)
        for k in range(16):
            idx = [(k + j) % 16 for j in range(9)]
            hit |= dark[idx].all(axis=0) | bright[idx].all(axis=0)
        return hit

    is_corner = corner(threshold)
    lo = np.full(c.shape, threshold, np.int32)            # largest t known to fire (for corners)
    hi = np.full(c.shape, 256, np.int32)                  # smallest t known not to
    while np.any(hi - lo > 1):
        mid = (lo + hi) // 2
        f = corner(mid)
        lo = np.where(f, mid, lo); hi = np.where(f, hi, mid)
    score = np.where(is_corner, lo, 0)
    if nonmax:
        pad = np.pad(score, 1)
        keep = is_corner.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= score > pad[1 + dy:1 + dy + score.shape[0], 1 + dx:1 + dx + score.shape[1]]
    else:
        keep = is_corner
    ys, xs = np.nonzero(keep)
    return xs + 3, ys + 3, score[ys, xs]


def test_fast9_against_definition(oracle, vsa):
    """The oracle's FAST (OpenCV's arc-minimum score and rolling-row non-maximum suppression) against the definition
    above, on noise, on the synthetic frames of the benchmarks, on low-contrast data and on narrow sub-images like the
    extractor's cells; thresholds as the settings files use them."""
    rng = np.random.default_rng(21)
    images = [rng.integers(0, 256, (48, 57), dtype=np.uint8),
              vsa.synth_frame(3, 160, 120),
              ((vsa.synth_frame(4, 120, 90).astype(np.int32) - 128) // 5 + 128).astype(np.uint8),
              rng.integers(100, 140, (36, 36), dtype=np.uint8),
              np.kron(rng.integers(0, 2, (8, 9)) * 90 + 40, np.ones((6, 6))).astype(np.uint8),
              rng.integers(0, 256, (7, 40), dtype=np.uint8),          # one detection row
              rng.integers(0, 256, (6, 40), dtype=np.uint8)]          # too small: nothing
    total = 0
    for img in images:
        for t in (7, 20, 1, 40):
            for nonmax in (False, True):
                gx, gy, gs = oracle.fast9(img, t, nonmax=nonmax)
                if img.shape[0] < 7:
                    assert len(gx) == 0
                    continue
                wx, wy, ws = numpy_fast9(img, t, nonmax=nonmax)
                assert len(gx) == len(wx), (img.shape, t, nonmax, len(gx), len(wx))
                assert np.array_equal(gx, wx) and np.array_equal(gy, wy), (img.shape, t, nonmax)
                if nonmax:                                # without suppression cv::FAST leaves the response 0
                    assert np.array_equal(gs, ws), (img.shape, t)
                else:
                    assert all(oracle.fast9_score(img, int(a), int(b), t) == int(c) for a, b, c in list(zip(wx, wy, ws))[:300])
                total += len(gx)
    assert total > 5000


def test_cell_loop_against_numpy(oracle, vsa):
    """The FAST cell loop of ComputeKeyPointsOctTree (src/ORBextractor.cc:764-828) restated a second time in Python over the
    numpy FAST above: borders, cell grid (`nCols = width / 30`, `wCell = ceil(width / nCols)`), the clamped 6-pixel overlap,
    the skipped cells, `FAST(ini)` then `FAST(min)` only for a cell left empty, cell offsets added to the coordinates, cells
    row by row.  The oracle's candidate list of every level must be the same keys in the same order."""
    import math
    rng = np.random.default_rng(8)
    flat = ((vsa.synth_frame(9, 260, 200).astype(np.int32) - 128) // 6 + 128).astype(np.uint8)      # low contrast: many cells fall back
    half = vsa.synth_frame(10, 300, 170).copy(); half[:, 150:] = 90                                 # cells with nothing at either threshold
    cases = [(vsa.synth_frame(8, 320, 240), 500, 4, 20, 7), (flat, 300, 3, 20, 7), (half, 300, 3, 20, 7),
             (rng.integers(0, 256, (131, 167), dtype=np.uint8), 400, 2, 40, 12)]
    fallbacks = empties = total = 0
    for img, nf, nl, ini, mn in cases:
        ex = oracle.Extractor(oracle.make_params(nf, 1.2, nl, ini, mn))
        ex.extract(img)
        for level in range(nl):
            im = ex.level_padded(level)[19:-19, 19:-19]            # mvImagePyramid[level]
            rows, cols = im.shape
            min_b, max_bx, max_by = 16, cols - 16, rows - 16
            width, height = np.float32(max_bx - min_b), np.float32(max_by - min_b)
            n_cols, n_rows = int(width / np.float32(30)), int(height / np.float32(30))
            want = []
            if n_cols > 0 and n_rows > 0:
                w_cell, h_cell = math.ceil(width / n_cols), math.ceil(height / n_rows)
                for i in range(n_rows):
                    ini_y = min_b + i * h_cell
                    max_y = ini_y + h_cell + 6
                    if ini_y >= max_by - 3:
                        continue
                    max_y = min(max_y, max_by)
                    for j in range(n_cols):
                        ini_x = min_b + j * w_cell
                        max_x = ini_x + w_cell + 6
                        if ini_x >= max_bx - 6:
                            continue
                        max_x = min(max_x, max_bx)
                        sub = im[ini_y:max_y, ini_x:max_x]
                        if sub.shape[0] < 7 or sub.shape[1] < 7:
                            continue
                        xs, ys, sc = numpy_fast9(sub, ini)
                        if len(xs) == 0:
                            xs, ys, sc = numpy_fast9(sub, mn)
                            fallbacks += len(xs) > 0
                            empties += len(xs) == 0
                        want += [(int(x) + j * w_cell, int(y) + i * h_cell, int(v)) for x, y, v in zip(xs, ys, sc)]
            cx, cy, cs = ex.level_candidates(level)
            got = list(zip(cx.tolist(), cy.tolist(), cs.tolist()))
            assert got == want, (img.shape, level, len(got), len(want))
            total += len(got)
        ex.close()
    assert total > 3000 and fallbacks > 10 and empties > 3, (total, fallbacks, empties)


def test_gauss_taps_and_blur(oracle):
    assert oracle.gauss_taps().tolist() == [18, 34, 49, 55, 49, 34, 18]
    const = np.full((20, 24), 77, np.uint8)
    # taps sum to 257 (not renormalised): 77*257*257 = 5085773; (x + 32768) >> 16 = 78
    assert (oracle.gauss7(const) == 78).all()
    assert (oracle.gauss7(np.full((9, 9), 255, np.uint8)) == 255).all()      # saturates
    imp = np.zeros((15, 15), np.uint8); imp[7, 7] = 255
    out = oracle.gauss7(imp)
    taps = np.array([18, 34, 49, 55, 49, 34, 18])
    want = (np.outer(taps, taps) * 255 + 32768) >> 16
    assert np.array_equal(out[4:11, 4:11], want)
    assert out.sum() == want.sum()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (41, 53), dtype=np.uint8)
    a, b = oracle.gauss7(img, 0), oracle.gauss7(img, 1)
    # the two rounding modes differ only on exact .5 ties and never in the last w%4 columns
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert np.array_equal(a[:, 52:], b[:, 52:])
    pad = np.pad(img, 3, mode="reflect").astype(np.int64)
    rows = sum(taps[k] * pad[3:-3, k:k + 53] for k in range(7))
    rows = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    full = sum(taps[k] * rows[k:k + 41, :] for k in range(7))
    assert np.array_equal(a, np.minimum((full + 32768) >> 16, 255).astype(np.uint8))


def test_border101(oracle):
    img = np.arange(5 * 7, dtype=np.uint8).reshape(5, 7)
    assert np.array_equal(oracle.border101(img, 3), np.pad(img, 3, mode="reflect"))
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    assert np.array_equal(oracle.border101(img, 19), np.pad(img, 19, mode="reflect"))


def test_resize_properties(oracle):
    const = np.full((48, 64), 201, np.uint8)
    assert (oracle.resize_linear(const, 53, 40) == 201).all()
    # exact 2x decimation is rerouted to INTER_AREA: rounded 2x2 mean
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    a = img.astype(int)
    want = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(oracle.resize_linear(img, 30, 20), want.astype(np.uint8))
    # horizontal ramp stays monotone and inside the source range
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))
    out = oracle.resize_linear(ramp, 100, 25)
    assert (np.diff(out.astype(int), axis=1) >= 0).all() and out.min() >= 0 and out.max() <= 238
    # independent restatement of the fixed-point formula in numpy for one random case
    src = rng.integers(0, 256, (37, 45), dtype=np.uint8)
    dw, dh = 38, 31
    sx_scale, sy_scale = 1.0 / (dw / 45), 1.0 / (dh / 37)
    out = oracle.resize_linear(src, dw, dh)
    ref = np.zeros((dh, dw), np.uint8)
    for dy in range(dh):
        fy = np.float32((dy + 0.5) * sy_scale - 0.5); sy = int(np.floor(fy)); fy = np.float32(fy - np.float32(sy))
        b0 = int(np.rint(np.float32(np.float32(1) - fy) * np.float32(2048))); b1 = int(np.rint(fy * np.float32(2048)))
        r0, r1 = min(max(sy, 0), 36), min(max(sy + 1, 0), 36)
        for dx in range(dw):
            fx = np.float32((dx + 0.5) * sx_scale - 0.5); sx = int(np.floor(fx)); fx = np.float32(fx - np.float32(sx))
            if sx >= 44:
                sx, fx = 44, np.float32(0)
            a0 = int(np.rint(np.float32(np.float32(1) - fx) * np.float32(2048))); a1 = int(np.rint(fx * np.float32(2048)))
            sx1 = min(sx + 1, 44)
            h0 = int(src[r0, sx]) * a0 + int(src[r0, sx1]) * a1
            h1 = int(src[r1, sx]) * a0 + int(src[r1, sx1]) * a1
            ref[dy, dx] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(out, ref)


def test_pyramid_chain_against_python(oracle, vsa):
    """ComputePyramid (src/ORBextractor.cc:1106-1131) a second time in Python over the resize primitive checked above: level
    sizes from the ORIGINAL size times the float inverse scale, round-half-even (:1111), each level resized from the level
    before it (:1119), every level inside a 19-pixel BORDER_REFLECT_101 frame taken from the level itself (:1121-1127)."""
    for w, h, nl, sf in ((320, 240, 5, 1.2), (203, 151, 4, 1.2), (256, 192, 3, 2.0), (300, 220, 4, 1.5)):
        img = vsa.synth_frame(40 + nl, w, h)
        ex = oracle.Extractor(oracle.make_params(200, sf, nl))
        ex.extract(img)
        inv = [np.float32(1)]
        scale = [np.float32(1)]
        for _ in range(1, nl):
            scale.append(np.float32(np.float64(scale[-1]) * sf))           # mvScaleFactor[i] = mvScaleFactor[i-1] * scaleFactor, the factor a double (:417)
            inv.append(np.float32(1) / scale[-1])
        prev = img
        for level in range(nl):
            lw, lh = int(np.rint(np.float32(w) * inv[level])), int(np.rint(np.float32(h) * inv[level]))
            assert ex.level_size(level) == (lw, lh)
            cur = img if level == 0 else oracle.resize_linear(prev, lw, lh)
            assert np.array_equal(ex.level_padded(level), np.pad(cur, 19, mode="reflect")), (w, h, level)
            prev = cur
        ex.close()


def test_fast_hand_cases(oracle):
    # constant image: no corner at any threshold
    assert len(oracle.fast9(np.full((40, 40), 90, np.uint8), 7)[0]) == 0
    # filled quadrant: 11 contiguous ring pixels are darker by 100 -> score 99, a corner for every T < 100
    img = np.zeros((31, 31), np.uint8); img[15:, 15:] = 100
    assert oracle.fast9_score(img, 15, 15, 20) == 99
    x, y, s = oracle.fast9(img, 20, nonmax=False)
    assert (15, 15) in set(zip(x.tolist(), y.tolist()))
    assert len(oracle.fast9(img, 100, nonmax=False)[0]) == 0
    # (16,16) scores 99 as well: equal neighbours suppress each other, so NMS leaves neither
    assert oracle.fast9_score(img, 16, 16, 20) == 99
    x, y, s = oracle.fast9(img, 20, nonmax=True)
    assert (15, 15) not in set(zip(x.tolist(), y.tolist()))
    # an isolated bright pixel: all 16 ring pixels darker by 255 -> one keypoint, score 254
    dot = np.zeros((21, 21), np.uint8); dot[10, 10] = 255
    x, y, s = oracle.fast9(dot, 20)
    assert (x.tolist(), y.tolist(), s.tolist()) == ([10], [10], [254])
    # a straight edge is not a corner (8 contiguous at most on the far side)
    edge = np.zeros((31, 31), np.uint8); edge[:, 15:] = 200
    assert len(oracle.fast9(edge, 20)[0]) == 0
    # score does not depend on the detection threshold; corner at T <=> score >= T
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (60, 60), dtype=np.uint8)
    x7, y7, s7 = oracle.fast9(img, 7, nonmax=False)
    x20, y20, s20 = oracle.fast9(img, 20, nonmax=False)
    assert len(x7) > len(x20) > 0
    pts7 = set(zip(x7.tolist(), y7.tolist())); pts20 = set(zip(x20.tolist(), y20.tolist()))
    assert pts20 <= pts7
    for xx, yy in list(pts7)[:200]:
        sc = oracle.fast9_score(img, xx, yy, 7)
        assert sc >= 7 and ((xx, yy) in pts20) == (sc >= 20)
        assert oracle.fast9_score(img, xx, yy, 0) == sc
    # detection range excludes the 3-pixel ring of the sub-image
    assert x7.min() >= 3 and y7.min() >= 3 and x7.max() <= 56 and y7.max() <= 56


def test_descriptor_distance(oracle, vsa):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 32, dtype=np.uint8)
    b = rng.integers(0, 256, 32, dtype=np.uint8)
    assert oracle.descriptor_distance(a, a) == 0
    assert oracle.descriptor_distance(a, ~a) == 256
    assert oracle.descriptor_distance(a, b) == oracle.descriptor_distance(b, a) == int(np.unpackbits(a ^ b).sum())
    assert vsa.descriptor_distance(a, b) == oracle.descriptor_distance(a, b)


def test_three_maxima_and_rotation_bins(oracle, vsa):
    for impl in (oracle, vsa):
        assert impl.compute_three_maxima([10] + [0] * 29) == (0, -1, -1)
        assert impl.compute_three_maxima([10, 1, 1] + [0] * 27) == (0, 1, 2)
        assert impl.compute_three_maxima([100, 9, 9] + [0] * 27) == (0, -1, -1)
        assert impl.compute_three_maxima([100, 10, 9] + [0] * 27) == (0, 1, -1)
        assert impl.compute_three_maxima([0] * 30) == (-1, -1, -1)
        assert impl.compute_three_maxima([1, 2, 3, 4] + [0] * 26) == (3, 2, 1)
    # factor = 1/HISTO_LENGTH: rot = 359 deg lands in bin round(359/30) = 12
    assert round(359 * (1.0 / 30)) == 12


def test_distribute_octree_properties(oracle):
    rng = np.random.default_rng(6)
    for it in range(30):
        W, H = int(rng.integers(100, 900)), int(rng.integers(100, 500))
        if round(W / H) < 1:
            continue
        n = int(rng.integers(1, 3000))
        pts = sorted({(int(rng.integers(3, H - 3)), int(rng.integers(3, W - 3))) for _ in range(n)})
        y = np.array([p[0] for p in pts], np.int16); x = np.array([p[1] for p in pts], np.int16)
        s = rng.integers(7, 255, len(pts)).astype(np.uint8)
        N = int(rng.integers(1, 300))
        sel = oracle.distribute_octree(x, y, s, 16, 16 + W, 16, 16 + H, N)
        assert len(set(sel.tolist())) == len(sel)
        assert len(sel) <= max(N + 3, 4 * round(W / H))
        if len(pts) <= N:
            assert len(sel) == len(pts)          # splitting stopped because every node holds one key
    assert len(oracle.distribute_octree(np.zeros(0, np.int16), np.zeros(0, np.int16), np.zeros(0, np.uint8), 16, 616, 16, 464, 100)) == 0


def python_distribute_octree(xs, ys, resp, min_x, max_x, min_y, max_y, N):
    """DistributeOctTree and DivideNode (src/ORBextractor.cc:480-762) written out a second time, statement by statement,
    over a Python list used as the std::list (push_front = insert(0)).  Where the reference orders equal-sized nodes by
    their heap address (:683, which no rerun reproduces) the later-created node counts as the larger, the convention the
    C oracle documents.  Returns the indices of the kept keypoints in result order."""
    import math
    f = np.float32

    class Node:
        __slots__ = ("ul", "ur", "bl", "br", "keys", "no_more", "seq")

        def __init__(self):
            self.keys = []; self.no_more = False; self.seq = -1

    counter = [0]

    def divide(nd):
        half_x = math.ceil(f(nd.ur[0] - nd.ul[0]) / 2)
        half_y = math.ceil(f(nd.br[1] - nd.ul[1]) / 2)
        n1, n2, n3, n4 = Node(), Node(), Node(), Node()
        n1.ul = nd.ul; n1.ur = (nd.ul[0] + half_x, nd.ul[1]); n1.bl = (nd.ul[0], nd.ul[1] + half_y); n1.br = (nd.ul[0] + half_x, nd.ul[1] + half_y)
        n2.ul = n1.ur; n2.ur = nd.ur; n2.bl = n1.br; n2.br = (nd.ur[0], nd.ul[1] + half_y)
        n3.ul = n1.bl; n3.ur = n1.br; n3.bl = nd.bl; n3.br = (n1.br[0], nd.bl[1])
        n4.ul = n3.ur; n4.ur = n2.br; n4.bl = n3.br; n4.br = nd.br
        for i in nd.keys:
            if xs[i] < n1.ur[0]:
                (n1 if ys[i] < n1.br[1] else n3).keys.append(i)
            elif ys[i] < n1.br[1]:
                n2.keys.append(i)
            else:
                n4.keys.append(i)
        for c in (n1, n2, n3, n4):
            if len(c.keys) == 1:
                c.no_more = True
        return n1, n2, n3, n4

    n_ini = int(np.round(f(max_x - min_x) / f(max_y - min_y)))        # round(): half away from zero; never a tie for these sizes
    h_x = f(max_x - min_x) / f(n_ini)
    nodes = []
    for i in range(n_ini):
        ni = Node()
        ni.ul = (int(h_x * f(i)), 0); ni.ur = (int(h_x * f(i + 1)), 0)
        ni.bl = (ni.ul[0], max_y - min_y); ni.br = (ni.ur[0], max_y - min_y)
        nodes.append(ni)
    ini = list(nodes)
    for i in range(len(xs)):
        ini[int(f(xs[i]) / h_x)].keys.append(i)
    kept = []
    for nd in nodes:
        if len(nd.keys) == 1:
            nd.no_more = True
        if nd.keys:
            kept.append(nd)
    nodes = kept
    finish = False
    while not finish:
        prev_size = len(nodes)
        to_expand = 0
        size_and_node = []
        pos = 0
        while pos < len(nodes):
            nd = nodes[pos]
            if nd.no_more:
                pos += 1
                continue
            for c in divide(nd):
                if c.keys:
                    nodes.insert(0, c); pos += 1
                    if len(c.keys) > 1:
                        to_expand += 1
                        c.seq = counter[0]; counter[0] += 1
                        size_and_node.append((len(c.keys), c))
            del nodes[pos]
        if len(nodes) >= N or len(nodes) == prev_size:
            finish = True
        elif len(nodes) + to_expand * 3 > N:
            while not finish:
                prev_size = len(nodes)
                prev = sorted(size_and_node, key=lambda t: (t[0], t[1].seq))
                size_and_node = []
                for size, nd in reversed(prev):
                    for c in divide(nd):
                        if c.keys:
                            nodes.insert(0, c)
                            if len(c.keys) > 1:
                                c.seq = counter[0]; counter[0] += 1
                                size_and_node.append((len(c.keys), c))
                    nodes.remove(nd)
                    if len(nodes) >= N:
                        break
                if len(nodes) >= N or len(nodes) == prev_size:
                    finish = True
    out = []
    for nd in nodes:
        best = nd.keys[0]
        for k in nd.keys[1:]:
            if resp[k] > resp[best]:
                best = k
        out.append(best)
    return out


def test_distribute_octree_against_python_list_version(oracle):
    rng = np.random.default_rng(60)
    checked = 0
    for it in range(40):
        W, H = int(rng.integers(90, 700)), int(rng.integers(90, 400))
        if it % 7 == 0:
            W, H = 608, 448
        if round(W / H) < 1:
            continue
        n = int(rng.integers(0, 1500 if it % 5 else 60))
        cx, cy = int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))
        pts = set()
        for _ in range(n):
            if it % 2 and rng.random() < 0.5:
                x, y = cx + int(rng.integers(-20, 21)), cy + int(rng.integers(-20, 21))
            else:
                x, y = int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))
            if 3 <= x <= W - 4 and 3 <= y <= H - 4:
                pts.add((y // 32, x // 31, y, x))
        pts = sorted(pts)
        x = np.array([p[3] for p in pts], np.int16); y = np.array([p[2] for p in pts], np.int16)
        s = rng.integers(7, 12 if it % 3 == 0 else 250, len(pts)).astype(np.uint8)
        N = int(rng.integers(1, 260))
        got = oracle.distribute_octree(x, y, s, 16, 16 + W, 16, 16 + H, N)
        want = python_distribute_octree(x.tolist(), y.tolist(), s.tolist(), 16, 16 + W, 16, 16 + H, N)
        assert got.tolist() == want, (it, W, H, len(pts), N)
        checked += 1
    assert checked > 30


def c_round(v):
    """C round(): half away from zero"""
    import math
    return int(math.floor(abs(float(v)) + 0.5)) * (1 if v >= 0 else -1)


class PythonFrameGrid:
    """Frame::AssignFeaturesToGrid, PosInGrid and GetFeaturesInArea (src/Frame.cc:230-245, :327-392) written out a second
    time in Python, float32 where the reference computes in float."""

    def __init__(self, k2, bounds):
        f = np.float32
        self.k2 = k2
        self.min_x, self.max_x, self.min_y, self.max_y = (f(v) for v in bounds)
        self.inv_w = f(64) / (self.max_x - self.min_x); self.inv_h = f(48) / (self.max_y - self.min_y)      # src/Frame.cc:101-102
        self.grid = [[[] for _ in range(48)] for _ in range(64)]
        for i in range(len(k2)):
            px = c_round((f(k2["x"][i]) - self.min_x) * self.inv_w); py = c_round((f(k2["y"][i]) - self.min_y) * self.inv_h)
            if 0 <= px < 64 and 0 <= py < 48:
                self.grid[px][py].append(i)

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        import math
        f = np.float32
        k2 = self.k2
        out = []
        x, y, r = f(x), f(y), f(r)
        c0 = max(0, int(math.floor((x - self.min_x - r) * self.inv_w)))
        if c0 >= 64:
            return out
        c1 = min(63, int(math.ceil((x - self.min_x + r) * self.inv_w)))
        if c1 < 0:
            return out
        r0 = max(0, int(math.floor((y - self.min_y - r) * self.inv_h)))
        if r0 >= 48:
            return out
        r1 = min(47, int(math.ceil((y - self.min_y + r) * self.inv_h)))
        if r1 < 0:
            return out
        check = min_level > 0 or max_level >= 0
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for j in self.grid[ix][iy]:
                    if check:
                        if k2["octave"][j] < min_level:
                            continue
                        if max_level >= 0 and k2["octave"][j] > max_level:
                            continue
                    if abs(f(k2["x"][j]) - x) < r and abs(f(k2["y"][j]) - y) < r:
                        out.append(j)
        return out


def python_hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def python_three_maxima(sizes):
    """ComputeThreeMaxima (src/ORBmatcher.cc:1604-1645)"""
    f = np.float32
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i, sz in enumerate(sizes):
        if sz > max1:
            max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i
        elif sz > max2:
            max3 = max2; max2 = sz; ind3 = ind2; ind2 = i
        elif sz > max3:
            max3 = sz; ind3 = i
    if f(max2) < f(0.1) * f(max1):
        ind2 = ind3 = -1
    elif f(max3) < f(0.1) * f(max1):
        ind3 = -1
    return ind1, ind2, ind3


def python_rot_bin(angle1, angle2):
    """src/ORBmatcher.cc:478-483"""
    f = np.float32
    rot = f(angle1) - f(angle2)
    if rot < 0.0:
        rot = rot + f(360.0)
    b = c_round(rot * (f(1.0) / f(30)))
    return 0 if b == 30 else b


def python_search_for_initialization(k1, d1, k2, d2, bounds, prev, window, nnratio, check_ori):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:408-523) written out a second time in Python over the grid,
    distance, histogram and three-maxima restatements above."""
    f = np.float32
    features_in_area = PythonFrameGrid(k2, bounds).features_in_area
    dist_of = python_hamming
    big = 2 ** 31 - 1
    nmatches = 0
    m12 = [-1] * len(k1); m21 = [-1] * len(k2); matched_dist = [big] * len(k2)
    hist = [[] for _ in range(30)]
    prev = np.array(prev, np.float32).copy()
    for i1 in range(len(k1)):
        if k1["octave"][i1] > 0:
            continue
        cand = features_in_area(prev[i1][0], prev[i1][1], window, 0, 0)
        if not cand:
            continue
        best, best2, best_idx = big, big, -1
        for i2 in cand:
            d = dist_of(d1[i1], d2[i2])
            if matched_dist[i2] <= d:
                continue
            if d < best:
                best2 = best; best = d; best_idx = i2
            elif d < best2:
                best2 = d
        if best <= 50 and f(best) < f(best2) * f(nnratio):
            if m21[best_idx] >= 0:
                m12[m21[best_idx]] = -1
                nmatches -= 1
            m12[i1] = best_idx; m21[best_idx] = i1; matched_dist[best_idx] = best
            nmatches += 1
            if check_ori:
                hist[python_rot_bin(k1["angle"][i1], k2["angle"][best_idx])].append(i1)
    if check_ori:
        ind1, ind2, ind3 = python_three_maxima([len(h) for h in hist])
        for i in range(30):
            if i in (ind1, ind2, ind3):
                continue
            for idx1 in hist[i]:
                if m12[idx1] >= 0:
                    m12[idx1] = -1
                    nmatches -= 1
    for i1 in range(len(k1)):
        if m12[i1] >= 0:
            prev[i1] = (k2["x"][m12[i1]], k2["y"][m12[i1]])
    return np.array(m12, np.int32), nmatches, prev


def test_search_for_initialization_against_python(oracle):
    from oracle_lib import KEYPOINT_DTYPE
    rng = np.random.default_rng(77)
    bounds = (0.0, 640.0, 0.0, 480.0)
    matched_total = 0
    for it in range(6):
        n1 = int(rng.integers(150, 500))
        k1 = np.zeros(n1, KEYPOINT_DTYPE)
        k1["x"] = rng.uniform(-5, 645, n1).astype(np.float32); k1["y"] = rng.uniform(-5, 485, n1).astype(np.float32)
        k1["octave"] = rng.integers(0, 3, n1) * (rng.random(n1) < 0.4); k1["angle"] = rng.uniform(0, 360, n1).astype(np.float32)
        d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
        if it % 2:
            d1[: n1 // 3] = d1[0]                                    # near-identical descriptors: ties and stolen matches
        # frame 2: moved and slightly altered copies, one common rotation, plus unrelated features
        take = rng.permutation(n1)[: int(n1 * 0.8)]
        extra = int(rng.integers(20, 120))
        n2 = len(take) + extra
        k2 = np.zeros(n2, KEYPOINT_DTYPE); d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        k2["x"][: len(take)] = k1["x"][take] + rng.uniform(-12, 12, len(take)).astype(np.float32)
        k2["y"][: len(take)] = k1["y"][take] + rng.uniform(-12, 12, len(take)).astype(np.float32)
        k2["octave"][: len(take)] = k1["octave"][take]
        rot = np.where(rng.random(len(take)) < 0.8, 40.0, rng.uniform(0, 360, len(take)))
        k2["angle"][: len(take)] = np.mod(k1["angle"][take] - rot + rng.uniform(-6, 6, len(take)), 360.0).astype(np.float32)
        flips = rng.random((len(take), 256)) < 0.06
        d2[: len(take)] = np.bitwise_xor(d1[take], np.packbits(flips, axis=1))
        k2["x"][len(take):] = rng.uniform(0, 640, extra).astype(np.float32); k2["y"][len(take):] = rng.uniform(0, 480, extra).astype(np.float32)
        k2["angle"][len(take):] = rng.uniform(0, 360, extra).astype(np.float32)
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        for window, ratio, check_ori in ((100, 0.9, True), (15, 0.9, True), (30, 0.7, False)):
            g12, gn, gprev = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev, window, ratio, check_ori)
            w12, wn, wprev = python_search_for_initialization(k1, d1, k2, d2, bounds, prev, window, ratio, check_ori)
            assert gn == wn and np.array_equal(g12, w12), (it, window, ratio, check_ori, gn, wn)
            assert np.array_equal(gprev.view(np.uint32), wprev.view(np.uint32))
            matched_total += gn
    assert matched_total > 500


def python_search_by_projection_mappoints(q, k2, d2, uright2, bounds, nnratio, claimed):
    """Candidate loop and acceptance of SearchByProjection(Frame&, vpMapPoints, th) (src/ORBmatcher.cc:68-125) a second time.
    q: per projected map point u, v, r (= r * scale[level], :69), ur (mTrackProjXR), level, desc, transient (the point
    has no observations, so its match does not block later ones: :87-89).  claimed[i] >= 0: feature i holds a map point
    with observations."""
    f = np.float32
    grid = PythonFrameGrid(k2, bounds)
    claimed = list(claimed)
    nmatches = 0
    for i in range(len(q["u"])):
        lvl = int(q["level"][i])
        cand = grid.features_in_area(q["u"][i], q["v"][i], q["r"][i], lvl - 1, lvl)
        if not cand:
            continue
        best, best_level, best2, best_level2, best_idx = 256, -1, 256, -1, -1
        for idx in cand:
            if claimed[idx] >= 0:
                continue
            if uright2 is not None and uright2[idx] > 0:
                if abs(f(q["ur"][i]) - f(uright2[idx])) > f(q["r"][i]):
                    continue
            d = python_hamming(q["desc"][i], d2[idx])
            if d < best:
                best2 = best; best = d; best_level2 = best_level; best_level = int(k2["octave"][idx]); best_idx = idx
            elif d < best2:
                best_level2 = int(k2["octave"][idx]); best2 = d
        if best <= 100:
            if best_level == best_level2 and f(best) > f(nnratio) * f(best2):
                continue
            claimed[best_idx] = -2 - i if q["transient"][i] else i
            nmatches += 1
    return np.array(claimed, np.int32), nmatches


def python_search_by_projection_lastframe(q, k2, d2, uright2, bounds, mode, th_dist, check_ori, claimed):
    """Candidate loop, acceptance and rotation histogram of SearchByProjection(CurrentFrame, LastFrame, th, bMono)
    (src/ORBmatcher.cc:1381-1470) a second time; mode 1 = forward (:1389), 2 = backward (:1391), 0 otherwise (:1393)."""
    f = np.float32
    grid = PythonFrameGrid(k2, bounds)
    claimed = list(claimed)
    nmatches = 0
    hist = [[] for _ in range(30)]
    for i in range(len(q["u"])):
        octv = int(q["level"][i]); radius = q["r"][i]
        if mode == 1:
            cand = grid.features_in_area(q["u"][i], q["v"][i], radius, octv)
        elif mode == 2:
            cand = grid.features_in_area(q["u"][i], q["v"][i], radius, 0, octv)
        else:
            cand = grid.features_in_area(q["u"][i], q["v"][i], radius, octv - 1, octv + 1)
        if not cand:
            continue
        best, best_idx = 256, -1
        for i2 in cand:
            if claimed[i2] >= 0:
                continue
            if uright2 is not None and uright2[i2] > 0:
                if abs(f(q["ur"][i]) - f(uright2[i2])) > f(radius):
                    continue
            d = python_hamming(q["desc"][i], d2[i2])
            if d < best:
                best = d; best_idx = i2
        if best <= th_dist:
            claimed[best_idx] = -2 - i if q["transient"][i] else i
            nmatches += 1
            if check_ori:
                hist[python_rot_bin(q["angle"][i], k2["angle"][best_idx])].append(best_idx)
    if check_ori:
        ind1, ind2, ind3 = python_three_maxima([len(h) for h in hist])
        for i in range(30):
            if i != ind1 and i != ind2 and i != ind3:
                for idx in hist[i]:
                    claimed[idx] = -1
                    nmatches -= 1
    return np.array(claimed, np.int32), nmatches


def python_search_by_projection_keyframe(q, k2, d2, bounds, claimed):
    """Candidate loop and acceptance of SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:362-401)
    a second time: KeyFrame::GetFeaturesInArea without levels (src/KeyFrame.cc:1311-1350), features that are matched
    already skipped, level band predicted-1 .. predicted, best distance within TH_LOW."""
    grid = PythonFrameGrid(k2, bounds)
    claimed = list(claimed)
    nmatches = 0
    for i in range(len(q["u"])):
        lvl = int(q["level"][i])
        best, best_idx = 256, -1
        for idx in grid.features_in_area(q["u"][i], q["v"][i], q["r"][i]):
            if claimed[idx] >= 0:
                continue
            if k2["octave"][idx] < lvl - 1 or k2["octave"][idx] > lvl:
                continue
            d = python_hamming(q["desc"][i], d2[idx])
            if d < best:
                best, best_idx = d, idx
        if best <= 50:
            claimed[best_idx] = i
            nmatches += 1
    return np.array(claimed, np.int32), nmatches


def test_search_by_projection_against_python(oracle):
    from oracle_lib import KEYPOINT_DTYPE
    rng = np.random.default_rng(91)
    bounds = (0.0, 752.0, 0.0, 480.0)
    scale = 1.2 ** np.arange(8)
    total = 0
    for it in range(6):
        n2 = int(rng.integers(300, 700))
        k2 = np.zeros(n2, KEYPOINT_DTYPE)
        k2["x"] = rng.uniform(-4, 756, n2).astype(np.float32); k2["y"] = rng.uniform(-4, 484, n2).astype(np.float32)
        k2["octave"] = rng.integers(0, 8, n2) * (rng.random(n2) < 0.6); k2["angle"] = rng.uniform(0, 360, n2).astype(np.float32)
        d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        uright2 = np.where(rng.random(n2) < 0.5, k2["x"] - rng.uniform(1, 40, n2), -1).astype(np.float32) if it % 2 else None
        nq = int(rng.integers(200, 500))
        src = rng.integers(0, n2, nq)                                   # each query comes from a feature of the frame ...
        src[: nq // 6] = src[0]                                        # ... several from the same one (overwrites, contested features)
        q = {"u": (k2["x"][src] + rng.uniform(-9, 9, nq)).astype(np.float32), "v": (k2["y"][src] + rng.uniform(-9, 9, nq)).astype(np.float32),
             "level": np.clip(k2["octave"][src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32),
             "angle": np.mod(k2["angle"][src] + np.where(rng.random(nq) < 0.8, 25.0, rng.uniform(0, 360, nq)) + rng.uniform(-5, 5, nq), 360).astype(np.float32),
             "transient": (rng.random(nq) < 0.3).astype(np.uint8)}
        q["r"] = (np.where(rng.random(nq) < 0.5, 2.5, 4.0) * rng.choice([1.0, 3.0, 7.0]) * scale[q["level"]]).astype(np.float32)
        q["ur"] = (q["u"] - rng.uniform(1, 40, nq)).astype(np.float32)
        flips = rng.random((nq, 256)) < rng.choice([0.05, 0.2])
        q["desc"] = np.bitwise_xor(d2[src], np.packbits(flips, axis=1))
        q["u"][:5] = rng.uniform(-300, -100, 5)                         # projections far outside the image
        claimed0 = np.where(rng.random(n2) < 0.15, 2 ** 31 - 1, -1).astype(np.int32)
        g, gn = oracle.search_by_projection_mappoints(q["u"], q["v"], q["r"], q["ur"], q["level"], q["desc"], k2, d2, uright2, bounds,
                                                      0.8, claimed0, transient=q["transient"])
        w, wn = python_search_by_projection_mappoints(q, k2, d2, uright2, bounds, 0.8, claimed0)
        assert gn == wn and np.array_equal(g, w), ("map points", it, gn, wn)
        total += gn
        for mode, th_dist, check_ori in ((0, 100, True), (1, 100, True), (2, 50, False)):
            g, gn = oracle.search_by_projection_lastframe(q["u"], q["v"], q["r"], q["ur"], q["level"], q["angle"], q["desc"], k2, d2, uright2,
                                                          bounds, mode, th_dist, check_ori, claimed0, transient=q["transient"])
            w, wn = python_search_by_projection_lastframe(q, k2, d2, uright2, bounds, mode, th_dist, check_ori, claimed0)
            assert gn == wn and np.array_equal(g, w), ("last frame", it, mode, gn, wn)
            total += gn
        g, gn = oracle.search_by_projection_keyframe(q["u"], q["v"], q["r"], q["level"], q["desc"], k2, d2, bounds, claimed0)
        w, wn = python_search_by_projection_keyframe(q, k2, d2, bounds, claimed0)
        assert gn == wn and np.array_equal(g, w), ("key frame", it, gn, wn)
        total += gn
    assert total > 1500


def test_stereo_row_search_against_python(oracle):
    """The Hamming part of Frame::ComputeStereoMatches (src/Frame.cc:476-549) a second time in Python: the row table of the
    right keypoints (every row within 2 * scale of the keypoint), per left keypoint the candidates of its row
    (`vRowIndices[vL]`, the float index truncated), the octave band, the disparity range, best distance below TH_HIGH."""
    import math
    from oracle_lib import KEYPOINT_DTYPE
    f = np.float32
    rng = np.random.default_rng(13)
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    n_rows = 480
    found = 0
    for it in range(4):
        nr = int(rng.integers(300, 700)); nl = int(rng.integers(300, 700))
        kr = np.zeros(nr, KEYPOINT_DTYPE)
        kr["x"] = rng.uniform(19, 733, nr).astype(np.float32); kr["y"] = rng.uniform(19, 461, nr).astype(np.float32)
        kr["octave"] = rng.integers(0, 8, nr)
        dr = rng.integers(0, 256, (nr, 32), dtype=np.uint8)
        src = rng.integers(0, nr, nl)
        kl = np.zeros(nl, KEYPOINT_DTYPE)
        kl["x"] = (kr["x"][src] + rng.uniform(-5, 70, nl)).astype(np.float32)
        kl["y"] = (kr["y"][src] + rng.uniform(-3, 3, nl)).astype(np.float32)
        kl["octave"] = np.clip(kr["octave"][src] + rng.integers(-2, 3, nl), 0, 7)
        dl = np.bitwise_xor(dr[src], np.packbits(rng.random((nl, 256)) < 0.1, axis=1))
        min_d, max_d = f(0), f(47.906 / 0.11)                                          # :496-498 with EuRoC's bf and b
        rows = [[] for _ in range(n_rows)]
        for i_r in range(nr):
            r = f(2.0) * scale[kr["octave"][i_r]]
            for yi in range(int(math.floor(f(kr["y"][i_r]) - r)), int(math.ceil(f(kr["y"][i_r]) + r)) + 1):
                rows[yi].append(i_r)
        want_d = np.full(nl, -1, np.int32); want_i = np.full(nl, -1, np.int32)
        for i_l in range(nl):
            cand = rows[int(kl["y"][i_l])]
            if not cand:
                continue
            min_u, max_u = f(kl["x"][i_l]) - max_d, f(kl["x"][i_l]) - min_d
            if max_u < 0:
                continue
            best, best_r = 100, 0
            for i_r in cand:
                if kr["octave"][i_r] < kl["octave"][i_l] - 1 or kr["octave"][i_r] > kl["octave"][i_l] + 1:
                    continue
                if min_u <= f(kr["x"][i_r]) <= max_u:
                    d = python_hamming(dl[i_l], dr[i_r])
                    if d < best:
                        best, best_r = d, i_r
            want_d[i_l], want_i[i_l] = best, best_r
        got_d, got_i = oracle.stereo_match(kl, dl, kr, dr, scale, n_rows, float(min_d), float(max_d))
        assert np.array_equal(got_d, want_d) and np.array_equal(got_i, want_i), it
        found += int(np.count_nonzero((got_d >= 0) & (got_d < 75)))
    assert found > 400


def test_extract_is_deterministic_and_bounded(oracle, vsa):
    img = vsa.synth_frame(0, 640, 480)
    ex = oracle.Extractor(oracle.make_params())
    k1, d1 = ex.extract(img)
    k2, d2 = ex.extract(img)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert 900 <= len(k1) <= 1000 + 3 * 8
    assert (k1["octave"][:-1] <= k1["octave"][1:]).all()
    assert k1["class_id"].tolist() == [-1] * len(k1)
    lvl0 = k1[k1["octave"] == 0]
    assert lvl0["x"].min() >= 19 and lvl0["x"].max() <= 640 - 20 and lvl0["y"].min() >= 19 and lvl0["y"].max() <= 480 - 20
    assert (k1["angle"] >= 0).all() and (k1["angle"] <= 360).all()
    kc, dc = ex.extract(np.full((480, 640), 128, np.uint8))
    assert len(kc) == 0


def test_levels_too_small_for_a_cell_yield_nothing(oracle, vsa):
    """src/ORBextractor.cc:783-786 with nCols or nRows == 0: `width/nCols` is a float division (inf, no trap), the cell loops
    run zero times and DistributeOctTree gets an empty list -- the level has no keypoints, the others are unaffected.
    160x120 with 8 levels: levels 4..7 are 77x58 and smaller (detection area under 30 rows).  Only a level smaller than
    its own 16-pixel borders is rejected."""
    img = vsa.synth_frame(6, 160, 120)
    k8, d8 = oracle.Extractor(oracle.make_params(500, 1.2, 8)).extract(img)
    per_level = np.bincount(k8["octave"], minlength=8)
    assert np.all(per_level[:4] > 0) and np.all(per_level[4:] == 0)
    # the four populated levels behave exactly as in a 4-level pyramid given the same per-level quotas
    ex = oracle.Extractor(oracle.make_params(500, 1.2, 8))
    ex.extract(img)
    for l in range(4, 8):
        assert len(ex.level_candidates(l)[0]) == 0 and len(ex.level_keypoints(l)) == 0
        assert ex.level_padded(l).shape[0] == vsa.level_size(vsa.make_params(500, 1.2, 8), 160, 120, l)[1] + 38
    with pytest.raises(Exception):
        oracle.Extractor(oracle.make_params(500, 1.2, 8)).extract(vsa.synth_frame(6, 100, 80))


def test_glibc_trig_restatement_equals_this_hosts_libm(oracle):
    """src/ORBextractor.cc:112 calls cos/sin on a float under `using namespace std`: glibc's cosf / sinf.  trig_mode 3 of
    the oracle calls this host's libm; trig_modes 1 and 2 restate its algorithm (with and without the fused multiply-adds
    of the x86-64 FMA build).  Walk EVERY float angle in [0, 360): libm equals both restatements bit for bit (the two
    differ in double precision for about 1 % of the angles, never after rounding to float), so the device, which follows
    the restatement, is pinned to the real callee.  The default trig_mode 0 (correctly rounded) differs from libm for
    about 0.13 % of the angles -- counted here so the figure in the documentation stays honest."""
    assert oracle.ANGLE_BITS_END == 0x43B40000
    assert oracle.trig_count_differences(3, 1) == 0
    assert oracle.trig_count_differences(3, 2) == 0
    n0 = oracle.trig_count_differences(3, 0)
    assert 1_000_000 < n0 < 2_000_000, n0
    # spot values: quadrant boundaries and the small-angle branches
    for deg, c, s in ((0.0, 1.0, 0.0), (90.0, -4.371139e-08, 1.0), (180.0, -1.0, -8.742278e-08)):
        cs, sn = oracle.trig_range(3, int(np.float32(deg).view(np.uint32)), 1)
        assert cs[0] == np.float32(c) and sn[0] == np.float32(s), (deg, cs[0], sn[0])


def test_distinctive_descriptor_known_answers(oracle):
    """src/MapPoint.cc:513-547 by hand: descriptors with 0, 1, 2, 3 and 40 set bits.  Rows of the distance matrix sorted:
    [0,1,2,3,40] [0,1,1,2,39] [0,1,1,2,38] [0,1,2,3,37] [0,37,38,39,40]; medians (element 2): 2, 1, 1, 2, 38 -> index 1
    (the first of the two rows with median 1)."""
    d = np.zeros((5, 32), np.uint8)
    d[1, 0] = 0x01; d[2, 0] = 0x03; d[3, 0] = 0x07; d[4, :5] = 0xFF
    assert oracle.distinctive_descriptor(d) == 1
    assert oracle.distinctive_descriptor(d[:1]) == 0
    assert oracle.distinctive_descriptor(d[:2]) == 0          # n = 2: the median is the zero on the diagonal for both rows
    assert oracle.distinctive_descriptor(d[:0]) == -1
    # n = 4: element (int)(0.5*3) = 1 of each sorted row = the distance to the nearest other descriptor
    assert oracle.distinctive_descriptor(d[[4, 0, 3, 1]]) == 1   # rows: [0,37,39,40] [0,1,3,40] [0,2,3,37] [0,1,2,39] -> medians 37,1,2,1


def test_vocabulary_transform_known_answers(oracle):
    """A two-level tree written by hand (TemplatedVocabulary.h:1230-1272): root -> A (all zero bits), B (all one bits);
    A -> words 0 (zero), 1 (low byte set); B is itself word 2.  L = 2."""
    parent = np.array([-1, 0, 0, 1, 1], np.int32)
    leaf = np.array([0, 0, 1, 1, 1], np.uint8)
    desc = np.zeros((5, 32), np.uint8); desc[2] = 0xFF; desc[4, 0] = 0xFF
    weight = np.array([0, 0, 2.0, 1.0, 0.5])
    feats = np.zeros((5, 32), np.uint8)
    feats[1] = 0xFF                       # -> B (word 2)
    feats[2, 0] = 0xFF                    # -> A -> word 1
    feats[3, 0] = 0x0F                    # 4 bits from either child of A: the first (word 0) wins the tie
    feats[4, :16] = 0xFF                  # 128 bits from A and from B: A wins the tie, then node 4 (120 vs 128)
    word, node, ids, vals, fn, fp, ff = oracle.vocabulary_transform(2, 2, parent, leaf, desc, weight, feats, levelsup=1, scoring=0, weighting=0)
    # word ids number the leaves in node order: node 2 (B) = word 0, node 3 = word 1, node 4 = word 2
    assert list(word) == [1, 0, 2, 1, 2]
    assert list(node) == [1, 2, 1, 1, 1]  # level L - levelsup = 1: A (1) or B (2, also the leaf reached)
    assert list(ids) == [0, 1, 2]
    # TF_IDF sums then L1: word 0 -> 2.0, word 1 -> 1+1, word 2 -> 0.5+0.5; norm 5
    assert np.array_equal(vals, np.array([2.0, 2.0, 1.0]) / 5.0)
    assert list(fn) == [1, 2] and list(fp) == [0, 4, 5] and list(ff) == [0, 2, 3, 4, 1]
    # BINARY weighting (3): first weight only; DOT_PRODUCT scoring (5) with TF (1): divided by the number of words
    vals_b = oracle.vocabulary_transform(2, 2, parent, leaf, desc, weight, feats, 1, 0, 3)[3]
    assert np.array_equal(vals_b, np.array([2.0, 1.0, 0.5]) / 3.5)
    vals_d = oracle.vocabulary_transform(2, 2, parent, leaf, desc, weight, feats, 1, 5, 1)[3]
    assert np.array_equal(vals_d, np.array([2.0, 2.0, 1.0]) / 3.0)
    # levelsup >= L: every feature files under the root
    assert list(oracle.vocabulary_transform(2, 2, parent, leaf, desc, weight, feats, 2)[1]) == [0] * 5
    # a stopped word (weight 0) leaves both vectors
    w0 = weight.copy(); w0[4] = 0.0
    r = oracle.vocabulary_transform(2, 2, parent, leaf, desc, w0, feats, 1)
    assert list(r[2]) == [0, 1] and list(r[6]) == [0, 3, 1]


def test_cvt_color_known_answers(oracle):
    """The 8-bit grey values cv::cvtColor is known for: pure red 76, green 150, blue 29, white 255, black 0."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 77]]], np.uint8)
    assert list(oracle.cvt_color_to_gray(px, blue_first=False)[0]) == [76, 150, 29, 255, 0, (10 * 4899 + 200 * 9617 + 77 * 1868 + 8192) >> 14]
    assert list(oracle.cvt_color_to_gray(px, blue_first=True)[0][:3]) == [29, 150, 76]
    rgba = np.concatenate([px, np.full((1, 6, 1), 99, np.uint8)], axis=2)
    assert np.array_equal(oracle.cvt_color_to_gray(rgba, False), oracle.cvt_color_to_gray(px, False))


def test_undistort_points_properties(oracle):
    """No reference fixture exists for cv::undistortPoints here (parity unpinned); what can be checked by reasoning:
    zero coefficients give the identity up to float rounding, the principal point is a fixed point, and distorting the
    result with the forward model returns the input (the five iterations converge for the TUM1 camera)."""
    fx, fy, cx, cy = 517.306408, 516.469215, 318.643040, 255.313989                     # Examples/Monocular/TUM1.yaml
    k1, k2, p1, p2, k3 = 0.262383, -0.953104, -0.005358, 0.002628, 1.163314
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    ident = oracle.undistort_points(xy, fx, fy, cx, cy, 0, 0, 0, 0, 0)
    assert np.abs(ident - xy).max() < 1e-4
    centre = oracle.undistort_points(np.array([[cx, cy]], np.float32), fx, fy, cx, cy, k1, k2, p1, p2, k3)
    assert np.allclose(centre, [[np.float32(cx), np.float32(cy)]], atol=1e-5)
    un = oracle.undistort_points(xy, fx, fy, cx, cy, k1, k2, p1, p2, k3).astype(np.float64)
    x = (un[:, 0] - np.float32(cx)) / np.float32(fx); y = (un[:, 1] - np.float32(cy)) / np.float32(fy)
    r2 = x * x + y * y
    cd = 1 + ((np.float32(k3) * r2 + np.float32(k2)) * r2 + np.float32(k1)) * r2
    xd = x * cd + 2 * np.float32(p1) * x * y + np.float32(p2) * (r2 + 2 * x * x)
    yd = y * cd + np.float32(p1) * (r2 + 2 * y * y) + 2 * np.float32(p2) * x * y
    back = np.stack([xd * np.float32(fx) + np.float32(cx), yd * np.float32(fy) + np.float32(cy)], 1)
    assert np.abs(back - xy).max() < 0.05


def test_remap_known_answers(oracle):
    """cv::remap INTER_LINEAR by hand: integer positions copy the source (weights 32767, 0, 0, 1), a half-pixel shift
    averages with rounding up, quarter positions blend 3:1, positions outside give the constant border 0 and the last
    row / column blend with it."""
    src = np.array([[10, 20, 40], [50, 90, 200], [0, 255, 255]], np.uint8)
    yy, xx = np.mgrid[0:3, 0:3].astype(np.float32)
    assert np.array_equal(oracle.remap_linear_u8(src, xx, yy), src)
    half = oracle.remap_linear_u8(src, xx + 0.5, yy)
    assert list(half[0]) == [15, 30, 20] and list(half[1]) == [70, 145, 100]            # last column: half of the pixel, half border
    quarter = oracle.remap_linear_u8(src, xx[:, :2] + 0.25, yy[:, :2] + 0.25)
    # (0,0): 10*.5625 + 20*.1875 + 50*.1875 + 90*.0625 = 24.375 -> 24
    assert quarter[0, 0] == 24
    out = oracle.remap_linear_u8(src, xx - 5, yy)
    assert not out.any()
    edge = oracle.remap_linear_u8(src, xx - 0.5, yy)
    assert list(edge[0]) == [5, 15, 30]                                                 # left neighbour outside: border 0
    # rounding of the map to 1/32 pixel: 0.49 * 32 = 15.68 -> 16/32 = the half-pixel result
    assert np.array_equal(oracle.remap_linear_u8(src, xx + np.float32(0.49), yy), half)
