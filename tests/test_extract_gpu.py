"""GPU parity: the HIP extractor (through the C ABI) against the CPU oracle, stage by stage.

Bar: bit-exact pyramid pixels, candidates, selected keypoints, coordinates, octaves, responses and
descriptors; angles bit-exact as well (the task's tolerance is 1e-4 rad, asserted separately)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [
    # (w, h, nfeatures, nlevels, scale, ini, min, frame_index)
    (640, 480, 1000, 8, 1.2, 20, 7, 0),      # TUM1.yaml
    (752, 480, 1000, 8, 1.2, 20, 7, 1),      # EuRoC shape
    (752, 480, 1200, 8, 1.2, 20, 7, 2),      # EuRoC.yaml nFeatures
    (1241, 376, 2000, 8, 1.2, 20, 7, 3),     # KITTI00-02.yaml
    (752, 480, 1000, 12, 1.2, 16, 5, 4),     # ORB_mv_25002112.yaml
    (640, 480, 2000, 12, 1.2, 14, 4, 5),     # TUM1realsense.yaml
    (160, 120, 300, 2, 1.2, 20, 7, 6),       # tiny
    (321, 243, 500, 5, 1.3, 25, 9, 7),       # odd sizes, other scale
    (640, 480, 1000, 3, 2.0, 20, 7, 8),      # exact 2x decimation (INTER_AREA reroute)
    (3840, 2160, 8000, 12, 1.2, 20, 7, 9),   # 4K stress shape: quadtree keys in device memory, 1500-keypoint quotas
    (480, 640, 1000, 8, 1.2, 20, 7, 10),     # portrait
    (1600, 1200, 3000, 8, 1.2, 20, 7, 11),   # between 1 and 2 Mpixel: 512-thread quadtree workgroups
    (1241, 376, 2000, 8, 1.2, 12, 12, 12),   # KITTI04-12-like thresholds (ini == min), cells above 38 pixels in one level
    (160, 120, 500, 8, 1.2, 20, 7, 13),      # levels 4..7 are under 62 px high: no FAST cell fits, they yield no keypoints
    (320, 180, 500, 8, 1.2, 20, 7, 14),      # same for levels 6 and 7
]


def run_both(vsa, oracle, cfg, blur_rounding=0):
    w, h, nf, nl, sf, ini, mn, idx = cfg
    img = vsa.synth_frame(idx, w, h)
    ex = vsa.Extractor(vsa.make_params(nf, sf, nl, ini, mn, blur_rounding), w, h, max_batch=1, device=0)
    oex = oracle.Extractor(oracle.make_params(nf, sf, nl, ini, mn, blur_rounding))
    kps, desc = ex.extract(img)
    okps, odesc = oex.extract(img, cap=nf + 64 * nl)
    return img, ex, oex, kps, desc, okps, odesc


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"{c[0]}x{c[1]}_n{c[2]}_l{c[3]}_s{c[4]}")
def test_stage_parity(vsa, oracle, cfg):
    img, ex, oex, kps, desc, okps, odesc = run_both(vsa, oracle, cfg)
    nl = cfg[3]
    for l in range(nl):
        got = ex.pyramid_level(0, l)
        want = oex.level_padded(l)
        assert got.shape == want.shape, l
        assert np.array_equal(got, want), f"pyramid level {l}: {np.count_nonzero(got != want)} pixels differ"
    for l, got in enumerate(ex.pyramid_levels(0)):                       # the one-trip form of the same download
        assert np.array_equal(got, oex.level_padded(l)), f"pyramid level {l} (batched download)"
    for l in range(nl):
        got, want = ex.debug_blurred_level(0, l), oex.level_blurred(l)
        assert want is None or np.array_equal(got, want), f"blurred level {l}: {np.count_nonzero(got != want)} pixels differ"
    for l in range(nl):
        gx, gy, gs = vsa.unpack_keys(ex.debug_candidates(0, l))
        ox, oy, os_ = oex.level_candidates(l)
        assert len(gx) == len(ox), f"level {l}: {len(gx)} candidates vs {len(ox)}"
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gs, os_), f"candidates level {l}"
    for l in range(nl):
        gx, gy, gs = vsa.unpack_keys(ex.debug_level_keypoints(0, l))
        ok = oex.level_keypoints(l)
        assert len(gx) == len(ok), f"level {l}: {len(gx)} selected vs {len(ok)}"
        assert np.array_equal(gx + 16, ok["x"].astype(np.int32)) and np.array_equal(gy + 16, ok["y"].astype(np.int32)), f"selection level {l}"
        assert np.array_equal(gs, ok["response"].astype(np.int32))
    assert len(kps) == len(okps) and len(kps) > 0
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    # tolerance stated by the task: 1e-4 rad; the implementation is expected to be bit-exact
    dang = np.abs(kps["angle"] - okps["angle"])
    dang = np.minimum(dang, 360.0 - dang)
    assert np.all(np.deg2rad(dang) <= 1e-4)
    assert np.array_equal(kps["angle"], okps["angle"])
    assert np.array_equal(desc, odesc), f"{np.count_nonzero(np.any(desc != odesc, axis=1))} descriptors differ"
    ex.close()


def test_blur_rounding_simd_mode(vsa, oracle):
    cfg = (640, 480, 1000, 8, 1.2, 20, 7, 11)
    _, ex, oex, kps, desc, okps, odesc = run_both(vsa, oracle, cfg, blur_rounding=1)
    for l in range(8):
        got, want = ex.debug_blurred_level(0, l), oex.level_blurred(l)
        assert want is not None and np.array_equal(got, want), f"blurred level {l}"
    assert np.array_equal(kps["x"], okps["x"]) and np.array_equal(desc, odesc)
    ex.close()


@pytest.mark.parametrize("own_blur", ["0", "1"])
@pytest.mark.parametrize("cfg,blur_rounding", [((640, 480, 1000, 8, 1.2, 20, 7, 21), 0), ((752, 480, 1200, 8, 1.2, 20, 7, 22), 1),
                                               ((1920, 1080, 2000, 8, 1.2, 20, 7, 23), 0), ((321, 243, 500, 5, 1.3, 25, 9, 24), 1)],
                         ids=["vga", "euroc_simd_rounding", "fullhd", "odd_simd_rounding"])
def test_descriptor_window_blur_forms(vsa, oracle, monkeypatch, cfg, blur_rounding, own_blur):
    """The two forms of the blur behind the descriptors -- every level blurred by its own kernel, or each keypoint's window
    blurred inside the describing kernel (the library picks by pixels per keypoint; ORBX_OWN_BLUR forces one) -- against the
    oracle, in both rounding modes; the blurred levels stay available to the debug download in either form."""
    monkeypatch.setenv("ORBX_OWN_BLUR", own_blur)
    _, ex, oex, kps, desc, okps, odesc = run_both(vsa, oracle, cfg, blur_rounding=blur_rounding)
    assert len(kps) == len(okps) and len(kps) > 100
    assert np.array_equal(kps["x"], okps["x"]) and np.array_equal(kps["y"], okps["y"]) and np.array_equal(kps["angle"], okps["angle"])
    assert np.array_equal(desc, odesc), f"{np.count_nonzero(np.any(desc != odesc, axis=1))} descriptors differ"
    got, want = ex.debug_blurred_level(0, 1), oex.level_blurred(1)
    assert want is None or np.array_equal(got, want)
    ex.close()


def test_trig_modes_exhaustive(vsa, oracle):
    """cos / sin of the keypoint angle (src/ORBextractor.cc:111-112) for EVERY float angle in [0, 360): the device under
    trig_mode 1 and 2 against this host's libm cosf / sinf (oracle trig_mode 3, the function the reference calls), and
    under trig_mode 0 against (float)cos((double)theta).  Bit for bit, no tolerance."""
    import torch
    chunk = 1 << 26
    d_cos = torch.empty(chunk, dtype=torch.float32, device="cuda"); d_sin = torch.empty_like(d_cos)
    st = torch.cuda.current_stream().cuda_stream
    bad = {0: 0, 1: 0, 2: 0}
    for first in range(0, oracle.ANGLE_BITS_END, chunk):
        n = min(chunk, oracle.ANGLE_BITS_END - first)
        want = {0: oracle.trig_range(0, first, n), 3: oracle.trig_range(3, first, n)}
        for mode in (0, 1, 2):
            vsa._check(vsa.lib.orbx_debug_trig_range(mode, first, n, d_cos.data_ptr(), d_sin.data_ptr(), st))
            torch.cuda.synchronize()
            wc, ws = want[0 if mode == 0 else 3]
            gc, gs = d_cos[:n].cpu().numpy(), d_sin[:n].cpu().numpy()
            bad[mode] += int(np.count_nonzero((gc.view(np.uint32) != wc.view(np.uint32)) | (gs.view(np.uint32) != ws.view(np.uint32))))
    assert bad == {0: 0, 1: 0, 2: 0}, bad


def test_trig_mode_glibc_end_to_end(vsa, oracle):
    """Descriptors with cos / sin taken from libm itself (oracle trig_mode 3) equal the device's under trig_mode 1 and 2;
    and they do differ from the default mode's somewhere in a large enough sample (the modes are not aliases)."""
    w, h = 752, 480
    ndiff = 0
    for idx in range(3):
        img = vsa.synth_frame(900 + idx, w, h)
        okps, odesc = oracle.Extractor(oracle.make_params(2000, 1.2, 8, 12, 5, 0, 3)).extract(img, cap=2600)
        for mode in (1, 2):
            ex = vsa.Extractor(vsa.make_params(2000, 1.2, 8, 12, 5, 0, mode), w, h, max_batch=1, device=0)
            kps, desc = ex.extract(img)
            ex.close()
            assert np.array_equal(kps, okps.astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, odesc), (idx, mode)
        o0k, o0d = oracle.Extractor(oracle.make_params(2000, 1.2, 8, 12, 5, 0, 0)).extract(img, cap=2600)
        ndiff += int(np.count_nonzero(np.any(o0d != odesc, axis=1)))
    print("descriptors that differ between trig_mode 0 and libm:", ndiff)
    with pytest.raises(vsa.OrbxError):
        vsa.Extractor(vsa.make_params(trig_mode=3), w, h, device=0)


def test_batch_matches_single(vsa, oracle):
    w, h = 640, 480
    p = vsa.make_params()
    B = 5
    imgs = np.stack([vsa.synth_frame(100 + i, w, h) for i in range(B)])
    ex = vsa.Extractor(p, w, h, max_batch=B, device=0)
    kps, desc, counts = ex.extract_batch(imgs)
    oex = oracle.Extractor(oracle.make_params())
    for i in range(B):
        okps, odesc = oex.extract(imgs[i])
        n = counts[i]
        assert n == len(okps)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(kps[i, :n][f], okps[f]), (i, f)
        assert np.array_equal(desc[i, :n], odesc)
    # a second call on the same context, fewer frames (stale state must not leak)
    kps2, desc2, counts2 = ex.extract_batch(imgs[3:])
    assert np.array_equal(counts2, counts[3:])
    assert np.array_equal(desc2[0, :counts2[0]], desc[3, :counts[3]])
    ex.close()


def test_extractors_of_different_sizes_coexist(vsa):
    """A large context keeps working after a small one was created (per-kernel limits such as the dynamic LDS size are
    shared by all contexts of a process: creating the small one must not lower them)."""
    big = vsa.Extractor(vsa.make_params(8000, 1.2, 12), 3840, 2160, device=0)     # its quadtree wants more LDS than the default limit
    img = vsa.synth_frame(31, 3840, 2160)
    kps, desc = big.extract(img)
    small = vsa.Extractor(vsa.make_params(200, 1.2, 3), 160, 120, device=0)
    small.extract(vsa.synth_frame(32, 160, 120))
    kps2, desc2 = big.extract(img)
    assert len(kps) > 6000 and np.array_equal(kps, kps2) and np.array_equal(desc, desc2)
    big.close(); small.close()


def test_noise_images(vsa, oracle):
    """Images unlike the benchmark's: white noise and a noisy scene, where a tenth of the pixels are corners (tens of
    thousands of candidates in a 640x480 level: dense FAST cells, long quadtree key lists, every node split many times)."""
    rng = np.random.default_rng(5)
    for w, h, nf, nl, ini, mn, kind in [(640, 480, 1000, 8, 7, 3, "noise"), (640, 480, 3000, 8, 20, 7, "noise"),
                                        (1241, 376, 2000, 8, 12, 5, "scene"), (800, 600, 5000, 6, 5, 2, "noise")]:
        if kind == "noise":
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img = (vsa.synth_frame(3, w, h).astype(np.int32) + rng.integers(-25, 26, (h, w))).clip(0, 255).astype(np.uint8)
        ex = vsa.Extractor(vsa.make_params(nf, 1.2, nl, ini, mn), w, h, max_batch=1, device=0)
        oex = oracle.Extractor(oracle.make_params(nf, 1.2, nl, ini, mn))
        kps, desc = ex.extract(img)
        okps, odesc = oex.extract(img, cap=nf + 64 * nl)
        assert len(oex.level_candidates(0)[0]) > 20000
        assert len(kps) == len(okps) >= nf, (w, h, kind)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(kps[f], okps[f]), (w, h, kind, f)
        assert np.array_equal(desc, odesc), (w, h, kind)
        ex.close(); oex.close()


def test_clustered_corners_leave_the_quadtree_tables(vsa, oracle):
    """Levels whose candidates do not fit the quadtree's LDS key arrays are selected from without a dense key copy, on
    the count tables alone, as long as the split steps stay above the tables' depth.  A frame whose corners sit in one
    small patch sends the steps far below it: the kernel must notice and take the dense form, with the same result.
    Shapes whose level 0 has thousands of candidates in a patch (noise) on a flat or on an ordinary background."""
    rng = np.random.default_rng(123)
    for w, h, nf, patch, background in [(1241, 376, 2000, (60, 200, 300, 560), "flat"), (1241, 376, 2000, (100, 260, 700, 980), "scene"),
                                        (1920, 1080, 4000, (400, 640, 800, 1200), "flat"), (1920, 1080, 4000, (0, 1080, 0, 1920), "scene")]:
        img = np.full((h, w), 117, np.uint8) if background == "flat" else vsa.synth_frame(31, w, h).copy()
        y0, y1, x0, x1 = patch
        if background == "flat" or (y1 - y0) < h:
            img[y0:y1, x0:x1] = rng.integers(0, 256, (y1 - y0, x1 - x0), dtype=np.uint8)
        ex = vsa.Extractor(vsa.make_params(nf, 1.2, 8, 20, 7), w, h, max_batch=1, device=0)
        oex = oracle.Extractor(oracle.make_params(nf, 1.2, 8, 20, 7))
        kps, desc = ex.extract(img)
        okps, odesc = oex.extract(img, cap=nf + 64 * 8)
        assert len(oex.level_candidates(0)[0]) > 3000, (w, h, background)
        assert len(kps) == len(okps) > 0, (w, h, background)
        assert np.array_equal(kps, okps.astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, odesc), (w, h, background)
        ex.close(); oex.close()


def test_constant_image_gives_no_keypoints(vsa):
    w, h = 320, 240
    ex = vsa.Extractor(vsa.make_params(500, 1.2, 4), w, h, device=0)
    kps, desc = ex.extract(np.full((h, w), 93, np.uint8))
    assert len(kps) == 0 and desc.shape == (0, 32)
    ex.close()


def test_fallback_threshold_cells(vsa, oracle):
    """Low-contrast image: most cells have no corner at iniThFAST and fall back to minThFAST."""
    w, h = 400, 300
    img = (vsa.synth_frame(21, w, h).astype(np.int32) - 128) // 6 + 128
    img = img.astype(np.uint8)
    ex = vsa.Extractor(vsa.make_params(800, 1.2, 4), w, h, device=0)
    oex = oracle.Extractor(oracle.make_params(800, 1.2, 4))
    kps, desc = ex.extract(img)
    okps, odesc = oex.extract(img)
    assert len(kps) == len(okps)
    assert np.array_equal(kps["x"], okps["x"]) and np.array_equal(kps["response"], okps["response"])
    assert np.array_equal(desc, odesc)
    assert (okps["response"] < 20).any() and (okps["response"] >= 20).any()
    ex.close()


def test_dense_cells(vsa, oracle):
    """Cells with more pretest survivors than the kernel's survivor list holds (noise at a low threshold: most pixels of
    a cell are corners) take the path that scores round by round and walks all pixels: same candidates, same output.
    Also a wide-cell geometry (cells above 38 pixels) and a frame that is half noise, half flat."""
    rng = np.random.default_rng(77)
    for w, h, ini, mn in ((320, 240, 5, 2), (400, 300, 3, 3), (1241, 376, 4, 1)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img[:, w // 2:] = np.where(rng.random((h, w - w // 2)) < 0.5, img[:, w // 2:], 128) if w == 400 else img[:, w // 2:]
        ex = vsa.Extractor(vsa.make_params(600, 1.2, 4, ini, mn), w, h, max_batch=1, device=0)
        oex = oracle.Extractor(oracle.make_params(600, 1.2, 4, ini, mn))
        kps, desc = ex.extract(img)
        okps, odesc = oex.extract(img)
        for l in range(4):
            gx, gy, gs = vsa.unpack_keys(ex.debug_candidates(0, l))
            ox, oy, os_ = oex.level_candidates(l)
            assert len(gx) == len(ox) and np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gs, os_), (w, h, l)
        assert len(oex.level_candidates(0)[0]) > 20 * (w // 30) * (h // 30)       # tens of maxima per cell: far beyond the usual
        assert np.array_equal(kps, okps.astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, odesc)
        ex.close()


def test_unsupported_geometry_is_an_error(vsa):
    with pytest.raises(vsa.OrbxError) as e:
        vsa.Extractor(vsa.make_params(500, 1.2, 8), 100, 300, device=0)   # W/H < 0.5 -> nIni == 0 in the reference
    assert e.value.status == 2
    with pytest.raises(vsa.OrbxError) as e:
        vsa.Extractor(vsa.make_params(500, 1.2, 8), 100, 80, device=0)    # top level (28x22) smaller than its own 16-px borders
    assert e.value.status == 2


def test_levels_without_cells_yield_nothing(vsa, oracle):
    """A level whose detection area is under 30 px on a side has nCols or nRows == 0 (src/ORBextractor.cc:783-786): the
    reference's cell loops run zero times and the level contributes no keypoints, while the other levels work normally."""
    w, h = 160, 120
    img = vsa.synth_frame(6, w, h)
    ex = vsa.Extractor(vsa.make_params(500, 1.2, 8), w, h, max_batch=2, device=0)
    kps, desc = ex.extract(img)
    okps, odesc = oracle.Extractor(oracle.make_params(500, 1.2, 8)).extract(img)
    assert len(kps) == len(okps) > 100 and np.array_equal(desc, odesc)
    assert np.array_equal(kps, okps.astype(vsa.KEYPOINT_DTYPE))
    per_level = np.bincount(kps["octave"], minlength=8)
    assert np.all(per_level[:4] > 0) and np.all(per_level[4:] == 0)
    for l in range(4, 8):
        assert len(ex.debug_candidates(0, l)) == 0 and len(ex.debug_level_keypoints(0, l)) == 0
        assert np.array_equal(ex.pyramid_level(0, l), oracle_level(oracle, img, l))
    bk, bd, bc = ex.extract_batch(np.stack([img, img]))
    assert bc[0] == bc[1] == len(okps) and np.array_equal(bd[1][:bc[1]], odesc)
    ex.close()


def oracle_level(oracle, img, l):
    oex = oracle.Extractor(oracle.make_params(500, 1.2, 8))
    oex.extract(img)
    return oex.level_padded(l)


def test_octree_count_tables(vsa, oracle):
    """Large key sets (the split steps run on the count tables), spread out and clustered (clusters make the steps
    outlast the tables' depth, so that the keys take over half-way)."""
    rng = np.random.default_rng(23)
    for it, (W, H, n, N) in enumerate([(1200, 400, 30000, 300), (1200, 400, 60000, 1500), (608, 448, 20000, 217), (2000, 900, 50000, 4000),
                                       (3808, 2128, 90000, 434), (900, 450, 25000, 40), (1200, 400, 30000, 0)]):
        xs = rng.integers(3, W - 3, n); ys = rng.integers(3, H - 3, n)
        if it % 2:
            k = n // 2
            xs[:k] = np.clip(W // 3 + rng.integers(-60, 61, k), 3, W - 4); ys[:k] = np.clip(H // 2 + rng.integers(-60, 61, k), 3, H - 4)
        pts = sorted(set((int(y) // 32, int(x) // 31, int(y), int(x)) for x, y in zip(xs, ys)))
        assert len(pts) >= 32 * 256
        x = np.array([p[3] for p in pts], np.int16); y = np.array([p[2] for p in pts], np.int16)
        s = rng.integers(7, 12 if it % 3 == 0 else 200, len(pts)).astype(np.uint8)
        keys = vsa.pack_keys(x, y, s)
        sel = oracle.distribute_octree(x, y, s, 16, 16 + W, 16, 16 + H, N)
        for storage in (0, 1):
            got = vsa.distribute_octree(keys, 16, 16 + W, 16, 16 + H, N, device=0, key_storage=storage)
            assert len(got) == len(sel), (it, W, H, len(pts), N, storage)
            assert np.array_equal(got, keys[sel]), (it, W, H, len(pts), N, storage)


def test_octree_standalone_random(vsa, oracle):
    rng = np.random.default_rng(5)
    for it in range(40):
        W = int(rng.integers(80, 1300)); H = int(rng.integers(80, 500))
        if round(W / H) < 1:
            continue
        n = int(rng.integers(0, (40000 if it % 8 == 0 else 6000) if it % 4 == 0 else 500))
        N = int(rng.integers(0, 400)) if it % 7 else (1500 if it % 14 else 4000)   # 4000: node tables beyond LDS
        pts = set()
        cx, cy = int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))
        for _ in range(n):
            if rng.random() < 0.5:
                x, y = cx + int(rng.integers(-20, 21)), cy + int(rng.integers(-20, 21))
            else:
                x, y = int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))
            if 3 <= x <= W - 4 and 3 <= y <= H - 4:
                pts.add((y // 32, x // 31, y, x))
        pts = sorted(pts)
        x = np.array([p[3] for p in pts], np.int16); y = np.array([p[2] for p in pts], np.int16)
        s = rng.integers(7, 12 if it % 2 else 200, len(pts)).astype(np.uint8)
        keys = vsa.pack_keys(x, y, s)
        sel = oracle.distribute_octree(x, y, s, 16, 16 + W, 16, 16 + H, N)
        for storage in ((0, 1) if it % 5 == 0 else (0,)):     # keys in registers (default) / in device memory
            got = vsa.distribute_octree(keys, 16, 16 + W, 16, 16 + H, N, device=0, key_storage=storage)
            assert len(got) == len(sel), (it, W, H, len(pts), N, storage)
            assert np.array_equal(got, keys[sel]), (it, W, H, len(pts), N, storage)


def test_random_geometries(vsa, oracle):
    """Seeded sweep over image sizes, pyramid shapes and thresholds: final keypoints and descriptors bit-exact, and the
    configurations the oracle rejects (a level narrower than the reference can handle) rejected by the library too."""
    rng = np.random.default_rng(2024)
    ran = 0
    for case in range(40):
        w = int(rng.integers(96, 1000)); h = int(rng.integers(96, 800))
        nf = int(rng.integers(50, 3000)); nl = int(rng.integers(1, 9))
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0])))
        ini = int(rng.integers(5, 40)); mn = int(rng.integers(1, ini + 1))
        mode = int(rng.integers(0, 2))
        img = vsa.synth_frame(1000 + case, w, h)
        op = oracle.make_params(nf, sf, nl, ini, mn, mode)
        try:
            okps, odesc = oracle.Extractor(op).extract(img, cap=nf + 64 * nl + 64)
        except Exception:
            okps = None
        try:
            ex = vsa.Extractor(vsa.make_params(nf, sf, nl, ini, mn, mode), w, h, max_batch=1, device=0)
        except RuntimeError:
            assert okps is None, f"case {case} ({w}x{h}, {nl} levels, scale {sf}): library rejects what the oracle accepts"
            continue
        assert okps is not None, f"case {case} ({w}x{h}, {nl} levels, scale {sf}): library accepts what the oracle rejects"
        kps, desc = ex.extract(img)
        ex.close()
        tag = f"case {case}: {w}x{h} nf={nf} nl={nl} sf={sf} ini={ini} min={mn} mode={mode}"
        assert len(kps) == len(okps), tag
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(kps[f], okps[f]), f"{tag}: {f}"
        assert np.array_equal(desc, odesc), tag
        ran += 1
    assert ran >= 20


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgba", "bgra"])
def test_colour_input(vsa, oracle, fmt):
    """Colour frames straight into the extractor: the grey conversion of src/Tracking.cc:254-267 happens in the first
    pyramid kernel.  Level 0 and the final outputs equal the oracle's on the oracle-converted image (host and batch paths,
    widths that end in a partial 16-pixel chunk)."""
    for w, h in ((640, 480), (331, 207)):
        rng = np.random.default_rng(w)
        chans = [vsa.synth_frame(70 + c, w, h) for c in range(3)]
        img = np.stack(chans + ([rng.integers(0, 256, (h, w), dtype=np.uint8)] if len(fmt) == 4 else []), axis=2)
        grey = oracle.cvt_color_to_gray(img, blue_first=fmt.startswith("b"))
        p = vsa.make_params(800, 1.2, 6)
        ex = vsa.Extractor(p, w, h, max_batch=3, device=0)
        ex.set_input_format(fmt)
        kps, desc = ex.extract(img)
        assert np.array_equal(ex.pyramid_level(0, 0)[19:-19, 19:-19], grey)
        okps, odesc = oracle.Extractor(oracle.make_params(800, 1.2, 6)).extract(grey)
        assert len(kps) == len(okps) > 100
        assert np.array_equal(kps, okps.astype(vsa.KEYPOINT_DTYPE)) and np.array_equal(desc, odesc)
        bk, bd, bc = ex.extract_batch(np.stack([img, img[::-1].copy(), img]))
        assert bc[0] == bc[2] == len(okps) and np.array_equal(bd[0][:bc[0]], odesc) and np.array_equal(bd[2][:bc[2]], odesc)
        o2k, o2d = oracle.Extractor(oracle.make_params(800, 1.2, 6)).extract(np.ascontiguousarray(grey[::-1]))
        assert bc[1] == len(o2k) and np.array_equal(bd[1][:bc[1]], o2d)
        ex.set_input_format("gray")                      # and back: the same context on grey input again
        k3, d3 = ex.extract(grey)
        assert np.array_equal(d3, odesc)
        ex.close()


def test_undistort_keypoints(vsa, oracle):
    """Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:404-456) on the device: coordinates bit-exact against
    the oracle's double-precision restatement of cv::undistortPoints; the batch form on the extractor's own output."""
    import torch
    w, h = 640, 480
    cams = [vsa.Camera(517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314),   # TUM1.yaml
            vsa.Camera(535.4, 539.2, 320.1, 247.6, 0.0, 0.0, 0.0, 0.0, 0.0),                                                  # TUM3.yaml: no distortion
            vsa.Camera(458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)]         # EuRoC.yaml
    ex = vsa.Extractor(vsa.make_params(), w, h, max_batch=3, device=0)
    rng = np.random.default_rng(3)
    pts = np.concatenate([np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32),
                          np.stack([rng.uniform(-50, 700, 5000), rng.uniform(-50, 530, 5000)], 1).astype(np.float32)])
    for cam in cams:
        args = [getattr(cam, n) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]
        got = ex.undistort_points(cam, pts)
        want = oracle.undistort_points(pts, *args)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # batch form behind a device-resident extraction
        B = 3
        imgs = torch.from_numpy(np.stack([vsa.synth_frame(40 + i, w, h) for i in range(B)])).cuda()
        cap = ex.cap
        d_k = torch.zeros(B * cap * 7, dtype=torch.float32, device="cuda"); d_u = torch.full_like(d_k, -7.0)
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"); d_c = torch.zeros(B, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        ex.extract_batch_device(imgs.data_ptr(), w, w * h, B, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
        ex.undistort_keypoints_batch_device(cam, d_k.data_ptr(), d_c.data_ptr(), cap, B, d_u.data_ptr(), st)
        torch.cuda.synchronize()
        k = d_k.cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(B, cap); u = d_u.cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(B, cap)
        c = d_c.cpu().numpy()
        for f in range(B):
            kk, uu = k[f, :c[f]], u[f, :c[f]]
            for name in ("size", "angle", "response", "octave", "class_id"):
                assert np.array_equal(kk[name], uu[name])
            xy = np.stack([kk["x"], kk["y"]], 1)
            want = xy if cam.k1 == 0.0 else oracle.undistort_points(xy, *args)      # k1 == 0: mvKeysUn = mvKeys (:406-410)
            assert np.array_equal(np.stack([uu["x"], uu["y"]], 1).view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
            assert np.all(u[f, c[f]:]["x"] == -7.0)                                # rows past the count untouched
    ex.close()


def test_full_size_batch_kitti_shape(vsa, oracle):
    """The largest workload of the benchmark plan at full size: 4096 frames of 1241x376 with 2000 features in one call
    (tens of GB of working set: every per-frame offset must be computed in 64 bits).  Sixteen distinct frames are tiled,
    so every group of sixteen outputs must repeat the first one exactly, and the first group equals the oracle."""
    import torch
    w, h, B, U = 1241, 376, 4096, 16
    p = vsa.make_params(2000)
    uniq = np.stack([vsa.synth_frame(300 + i, w, h) for i in range(U)])
    frames = torch.from_numpy(uniq).cuda().repeat(B // U, 1, 1).contiguous()
    ex = vsa.Extractor(p, w, h, max_batch=B, device=0)
    cap = ex.cap
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(frames.data_ptr(), w, w * h, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap,
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = counts.cpu().numpy().reshape(B // U, U)
    assert np.all(c == c[0]) and c.min() > 1500
    assert torch.equal(desc.view(B // U, U, cap, 32), desc[:U].unsqueeze(0).expand(B // U, U, cap, 32))
    kbits = kps.view(torch.int32)                                      # class_id -1 reads as NaN in the float view
    assert torch.equal(kbits.view(B // U, U, cap, 7), kbits[:U].unsqueeze(0).expand(B // U, U, cap, 7))
    oex = oracle.Extractor(oracle.make_params(2000))
    for f in (0, 7, 15):
        ok, od = oex.extract(uniq[f])
        n = int(c[0, f])
        assert n == len(ok)
        assert np.array_equal(desc[B - U + f, :n].cpu().numpy(), od)
        assert np.array_equal(kps[B - U + f, :n].cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(n), ok.astype(vsa.KEYPOINT_DTYPE))
    ex.close()


def test_error_paths(vsa, oracle):
    """Status codes instead of faults: output capacity too small (the first `cap` keypoints are still returned and the
    count reports how many there were), bad batch sizes and strides, wrong input layout."""
    import ctypes as C
    w, h = 320, 240
    img = vsa.synth_frame(5, w, h)
    ex = vsa.Extractor(vsa.make_params(500, 1.2, 4), w, h, max_batch=2, device=0)
    okps, odesc = oracle.Extractor(oracle.make_params(500, 1.2, 4)).extract(img)
    cap = 100
    kps = np.zeros(cap, vsa.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
    st = vsa.lib.orbx_extract(ex.h, img.ctypes.data_as(C.c_void_p), w, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    assert st == 4 and n.value == len(okps) > cap                                  # ORBX_ERR_CAPACITY
    assert np.array_equal(desc, odesc[:cap]) and np.array_equal(kps, okps[:cap].astype(vsa.KEYPOINT_DTYPE))
    with pytest.raises(vsa.OrbxError) as e:
        ex.extract_batch(np.stack([img, img, img]))                                # more frames than max_batch
    assert e.value.status == 1
    counts = np.zeros(1, np.int32)
    st = vsa.lib.orbx_extract_batch(ex.h, img.ctypes.data_as(C.c_void_p), w - 1, w * h, 1, kps.ctypes.data_as(C.c_void_p),
                                    desc.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), cap)
    assert st == 1                                                                 # row stride below the width
    with pytest.raises(vsa.OrbxError):
        vsa.lib.orbx_extractor_set_input_format.restype = C.c_int
        vsa._check(vsa.lib.orbx_extractor_set_input_format(ex.h, 9))
    k2, d2 = ex.extract(img)                                                       # the context is still usable
    assert np.array_equal(d2, odesc)
    # a null image is the reference's empty-image early return: success, nothing written
    st = vsa.lib.orbx_extract(ex.h, None, w, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    assert st == 0 and n.value == 0
    ex.close()
    m = vsa.Matcher(device=0)
    with pytest.raises(vsa.OrbxError):
        m.hamming_csr(odesc[:4], odesc, np.array([0, 2, 1, 3, 3], np.int32), np.array([0, 1, 2], np.int32))   # rowptr not monotone
    with pytest.raises(vsa.OrbxError):
        m.hamming_csr(odesc[:2], odesc[:10], np.array([0, 1, 2], np.int32), np.array([3, 10], np.int32))      # candidate out of range
    m.close()


def test_host_batch_pipeline(vsa, oracle):
    """More than eight host frames take the chunked path (pinned mirror filled by several threads, uploads overlapped with
    the kernels of the previous chunk): 70 frames with padded rows, every frame against the oracle; then a short batch on
    the same context."""
    w, h, B = 331, 207, 70
    p = vsa.make_params(400, 1.2, 5)
    padded = np.zeros((B, h, w + 13), np.uint8)
    for i in range(B):
        padded[i, :, :w] = vsa.synth_frame(500 + i % 9, w, h)
        padded[i, :, w:] = 255 - i                                         # garbage past the row end must not matter
    ex = vsa.Extractor(p, w, h, max_batch=B, device=0)
    cap = ex.cap
    kps = np.zeros((B, cap), vsa.KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); counts = np.zeros(B, np.int32)
    vsa._check(vsa.lib.orbx_extract_batch(ex.h, padded.ctypes.data, padded.strides[1], padded.strides[0], B, kps.ctypes.data, desc.ctypes.data,
                                          counts.ctypes.data, cap))
    oex = oracle.Extractor(oracle.make_params(400, 1.2, 5))
    want = [oex.extract(np.ascontiguousarray(padded[i, :, :w])) for i in range(9)]
    for i in range(B):
        ok, od = want[i % 9]
        assert counts[i] == len(ok) and np.array_equal(desc[i, :counts[i]], od) and np.array_equal(kps[i, :counts[i]], ok.astype(vsa.KEYPOINT_DTYPE)), i
    k2, d2, c2 = ex.extract_batch(np.ascontiguousarray(padded[3:6, :, :w]))
    assert np.array_equal(c2, counts[3:6]) and np.array_equal(d2[1, :c2[1]], want[4][1])
    ex.close()


def test_input_view_skips_the_copy_and_changes_nothing(vsa, oracle):
    """Frames written straight into the context's input view (level 0 of the pyramid) give the same keypoints,
    descriptors and padded levels as the same frames passed in a plain buffer, and both equal the oracle."""
    import torch
    w, h, B = 640, 480, 3
    p = vsa.make_params(1000, 1.2, 8, 20, 7)
    imgs = np.stack([vsa.synth_frame(40 + i, w, h) for i in range(B)])
    ex = vsa.Extractor(p, w, h, max_batch=B, device=0)
    cap = ex.cap
    d = torch.from_numpy(imgs).cuda()
    outs = []
    for use_view in (False, True):
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        counts = torch.zeros(B, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        if use_view:
            ptr, rs, fs = ex.input_view()
            assert rs >= w and fs >= rs * h
            # a producer on the device: strided copy of the frames into the view (rows rs apart, frames fs apart)
            torch.as_tensor(ex.input_view_array(B), device="cuda").copy_(d)
            torch.cuda.synchronize()
            ex.extract_batch_device(ptr, rs, fs, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
        else:
            ex.extract_batch_device(d.data_ptr(), w, w * h, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
        torch.cuda.synchronize()
        levels = [[ex.pyramid_level(f, l) for l in range(8)] for f in range(B)]
        outs.append((kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy(), levels))
        # spoil level 0 between the two runs so that the second cannot pass on what the first left there
        if not use_view:
            ex.extract_batch_device(torch.zeros_like(d).data_ptr(), w, w * h, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
            torch.cuda.synchronize()
    (k0, d0, c0, l0), (k1, d1, c1, l1) = outs
    assert np.array_equal(c0, c1) and c0.min() > 500
    assert np.array_equal(k0.view(np.uint32), k1.view(np.uint32)) and np.array_equal(d0, d1)
    for f in range(B):
        for l in range(8):
            assert np.array_equal(l0[f][l], l1[f][l]), (f, l)
    oex = oracle.Extractor(oracle.make_params(1000, 1.2, 8, 20, 7))
    for f in range(B):
        okps, odesc = oex.extract(imgs[f], cap=cap)
        assert c1[f] == len(okps)
        assert np.array_equal(d1[f, :c1[f]], odesc)
        assert np.array_equal(l1[f][0], oex.level_padded(0))
    ex.close()


def test_geometry_beyond_the_launch_grid_is_refused_at_create(vsa):
    """More FAST cells per frame than a launch grid's second dimension holds (65535): refused when the context is
    created, with a message that names the stage, not at the first extraction with a bare HIP error."""
    with pytest.raises(vsa.OrbxError) as e:
        vsa.Extractor(vsa.make_params(20000, 1.1, 16, 20, 7), 4096, 4096, max_batch=1, device=0)
    assert e.value.status == 2 and "workgroups per frame" in str(e.value), str(e.value)


def test_one_call_over_several_contexts(vsa):
    """orbx_extract_batch_device_multi: one call, four contexts on their own streams, joined on the caller's stream --
    the outputs equal those of one context working through the whole batch, from a plain buffer and from the input views."""
    import torch
    w, h, B, C = 640, 480, 10, 4                       # shares of 3, 3, 3, 1
    p = vsa.make_params(1000, 1.2, 8, 20, 7)
    imgs = np.stack([vsa.synth_frame(60 + i, w, h) for i in range(B)])
    d = torch.from_numpy(imgs).cuda()
    one = vsa.Extractor(p, w, h, max_batch=B, device=0)
    cap = one.cap
    def outputs():
        return (torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
                torch.zeros(B, dtype=torch.int32, device="cuda"))
    st = torch.cuda.current_stream().cuda_stream
    k0, d0, c0 = outputs()
    one.extract_batch_device(d.data_ptr(), w, w * h, B, k0.data_ptr(), d0.data_ptr(), c0.data_ptr(), cap, st)
    torch.cuda.synchronize()
    exs = [vsa.Extractor(p, w, h, max_batch=3, device=0) for _ in range(C)]
    k1, d1, c1 = outputs()
    vsa.extract_batch_device_multi(exs, d.data_ptr(), w, w * h, B, k1.data_ptr(), d1.data_ptr(), c1.data_ptr(), cap, st)
    got_counts = c1.cpu().numpy()                      # a read on the caller's stream: the join makes it safe without a device synchronise
    assert np.array_equal(got_counts, c0.cpu().numpy()) and got_counts.min() > 500
    assert torch.equal(k0.view(torch.int32), k1.view(torch.int32)) and torch.equal(d0, d1)
    # the same through the input views: context c holds frames 3c .. 3c+2 of the batch
    for c, ex in enumerate(exs):
        n = min(3, B - 3 * c)
        torch.as_tensor(ex.input_view_array(n), device="cuda").copy_(d[3 * c:3 * c + n])
    k2, d2, c2 = outputs()
    vsa.extract_batch_device_multi(exs, None, 0, 0, B, k2.data_ptr(), d2.data_ptr(), c2.data_ptr(), cap, st)
    torch.cuda.synchronize()
    assert torch.equal(c0, c2) and torch.equal(k0.view(torch.int32), k2.view(torch.int32)) and torch.equal(d0, d2)
    with pytest.raises(vsa.OrbxError):                 # a share larger than a context's max_batch
        vsa.extract_batch_device_multi(exs[:2], d.data_ptr(), w, w * h, B, k2.data_ptr(), d2.data_ptr(), c2.data_ptr(), cap, st)
    # deferred join: two calls back to back, one join
    k3, d3, c3 = outputs()
    vsa.extract_batch_device_multi(exs, d.data_ptr(), w, w * h, B, k3.data_ptr(), d3.data_ptr(), c3.data_ptr(), cap, st, join=False)
    vsa.extract_batch_device_multi(exs, d.data_ptr(), w, w * h, B, k3.data_ptr(), d3.data_ptr(), c3.data_ptr(), cap, st, join=False)
    vsa.extract_multi_join(exs, st)
    assert torch.equal(c0.cpu(), c3.cpu()) and torch.equal(k0.view(torch.int32), k3.view(torch.int32)) and torch.equal(d0, d3)
    for e in exs + [one]:
        e.close()
