"""GPU parity: the HIP matcher (through the C ABI) against the CPU oracle.  Integer results: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher(vsa):
    m = vsa.Matcher(device=0)
    yield m
    m.close()


@pytest.fixture(scope="module")
def stereo_features(vsa, oracle):
    """Keypoints/descriptors of a synthetic stereo pair, extracted by the ORACLE (independent of the GPU extractor)."""
    w, h = 752, 480
    left, right = vsa.synth_stereo(3, w, h)
    oex = oracle.Extractor(oracle.make_params())
    kl, dl = oex.extract(left)
    kr, dr = oex.extract(right)
    return w, h, kl, dl, kr, dr


@pytest.fixture(scope="module")
def motion_features(vsa, oracle):
    """Two views related by a small translation plus fresh noise: many true correspondences."""
    w, h = 640, 480
    a = vsa.synth_frame(12, w, h)
    rng = np.random.default_rng(1)
    b = np.roll(a, (3, 5), axis=(0, 1)).astype(np.int32) + rng.integers(-2, 3, a.shape)
    b = np.clip(b, 0, 255).astype(np.uint8)
    oex = oracle.Extractor(oracle.make_params(2000))
    k1, d1 = oex.extract(a)
    k2, d2 = oex.extract(b)
    return w, h, k1, d1, k2, d2


def eq_best2(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("best", "best_idx", "second"))


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 300), (64, 5000), (513, 20000), (1000, 257), (4096, 70000)])
def test_best2_random(vsa, oracle, matcher, nq, nt):
    table = vsa.synth_descriptors(1000 + nt, nt)
    q, _ = vsa.synth_queries(7 + nq, table, nq)
    assert eq_best2(matcher.hamming_best2(q, table), oracle.hamming_best2(q, table))


def test_best2_ties_and_extremes(vsa, oracle, matcher):
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    table = np.concatenate([base, base, base[::-1], ~base])          # exact duplicates: lowest index must win
    q = np.concatenate([base[:20], ~base[:5], np.zeros((1, 32), np.uint8)])
    got, want = matcher.hamming_best2(q, table), oracle.hamming_best2(q, table)
    assert eq_best2(got, want)
    assert (got["best"][:20] == 0).all() and (got["second"][:20] == 0).all()
    # a table holding only the complement of the query: distance 256 never beats the start value
    one = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    got = matcher.hamming_best2(one, ~one)
    assert got["best"][0] == 256 and got["best_idx"][0] == -1 and got["second"][0] == 256
    assert eq_best2(got, oracle.hamming_best2(one, ~one))


@pytest.mark.parametrize("nt", [15, 16, 17, 255, 4099, 70001, 300001])     # the last one takes the seeded two-pass route
def test_best2_few_patterns(vsa, oracle, matcher, nt):
    """Rows and queries drawn from a handful of patterns (all zeros, all ones, a nibble pattern, a random row, its
    complement and a one-bit neighbour): nearly every comparison is a tie, so the lowest-index rule and the second-best of
    equal distances are exercised across every tile, lane group and row-run boundary; distances 0 and 256 both occur."""
    rng = np.random.default_rng(nt)
    a = rng.integers(0, 256, 32, dtype=np.uint8)
    a1 = a.copy(); a1[7] ^= 0x10
    pats = np.stack([np.zeros(32, np.uint8), np.full(32, 255, np.uint8), np.full(32, 0x0F, np.uint8), a, ~a, a1])
    table = pats[rng.integers(0, len(pats), nt)]
    q = np.concatenate([pats, pats[rng.integers(0, len(pats), 125)], rng.integers(0, 256, (9, 32), dtype=np.uint8)])
    assert eq_best2(matcher.hamming_best2(q, table), oracle.hamming_best2(q, table))


def test_best2_seeded_pass_late_winners(vsa, oracle, matcher):
    """Large map whose best rows lie far behind the rows the threshold seed is taken from, some of them exact duplicates
    of each other: the seeded pass must still find them, lowest index first."""
    nt, nq = 400_000, 200
    table = vsa.synth_descriptors(77, nt)
    q = vsa.synth_descriptors(78, nq)
    rng = np.random.default_rng(5)
    for i in range(nq):                               # plant near-copies of the queries at the back of the map
        for pos in rng.integers(nt // 2, nt, 3):
            row = q[i].copy(); row[rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8)) if i % 3 else np.uint8(0)
            table[pos] = row
    assert eq_best2(matcher.hamming_best2(q, table), oracle.hamming_best2(q, table))


def test_best2_rejects_misaligned_pointers(vsa, matcher):
    """The matrix-core kernel reads 16-byte (queries) and 8-byte (rows) vectors: other alignments are refused on the host."""
    import torch
    buf = torch.zeros(64 * 32 + 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros((4, 4), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(vsa.OrbxError):
        matcher.hamming_best2_device(buf.data_ptr() + 8, 4, buf.data_ptr(), 16, 0, out.data_ptr(), st)
    with pytest.raises(vsa.OrbxError):
        matcher.hamming_best2_device(buf.data_ptr(), 4, buf.data_ptr() + 4, 16, 0, out.data_ptr(), st)
    matcher.hamming_best2_device(buf.data_ptr() + 16, 4, buf.data_ptr() + 8, 16, 0, out.data_ptr(), st)
    torch.cuda.synchronize()
    assert (out[:, 0] == 0).all().item() and (out[:, 1] == 0).all().item()


def test_best2_empty_table(vsa, matcher):
    q = vsa.synth_descriptors(1, 5)
    got = matcher.hamming_best2(q, np.zeros((0, 32), np.uint8))
    assert (got["best"] == 256).all() and (got["best_idx"] == -1).all() and (got["second"] == 256).all()


def test_best2_large_property(vsa, oracle, matcher):
    """2^20 rows: every query is a corrupted copy of a known row; spot-check a slice against the oracle."""
    nt, nq = 1 << 20, 512
    table = vsa.synth_descriptors(42, nt)
    q, src = vsa.synth_queries(43, table, nq)
    got = matcher.hamming_best2(q, table)
    d_src = np.array([oracle.descriptor_distance(q[i], table[src[i]]) for i in range(nq)])
    assert (got["best"] <= d_src).all()
    assert (got["best_idx"] == src).mean() > 0.99
    assert (got["second"] >= got["best"]).all()
    want = oracle.hamming_best2(q[:8], table)
    assert eq_best2(got[:8], want)


def test_best2_full_size_and_sharded(vsa, oracle, matcher):
    """The map-matching workload at full size (4096 queries x 2^24 rows, 512 MiB of descriptors) in one pass, and the
    same map cut into 8 contiguous shards whose results are merged the way the ranks of a multi-GPU run merge theirs:
    both must agree exactly; planted rows are found; 64 queries are checked against the oracle."""
    import torch
    nt, nq, shards = 1 << 24, 4096, 8
    table = vsa.synth_descriptors(42, nt)
    q, src = vsa.synth_queries(43, table, nq)
    d_t = torch.from_numpy(table).cuda(); d_q = torch.from_numpy(q).cuda()
    st = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros((nq, 4), dtype=torch.int32, device="cuda")
    matcher.hamming_best2_device(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 0, whole.data_ptr(), st)
    parts = torch.zeros((shards, nq, 4), dtype=torch.int32, device="cuda")
    rows = nt // shards
    for s in range(shards):
        matcher.hamming_best2_device(d_q.data_ptr(), nq, d_t[s * rows:].data_ptr(), rows, s * rows, parts[s].data_ptr(), st)
    merged = torch.zeros((nq, 4), dtype=torch.int32, device="cuda")
    matcher.best2_merge_device(parts.data_ptr(), shards, nq, merged.data_ptr(), st)
    torch.cuda.synchronize()
    w = whole.cpu().numpy(); mg = merged.cpu().numpy()
    assert np.array_equal(w[:, :3], mg[:, :3])
    assert (w[:, 1] == src).mean() > 0.99 and (w[:, 2] >= w[:, 0]).all()
    # 64 queries against the oracle over the whole map: every position n of a 16-query tile four times, every one of the
    # eight query tiles of a workgroup, spread over the 32 workgroup rows of the grid (every query meets all four lane
    # groups, which hold different rows of the same queries)
    pick = np.array([((j * 5) % 32) * 128 + ((j // 3) % 8) * 16 + (j % 16) for j in range(64)])
    assert len(set(pick.tolist())) == 64 and set((pick % 16).tolist()) == set(range(16)) and set(((pick % 128) // 16).tolist()) == set(range(8))
    want = oracle.hamming_best2(np.ascontiguousarray(q[pick]), table)
    assert np.array_equal(w[pick, 0], want["best"]) and np.array_equal(w[pick, 1], want["best_idx"]) and np.array_equal(w[pick, 2], want["second"])


def test_csr_distances(vsa, oracle, matcher):
    rng = np.random.default_rng(3)
    t = vsa.synth_descriptors(5, 900)
    q = vsa.synth_descriptors(6, 300)
    lens = rng.integers(0, 90, 300); lens[::17] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, 900, rowptr[-1]).astype(np.int32)
    dist = matcher.hamming_csr(q, t, rowptr, cand)
    want = np.array([oracle.descriptor_distance(q[i], t[c]) for i in range(300) for c in cand[rowptr[i]:rowptr[i + 1]]], np.int32)
    assert np.array_equal(dist, want)


def test_search_for_initialization(vsa, oracle, matcher, motion_features):
    w, h, kl, dl, kr, dr = motion_features
    bounds = (0.0, float(w), 0.0, float(h))
    prev = np.stack([kl["x"], kl["y"]], 1).astype(np.float32)
    for check_ori in (True, False):
        for window, ratio in ((100, 0.9), (40, 0.8), (300, 1.0)):
            g = matcher.search_for_initialization(kl, dl, kr, dr, bounds, prev, window, ratio, check_ori)
            o = oracle.search_for_initialization(kl, dl, kr, dr, bounds, prev, window, ratio, check_ori)
            assert g[1] == o[1] and np.array_equal(g[0], o[0]) and np.array_equal(g[2], o[2])
    assert o[1] >= 0
    g = matcher.search_for_initialization(kl, dl, kr, dr, bounds, prev, 100, 0.9, True)
    assert g[1] > 20 and (g[0] >= 0).sum() == g[1]


@pytest.mark.parametrize("shape", [(640, 480, 1000, 4, 64), (1241, 376, 4000, 8, 6), (1920, 1080, 8000, 8, 3)],
                         ids=["cap1000_lds", "cap4000_arena", "cap8000_arena"])
def test_search_for_initialization_batch_device(vsa, oracle, matcher, shape):
    """Frame pairs (64 at the usual size; a few at the capacities of the reference's initialisation extractor, which runs
    at twice nFeatures, src/Tracking.cc:136-137: 4000 for the KITTI settings, 8000 for the 3840x2160 sheet -- there the
    frame's descriptors live in device memory instead of LDS) straight from orbx_extract_batch_device's output, matched on the device without a host round trip:
    every pair's vnMatches12, match count and updated vbPrevMatched equal the oracle's sequential routine run on that pair.
    Second round: the updated vbPrevMatched fed back in with a smaller window (the way the tracker calls it frame after
    frame); also a pair of empty frames and equal frames (ties everywhere: the visiting order decides)."""
    import torch
    w, h, nf, nlev, P = shape
    rng = np.random.default_rng(33)
    frames = []
    for p in range(P):
        a = vsa.synth_frame(700 + p % 20, w, h)
        if p == 5:
            a = np.full((h, w), 77, np.uint8)                                       # no keypoints at all
        b = a if p == 6 else np.clip(np.roll(a, (p % 7 - 3, p % 11 - 5), axis=(0, 1)).astype(np.int32) + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
        frames += [a, b]
    ex = vsa.Extractor(vsa.make_params(nf, 1.2, nlev), w, h, max_batch=2 * P, device=0)
    cap = ex.cap
    assert cap >= nf
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    d_k = torch.zeros((2 * P, cap, 7), dtype=torch.float32, device="cuda")
    d_d = torch.zeros((2 * P, cap, 32), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(2 * P, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(imgs.data_ptr(), w, w * h, 2 * P, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), cap, st)
    # frames 2p and 2p+1 form pair p: de-interleave views (stride 2 over frames) into contiguous per-pair arrays
    k1 = d_k[0::2].contiguous(); k2 = d_k[1::2].contiguous(); dd1 = d_d[0::2].contiguous(); dd2 = d_d[1::2].contiguous()
    n1 = d_c[0::2].contiguous(); n2 = d_c[1::2].contiguous()
    prev = k1[:, :, :2].contiguous()                                              # vbPrevMatched starts at F1's keypoints (src/Tracking.cc:628-630)
    m12 = torch.full((P, cap), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    bounds = (0.0, float(w), 0.0, float(h))
    hk1 = k1.cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(P, cap); hk2 = k2.cpu().numpy().view(vsa.KEYPOINT_DTYPE).reshape(P, cap)
    hd1 = dd1.cpu().numpy(); hd2 = dd2.cpu().numpy(); hn1 = n1.cpu().numpy(); hn2 = n2.cpu().numpy()
    hprev = prev.cpu().numpy().copy()
    total = 0
    for window, ratio, ori in ((100, 0.9, True), (30, 0.8, True), (100, 0.9, False)):
        matcher.search_for_initialization_batch_device(k1.data_ptr(), dd1.data_ptr(), n1.data_ptr(), k2.data_ptr(), dd2.data_ptr(), n2.data_ptr(),
                                                       cap, P, bounds, prev.data_ptr(), window, ratio, ori, m12.data_ptr(), nm.data_ptr(), st)
        torch.cuda.synchronize()
        gm, gn, gp = m12.cpu().numpy(), nm.cpu().numpy(), prev.cpu().numpy()
        for p in range(P):
            a, b = int(hn1[p]), int(hn2[p])
            om, on, op = oracle.search_for_initialization(hk1[p, :a], hd1[p, :a], hk2[p, :b], hd2[p, :b], bounds, hprev[p, :a], window, ratio, ori)
            assert gn[p] == on, (p, window, gn[p], on)
            assert np.array_equal(gm[p, :a], om) and np.all(gm[p, a:] == -1), (p, window)
            assert np.array_equal(gp[p, :a].view(np.uint32), np.asarray(op, np.float32).reshape(-1, 2).view(np.uint32)), (p, window)
            hprev[p, :a] = np.asarray(op, np.float32).reshape(-1, 2)
            total += on
    assert (P <= 5 or hn1[5] == 0) and total > 50 * P
    with pytest.raises(vsa.OrbxError) as e:      # refused before anything is launched: the arrays are not that large
        matcher.search_for_initialization_batch_device(k1.data_ptr(), dd1.data_ptr(), n1.data_ptr(), k2.data_ptr(), dd2.data_ptr(), n2.data_ptr(),
                                                       20000, P, bounds, prev.data_ptr(), 100, 0.9, True, m12.data_ptr(), nm.data_ptr(), st)
    assert e.value.status == 4
    ex.close()


def test_search_by_projection_variants(vsa, oracle, matcher, motion_features):
    w, h, kl, dl, kr, dr = motion_features
    rng = np.random.default_rng(9)
    bounds = (0.0, float(w), 0.0, float(h))
    scale = vsa.scale_tables(vsa.make_params())["scale"]
    nq = len(kl)
    qu = kl["x"] + 5.0 + rng.uniform(-3, 3, nq).astype(np.float32)
    qv = kl["y"] + 3.0 + rng.uniform(-3, 3, nq).astype(np.float32)
    lvl = kl["octave"].astype(np.int32)
    uright = np.where(rng.random(len(kr)) < 0.5, kr["x"] - 5.0, -1.0).astype(np.float32)
    qur = (qu - 5.0).astype(np.float32)
    claimed0 = np.full(len(kr), -1, np.int32); claimed0[::13] = 0
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.6)):
        qr = (np.where(rng.random(nq) < 0.5, 2.5, 4.0) * th * scale[lvl]).astype(np.float32)
        g = matcher.search_by_projection_mappoints(qu, qv, qr, qur, lvl, dl, kr, dr, uright, bounds, ratio, claimed0)
        o = oracle.search_by_projection_mappoints(qu, qv, qr, qur, lvl, dl, kr, dr, uright, bounds, ratio, claimed0)
        assert g[1] == o[1] and np.array_equal(g[0], o[0])
    assert o[1] > 10
    for mode in (0, 1, 2):
        for th_dist, check_ori in ((100, True), (64, False)):
            radius = (15.0 * scale[lvl]).astype(np.float32)
            g = matcher.search_by_projection_lastframe(qu, qv, radius, qur, lvl, kl["angle"], dl, kr, dr, uright, bounds, mode, th_dist, check_ori, claimed0)
            o = oracle.search_by_projection_lastframe(qu, qv, radius, qur, lvl, kl["angle"], dl, kr, dr, uright, bounds, mode, th_dist, check_ori, claimed0)
            assert g[1] == o[1] and np.array_equal(g[0], o[0])
    assert o[1] > 10
    # map points without observations (the temporal points of stereo / RGB-D tracking): their matches do not block the
    # feature (src/ORBmatcher.cc:87-89, :1406-1408), so later queries can overwrite it; the count still goes up each time
    transient = (rng.random(nq) < 0.5).astype(np.uint8)
    dup = np.concatenate([np.arange(nq), np.arange(0, nq, 3)])              # repeated queries make overwrites certain
    a = lambda v: np.ascontiguousarray(np.asarray(v)[dup])
    radius = (15.0 * scale[lvl]).astype(np.float32)
    saw_overwrite = False
    for mode, check_ori in ((0, True), (1, False)):
        g = matcher.search_by_projection_lastframe(a(qu), a(qv), a(radius), a(qur), a(lvl), a(kl["angle"]), a(dl), kr, dr, uright, bounds, mode, 100, check_ori, claimed0, a(transient))
        o = oracle.search_by_projection_lastframe(a(qu), a(qv), a(radius), a(qur), a(lvl), a(kl["angle"]), a(dl), kr, dr, uright, bounds, mode, 100, check_ori, claimed0, a(transient))
        assert g[1] == o[1] and np.array_equal(g[0], o[0])
        assert (o[0] <= -2).any()
        saw_overwrite |= o[1] > np.count_nonzero((o[0] != -1) & (o[0] != 0) | ((o[0] == 0) & (claimed0 == -1)))
    assert saw_overwrite
    g = matcher.search_by_projection_mappoints(a(qu), a(qv), a(radius), a(qur), a(lvl), a(dl), kr, dr, uright, bounds, 0.9, claimed0, a(transient))
    o = oracle.search_by_projection_mappoints(a(qu), a(qv), a(radius), a(qur), a(lvl), a(dl), kr, dr, uright, bounds, 0.9, claimed0, a(transient))
    assert g[1] == o[1] and np.array_equal(g[0], o[0]) and (o[0] <= -2).any()


@pytest.mark.parametrize("cap_target", [0, 4100, 8100, 16000], ids=["tight_lds", "cap4100_arena", "cap8100_arena", "cap16000_arena"])
def test_search_by_projection_batch_device(vsa, oracle, matcher, motion_features, stereo_features, cap_target):
    """(cap_target: per-frame capacity of the arrays; above about 2400 the frame's descriptors live in device memory
    instead of LDS.)  The projection searches for a batch of calls on the device, against the oracle's sequential routines call by call:
    every motion mode, both acceptance distances, with and without the orientation check, stereo right coordinates,
    pre-claimed features, queries without observations (overwrites), repeated queries (ties: the visiting order decides),
    queries outside the image, an empty query list and an empty frame."""
    import torch
    w, h, kl, dl, kr, dr = motion_features
    rng = np.random.default_rng(19)
    scale = vsa.scale_tables(vsa.make_params())["scale"]
    bounds = (0.0, float(w), 0.0, float(h))
    C = 16                                                        # calls in the batch
    cap = max(len(kr) + 7, cap_target); qcap = len(kl) + len(kl) // 3 + 5
    calls = []
    for c in range(C):
        n2 = len(kr) if c != 3 else 0
        sel = np.concatenate([np.arange(len(kl)), np.arange(0, len(kl), 3)]) if c % 2 else rng.permutation(len(kl))[:max(1, len(kl) // (1 + c % 4))]
        if c == 5:
            sel = sel[:0]
        nq = len(sel)
        qu = (kl["x"][sel] + 5.0 + rng.uniform(-3, 3, nq)).astype(np.float32); qv = (kl["y"][sel] + 3.0 + rng.uniform(-3, 3, nq)).astype(np.float32)
        if nq > 10:
            qu[:3] = [-500.0, 5000.0, 10.0]; qv[:3] = [20.0, 30.0, -900.0]            # windows outside the grid
        lvl = kl["octave"][sel].astype(np.int32)
        rad = (rng.choice([7.0, 15.0, 30.0]) * scale[lvl]).astype(np.float32)
        calls.append(dict(n2=n2, nq=nq, qu=qu, qv=qv, rad=rad, qur=(qu - 5.0).astype(np.float32), lvl=lvl, ang=kl["angle"][sel].astype(np.float32),
                          qd=dl[sel], tr=(rng.random(nq) < (0.5 if c % 3 == 0 else 0.0)).astype(np.uint8),
                          ur2=np.where(rng.random(len(kr)) < 0.5, kr["x"] - 5.0, -1.0).astype(np.float32),
                          cl=np.where(rng.random(len(kr)) < 0.08, 1 << 30, -1).astype(np.int32)))
    def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    def pack(key, width, dtype, fill=0):
        out = np.full((C, width) + calls[0][key].shape[1:], fill, dtype)
        for c, d in enumerate(calls):
            out[c, :len(d[key])] = d[key]
        return out
    d_qu, d_qv, d_rad, d_qur = dev(pack("qu", qcap, np.float32)), dev(pack("qv", qcap, np.float32)), dev(pack("rad", qcap, np.float32)), dev(pack("qur", qcap, np.float32))
    d_lvl, d_ang, d_qd, d_tr = dev(pack("lvl", qcap, np.int32)), dev(pack("ang", qcap, np.float32)), dev(pack("qd", qcap, np.uint8)), dev(pack("tr", qcap, np.uint8))
    d_nq = dev(np.array([d["nq"] for d in calls], np.int32)); d_n2 = dev(np.array([d["n2"] for d in calls], np.int32))
    k2 = np.zeros((C, cap), vsa.KEYPOINT_DTYPE); k2[:, :len(kr)] = kr
    dd2 = np.zeros((C, cap, 32), np.uint8); dd2[:, :len(kr)] = dr
    d_k2, d_d2, d_ur2 = dev(k2.view(np.float32).reshape(C, cap, 7)), dev(dd2), dev(pack("ur2", cap, np.float32, -1.0))
    st = torch.cuda.current_stream().cuda_stream
    seen_overwrite = False
    for form, mode, th_dist, ratio, ori, stereo in ((0, 0, 100, 0.0, True, True), (0, 1, 100, 0.0, True, False), (0, 2, 64, 0.0, False, True),
                                                    (0, 0, 64, 0.0, True, False), (1, 0, 0, 0.8, False, True), (1, 0, 0, 0.6, False, False)):
        d_cl = dev(pack("cl", cap, np.int32, -1)); d_nm = torch.zeros(C, dtype=torch.int32, device="cuda")
        matcher.search_by_projection_batch_device(form, d_qu.data_ptr(), d_qv.data_ptr(), d_rad.data_ptr(), d_qur.data_ptr(), d_lvl.data_ptr(),
                                                  d_ang.data_ptr(), d_qd.data_ptr(), d_tr.data_ptr(), d_nq.data_ptr(), qcap, d_k2.data_ptr(), d_d2.data_ptr(),
                                                  d_ur2.data_ptr() if stereo else None, d_n2.data_ptr(), cap, C, bounds, mode, th_dist, ratio, ori,
                                                  d_cl.data_ptr(), d_nm.data_ptr(), st)
        torch.cuda.synchronize()
        gcl, gnm = d_cl.cpu().numpy(), d_nm.cpu().numpy()
        for c, d in enumerate(calls):
            n2 = d["n2"]
            ur = d["ur2"][:n2] if stereo else None
            if form == 0:
                ocl, on = oracle.search_by_projection_lastframe(d["qu"], d["qv"], d["rad"], d["qur"], d["lvl"], d["ang"], d["qd"], kr[:n2], dr[:n2], ur, bounds,
                                                                mode, th_dist, ori, d["cl"][:n2], d["tr"]) if d["nq"] and n2 else (d["cl"][:n2], 0)
            else:
                ocl, on = oracle.search_by_projection_mappoints(d["qu"], d["qv"], d["rad"], d["qur"], d["lvl"], d["qd"], kr[:n2], dr[:n2], ur, bounds,
                                                                ratio, d["cl"][:n2], d["tr"]) if d["nq"] and n2 else (d["cl"][:n2], 0)
            assert gnm[c] == on, (form, mode, c, gnm[c], on)
            assert np.array_equal(gcl[c, :n2], ocl), (form, mode, c, int(np.count_nonzero(gcl[c, :n2] != ocl)))
            assert np.array_equal(gcl[c, n2:], pack("cl", cap, np.int32, -1)[c, n2:])
            seen_overwrite |= on > np.count_nonzero((ocl != -1) & (ocl != (1 << 30)))
    assert seen_overwrite
    with pytest.raises(vsa.OrbxError) as e:
        matcher.search_by_projection_batch_device(0, d_qu.data_ptr(), d_qv.data_ptr(), d_rad.data_ptr(), None, d_lvl.data_ptr(), d_ang.data_ptr(), d_qd.data_ptr(),
                                                  None, d_nq.data_ptr(), qcap, d_k2.data_ptr(), d_d2.data_ptr(), None, d_n2.data_ptr(), 40000, C, bounds, 0, 100, 0.0,
                                                  True, d_cl.data_ptr(), d_nm.data_ptr(), st)
    assert e.value.status == 4


def test_stereo_match(vsa, oracle, matcher, stereo_features):
    w, h, kl, dl, kr, dr = stereo_features
    scale = vsa.scale_tables(vsa.make_params())["scale"]
    bf, fx = 47.90639384423901, 435.2046959714599       # EuRoC.yaml
    mb = bf / fx
    g = matcher.stereo_match(kl, dl, kr, dr, scale, h, 0.0, bf / mb)
    o = oracle.stereo_match(kl, dl, kr, dr, scale, h, 0.0, bf / mb)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    assert ((o[0] >= 0) & (o[0] < 75)).sum() > 50
    g = matcher.stereo_match(kl, dl, kr[:0], dr[:0], scale, h, 0.0, bf / mb)
    assert (g[0] == -1).all()


@pytest.mark.parametrize("nfeatures,lists_in_lds", [(1000, True), (1000, False), (2600, True)])
def test_stereo_pipeline_device_resident(vsa, oracle, matcher, monkeypatch, nfeatures, lists_in_lds):
    """The 752x480 stereo shape: left+right extraction then stereo Hamming, all buffers in HBM.  The row search with its
    index lists in LDS and in device memory; 2600 features: more left keypoints than the median filter holds in registers."""
    torch = pytest.importorskip("torch")
    w, h, B = 752, 480, 2
    p = vsa.make_params(nfeatures=nfeatures)
    if not lists_in_lds:
        monkeypatch.setenv("ORBX_STEREO_LIST_IN_LDS", "0")
    exl = vsa.Extractor(p, w, h, max_batch=B, device=0)
    exr = vsa.Extractor(p, w, h, max_batch=B, device=0)
    pairs = [vsa.synth_stereo(30 + i, w, h) for i in range(B)]
    dev = torch.device("cuda:0")
    L = torch.from_numpy(np.stack([p_[0] for p_ in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p_[1] for p_ in pairs])).to(dev)
    cap = exl.cap
    out = {}
    for name, ex, imgs in (("l", exl, L), ("r", exr, R)):
        k = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        c = torch.zeros(B, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        ex.extract_batch_device(imgs.data_ptr(), w, w * h, B, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, s)
        out[name] = (k, d, c)
    bd = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    bi = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    scale = vsa.scale_tables(p)["scale"]
    bf, fx = 47.90639384423901, 435.2046959714599
    matcher.stereo_match_batch_device(out["l"][0].data_ptr(), out["l"][1].data_ptr(), out["l"][2].data_ptr(),
                                      out["r"][0].data_ptr(), out["r"][1].data_ptr(), out["r"][2].data_ptr(),
                                      cap, B, scale, h, 0.0, bf / (bf / fx), bd.data_ptr(), bi.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    # sub-pixel refinement and median filter on the two contexts' device pyramids (src/Frame.cc:551-639)
    ur = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    sad = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    mb = bf / fx
    exl.stereo_refine_batch_device(exr, out["l"][0].data_ptr(), out["l"][2].data_ptr(), out["r"][0].data_ptr(), cap, B,
                                   bd.data_ptr(), bi.data_ptr(), bf, 0.0, bf / mb, ur.data_ptr(), dp.data_ptr(), sad.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oexl, oexr = oracle.Extractor(oracle.make_params(nfeatures=nfeatures)), oracle.Extractor(oracle.make_params(nfeatures=nfeatures))
    for i in range(B):
        kl, dl = oexl.extract(pairs[i][0]); kr, dr = oexr.extract(pairs[i][1])
        nl = int(out["l"][2][i])
        assert nl == len(kl) and int(out["r"][2][i]) == len(kr)
        assert nfeatures < 2000 or nl > 64 * 24
        assert np.array_equal(out["l"][1][i, :nl].cpu().numpy(), dl)
        o = oracle.stereo_match(kl, dl, kr, dr, scale, h, 0.0, bf / mb)
        assert np.array_equal(bd[i, :nl].cpu().numpy(), o[0]) and np.array_equal(bi[i, :nl].cpu().numpy(), o[1])
        our, odp, osad = oracle.stereo_refine(oexl, oexr, kl, kr, o[0], o[1], bf, 0.0, bf / mb)
        gur, gdp = ur[i, :nl].cpu().numpy(), dp[i, :nl].cpu().numpy()
        assert np.count_nonzero(our >= 0) > 50, "the synthetic pair should give stereo matches"
        assert np.array_equal(gur, our) and np.array_equal(gdp, odp)          # float results bit-exact
        assert np.array_equal(sad[i, :nl].cpu().numpy(), osad)
    # host-array form on the last pair (frame index 0 of each context must hold the pair: re-extract it alone)
    kl_g, dl_g = exl.extract(pairs[0][0]); kr_g, dr_g = exr.extract(pairs[0][1])
    kl, dl = oexl.extract(pairs[0][0]); kr, dr = oexr.extract(pairs[0][1])
    o = oracle.stereo_match(kl, dl, kr, dr, scale, h, 0.0, bf / mb)
    our, odp, _ = oracle.stereo_refine(oexl, oexr, kl, kr, o[0], o[1], bf, 0.0, bf / mb)
    gur, gdp = exl.stereo_refine(exr, kl_g, kr_g, o[0], o[1], bf, 0.0, bf / mb)
    assert np.array_equal(gur, our) and np.array_equal(gdp, odp)
    exl.close(); exr.close()


def _nodes(desc):
    """A stand-in vocabulary: the node of a feature is read off ten descriptor bits, so most true correspondences share it."""
    return ((desc[:, 0].astype(np.int64) >> 3) << 5) | (desc[:, 16].astype(np.int64) >> 3)


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow(vsa, oracle, matcher, motion_features, form, check_ori):
    w, h, k1, d1, k2, d2 = motion_features
    rng = np.random.default_rng(5 + form)
    e1 = rng.random(len(k1)) < 0.8                       # key-frame features with a good map point
    e2 = (rng.random(len(k2)) < 0.8) if form == 1 else None
    n1, n2 = _nodes(d1), _nodes(d2)
    for ratio in (0.7, 0.9):
        g = matcher.search_by_bow(d1, k1["angle"], e1, n1, d2, k2["angle"], e2, n2, ratio, check_ori, form)
        o = oracle.search_by_bow(d1, k1["angle"], e1, n1, d2, k2["angle"], e2, n2, ratio, check_ori, form)
        assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
        assert g[2] > 50
    # empty sides and disjoint vocabularies
    g = matcher.search_by_bow(d1[:0], k1["angle"][:0], e1[:0], n1[:0], d2, k2["angle"], e2, n2, 0.7, check_ori, form)
    assert g[2] == 0 and np.all(g[1] == -1)
    g = matcher.search_by_bow(d1, k1["angle"], e1, n1, d2, k2["angle"], e2, n2 + 100000, 0.7, check_ori, form)
    assert g[2] == 0


@pytest.mark.parametrize("only_stereo", [False, True])
def test_search_for_triangulation(vsa, oracle, matcher, motion_features, only_stereo):
    w, h, k1, d1, k2, d2 = motion_features
    rng = np.random.default_rng(9)
    has1 = rng.random(len(k1)) < 0.5
    has2 = rng.random(len(k2)) < 0.5
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 10.0, -1.0).astype(np.float32)
    ur2 = np.where(rng.random(len(k2)) < 0.5, k2["x"] - 10.0, -1.0).astype(np.float32)
    # fundamental matrix of a pure image translation (x2 = x1 + t): epipolar lines parallel to t; a scale keeps values generic
    t = np.array([5.0, 3.0, 0.0])
    F12 = (1e-3 * np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])).T.astype(np.float32)
    tabs = vsa.scale_tables(vsa.make_params(2000))
    n1, n2 = _nodes(d1), _nodes(d2)
    for check_ori in (True, False):
        g = matcher.search_for_triangulation(k1, d1, has1, ur1, n1, k2, d2, has2, ur2, n2, F12, 320.0, 240.0, tabs["scale"], tabs["sigma2"],
                                             only_stereo, check_ori)
        o = oracle.search_for_triangulation(k1, d1, has1, ur1, n1, k2, d2, has2, ur2, n2, F12, 320.0, 240.0, tabs["scale"], tabs["sigma2"],
                                            only_stereo, check_ori)
        assert g[1] == o[1] and np.array_equal(g[0], o[0])
        assert g[1] > 10


def test_search_by_projection_keyframe(vsa, oracle, matcher, motion_features):
    """The loop-closing form: map points projected with a Sim3 into a key frame, best-1 within TH_LOW."""
    w, h, k1, d1, k2, d2 = motion_features
    rng = np.random.default_rng(11)
    sel = rng.permutation(len(k1))[:1200]
    scale = vsa.scale_tables(vsa.make_params(2000))["scale"]
    qu = k1["x"][sel] + 5.0 + rng.normal(0, 1.0, len(sel)).astype(np.float32)      # the second view is the first shifted by (5, 3)
    qv = k1["y"][sel] + 3.0 + rng.normal(0, 1.0, len(sel)).astype(np.float32)
    qlevel = np.clip(k1["octave"][sel] + rng.integers(0, 2, len(sel)), 0, 7).astype(np.int32)
    qradius = (10.0 * scale[qlevel]).astype(np.float32)
    claimed = np.where(rng.random(len(k2)) < 0.2, 1 << 20, -1).astype(np.int32)       # features already matched before the call
    g = matcher.search_by_projection_keyframe(qu, qv, qradius, qlevel, d1[sel], k2, d2, (0.0, float(w), 0.0, float(h)), claimed)
    o = oracle.search_by_projection_keyframe(qu, qv, qradius, qlevel, d1[sel], k2, d2, (0.0, float(w), 0.0, float(h)), claimed)
    assert g[1] == o[1] and np.array_equal(g[0], o[0]) and g[1] > 100


@pytest.mark.parametrize("reproj", [False, True])
def test_fuse_candidate_loop(vsa, oracle, matcher, motion_features, reproj):
    """ORBmatcher::Fuse, both forms: per projected map point the nearest key-frame feature in the radius and level band,
    with the chi-square reprojection test of the local-mapping form."""
    w, h, k1, d1, k2, d2 = motion_features
    rng = np.random.default_rng(21 + reproj)
    sel = rng.permutation(len(k1))[:1500]
    tabs = vsa.scale_tables(vsa.make_params(2000))
    qu = k1["x"][sel] + 5.0 + rng.normal(0, 1.5, len(sel)).astype(np.float32)
    qv = k1["y"][sel] + 3.0 + rng.normal(0, 1.5, len(sel)).astype(np.float32)
    qlevel = np.clip(k1["octave"][sel] + rng.integers(0, 2, len(sel)), 0, 7).astype(np.int32)
    qradius = (3.0 * tabs["scale"][qlevel]).astype(np.float32)
    bounds = (0.0, float(w), 0.0, float(h))
    if reproj:
        uright2 = np.where(rng.random(len(k2)) < 0.6, k2["x"] - rng.uniform(2, 30, len(k2)), -1).astype(np.float32)
        qur = (qu - rng.uniform(2, 30, len(sel))).astype(np.float32)
        g = matcher.project_best1(qu, qv, qradius, qlevel, d1[sel], k2, d2, bounds, qur, uright2, tabs["inv_sigma2"])
        o = oracle.fuse_candidates(qu, qv, qradius, qlevel, d1[sel], k2, d2, bounds, qur, uright2, tabs["inv_sigma2"])
    else:
        g = matcher.project_best1(qu, qv, qradius, qlevel, d1[sel], k2, d2, bounds)
        o = oracle.fuse_candidates(qu, qv, qradius, qlevel, d1[sel], k2, d2, bounds)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    assert np.count_nonzero(g[0] >= 0) > 200 and np.count_nonzero(g[0] < 0) > 20
    assert np.count_nonzero(g[1][g[0] >= 0] <= 50) > 100            # fused at TH_LOW


def test_search_by_sim3(vsa, oracle, matcher, motion_features):
    """ORBmatcher::SearchBySim3: both projection directions, best-1 within TH_HIGH, mutual agreement."""
    w, h, k1, d1, k2, d2 = motion_features
    rng = np.random.default_rng(31)
    scale = vsa.scale_tables(vsa.make_params(2000))["scale"]

    def side(ks, ds, dx, dy):
        n = len(ks)
        valid = (rng.random(n) < 0.8).astype(np.uint8)
        u = ks["x"] + dx + rng.normal(0, 1.0, n).astype(np.float32); v = ks["y"] + dy + rng.normal(0, 1.0, n).astype(np.float32)
        lvl = np.clip(ks["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
        flips = np.packbits(rng.random((n, 256)) < 0.03, axis=1)                     # the map point's descriptor is not the feature's own
        return valid, u, v, (7.5 * scale[lvl]).astype(np.float32), lvl, ds ^ flips

    s1, s2 = side(k1, d1, 5.0, 3.0), side(k2, d2, -5.0, -3.0)
    bounds = (0.0, float(w), 0.0, float(h))
    g = matcher.search_by_sim3(s1, k1, d1, s2, k2, d2, bounds)
    o = oracle.search_by_sim3(s1, k1, d1, s2, k2, d2, bounds)
    assert g[1] == o[1] and np.array_equal(g[0], o[0]) and g[1] > 200
