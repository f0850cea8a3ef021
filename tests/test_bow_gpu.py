"""GPU parity of the Hamming consumers that follow extraction (distinctive descriptors of map points, vocabulary
transform) against the CPU oracle, through the C ABI.  Ids and indices bit-exact; BowVector values are doubles computed
in the reference's order on the host, compared exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher(vsa):
    m = vsa.Matcher(device=0)
    yield m
    m.close()


def clustered_descriptors(vsa, rng, n_groups, max_n):
    """Observation sets: a base descriptor per map point, each observation a noisy copy of it."""
    sizes = rng.integers(0, max_n + 1, n_groups)
    sizes[rng.integers(0, n_groups, max(1, n_groups // 50))] = 0           # map points without observations
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = vsa.synth_descriptors(int(rng.integers(1, 1 << 30)), n_groups)
    desc = np.repeat(base, sizes, axis=0)
    flips = rng.random((desc.shape[0], 256)) < 0.1
    desc ^= np.packbits(flips, axis=1, bitorder="little")
    return desc, ptr


@pytest.mark.parametrize("n_groups,max_n", [(1, 1), (1, 2), (300, 12), (50, 70), (3, 400)])
def test_distinctive_descriptors(vsa, oracle, matcher, n_groups, max_n):
    rng = np.random.default_rng(100 + n_groups + max_n)
    desc, ptr = clustered_descriptors(vsa, rng, n_groups, max_n)
    got = matcher.distinctive_descriptors(desc, ptr)
    want = np.array([oracle.distinctive_descriptor(desc[ptr[g]:ptr[g + 1]]) for g in range(n_groups)], np.int32)
    assert np.array_equal(got, want)


def test_distinctive_descriptors_ties(vsa, oracle, matcher):
    # identical observations: every median is 0, the first one wins; two observations: both medians are the zero on the diagonal
    one = vsa.synth_descriptors(5, 1)
    desc = np.concatenate([np.repeat(one, 5, axis=0), vsa.synth_descriptors(6, 2), one, ~one, one])
    ptr = np.array([0, 5, 7, 10], np.int32)
    got = matcher.distinctive_descriptors(desc, ptr)
    want = [oracle.distinctive_descriptor(desc[ptr[g]:ptr[g + 1]]) for g in range(3)]
    assert list(got) == want == [0, 0, 0]
    assert len(matcher.distinctive_descriptors(np.zeros((0, 32), np.uint8), np.array([0], np.int32))) == 0


def test_distinctive_descriptors_device_large(vsa, oracle, matcher):
    """100 000 map points in device memory; checked against the oracle on a sample and through a property that holds for
    every group: the winner's median is the minimum of its group's medians (recomputed with numpy)."""
    import torch
    rng = np.random.default_rng(8)
    n_groups = 100_000
    desc, ptr = clustered_descriptors(vsa, rng, n_groups, 14)
    d_desc = torch.from_numpy(desc).cuda(); d_ptr = torch.from_numpy(ptr).cuda()
    d_best = torch.empty(n_groups, dtype=torch.int32, device="cuda")
    matcher.distinctive_descriptors_device(d_desc.data_ptr(), d_ptr.data_ptr(), n_groups, desc.shape[0], d_best.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    best = d_best.cpu().numpy()
    sizes = np.diff(ptr)
    assert np.array_equal(best < 0, sizes == 0)
    assert np.all(best[sizes > 0] < sizes[sizes > 0])
    for g in rng.integers(0, n_groups, 400):
        assert best[g] == oracle.distinctive_descriptor(desc[ptr[g]:ptr[g + 1]])
    bits = np.unpackbits(desc, axis=1).astype(np.int16)
    for g in rng.integers(0, n_groups, 400):
        b = bits[ptr[g]:ptr[g + 1]]
        if len(b) == 0:
            continue
        D = (b[:, None, :] != b[None, :, :]).sum(axis=2)
        med = np.sort(D, axis=1)[:, (len(b) - 1) // 2]
        assert med[best[g]] == med.min() and best[g] == int(np.argmin(med))


def random_tree(rng, k, L, n_nodes_cap, p_leaf):
    """An unbalanced tree in file order (a node after its parent): every inner node gets 1..k children."""
    parent = [-1]; depth = [0]; leaf = [0]
    frontier = [0]
    while frontier:
        p = frontier.pop(0)
        nch = int(rng.integers(1, k + 1))
        for _ in range(nch):
            i = len(parent)
            parent.append(p); depth.append(depth[p] + 1)
            is_leaf = depth[i] >= L or rng.random() < p_leaf or len(parent) > n_nodes_cap
            leaf.append(1 if is_leaf else 0)
            if not is_leaf:
                frontier.append(i)
    n = len(parent)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = np.where(np.array(leaf) == 1, rng.random(n) * 3, 0.0)
    weight[(np.array(leaf) == 1) & (rng.random(n) < 0.1)] = 0.0             # stopped words
    return np.array(parent, np.int32), np.array(leaf, np.uint8), desc, weight


def check_transform(vsa, oracle, k, L, tree, feats, levelsup, scoring, weighting):
    parent, leaf, desc, weight = tree
    voc = vsa.Vocabulary(k, L, parent, leaf, desc, weight, scoring=scoring, weighting=weighting, device=0)
    try:
        assert voc.words() == int(leaf.sum())
        ids, vals, fn, fp, ff = voc.transform(feats, levelsup)
        word, node, oids, ovals, ofn, ofp, off = oracle.vocabulary_transform(k, L, parent, leaf, desc, weight, feats, levelsup, scoring, weighting)
        assert np.array_equal(ids, oids)
        assert np.array_equal(vals, ovals)                                  # doubles, same operations in the same order
        assert np.array_equal(fn, ofn) and np.array_equal(fp, ofp) and np.array_equal(ff, off)
        # the per-feature ids through the device-resident entry point
        import torch
        n = feats.shape[0]
        if n:
            d = torch.from_numpy(feats).cuda()
            w = torch.empty(n, dtype=torch.int32, device="cuda"); nd = torch.empty(n, dtype=torch.int32, device="cuda")
            voc.descend_device(d.data_ptr(), None, n, 1, levelsup, w.data_ptr(), nd.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(w.cpu().numpy(), word) and np.array_equal(nd.cpu().numpy(), node)
    finally:
        voc.close()


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (10, 4, 2), (3, 6, 4), (20, 2, 4), (2, 10, 4), (10, 3, 0)])
def test_vocabulary_transform_full_trees(vsa, oracle, k, L, levelsup):
    tree = vsa.synth_vocabulary(7 + k, k, L)
    rng = np.random.default_rng(k * 100 + L)
    leaves = np.flatnonzero(tree[1])
    feats = tree[2][rng.choice(leaves, 1500)] ^ np.packbits(rng.random((1500, 256)) < 0.05, axis=1)
    check_transform(vsa, oracle, k, L, tree, feats, levelsup, 0, 0)


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (0, 1), (1, 0), (0, 2), (2, 3), (5, 0), (5, 1), (5, 3)])
def test_vocabulary_transform_unbalanced_and_modes(vsa, oracle, scoring, weighting):
    """Unbalanced trees (leaves above the feature-vector level, 1..k children), stopped words, every weighting type and
    the normalising / non-normalising scorings; tied children (duplicate descriptors) exercise first-wins."""
    rng = np.random.default_rng(50 + scoring * 10 + weighting)
    tree = random_tree(rng, 17, 5, 4000, 0.3)
    parent, leaf, desc, weight = tree
    dup = rng.integers(2, len(parent), 300)
    desc[dup] = desc[dup - 1]                                                # often siblings: equal distances
    feats = np.concatenate([rng.integers(0, 256, (700, 32), dtype=np.uint8), desc[rng.integers(1, len(parent), 300)]])
    check_transform(vsa, oracle, 17, 5, tree, feats, 2, scoring, weighting)
    check_transform(vsa, oracle, 17, 5, tree, feats[:0], 2, scoring, weighting)


def test_vocabulary_rejects_malformed(vsa):
    parent, leaf, desc, weight = vsa.synth_vocabulary(1, 3, 2)
    bad = parent.copy(); bad[5] = 7
    with pytest.raises(vsa.OrbxError):
        vsa.Vocabulary(3, 2, bad, leaf, desc, weight, device=0)
    bad_leaf = leaf.copy(); bad_leaf[1] = 1
    with pytest.raises(vsa.OrbxError):
        vsa.Vocabulary(3, 2, parent, bad_leaf, desc, weight, device=0)


def test_vocabulary_descend_batch_layout(vsa, oracle):
    """The ORB vocabulary's shape (k = 10, L = 6: 1 111 111 nodes) on the output layout of a batched extraction: blocks
    of `cap` rows with a count per frame; rows past the count are marked -1."""
    import torch
    k, L = 10, 6
    tree = vsa.synth_vocabulary(3, k, L)
    parent, leaf, desc, weight = tree
    voc = vsa.Vocabulary(k, L, parent, leaf, desc, weight, device=0)
    rng = np.random.default_rng(4)
    frames, cap = 6, 500
    counts = np.array([500, 0, 137, 499, 1, 256], np.int32)
    leaves = np.flatnonzero(leaf)
    feats = desc[rng.choice(leaves, frames * cap)] ^ np.packbits(rng.random((frames * cap, 256)) < 0.04, axis=1)
    d = torch.from_numpy(feats).cuda(); dc = torch.from_numpy(counts).cuda()
    w = torch.empty(frames * cap, dtype=torch.int32, device="cuda"); nd = torch.empty_like(w)
    voc.descend_device(d.data_ptr(), dc.data_ptr(), cap, frames, 4, w.data_ptr(), nd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    w = w.cpu().numpy().reshape(frames, cap); nd = nd.cpu().numpy().reshape(frames, cap)
    for f in range(frames):
        c = counts[f]
        assert np.all(w[f, c:] == -1) and np.all(nd[f, c:] == -1)
        if c:
            word, node = oracle.vocabulary_transform(k, L, parent, leaf, desc, weight, feats[f * cap:f * cap + c], 4)[:2]
            assert np.array_equal(w[f, :c], word) and np.array_equal(nd[f, :c], node)
    voc.close()
