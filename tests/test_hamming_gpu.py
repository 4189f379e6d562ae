"""Binary descriptors under the Hamming distance (sfm_binset_*, sfm_match_batch_hamming).  The
reference has no binary descriptors, so the only checker is this repository's CPU restatement (parity
unpinned by the reference).  Integer work: the bar is bit-exact match lists."""
import numpy as np
import pytest

from sfm_amd import capi

pytestmark = pytest.mark.gpu


def binary_views(seed, n_images, n, nbytes, flip=0.06, shared=0.6, dup=0.05):
    """Images share a pool of random bit strings (each copy with a few bits flipped); some descriptors
    occur twice inside an image, so equal distances and lowest-index ties happen."""
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 256, size=(int(n / shared) + 8, nbytes), dtype=np.uint8)
    pool_pos = rng.uniform(0, 1000, size=(len(pool), 2))
    descs, ptss, counts = [], [], []
    for _ in range(n_images):
        m = int(n) if np.isscalar(n) else int(n)
        pick = rng.choice(len(pool), size=m, replace=False)
        d = pool[pick].copy()
        noise = rng.random((m, nbytes * 8)) < flip
        d ^= np.packbits(noise, axis=1)
        twins = rng.random(m) < dup
        d[twins] = d[rng.integers(0, m, size=int(twins.sum()))]
        descs.append(d)
        ptss.append(pool_pos[pick] + rng.normal(0, 2, size=(m, 2)))
        counts.append(m)
    return descs, ptss, np.array(counts, np.int32)


def run(ctx, descs, ptss, counts, pl, **kw):
    s = capi.BinarySet(ctx, counts, np.concatenate(descs) if len(descs) else np.zeros((0, 32), np.uint8),
                       np.concatenate(ptss) if len(ptss) else np.zeros((0, 2)))
    res = s.match(pl, **kw)
    off, pairs = res.offsets(), res.pairs()
    res.close()
    s.close()
    return [pairs[off[k]:off[k + 1]] for k in range(len(pl))]


@pytest.mark.parametrize("nbytes", [16, 32, 64])
@pytest.mark.parametrize("max_distance,gate", [(40, 1e8), (60, 60.0 ** 2), (10, 1e8)])
def test_hamming_matches_restatement(ctx, oracle, nbytes, max_distance, gate):
    descs, ptss, counts = binary_views(3 + nbytes, 4, 600, nbytes)
    descs[2], ptss[2], counts[2] = descs[2][:333], ptss[2][:333], 333  # ragged sizes
    thr = max_distance * nbytes // 32
    pl = np.array([[0, 1], [1, 2], [2, 0], [3, 3], [0, 3]], np.int32)
    got = run(ctx, descs, ptss, counts, pl, max_distance=thr, max_dist_sq=gate)
    total = 0
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match_hamming(descs[a], ptss[a], descs[b], ptss[b], max_distance=thr, max_dist_sq=gate)
        assert np.array_equal(got[k], exp), f"pair {k}"
        total += len(exp)
    assert total > 300


def test_hamming_everything_admissible(ctx, oracle):
    """max_distance = all bits: every cell goes through the slow branch; ties everywhere."""
    rng = np.random.default_rng(9)
    A = rng.integers(0, 4, size=(300, 32), dtype=np.uint8)  # few distinct bits: many equal distances
    B = rng.integers(0, 4, size=(257, 32), dtype=np.uint8)
    pA, pB = rng.uniform(0, 50, size=(300, 2)), rng.uniform(0, 50, size=(257, 2))
    got = run(ctx, [A, B], [pA, pB], np.array([300, 257], np.int32), np.array([[0, 1], [1, 0]], np.int32),
              max_distance=256, max_dist_sq=30.0 ** 2)
    for k, (x, px, y, py) in enumerate([(A, pA, B, pB), (B, pB, A, pA)]):
        exp, _, _ = oracle.match_hamming(x, px, y, py, max_distance=256, max_dist_sq=30.0 ** 2)
        assert np.array_equal(got[k], exp) and len(exp) > 20


def test_hamming_edges(ctx, oracle):
    rng = np.random.default_rng(1)
    A = rng.integers(0, 256, size=(5, 32), dtype=np.uint8)
    pA = np.zeros((5, 2))
    E = np.zeros((0, 32), np.uint8)
    got = run(ctx, [A, E, A], [pA, np.zeros((0, 2)), pA], np.array([5, 0, 5], np.int32),
              np.array([[0, 1], [1, 0], [1, 1], [0, 2]], np.int32), max_distance=0, max_dist_sq=1.0)
    assert [len(g) for g in got[:3]] == [0, 0, 0]
    assert got[3].tolist() == [[i, i] for i in range(5)]  # identical sets at distance 0
    with pytest.raises(capi.SfmError, match="bytes per descriptor"):
        capi.BinarySet(ctx, np.array([2], np.int32), np.zeros((2, 24), np.uint8), np.zeros((2, 2)))
    s = capi.BinarySet(ctx, np.array([5], np.int32), A, pA)
    with pytest.raises(capi.SfmError, match="outside the set"):
        s.match(np.array([[0, 1]], np.int32))
    with pytest.raises(capi.SfmError, match="max_distance"):
        s.match(np.array([[0, 0]], np.int32), max_distance=-1)
    s.close()


def test_hamming_full_size_and_chunks(ctx, oracle):
    """8192 x 8192 x 256 bits against the restatement, and a pair list cut into launches of two."""
    descs, ptss, counts = binary_views(77, 3, 8192, 32, dup=0.0)
    pl = np.array([[0, 1], [1, 2], [0, 2]], np.int32)
    ctx.set_option("pairs_per_launch", 2)
    try:
        got = run(ctx, descs, ptss, counts, pl, max_distance=64, max_dist_sq=1e8)
    finally:
        ctx.set_option("pairs_per_launch", 2048)
    exp, _, _ = oracle.match_hamming(descs[0], ptss[0], descs[1], ptss[1], max_distance=64, max_dist_sq=1e8)
    assert np.array_equal(got[0], exp) and len(exp) > 2000
    exp, _, _ = oracle.match_hamming(descs[0], ptss[0], descs[2], ptss[2], max_distance=64, max_dist_sq=1e8)
    assert np.array_equal(got[2], exp)


@pytest.mark.parametrize("mfma", [0, 1])
def test_hamming_both_paths_agree(ctx, oracle, mfma):
    """256-bit descriptors go through the matrix cores (signs, int8 sums = 256 - 2 d) unless turned off;
    either way the lists are the restatement's."""
    descs, ptss, counts = binary_views(91, 3, 1500, 32)
    pl = np.array([[0, 1], [1, 2], [2, 0]], np.int32)
    ctx.set_option("hamming_mfma", mfma)
    try:
        got = run(ctx, descs, ptss, counts, pl, max_distance=50, max_dist_sq=200.0 ** 2)
    finally:
        ctx.set_option("hamming_mfma", 1)
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match_hamming(descs[a], ptss[a], descs[b], ptss[b], max_distance=50, max_dist_sq=200.0 ** 2)
        assert np.array_equal(got[k], exp) and len(exp) > 300


def test_hamming_overflowing_list_falls_back(ctx, oracle):
    """A candidate list too small for the cells within max_distance: the pair is redone by the popcount
    kernel; its neighbour in the batch is not."""
    descs, ptss, counts = binary_views(92, 3, 1200, 32, flip=0.02)
    descs[1][:] = descs[1][0]  # image 1: one descriptor 1200 times -> every cell of (1, 1) is within reach
    pl = np.array([[1, 1], [0, 2]], np.int32)
    ctx.set_option("cand_capacity", 2048)
    try:
        got = run(ctx, descs, ptss, counts, pl, max_distance=40, max_dist_sq=1e8)
    finally:
        ctx.set_option("cand_capacity", 524288)
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match_hamming(descs[a], ptss[a], descs[b], ptss[b], max_distance=40, max_dist_sq=1e8)
        assert np.array_equal(got[k], exp), k
    assert len(got[0]) == 1 and len(got[1]) > 300
