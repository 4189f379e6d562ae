"""Lowe's second-best ratio test on top of the reference's matcher (sfm_match_batch_ratio).  The
reference has no such test, so the only checker is this repository's CPU restatement (parity unpinned by
the reference); with the test off the call must reproduce matchTwoSided exactly.  Bar: bit-exact match
lists."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu


def crowded_pair(seed, n=700, dup=0.35, noise=10.0, integer=True):
    """Two views of n features where a share of the descriptors have a near twin in the other image, so
    that the second best comes close to the best for many rows."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 120, size=(n, 128)).astype(np.float32)
    twin = rng.random(n) < dup
    src = rng.integers(0, n, size=n)
    base[twin] = base[src[twin]] + rng.normal(0, noise * 0.5, size=(int(twin.sum()), 128)).astype(np.float32)
    A = base + rng.normal(0, noise, size=base.shape).astype(np.float32)
    perm = rng.permutation(n)
    B = base[perm] + rng.normal(0, noise, size=base.shape).astype(np.float32)
    if integer:
        A, B = np.clip(np.rint(A), 0, 255), np.clip(np.rint(B), 0, 255)
    else:
        A, B = np.abs(A) + 0.25, np.abs(B) + 0.25
    pA = rng.uniform(0, 1000, size=(n, 2))
    pB = pA[perm] + rng.normal(0, 3, size=(n, 2))
    return A.astype(np.float32), pA, B.astype(np.float32), pB


def run_pairs(ctx, views, pl, **kw):
    counts = np.array([len(v[1]) for v in views], np.int32)
    desc = np.concatenate([v[0] for v in views])
    pts = np.concatenate([v[1] for v in views])
    s = ctx.imageset(counts, desc, pts)
    res = s.match(pl, **kw)
    off, pairs = res.offsets(), res.pairs()
    stats = res.stats_ex()
    res.close()
    s.close()
    return [pairs[off[k]:off[k + 1]] for k in range(len(pl))], stats


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("ratio", [0.6, 0.8, 0.95, 1.0])
def test_ratio_matches_restatement(ctx, oracle, ratio, integer):
    A, pA, B, pB = crowded_pair(11, integer=integer)
    C, pC, D, pD = crowded_pair(12, n=450, integer=integer)
    views = [(A, pA), (B, pB), (C, pC), (D, pD)]
    pl = np.array([[0, 1], [2, 3], [1, 0], [0, 3]], np.int32)
    got, _ = run_pairs(ctx, views, pl, threshold=0.9, max_dist_sq=400.0 ** 2, ratio=ratio)
    plain, _ = run_pairs(ctx, views, pl, threshold=0.9, max_dist_sq=400.0 ** 2)
    dropped = 0
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match(views[a][0], views[a][1], views[b][0], views[b][1], threshold=0.9,
                                 max_dist_sq=400.0 ** 2, ratio=ratio)
        assert np.array_equal(got[k], exp), f"pair {k}"
        kept = {tuple(r) for r in got[k]}
        assert kept <= {tuple(r) for r in plain[k]}
        dropped += len(plain[k]) - len(got[k])
    assert sum(len(g) for g in got) > 100
    if ratio < 0.9:
        assert dropped > 20, "the test never bites: the case tests nothing"


def test_ratio_with_loose_threshold_takes_the_row_walk(ctx, oracle):
    """A threshold so low that the bound pass floods: the pair goes through the filtered passes, whose
    lists hold possible bests only, and the ratio test walks the rows instead."""
    A, pA, B, pB = crowded_pair(21, n=600)
    ctx.set_option("cand_capacity", 4096)
    try:
        got, stats = run_pairs(ctx, [(A, pA), (B, pB)], np.array([[0, 1]], np.int32), threshold=0.3,
                               max_dist_sq=1e8, ratio=0.8)
    finally:
        ctx.set_option("cand_capacity", 524288)
    exp, _, _ = oracle.match(A, pA, B, pB, threshold=0.3, max_dist_sq=1e8, ratio=0.8)
    assert stats[2] + stats[1] >= 1, "the pair was expected to overflow its candidate list"
    assert np.array_equal(got[0], exp) and len(exp) > 50


def test_ratio_duplicates_and_edges(ctx, oracle):
    """Identical descriptors in B: the second best equals the best, so the row fails for every ratio < 1
    and also for ratio = 1 (strict inequality); empty images; a gate that hides the twin."""
    rng = np.random.default_rng(5)
    A = rng.integers(1, 200, size=(40, 128)).astype(np.float32)
    B = np.concatenate([A, A[:10]])  # rows 0..9 of A have two identical partners
    pA = rng.uniform(0, 100, size=(40, 2))
    pB = np.concatenate([pA, pA[:10] + 500.0])
    for gate, want_first_ten in ((1e8, False), (50.0 ** 2, True)):
        got, _ = run_pairs(ctx, [(A, pA), (B, pB)], np.array([[0, 1]], np.int32), threshold=0.9, max_dist_sq=gate,
                           ratio=1.0)
        exp, _, _ = oracle.match(A, pA, B, pB, threshold=0.9, max_dist_sq=gate, ratio=1.0)
        assert np.array_equal(got[0], exp)
        assert (set(range(10)) <= set(got[0][:, 0])) == want_first_ten
    E = np.zeros((0, 128), np.float32)
    got, _ = run_pairs(ctx, [(A, pA), (E, np.zeros((0, 2)))], np.array([[0, 1], [1, 0]], np.int32), threshold=0.9,
                       max_dist_sq=1e8, ratio=0.8)
    assert len(got[0]) == 0 and len(got[1]) == 0
    s = ctx.imageset(np.array([40], np.int32), A, pA)
    with pytest.raises(capi.SfmError, match="ratio"):
        s.match(np.array([[0, 0]], np.int32), ratio=1.5)
    with pytest.raises(capi.SfmError, match="cosine"):
        s.match(np.array([[0, 0]], np.int32), ratio=0.8, metric=capi.METRIC_DOT)
    s.close()


def test_ratio_full_size(ctx, oracle):
    """8192 x 8192 at the benchmark's threshold, one pair against the restatement."""
    desc, pts, counts, _ = synth.frame_sequence(2, 8192, seed=77)
    s = ctx.imageset(counts, desc, pts)
    res = s.match(np.array([[0, 1]], np.int32), threshold=0.95, max_dist_sq=1e4, ratio=0.8)
    got = res.pairs()
    exp, _, _ = oracle.match(desc[:8192], pts[:8192], desc[8192:], pts[8192:], threshold=0.95, max_dist_sq=1e4, ratio=0.8)
    assert np.array_equal(got, exp) and len(exp) > 2000
    res.close()
    s.close()
