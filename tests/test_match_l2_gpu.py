"""The L2 mode (extension: mutual nearest neighbours under the squared Euclidean distance in the form
|a|^2 + |b|^2 - 2 a.b, with an optional second-best ratio test), for descriptors that are not unit-norm.
No counterpart in the reference: parity is with this repository's CPU restatement only.  Bar: bit-exact
pair lists and per-row / per-column bests."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu


def check(ctx, oracle, dA, pA, dB, pB, T, gate=1e8, ratio=0.0):
    exp, e1, e2 = oracle.match_l2(dA, pA, dB, pB, T, max_dist_sq=gate, ratio=ratio)
    counts = np.array([len(dA), len(dB)], np.int32)
    s = ctx.imageset(counts, np.concatenate([dA, dB]), np.concatenate([pA, pB]))
    res = s.match(np.array([[0, 1]], np.int32), threshold=T, max_dist_sq=gate, metric=capi.METRIC_L2, ratio=ratio)
    got = res.pairs()
    stats = res.stats_ex()
    res.close()
    s.close()
    assert np.array_equal(got, exp), (len(got), len(exp))
    if ratio == 0.0:  # the single-pair entry point also returns the per-row / per-column bests
        got2, g1, g2 = ctx.match_pair(dA, pA, dB, pB, threshold=T, max_dist_sq=gate, metric=capi.METRIC_L2, return_best=True)
        assert np.array_equal(got2, exp) and np.array_equal(g1, e1) and np.array_equal(g2, e2)
    return got, stats


def scaled_sift(n, seed, spread):
    """SIFT-like descriptors whose norms differ by up to `spread` (L2 order != cosine order)."""
    d, p, _ = synth.image_set(2, n, seed=seed)
    rng = np.random.default_rng(seed)
    scale = rng.uniform(1.0 - spread, 1.0, size=(2 * n, 1)).astype(np.float32)
    d = np.floor(d * scale).astype(np.float32)
    return d[:n], p[:n], d[n:], p[n:]


@pytest.mark.parametrize("n,spread", [(700, 0.0), (1500, 0.1), (900, 0.5)])
@pytest.mark.parametrize("ratio", [0.0, 0.8])
def test_l2_sift_like(ctx, oracle, n, spread, ratio):
    dA, pA, dB, pB = scaled_sift(n, 11, spread)
    got, _ = check(ctx, oracle, dA, pA, dB, pB, 250.0 ** 2, ratio=ratio)
    assert len(got) > 20
    check(ctx, oracle, dA, pA, dB, pB, 250.0 ** 2, gate=120.0 ** 2, ratio=ratio)


def test_l2_order_differs_from_cosine_order(ctx, oracle):
    """A scaled copy of a descriptor has cosine 1 but a large distance: the L2 mode must not match it."""
    dA, pA, dB, pB = scaled_sift(300, 3, 0.0)
    dB[7] = np.floor(dA[5] * 0.4)  # same direction, much shorter
    dB[9] = np.clip(dA[5] + np.random.default_rng(1).integers(-4, 5, 128), 0, 255)  # near in L2
    got, _ = check(ctx, oracle, dA, pA, dB, pB, 200.0 ** 2)
    assert [5, 9] in got.tolist()
    cos = ctx.match_pair(dA, pA, dB, pB, threshold=0.95)
    assert [5, 7] in cos.tolist()


@pytest.mark.parametrize("T", [4.0, 9.0, 1e9, 0.0, float("nan")])
def test_l2_general_floats(ctx, oracle, T):
    d, p, _ = synth.image_set(2, 600, seed=5, kind="gauss")
    check(ctx, oracle, d[:600], p[:600], d[600:], p[600:], T)
    if T == 9.0:
        check(ctx, oracle, d[:600], p[:600], d[600:], p[600:], T, ratio=0.7)


def test_l2_ties_zero_rows_and_ragged_sizes(ctx, oracle):
    dA, pA, dB, pB = scaled_sift(400, 17, 0.2)
    dA, pA = dA[:257], pA[:257]
    dA[10] = dA[40] = dA[7]
    dB[5] = dB[33] = dB[60] = dA[7]  # distance 0 three times: lowest index wins on both sides
    dA[3] = 0                        # a zero descriptor is at distance |b| from b: admissible under a large T
    dB[8] = 0
    got, stats = check(ctx, oracle, dA, pA, dB, pB, 300.0 ** 2)
    assert [7, 5] in got.tolist()
    check(ctx, oracle, dA, pA, dB, pB, 300.0 ** 2, ratio=0.9)
    e = np.zeros((0, 128), np.float32)
    assert len(check(ctx, oracle, e, np.zeros((0, 2)), dB, pB, 1e4)[0]) == 0
    assert len(check(ctx, oracle, dA[:1], pA[:1], dB[:1], pB[:1], 1e12)[0]) == 1


def test_l2_list_overflow_goes_cell_by_cell(ctx, oracle):
    dA, pA, dB, pB = scaled_sift(500, 23, 0.3)
    ctx.set_option("cand_capacity", 256)
    try:
        _, stats = check(ctx, oracle, dA, pA, dB, pB, 600.0 ** 2, ratio=0.85)
        assert stats[1] == 1  # scored cell by cell
    finally:
        ctx.set_option("cand_capacity", 524288)


def test_l2_benchmark_size(ctx, oracle):
    d, p, _ = synth.image_set(2, 8192, seed=2026)
    got, stats = check(ctx, oracle, d[:8192], p[:8192], d[8192:], p[8192:], 160.0 ** 2, ratio=0.8)
    assert len(got) > 2000 and stats[1] == 0
