"""Host-side logic that needs no GPU: option structs mirror the reference's defaults, the plug-in
points reject user functors, pair-list partitioning, synthetic generators."""
import numpy as np
import pytest

import sfm_amd
from sfm_amd import api, shard, synth


def test_option_defaults_mirror_the_reference():
    m = sfm_amd.MATCHING_OPTIONS()        # match/match.hpp:55-61
    assert (m.algorithm, m.threshold, m.maxDistancePxSquared) == (0, 0.5, 1e8)
    assert m.similarityMetric is sfm_amd.cosineSimilarity
    r = sfm_amd.RANSAC_OPTIONS()          # solve/solve.hpp:108-125
    assert (r.minN, r.maxIter, r.maxTimeMs, r.maxSquaredError, r.successProb) == (8, 512, 1000, 10.0, 0.99)
    assert r.model is api.eightPointAlgorithmFromSubset and r.loss is api.defaultLoss


def test_unknown_algorithm_raises_like_the_reference():
    with pytest.raises(ValueError, match="Unknown matching algorithm"):   # match.cpp:18
        sfm_amd.match([], [], sfm_amd.MATCHING_OPTIONS(algorithm=3))


def test_user_functors_have_no_kernel():
    with pytest.raises(NotImplementedError):
        sfm_amd.match([], [], sfm_amd.MATCHING_OPTIONS(similarityMetric=lambda a, b: 0.0))
    x = np.zeros((9, 2))
    with pytest.raises(NotImplementedError):
        sfm_amd.RANSAC(x, x, np.eye(3), sfm_amd.RANSAC_OPTIONS(model=lambda *a: None))
    with pytest.raises(NotImplementedError):
        sfm_amd.RANSAC(x, x, np.eye(3), sfm_amd.RANSAC_OPTIONS(loss=lambda *a: 0.0))


def test_ransac_guard_returns_empty():
    x = np.zeros((7, 2))
    assert sfm_amd.RANSAC(x, x, np.eye(3)) == []                  # RANSAC.cpp:148-149
    assert sfm_amd.RANSAC(np.zeros((9, 2)), np.zeros((8, 2)), np.eye(3)) == []


def test_safe_cast_to_int():
    assert api.safeCastToInt(float("nan")) == 0
    assert api.safeCastToInt(1e300) == 2147483647 and api.safeCastToInt(-1e300) == -2147483648
    assert api.safeCastToInt(12.9) == 12 and api.safeCastToInt(float("inf")) == 2147483647


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (63, 2), (64, 2), (1000, 2), (19900, 8), (130, 4)])
def test_partition_is_a_partition(n, world):
    parts = [shard.partition_pairs(n, r, world) for r in range(world)]
    allidx = np.sort(np.concatenate(parts)) if n else np.zeros(0, np.int64)
    assert np.array_equal(allidx, np.arange(n))
    sizes = [len(p) for p in parts]
    assert max(sizes) - min(sizes) <= 64
    for p in parts:
        assert np.all(np.diff(p) > 0)


def test_images_for_pairs():
    for n_pairs in (1, 3, 19900, 39800, 79600, 159200):
        n = shard.images_for_pairs(n_pairs)
        assert n * (n - 1) // 2 >= n_pairs > (n - 1) * (n - 2) // 2
    assert shard.images_for_pairs(19900) == 200


def test_exhaustive_pairs():
    p = synth.exhaustive_pairs(200)
    assert p.shape == (19900, 2) and np.all(p[:, 0] < p[:, 1])
    assert len({tuple(r) for r in p}) == 19900


def test_sample_table_rows_are_distinct_and_in_range():
    for n in (8, 20, 300, 20000):
        S = synth.sample_table(n, 500, seed=n)
        assert S.shape == (500, 8) and S.min() >= 0 and S.max() < n
        assert all(len(set(r)) == 8 for r in S)
    assert np.array_equal(synth.sample_table(1000, 10, seed=5), synth.sample_table(1000, 10, seed=5))


def test_sift_like_descriptors_are_exact_in_fp16_mfma():
    d = synth.sift_like(np.random.default_rng(0), 1000)
    assert d.dtype == np.float32 and np.array_equal(d, np.rint(d)) and d.min() >= 0 and d.max() <= 255
    assert (d.astype(np.float64) ** 2).sum(axis=1).max() < 2 ** 24


def test_two_view_geometry_is_consistent():
    tv = synth.two_view(500, inlier_ratio=0.4, seed=3, noise_px=0.0)
    x = np.concatenate([tv["x"], np.ones((500, 1))], axis=1)
    y = np.concatenate([tv["y"], np.ones((500, 1))], axis=1)
    T = tv["T"]
    E = np.array([[0, -T[2], T[1]], [T[2], 0, -T[0]], [-T[1], T[0], 0]]) @ tv["R"]
    res = np.abs(np.einsum("ni,ij,nj->n", y, E, x))
    assert res[tv["inlier_mask"]].max() < 1e-9 and np.median(res[~tv["inlier_mask"]]) > 1e-3
    assert tv["inlier_mask"].sum() == 200


def _int8_bound_holds(A, B, rng=None):
    """Emulates the ingest of the int8 table (per-image scale from the largest unit-normalised element,
    round to nearest) and checks the error bound the matcher's integer threshold relies on:
    |cos * sA * sB - S| <= 0.5001 (sA L1_A + sB L1_B) + 0.7505 D for every cell."""
    def ingest(X):
        n = np.sqrt((X.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
        hat = (X.astype(np.float32) / n[:, None]).astype(np.float32)
        scale = np.float32(127.0) / np.abs(hat).max()
        q = np.rint((hat * scale).astype(np.float32)).clip(-127, 127).astype(np.int64)
        return hat.astype(np.float64), float(scale), q, float(np.abs(hat).sum(axis=1).max() * 1.0001)
    ha, sa, qa, l1a = ingest(A)
    hb, sb, qb, l1b = ingest(B)
    S = qa @ qb.T
    cos = ha @ hb.T
    err = 0.5001 * (sa * l1a + sb * l1b) + 0.7505 * A.shape[1]
    worst = np.abs(cos * sa * sb - S).max()
    assert worst <= err, (worst, err)
    return worst / err, err / (sa * sb)


def test_int8_bound_of_the_matcher_is_rigorous():
    from sfm_amd import synth
    rng = np.random.default_rng(3)
    sift = synth.sift_like(rng, 600)
    frac, margin = _int8_bound_holds(sift[:300], sift[300:])
    assert margin < 0.05  # a few 1e-2 in cosine units on SIFT-like data
    _int8_bound_holds(rng.standard_normal((300, 128)).astype(np.float32), rng.standard_normal((300, 128)).astype(np.float32))
    # adversarial: every element sits just off a rounding boundary, all errors aligned
    base = (np.arange(1, 129) % 7 + 0.5).astype(np.float32)
    A = np.tile(base, (4, 1)) * np.array([[1.0], [1.0001], [0.9999], [1.00005]], np.float32)
    _int8_bound_holds(A, A[::-1].copy())
    # sparse and spiky descriptors (one dominant element): the scale is small, the bound must still hold
    spiky = rng.random((200, 128)).astype(np.float32) * 0.01
    spiky[np.arange(200), rng.integers(0, 128, 200)] = 1.0
    _int8_bound_holds(spiky[:100], spiky[100:])
    _int8_bound_holds(sift[:100], rng.standard_normal((100, 128)).astype(np.float32))


# ---- the library's static partition and gather layout (host-only entry points) ---------------------
@pytest.mark.parametrize("n_pairs", [0, 1, 63, 64, 65, 1000, 19900, 499500])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_owns_every_pair_once(n_pairs, world):
    """sfm_shard_indices / sfm_shard_count: every pair has exactly one owner, shares differ by at most one
    block of 64, owned indices ascend, consecutive pairs of a block stay together."""
    from sfm_amd import capi
    owned = [capi.shard_indices(n_pairs, r, world) for r in range(world)]
    assert [len(o) for o in owned] == [capi.shard_count(n_pairs, r, world) for r in range(world)]
    allidx = np.concatenate(owned) if n_pairs else np.zeros(0, np.int64)
    assert np.array_equal(np.sort(allidx), np.arange(n_pairs))
    sizes = [len(o) for o in owned]
    assert max(sizes) - min(sizes) <= 64
    for r, o in enumerate(owned):
        assert np.all(np.diff(o) > 0)
        assert np.all((o // 64) % world == r)


def test_gather_layout_against_numpy():
    """sfm_shard_layout: position of every pair's list in the rank-major buffer, against the same rule
    written in numpy (rank 0's lists, then rank 1's, ..., each in the order of the rank's pairs)."""
    from sfm_amd import capi
    rng = np.random.default_rng(5)
    for n_pairs, world in [(1000, 8), (64 * 8 + 3, 8), (130, 2), (5, 4), (7, 1)]:
        owned = [capi.shard_indices(n_pairs, r, world) for r in range(world)]
        max_owned = max(max(len(o) for o in owned), 1)
        counts = np.zeros((world, max_owned), np.int32)
        for r, o in enumerate(owned):
            counts[r, :len(o)] = rng.integers(0, 50, len(o))
        starts, cnt = capi.shard_layout(n_pairs, world, counts)
        run = 0
        for r, o in enumerate(owned):
            for k, i in enumerate(o):
                assert starts[i] == run and cnt[i] == counts[r, k]
                run += counts[r, k]


@pytest.mark.parametrize("n_pairs,world", [(1000, 8), (64 * 8 + 3, 8), (130, 2), (5, 4), (7, 1), (0, 3), (19900, 8)])
def test_ransac_gather_unpack_against_numpy(n_pairs, world):
    """sfm_shard_unpack_ransac: what the root does with the buffers of a RANSAC gather -- per rank the winners'
    records (padded to the largest share) and, rank after rank, the inlier lists -- against per-pair results
    dealt to the ranks in numpy and packed the way the ranks send them."""
    from sfm_amd import capi
    rng = np.random.default_rng(n_pairs + world)
    counts_true = rng.integers(0, 40, n_pairs).astype(np.int32)
    counts_true[rng.random(n_pairs) < 0.2] = 0
    rec_true = np.zeros(n_pairs, capi.RANSAC_RECORD)
    rec_true["best_h"] = np.where(counts_true > 0, rng.integers(0, 4096, n_pairs), -1)
    rec_true["count"] = counts_true
    rec_true["iterations"] = rng.integers(1, 4097, n_pairs)
    rec_true["toterr"] = rng.random(n_pairs)
    rec_true["pose12"] = rng.standard_normal((n_pairs, 12))
    lists_true = [np.sort(rng.choice(20000, c, replace=False)).astype(np.int32) for c in counts_true]
    owned = [capi.shard_indices(n_pairs, r, world) for r in range(world)]
    max_owned = max(max(len(o) for o in owned), 1)
    counts = np.zeros((world, max_owned), np.int32)
    records = np.zeros((world, max_owned), capi.RANSAC_RECORD)
    payload = []
    for r, o in enumerate(owned):
        counts[r, :len(o)] = counts_true[o]
        records[r, :len(o)] = rec_true[o]
        payload += [lists_true[i] for i in o]
    payload = np.concatenate(payload) if payload else np.zeros(0, np.int32)
    got = capi.shard_unpack_ransac(n_pairs, world, counts, records, payload)
    assert np.array_equal(got["best_h"], rec_true["best_h"]) and np.array_equal(got["count"], counts_true)
    assert np.array_equal(got["iterations"], rec_true["iterations"]) and np.array_equal(got["toterr"], rec_true["toterr"])
    assert np.array_equal(got["pose12"], rec_true["pose12"])
    assert np.array_equal(got["inlier_offsets"], np.concatenate([[0], np.cumsum(counts_true)]))
    for i in range(n_pairs):
        assert np.array_equal(got["inliers"][got["inlier_offsets"][i]:got["inlier_offsets"][i + 1]], lists_true[i])
