"""Parity of the HIP matcher (through the C-ABI) with the CPU restatement of
matchTwoSided (reference submodules/match/match.cpp:24-94).  Bar: bit-exact pair lists
and bit-exact matches1 / matches2 arrays, for every input class below."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu

DEFAULT_CAND_CAPACITY = 524288


def check_pair(ctx, oracle, dA, pA, dB, pB, thr, gate=1e8, metric=0):
    got, g1, g2 = ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric, return_best=True)
    exp, e1, e2 = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric)
    assert np.array_equal(g1, e1), "matches1 differs"
    assert np.array_equal(g2, e2), "matches2 differs"
    assert np.array_equal(got, exp), "pair list differs"
    return got


def two_images(nA, nB, kind="sift", seed=3, dim=128, shared=0.3):
    n = max(nA, nB)
    d, p, _ = synth.image_set(2, n, dim=dim, shared_frac=shared, seed=seed, kind=kind)
    return d[:nA].copy(), p[:nA].copy(), d[n:n + nB].copy(), p[n:n + nB].copy()


@pytest.mark.parametrize("nA,nB", [(257, 1000), (1000, 257), (2000, 2000), (1, 1), (1, 700), (513, 129)])
@pytest.mark.parametrize("gate", [1e8, 100.0 * 100.0])
def test_sift_like(ctx, oracle, nA, nB, gate):
    dA, pA, dB, pB = two_images(nA, nB)
    got = check_pair(ctx, oracle, dA, pA, dB, pB, 0.95, gate)
    if min(nA, nB) >= 257:
        assert len(got) > 10


@pytest.mark.parametrize("thr", [0.5, 0.95, 0.0, -1.0])
def test_general_float(ctx, oracle, thr):
    dA, pA, dB, pB = two_images(700, 900, kind="gauss", seed=5)
    check_pair(ctx, oracle, dA, pA, dB, pB, thr)


def test_default_options_all_cells_pass(ctx, oracle):
    # MATCHING_OPTIONS defaults (threshold 0.5) on SIFT-like data: most cells clear the
    # threshold, so the arg-max decides everything.
    dA, pA, dB, pB = two_images(600, 640, shared=0.5)
    dA = np.maximum(dA, 20)  # raise the floor: cosine of unrelated descriptors ~0.8
    dB = np.maximum(dB, 20)
    check_pair(ctx, oracle, dA, pA, dB, pB, 0.5)


def test_candidate_list_overflow(ctx, oracle):
    dA, pA, dB, pB = two_images(500, 500, kind="gauss", seed=9)
    ctx.set_option("cand_capacity", 64)
    try:
        check_pair(ctx, oracle, dA, pA, dB, pB, -1.0)
        check_pair(ctx, oracle, dA, pA, dB, pB, 0.05)
        ctx.set_option("cand_capacity", 0)
        check_pair(ctx, oracle, dA, pA, dB, pB, 0.2)
    finally:
        ctx.set_option("cand_capacity", DEFAULT_CAND_CAPACITY)


def test_ties_pick_lowest_index(ctx, oracle):
    dA, pA, dB, pB = two_images(300, 300)
    dA[10] = dA[200] = dA[7]          # three identical rows in A
    dB[5] = dB[99] = dB[250] = dA[7]  # and three copies in B
    dB[17] = dB[18] = dA[40]
    got = check_pair(ctx, oracle, dA, pA, dB, pB, 0.9)
    assert (7, 5) in [tuple(r) for r in got]


def test_zero_descriptor_never_matches(ctx, oracle):
    dA, pA, dB, pB = two_images(200, 200)
    dA[3] = 0
    dB[8] = 0
    for thr in (0.95, 0.0, -1.0):
        got = check_pair(ctx, oracle, dA, pA, dB, pB, thr)
        assert 3 not in got[:, 0] and 8 not in got[:, 1]


def test_threshold_edge_is_strict(ctx, oracle):
    dA, pA, dB, pB = two_images(64, 64, shared=1.0)
    s = np.float32(oracle.similarity(dA[0], dB[np.argmax([oracle.similarity(dA[0], b) for b in dB])]))
    check_pair(ctx, oracle, dA, pA, dB, pB, float(s))                       # sim == thr must fail
    check_pair(ctx, oracle, dA, pA, dB, pB, float(np.nextafter(s, np.float32(-1))))


def test_pixel_gate_edge_is_strict(ctx, oracle):
    dA, pA, dB, pB = two_images(50, 50, shared=1.0)
    pA[:] = 0.0
    pB[:] = 0.0
    pB[:, 0] = 3.0
    pB[:, 1] = 4.0  # squared distance exactly 25
    assert len(check_pair(ctx, oracle, dA, pA, dB, pB, 0.9, gate=25.0)) == 0
    assert len(check_pair(ctx, oracle, dA, pA, dB, pB, 0.9, gate=25.000002)) > 0


def test_empty_inputs(ctx, oracle):
    dA, pA, dB, pB = two_images(100, 100)
    e = np.zeros((0, 128), np.float32)
    ep = np.zeros((0, 2))
    assert len(ctx.match_pair(e, ep, dB, pB)) == 0
    assert len(ctx.match_pair(dA, pA, e, ep)) == 0
    assert len(ctx.match_pair(e, ep, e, ep)) == 0


@pytest.mark.parametrize("dim", [64, 100, 3])
def test_other_descriptor_lengths(ctx, oracle, dim):
    dA, pA, dB, pB = two_images(300, 400, dim=dim, kind="gauss", seed=21)
    check_pair(ctx, oracle, dA, pA, dB, pB, 0.8)
    check_pair(ctx, oracle, dA, pA, dB, pB, 0.0)


@pytest.mark.parametrize("thr", [0.0, 50.0, -3.0])
def test_dot_metric(ctx, oracle, thr):
    dA, pA, dB, pB = two_images(300, 280, kind="gauss", seed=8)
    check_pair(ctx, oracle, dA, pA, dB, pB, thr, metric=capi.METRIC_DOT)


def test_dot_metric_sift(ctx, oracle):
    dA, pA, dB, pB = two_images(300, 280)
    check_pair(ctx, oracle, dA, pA, dB, pB, 250000.0, metric=capi.METRIC_DOT)


def test_extreme_scales(ctx, oracle):
    dA, pA, dB, pB = two_images(200, 220, kind="gauss", seed=12)
    dA[:50] *= 1e-18   # squares underflow towards denormals
    dB[:50] *= 1e15    # squares near overflow
    dA[60] *= 1e25     # norm overflows to inf
    dB[70, 0] = np.inf
    dB[71, 5] = np.nan
    for metric in (capi.METRIC_COSINE, capi.METRIC_DOT):
        check_pair(ctx, oracle, dA, pA, dB, pB, 0.3, metric=metric)


def test_unknown_algorithm_is_rejected(ctx):
    dA, pA, dB, pB = two_images(10, 10)
    with pytest.raises(capi.SfmError) as e:
        ctx.match_pair(dA, pA, dB, pB, algorithm=7)
    assert e.value.code == capi.ERR_INVALID_ARG


def test_descriptor_too_long_is_rejected(ctx):
    with pytest.raises(capi.SfmError) as e:
        ctx.match_pair(np.ones((4, 256), np.float32), np.zeros((4, 2)), np.ones((4, 256), np.float32), np.zeros((4, 2)))
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_batch_ragged_exhaustive(ctx, oracle):
    counts = np.array([300, 0, 513, 1, 1000, 777], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=31)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    starts = np.concatenate([[0], np.cumsum(counts)])
    pl = synth.exhaustive_pairs(len(counts))
    pl = np.concatenate([pl, pl[:, ::-1], [[2, 2]]]).astype(np.int32)
    ctx.set_option("pairs_per_launch", 7)  # several chunks
    try:
        s = ctx.imageset(counts, desc, pts)
        res = s.match(pl, threshold=0.95, max_dist_sq=100.0 * 100.0)
        off, pairs = res.offsets(), res.pairs()
        res2 = s.match(pl, threshold=0.95, max_dist_sq=100.0 * 100.0, result=res)  # buffer reuse
        assert np.array_equal(off, res2.offsets()) and np.array_equal(pairs, res2.pairs())
    finally:
        ctx.set_option("pairs_per_launch", 2048)
    assert off[0] == 0 and len(off) == len(pl) + 1 and off[-1] == len(pairs)
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]],
                                 desc[starts[b]:starts[b + 1]], pts[starts[b]:starts[b + 1]],
                                 threshold=0.95, max_dist_sq=100.0 * 100.0)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"pair {k} ({a},{b})"
    s.close()


@pytest.mark.parametrize("metric,threshold,chunk", [(0, 0.95, 1), (0, 0.95, 3), (0, 0.95, 11), (0, 0.3, 2),
                                                    (1, 0.0, 2)])
def test_chunks_on_two_streams_give_the_same_lists(ctx, oracle, metric, threshold, chunk):
    """A batch of several chunks runs what follows a chunk's bound passes on a second stream while the next
    chunk starts (up to three sets of per-chunk buffers: 21 pairs in chunks of 11 take two, the others three).  Lists with and without that overlap, and across repeated
    calls into one result object, are identical, and equal the CPU restatement's."""
    counts = np.array([640, 257, 0, 900, 64, 511, 333], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=131)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    starts = np.concatenate([[0], np.cumsum(counts)])
    pl = synth.exhaustive_pairs(len(counts)).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    got = {}
    ctx.set_option("pairs_per_launch", chunk)
    try:
        for ov in (1, 0):
            ctx.set_option("match_overlap", ov)
            res = None
            for _ in range(3):  # reuse: the second and third call find both streams and both buffer sets used
                res = s.match(pl, metric=metric, threshold=threshold, max_dist_sq=150.0 * 150.0, result=res)
            got[ov] = (res.offsets().copy(), res.pairs().copy())
            res.close()
    finally:
        ctx.set_option("pairs_per_launch", 2048)
        ctx.set_option("match_overlap", 1)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    off, pairs = got[1]
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]],
                                 desc[starts[b]:starts[b + 1]], pts[starts[b]:starts[b + 1]],
                                 metric=metric, threshold=threshold, max_dist_sq=150.0 * 150.0)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"pair {k} ({a},{b})"
    s.close()


def test_lists_sized_from_the_first_chunk(ctx, oracle):
    """A call of several chunks sizes the candidate lists of the later chunks from what the first one needed.  Here
    the first chunk holds unrelated images (next to no candidates) and the later ones pairs of images whose rows
    all come four times: their slices overflow the small lists, the more selective passes take over, and the lists are
    still the restatement's; the next call on the same result object starts from larger lists.  The result object
    holds far less memory than with lists of the maximum size."""
    n = 768
    d, p, _ = synth.image_set(6, n, seed=91, shared_frac=0.0)
    desc, pts = d.copy(), p.copy()
    quarter = desc[4 * n:4 * n + n // 4].copy()
    for k in (4, 5):  # every row of these two images four times, and image 5 = image 4
        for q in range(4):
            desc[k * n + q * (n // 4):k * n + (q + 1) * (n // 4)] = quarter
        pts[k * n:(k + 1) * n] = pts[4 * n:5 * n]
    counts = np.full(6, n, np.int32)
    pl = np.array([[0, 1], [2, 3], [4, 5], [5, 4], [0, 4], [4, 4]], np.int32)
    s = ctx.imageset(counts, desc, pts)
    ctx.set_option("pairs_per_launch", 2)
    try:
        sizes = {}
        for observe in (1, 0):
            ctx.set_option("cand_observe", observe)
            res = None
            for call in range(3):
                res = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res)
                off, pairs = res.offsets(), res.pairs()
                for k, (a, b) in enumerate(pl):
                    exp, _, _ = oracle.match(desc[a * n:(a + 1) * n], pts[a * n:(a + 1) * n], desc[b * n:(b + 1) * n],
                                             pts[b * n:(b + 1) * n], threshold=0.95, max_dist_sq=1e8)
                    assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"observe {observe}, call {call}: pair {k} ({a},{b})"
                if call == 0:
                    sizes[observe] = res.device_bytes()
                    first_stats = res.stats_ex()
            if observe:
                assert sum(first_stats[1:]) > 0, "the small lists were meant to overflow in the first call"
            res.close()
        assert sizes[1] * 2 < sizes[0], sizes  # (the first chunk keeps lists of the maximum size)
    finally:
        ctx.set_option("pairs_per_launch", 2048)
        ctx.set_option("cand_observe", 1)
    s.close()


@pytest.mark.parametrize("split", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("sim_variant", [1, 0])
def test_column_parts_give_the_same_lists(ctx, oracle, split, sim_variant):
    """Small launches of the strict passes cut each row block's columns into parts, one block per part with its
    own slices of the candidate list.  Whatever the number of parts (also more parts than a short image has
    steps of 64 columns, and lists so short that the later passes take over), the lists are the restatement's."""
    counts = np.array([640, 257, 0, 1500, 64, 2100, 1], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=77)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    starts = np.concatenate([[0], np.cumsum(counts)])
    pl = synth.exhaustive_pairs(len(counts))
    pl = np.concatenate([pl, pl[:, ::-1]]).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    ctx.set_option("match_col_split", split)
    ctx.set_option("sim_variant", sim_variant)
    ctx.set_option("pairs_per_launch", 16)
    try:
        for thr, gate, cap in ((0.95, 1e8, DEFAULT_CAND_CAPACITY), (0.5, 200.0 * 200.0, DEFAULT_CAND_CAPACITY), (0.9, 1e8, 4096)):
            ctx.set_option("cand_capacity", cap)
            res = s.match(pl, threshold=thr, max_dist_sq=gate)
            off, pairs = res.offsets(), res.pairs()
            res.close()
            for k, (a, b) in enumerate(pl):
                exp, _, _ = oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]],
                                         desc[starts[b]:starts[b + 1]], pts[starts[b]:starts[b + 1]],
                                         threshold=thr, max_dist_sq=gate)
                assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"threshold {thr}: pair {k} ({a},{b})"
    finally:
        ctx.set_option("match_col_split", 0)
        ctx.set_option("sim_variant", 1)
        ctx.set_option("pairs_per_launch", 2048)
        ctx.set_option("cand_capacity", DEFAULT_CAND_CAPACITY)
    s.close()


@pytest.mark.parametrize("sim_variant", [1, 0])
@pytest.mark.parametrize("gate", [1e8, 1e4])
def test_benchmark_pair_against_restatement(ctx, oracle, sim_variant, gate):
    """The benchmark's own configuration at full size: images 0 and 1 of its generator (seed 2026, 8192
    features each), threshold 0.95, with both similarity bounds and both pixel gates.  Pair list, matches1
    and matches2 equal those of matchTwoSided (match.cpp:24-94), through the batch entry point and the
    single-pair one."""
    d, p, c = synth.image_set(2, 8192, seed=2026)
    exp, e1, e2 = oracle.match(d[:8192], p[:8192], d[8192:], p[8192:], threshold=0.95, max_dist_sq=gate)
    assert len(exp) > 2000
    ctx.set_option("sim_variant", sim_variant)
    try:
        got, g1, g2 = ctx.match_pair(d[:8192], p[:8192], d[8192:], p[8192:], threshold=0.95, max_dist_sq=gate,
                                     return_best=True)
        assert np.array_equal(g1, e1) and np.array_equal(g2, e2) and np.array_equal(got, exp)
        s = ctx.imageset(c, d, p)
        res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=0.95, max_dist_sq=gate)
        off, pairs = res.offsets(), res.pairs()
        assert np.array_equal(pairs[off[0]:off[1]], exp)
        rev, _, _ = oracle.match(d[8192:], p[8192:], d[:8192], p[:8192], threshold=0.95, max_dist_sq=gate)
        assert np.array_equal(pairs[off[1]:off[2]], rev)
        assert res.stats()[1] == 0  # nothing fell back to cell-by-cell scoring
        res.close()
        s.close()
    finally:
        ctx.set_option("sim_variant", int(__import__("os").environ.get("SFM_SIM_VARIANT", "1")))


def test_full_size_properties(ctx):
    """8192 x 8192 (the benchmark shape): size-independent properties instead of the oracle.
    Swapping the images transposes the match set; matching an image with itself is the identity
    on rows with a unique descriptor; results are ascending and one-to-one."""
    d, p, c = synth.image_set(3, 8192, seed=77)
    s = ctx.imageset(c, d, p)
    res = s.match(np.array([[0, 1], [1, 0], [2, 2]], np.int32), threshold=0.95)
    off, pairs = res.offsets(), res.pairs()
    ab, ba, cc = pairs[off[0]:off[1]], pairs[off[1]:off[2]], pairs[off[2]:off[3]]
    assert len(ab) > 2000
    assert np.all(np.diff(ab[:, 0]) > 0) and len(np.unique(ab[:, 1])) == len(ab)
    assert set(map(tuple, ab)) == set(map(tuple, ba[:, ::-1]))
    assert np.array_equal(cc[:, 0], cc[:, 1]) and len(cc) > 8000
    s.close()


@pytest.mark.parametrize("thr", [0.5, 0.0])
def test_loose_threshold_with_tight_pixel_gate(ctx, oracle, thr):
    """Most cells clear the similarity threshold but few are inside the pixel gate: a row's most
    similar cell is usually outside the gate and must not shadow the best cell inside it."""
    dA, pA, dB, pB = two_images(900, 1000, shared=0.4, seed=41)
    dA = np.maximum(dA, 15)
    dB = np.maximum(dB, 15)
    ctx.set_option("cand_capacity", 4096)  # force the filtered second pass
    try:
        for gate in (60.0 * 60.0, 300.0 * 300.0, 1500.0 * 1500.0):
            check_pair(ctx, oracle, dA, pA, dB, pB, thr, gate)
    finally:
        ctx.set_option("cand_capacity", DEFAULT_CAND_CAPACITY)


@pytest.mark.parametrize("nA,nB", [(700, 1300), (1300, 700), (64, 2000), (2000, 1), (1025, 1025)])
@pytest.mark.parametrize("metric", [0, 1])
def test_second_pass_both_orientations(ctx, oracle, nA, nB, metric):
    """Every cell clears the threshold, the list is too small for all of them but large enough for
    the cells that can be a row best or a column best: the second pass (row-best filter, once with
    each image as the row image) must deliver the reference's result without cell-by-cell scoring."""
    n = max(nA, nB)
    d, p, _ = synth.image_set(2, n, shared_frac=0.4, seed=53)
    d = np.maximum(d, 15)
    counts = np.array([nA, nB], np.int32)
    desc = np.concatenate([d[:nA], d[n:n + nB]])
    pts = np.concatenate([p[:nA], p[n:n + nB]])
    ctx.set_option("cand_capacity", 65536)
    try:
        s = ctx.imageset(counts, desc, pts)
        for gate in (1e8, 400.0 * 400.0):
            res = s.match(np.array([[0, 1]], np.int32), threshold=0.0, max_dist_sq=gate, metric=metric)
            exp, _, _ = oracle.match(desc[:nA], pts[:nA], desc[nA:], pts[nA:], threshold=0.0, max_dist_sq=gate,
                                     metric=metric)
            assert np.array_equal(res.pairs(), exp)
            cand, cell_by_cell = res.stats()
            assert cell_by_cell == 0
            if nA * nB > 65536:
                assert cand < nA * nB // 4, "the filter should have dropped most cells"
        s.close()
    finally:
        ctx.set_option("cand_capacity", DEFAULT_CAND_CAPACITY)


def test_32k_features_properties(ctx):
    """32768 x 32768 (the largest configured image size): the image swap transposes the match set,
    results are ascending and one-to-one, and nothing is scored cell by cell."""
    d, p, c = synth.image_set(2, 32768, seed=78)
    s = ctx.imageset(c, d, p)
    res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=0.95, max_dist_sq=100.0 * 100.0)
    off, pairs = res.offsets(), res.pairs()
    ab, ba = pairs[off[0]:off[1]], pairs[off[1]:off[2]]
    assert len(ab) > 5000
    assert np.all(np.diff(ab[:, 0]) > 0) and len(np.unique(ab[:, 1])) == len(ab)
    assert set(map(tuple, ab)) == set(map(tuple, ba[:, ::-1]))
    assert res.stats()[1] == 0
    s.close()


@pytest.mark.parametrize("gate", [1e4, 1e8])
def test_32k_pair_against_restatement(ctx, oracle, gate):
    """One pair of the high-density configuration at full size (2 x 32768 features, 1.07e9 cells): pair list,
    matches1 and matches2 equal those of matchTwoSided (match.cpp:24-94), whose restatement materialises the
    4 GiB similarity matrix as the reference does; through the single-pair entry point and the batch one."""
    n = 32768
    d, p, c = synth.image_set(2, n, seed=78)
    exp, e1, e2 = oracle.match(d[:n], p[:n], d[n:], p[n:], threshold=0.95, max_dist_sq=gate)
    assert len(exp) > 5000
    got, g1, g2 = ctx.match_pair(d[:n], p[:n], d[n:], p[n:], threshold=0.95, max_dist_sq=gate, return_best=True)
    assert np.array_equal(g1, e1) and np.array_equal(g2, e2) and np.array_equal(got, exp)
    s = ctx.imageset(c, d, p)
    res = s.match(np.array([[0, 1]], np.int32), threshold=0.95, max_dist_sq=gate)
    assert np.array_equal(res.pairs(), exp) and res.stats()[1] == 0
    res.close()
    s.close()


def test_norm_square_root_is_correctly_rounded(ctx, oracle):
    """A descriptor matched with its copy at threshold 1.0: sim = n / (sqrtf(n) * sqrtf(n)) is 0.99999994 for
    this n (match.hpp:45), so the reference rejects it; a square root that is one ulp low would accept."""
    a = np.array([[163, 6, 172, 131, 41, 116, 2, 172, 91, 23, 172, 172, 124, 21, 172, 0, 172]], np.float32)
    p = np.zeros((1, 2))
    for thr in (1.0, 0.99999994, 0.9999999):
        check_pair(ctx, oracle, a, p, a.copy(), p, thr)
    assert len(ctx.match_pair(a, p, a.copy(), p, threshold=1.0)) == 0


def test_repeated_pair_calls_reuse_their_buffers(ctx, oracle):
    """sfm_match_pair keeps its device tables in the context: calls of changing size, descriptor length
    and value range (byte-valued or not) in a row must each give the reference's list."""
    rng = np.random.default_rng(123)
    for nA, nB, dim, integer in [(900, 700, 128, True), (50, 1300, 128, False), (1300, 40, 64, True), (0, 10, 128, True),
                                 (600, 600, 128, True), (300, 300, 7, False)]:
        base = rng.integers(0, 200, size=(max(nA, nB, 1), dim)).astype(np.float32)
        A = base[:nA] + (0 if integer else rng.normal(0, 0.3, size=(nA, dim)).astype(np.float32))
        B = base[:nB][::-1].copy() + (0 if integer else rng.normal(0, 0.3, size=(nB, dim)).astype(np.float32))
        pA, pB = rng.uniform(0, 100, size=(nA, 2)), rng.uniform(0, 100, size=(nB, 2))
        got, m1, m2 = ctx.match_pair(A, pA, B, pB, threshold=0.9, max_dist_sq=1e8, return_best=True)
        exp, e1, e2 = oracle.match(A, pA, B, pB, threshold=0.9, max_dist_sq=1e8)
        assert np.array_equal(got, exp) and np.array_equal(m1, e1) and np.array_equal(m2, e2), (nA, nB, dim)
    with pytest.raises(capi.SfmError):
        ctx.match_pair(np.zeros((4, 200), np.float32), np.zeros((4, 2)), np.zeros((4, 200), np.float32), np.zeros((4, 2)))
    A = rng.integers(1, 200, size=(64, 128)).astype(np.float32)
    got = ctx.match_pair(A, np.zeros((64, 2)), A, np.zeros((64, 2)), threshold=0.9)  # the call after a refused one
    assert got.tolist() == [[i, i] for i in range(64)]


def test_calls_queue_behind_each_other_without_a_wait(ctx, oracle):
    """Batch calls return before their work is done, and a further call on the same result object waits only for
    the uploads of the one before.  Calls with different pair lists and thresholds, issued back to back into one
    and into two result objects, each leave the lists of the last call made on the object."""
    counts = np.array([900, 640, 257, 1100, 64, 511], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=321)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    starts = np.concatenate([[0], np.cumsum(counts)])
    pl_a = synth.exhaustive_pairs(len(counts)).astype(np.int32)
    pl_b = pl_a[::-1, ::-1].copy()[:11]
    s = ctx.imageset(counts, desc, pts)

    def expect(pl, thr, gate):
        return [oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]], desc[starts[b]:starts[b + 1]],
                             pts[starts[b]:starts[b + 1]], threshold=thr, max_dist_sq=gate)[0] for a, b in pl]

    def check(res, pl, want):
        off, pairs = res.offsets(), res.pairs()
        assert len(off) == len(pl) + 1
        for k in range(len(pl)):
            assert np.array_equal(pairs[off[k]:off[k + 1]], want[k]), k

    want_a, want_b = expect(pl_a, 0.95, 1e8), expect(pl_b, 0.9, 150.0 ** 2)
    ctx.set_option("pairs_per_launch", 4)  # several chunks: three streams, three buffer sets
    try:
        r1 = r2 = None
        for _ in range(4):
            r1 = s.match(pl_a, threshold=0.95, max_dist_sq=1e8, result=r1)
            r2 = s.match(pl_b, threshold=0.9, max_dist_sq=150.0 ** 2, result=r2)
            r1 = s.match(pl_b, threshold=0.9, max_dist_sq=150.0 ** 2, result=r1)
            r1 = s.match(pl_a, threshold=0.95, max_dist_sq=1e8, result=r1)
        check(r2, pl_b, want_b)
        check(r1, pl_a, want_a)
        r1 = s.match(pl_b, threshold=0.9, max_dist_sq=150.0 ** 2, result=r1)
        check(r1, pl_b, want_b)
        r1.close()
        r2.close()
    finally:
        ctx.set_option("pairs_per_launch", 2048)
    s.close()


def test_tables_in_page_locked_memory(ctx, oracle):
    """sfm_host_alloc: tables kept in page-locked memory go through the same entry point and give the same lists."""
    dA, pA, dB, pB = two_images(700, 900)
    exp, e1, e2 = oracle.match(dA, pA, dB, pB, threshold=0.95, max_dist_sq=1e8)
    tabs = [ctx.pinned_like(a) for a in (dA, pA, dB, pB)]
    assert all(np.array_equal(t, a) for t, a in zip(tabs, (dA, pA, dB, pB)))
    for _ in range(3):
        got, g1, g2 = ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8, return_best=True)
        assert np.array_equal(got, exp) and np.array_equal(g1, e1) and np.array_equal(g2, e2)
    del tabs


@pytest.mark.parametrize("thr", [0.5, 0.0])
def test_two_phase_sweeps_and_the_gate(ctx, oracle, thr):
    """Pairs whose candidate lists overflow are redone with row maxima and a strict sweep against them when
    their pixel gate excludes nothing, and with the filtered sweep otherwise.  Gates far above, just above and
    just below the extent of the points, a point that is not finite, and the option that turns the two-phase
    form off: always the restatement's lists."""
    dA, pA, dB, pB = two_images(900, 1100, shared=0.5)
    dA, dB = np.maximum(dA, 20), np.maximum(dB, 20)  # unrelated cosines ~0.8: every cell clears the threshold
    allp = np.concatenate([pA, pB])
    ext2 = float(((allp.max(axis=0) - allp.min(axis=0)) ** 2).sum())
    for gate in (1e8, ext2 * 1.01, ext2 * 0.999, ext2 * 0.25):
        for two_phase in (1, 0):
            ctx.set_option("match_two_phase", two_phase)
            try:
                check_pair(ctx, oracle, dA, pA, dB, pB, thr, np.float32(gate))
            finally:
                ctx.set_option("match_two_phase", 1)
    pB2 = pB.copy()
    pB2[17, 0] = np.inf
    pB2[33, 1] = np.nan
    check_pair(ctx, oracle, dA, pA, dB, pB2, thr, 1e8)
    counts = np.array([len(dA), len(dB)], np.int32)
    s = ctx.imageset(counts, np.concatenate([dA, dB]), np.concatenate([pA, pB]))
    res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=thr, max_dist_sq=1e8)
    assert res.stats_ex()[2] == 2, "both pairs were meant to need the filtered passes"
    res.close()
    s.close()


def test_byte_tables_through_the_pair_call(ctx, oracle):
    """sfm_match_pair_u8: descriptors handed over as bytes (OpenCV's SIFT, detectOpenCv.cpp:166-169) give the
    lists and bests of the float call and of the restatement, bit for bit -- for 128 elements per row (the bytes
    serve the exact pass as they are) and for other lengths, for both metrics, thresholds that flood the bound
    passes, empty images, and between float calls on the same context."""
    rng = np.random.default_rng(321)
    cases = [(900, 700, 128, 0.9, 0), (50, 1300, 128, 0.5, 0), (1300, 40, 64, 0.9, 0), (0, 10, 128, 0.9, 0), (10, 0, 128, 0.9, 0),
             (2048, 2048, 128, 0.95, 0), (300, 300, 7, 0.8, 0), (700, 900, 128, 1e5, 1), (400, 400, 128, 0.0, 0)]
    for nA, nB, dim, thr, metric in cases:
        base = rng.integers(0, 256, size=(max(nA, nB, 1), dim)).astype(np.uint8)
        A = base[:nA].copy()
        B = base[:nB][::-1].copy()
        if nB:
            B[::3] = rng.integers(0, 256, size=B[::3].shape)  # a third of the rows unrelated
        pA, pB = rng.uniform(0, 100, size=(nA, 2)), rng.uniform(0, 100, size=(nB, 2))
        got, m1, m2 = ctx.match_pair(A, pA, B, pB, threshold=thr, max_dist_sq=1e8, metric=metric, return_best=True)
        flt, f1, f2 = ctx.match_pair(A.astype(np.float32), pA, B.astype(np.float32), pB, threshold=thr, max_dist_sq=1e8,
                                     metric=metric, return_best=True)
        exp, e1, e2 = oracle.match(A.astype(np.float32), pA, B.astype(np.float32), pB, threshold=thr, max_dist_sq=1e8, metric=metric)
        assert np.array_equal(got, exp) and np.array_equal(m1, e1) and np.array_equal(m2, e2), (nA, nB, dim, thr)
        assert np.array_equal(flt, exp) and np.array_equal(f1, e1) and np.array_equal(f2, e2), (nA, nB, dim, thr)
    with pytest.raises(capi.SfmError):
        ctx.match_pair(np.zeros((4, 200), np.uint8), np.zeros((4, 2)), np.zeros((4, 200), np.uint8), np.zeros((4, 2)))


def test_flood_probe_changes_nothing(ctx, oracle):
    """Pairs the int8 bound floods are probed (64 x 256 cells under the fp16 bound) and, when that bound would flood
    as well, handed straight to the filtered passes instead of trying the fp16 strict pass first.  With the probe on
    or off the lists are the restatement's: at a loose threshold (the probe skips the attempt), at a tight one on
    descriptors whose unrelated cosines sit inside the int8 margin only (it must not), and on images too small to probe."""
    rng = np.random.default_rng(11)
    n = 1100
    d, p, _ = synth.image_set(3, n, shared_frac=0.4, seed=77)
    base = synth.sift_like(rng, 1)[0]
    hard = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, 2 * n) * 0.42, 0, 255).astype(np.float32)
    hard[n:n + 300] = hard[:300]
    cases = [(d[:n], p[:n], d[n:2 * n], p[n:2 * n], 0.5), (d[:n], p[:n], d[2 * n:], p[2 * n:], 0.2),
             (hard[:n], p[:n], hard[n:], p[n:2 * n], 0.95), (d[:60], p[:60], d[n:n + 200], p[n:n + 200], 0.3)]
    for dA, pA, dB, pB, thr in cases:
        exp, e1, e2 = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=1e8)
        for probe in (1, 0):
            ctx.set_option("match_flood_probe", probe)
            try:
                got, g1, g2 = ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=1e8, return_best=True)
            finally:
                ctx.set_option("match_flood_probe", 1)
            assert np.array_equal(got, exp) and np.array_equal(g1, e1) and np.array_equal(g2, e2), (thr, probe)


def _mixed_set(rng, n, n_img, weight):
    """Images of SIFT-like descriptors that share a common part: the cosine between unrelated descriptors grows as
    the weight of the individual part shrinks (0.42: about 0.93, 0.6: about 0.88); the first 30 % of every image
    re-detects image 0's features."""
    base = synth.sift_like(rng, 1)[0]
    d = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n_img * n) * weight, 0, 255).astype(np.float32)
    shared = int(0.3 * n)
    for im in range(1, n_img):
        d[im * n:im * n + shared] = d[:shared]
    return d, rng.uniform(0, 4000, (n_img * n, 2))


@pytest.mark.parametrize("weight", [None, 0.6, 0.42])
def test_six_bit_first_pass_changes_nothing(ctx, oracle, weight):
    """The first bound pass of a batch call runs on 6-bit floats (e2m3) with a margin of about 0.1; the pairs it
    floods are redone under the int8 bound (margin 0.02-0.04), those under the fp16 bound, and so on.  Lists with
    the 6-bit pass equal the lists without it and the restatement's: on the usual descriptors (nothing floods), on
    descriptors whose unrelated cosines (0.88) are inside the 6-bit margin of the threshold only, and on ones inside
    the int8 margin too (0.93) -- in one chunk and in chunks of three pairs, where the later chunks learn from the
    first whether to start with the 6-bit pass at all."""
    rng = np.random.default_rng(5)
    n, n_img = 1100, 5
    if weight is None:
        d, p, _ = synth.image_set(n_img, n, shared_frac=0.4, seed=31)
    else:
        d, p = _mixed_set(rng, n, n_img, weight)
    counts = np.full(n_img, n, np.int32)
    pl = synth.exhaustive_pairs(n_img)
    s = ctx.imageset(counts, d, p)
    lists = {}
    try:
        for six in (1, 0):
            for per in (2048, 3):
                ctx.set_option("match_six_bit", six)
                ctx.set_option("pairs_per_launch", per)
                res = s.match(pl, threshold=0.95, max_dist_sq=1e8)
                lists[six, per] = (res.offsets(), res.pairs(), res.stats_ex())
                res.close()
    finally:
        ctx.set_option("match_six_bit", 1)
        ctx.set_option("pairs_per_launch", 2048)
        s.close()
    off, pairs, _ = lists[1, 2048]
    for key, (o, q, _) in lists.items():
        assert np.array_equal(o, off) and np.array_equal(q, pairs), key
    for k in (0, 4, 9):
        a, b = pl[k]
        exp, _, _ = oracle.match(d[a * n:(a + 1) * n], p[a * n:(a + 1) * n], d[b * n:(b + 1) * n], p[b * n:(b + 1) * n],
                                 threshold=0.95, max_dist_sq=1e8)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), k
    # what the passes after the first had to do: nothing on the usual descriptors, the fp16 bound (or more) at 0.93
    _, cells, filtered, redone = lists[1, 2048][2]
    if weight is None:
        assert cells == 0 and filtered == 0 and redone == 0
    if weight == 0.42:
        assert redone + filtered + cells > 0


def test_six_bit_table_is_optional(ctx, oracle):
    """An image set made while match_six_bit is 0 has no 6-bit table: its calls start with the int8 pass whatever the
    option says later, and give the same lists as a set that has one."""
    d, p, c = synth.image_set(3, 900, shared_frac=0.5, seed=8)
    pl = synth.exhaustive_pairs(3)
    with_table = ctx.imageset(c, d, p)
    ctx.set_option("match_six_bit", 0)
    try:
        without = ctx.imageset(c, d, p)
    finally:
        ctx.set_option("match_six_bit", 1)
    assert without.device_bytes() < with_table.device_bytes()
    a, b = with_table.match(pl, threshold=0.9, max_dist_sq=1e8), without.match(pl, threshold=0.9, max_dist_sq=1e8)
    assert np.array_equal(a.offsets(), b.offsets()) and np.array_equal(a.pairs(), b.pairs())
    exp, _, _ = oracle.match(d[:900], p[:900], d[900:1800], p[900:1800], threshold=0.9, max_dist_sq=1e8)
    assert np.array_equal(a.pairs()[a.offsets()[0]:a.offsets()[1]], exp)
    for x in (a, b, with_table, without):
        x.close()


def test_later_chunks_skip_an_int8_pass_that_floods(ctx, oracle):
    """A call of several chunks at the reference's default threshold (match.hpp:59): the observed first chunk floods
    its int8 lists, so the later chunks skip that pass and start from the flood probe -- including pairs that would
    not have flooded (tiny images, images of duplicates the threshold keeps apart), which then take the fp16 strict
    pass.  A second call on the same result object skips it from the first chunk on.  Lists equal the restatement's
    for every pair, and equal a run in one chunk."""
    counts = np.array([1100, 1100, 1100, 1100, 1100, 1100, 1100, 40, 300], np.int32)
    n = 1100
    d, p, _ = synth.image_set(len(counts), n, shared_frac=0.4, seed=123)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    starts = np.concatenate([[0], np.cumsum(counts)])
    big = synth.exhaustive_pairs(7)                       # 21 pairs of full images: the observed chunk
    rest = np.array([[i, j] for i in range(9) for j in (7, 8) if i != j] + [[k, k] for k in range(7)], np.int32)
    pl = np.concatenate([big, rest, big[::-1]]).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    ctx.set_option("pairs_per_launch", 21)
    try:
        res = s.match(pl, threshold=0.5, max_dist_sq=1e8)
        off, pairs = res.offsets(), res.pairs()
        again = s.match(pl, threshold=0.5, max_dist_sq=1e8, result=res)   # the observation is kept with the result
        assert np.array_equal(again.offsets(), off) and np.array_equal(again.pairs(), pairs)
    finally:
        ctx.set_option("pairs_per_launch", 2048)
    one = s.match(pl, threshold=0.5, max_dist_sq=1e8)
    assert np.array_equal(one.offsets(), off) and np.array_equal(one.pairs(), pairs)
    for k in list(range(0, len(pl), 5)) + [21, 22, 23, len(pl) - 1]:
        a, b = pl[k]
        exp, _, _ = oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]],
                                 desc[starts[b]:starts[b + 1]], pts[starts[b]:starts[b + 1]], threshold=0.5, max_dist_sq=1e8)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), k
    for o in (res, one, s):
        o.close()


@pytest.mark.parametrize("spread", [0.0, 0.2])
def test_dot_metric_through_the_strict_passes(ctx, oracle, spread):
    """dotProduct (match.hpp:18-27) on descriptors of unit or nearly unit norm: a.b > T implies a cosine above
    T / (maxA maxB), so the pairs take the strict passes under that bound and the exact pass scores the dot
    product.  Lists and bests are the restatement's with that path on or off, in a single call and in a batch,
    also where the bound is useless (raw rows, threshold 0) and where nothing can pass (threshold above every
    product of norms)."""
    rng = np.random.default_rng(5)
    n = 1200
    d, p, _ = synth.image_set(3, n, shared_frac=0.5, seed=91)
    u = d / np.linalg.norm(d, axis=1, keepdims=True)
    u = (u * (1.0 + spread * rng.uniform(-1, 1, size=(len(u), 1)))).astype(np.float32)
    for thr in (0.9, 0.6, 0.0, 2.5):
        exp, e1, e2 = oracle.match(u[:n], p[:n], u[n:2 * n], p[n:2 * n], threshold=thr, max_dist_sq=1e8, metric=1)
        for strict in (1, 0):
            ctx.set_option("match_dot_strict", strict)
            try:
                got, g1, g2 = ctx.match_pair(u[:n], p[:n], u[n:2 * n], p[n:2 * n], threshold=thr, max_dist_sq=1e8, metric=1,
                                             return_best=True)
            finally:
                ctx.set_option("match_dot_strict", 1)
            assert np.array_equal(got, exp) and np.array_equal(g1, e1) and np.array_equal(g2, e2), (thr, strict)
        if thr == 0.9:
            assert len(exp) > 100
    counts = np.array([n, n, n], np.int32)
    s = ctx.imageset(counts, u, p)
    pl = np.array([[0, 1], [0, 2], [1, 2], [2, 2]], np.int32)
    res = s.match(pl, threshold=0.9, max_dist_sq=1e8, metric=1)
    off, pairs = res.offsets(), res.pairs()
    cand, cells, filtered, redone = res.stats_ex()
    assert cells == 0 and filtered == 0, "unit rows at a tight threshold need neither fallback"
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match(u[a * n:(a + 1) * n], p[a * n:(a + 1) * n], u[b * n:(b + 1) * n], p[b * n:(b + 1) * n],
                                 threshold=0.9, max_dist_sq=1e8, metric=1)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), k
    s.close()


@pytest.mark.parametrize("as_bytes,dim", [(False, 128), (True, 128), (True, 64), (False, 42)])
def test_pair_calls_reuse_the_shared_frame(ctx, oracle, as_bytes, dim):
    """Option match_pair_reuse: image A of a call is taken from the device when it is the last call's image B (same
    host tables).  A video's order (scene.cpp:60: frame i against i + 1, then i + 1 against i + 2), one call out of
    that order, frames of unequal sizes and a table of another size at a reused address: every list equals the
    restatement's and the lists of the same calls with the option off."""
    d, p, _ = synth.image_set(4, 900, seed=77)
    sizes = [900, 640, 900, 777]
    frames = []
    for k, n in enumerate(sizes):
        dk = np.ascontiguousarray(d[900 * k:900 * k + n, :dim])
        if as_bytes:
            dk = np.ascontiguousarray(np.clip(np.rint(dk * 400.0), 0, 255).astype(np.uint8))
        frames.append((dk, np.ascontiguousarray(p[900 * k:900 * k + n])))
    order = [(0, 1), (1, 2), (2, 3), (0, 3), (3, 2), (2, 1), (1, 0), (0, 1)]

    def run():
        return [ctx.match_pair(frames[a][0], frames[a][1], frames[b][0], frames[b][1], threshold=0.9, max_dist_sq=1e8)
                for a, b in order]

    plain = run()
    ctx.set_option("match_pair_reuse", 1)
    try:
        kept = run()
        # the promise covers unchanged tables only: a changed table must come under another address, and a table of
        # another size at the same address is not taken for the kept one
        other = (frames[1][0][:500], frames[1][1][:500])
        kept_other = ctx.match_pair(other[0], other[1], frames[2][0], frames[2][1], threshold=0.9, max_dist_sq=1e8)
    finally:
        ctx.set_option("match_pair_reuse", 0)
    for (a, b), x, y in zip(order, plain, kept):
        exp, _, _ = oracle.match(frames[a][0].astype(np.float32), frames[a][1], frames[b][0].astype(np.float32), frames[b][1],
                                 threshold=0.9, max_dist_sq=1e8)
        assert np.array_equal(x, exp) and np.array_equal(y, exp), (a, b)
        assert len(exp) > (10 if dim == 128 else 0)
    exp, _, _ = oracle.match(other[0].astype(np.float32), other[1], frames[2][0].astype(np.float32), frames[2][1], threshold=0.9,
                             max_dist_sq=1e8)
    assert np.array_equal(kept_other, exp)


def test_reused_frame_is_the_copy_on_the_device(ctx, oracle):
    """That the option takes effect at all: a table changed in place between two calls - against the promise the
    option asks for - is matched as it was when the call before uploaded it; with the option off, as it is now."""
    d, p, _ = synth.image_set(3, 800, seed=78)
    f = [(np.ascontiguousarray(d[800 * k:800 * (k + 1)]), np.ascontiguousarray(p[800 * k:800 * (k + 1)])) for k in range(3)]
    ctx.set_option("match_pair_reuse", 1)
    try:
        ctx.match_pair(f[0][0], f[0][1], f[1][0], f[1][1], threshold=0.9, max_dist_sq=1e8)
        before = f[1][0].copy()
        f[1][0][...] = before[::-1]
        kept = ctx.match_pair(f[1][0], f[1][1], f[2][0], f[2][1], threshold=0.9, max_dist_sq=1e8)
    finally:
        ctx.set_option("match_pair_reuse", 0)
    now = ctx.match_pair(f[1][0], f[1][1], f[2][0], f[2][1], threshold=0.9, max_dist_sq=1e8)
    exp_before, _, _ = oracle.match(before, f[1][1], f[2][0], f[2][1], threshold=0.9, max_dist_sq=1e8)
    exp_now, _, _ = oracle.match(f[1][0], f[1][1], f[2][0], f[2][1], threshold=0.9, max_dist_sq=1e8)
    assert len(exp_before) > 10 and not np.array_equal(exp_before, exp_now)
    assert np.array_equal(kept, exp_before)
    assert np.array_equal(now, exp_now)


@pytest.mark.parametrize("threshold", [0.95, 0.5])
def test_lists_sized_as_the_call_goes_are_the_same_lists(ctx, threshold):
    """A batch of many chunks starts with room for the worst case of its first eight chunks and keeps its list
    buffer ahead of the running total the device reports (growing it once, the lists so far moving over), instead
    of allocating min(nA, nB) entries per pair up front.  Same offsets and lists as with the worst case allocated,
    in less memory; a second call on the result object finds the room it needs and allocates nothing."""
    n_img, n = 28, 400
    d, p, c = synth.image_set(n_img, n, seed=31)
    pl = synth.exhaustive_pairs(n_img)  # 378 pairs, 48 chunks of 8
    s = ctx.imageset(c, d, p)
    ctx.set_option("pairs_per_launch", 8)
    try:
        ctx.set_option("match_list_observe", 0)
        ref = s.match(pl, threshold=threshold)
        off0, pairs0, bytes0 = ref.offsets(), ref.pairs(), ref.device_bytes()
        ref.close()
        ctx.set_option("match_list_observe", 2)  # (1, the default, waits for calls of more than 32 chunks: see below)
        res = s.match(pl, threshold=threshold)
        off1, pairs1, bytes1 = res.offsets(), res.pairs(), res.device_bytes()
        res = s.match(pl, threshold=threshold, result=res)
        off2, pairs2, bytes2 = res.offsets(), res.pairs(), res.device_bytes()
        res.close()
    finally:
        ctx.set_option("pairs_per_launch", 2048)
        ctx.set_option("match_list_observe", 1)
    s.close()
    assert int(off0[-1]) > 8 * 8 * n  # more matches than the first eight chunks could hold at worst: the buffer grew
    assert np.array_equal(off0, off1) and np.array_equal(pairs0, pairs1)
    assert np.array_equal(off0, off2) and np.array_equal(pairs0, pairs2)
    worst = len(pl) * n * 8
    print("list bytes: worst case", worst, "result objects", bytes0, bytes1, bytes2, "matches", int(off0[-1]))
    assert bytes2 == bytes1 and bytes1 < bytes0  # (at threshold 0.5 most keypoints find a partner: little to save)
    if threshold > 0.9:
        assert bytes0 - bytes1 > worst // 3


def test_single_pair_calls_remember_where_the_last_one_ended(ctx, oracle):
    """Single-pair calls at the reference's default threshold (match.hpp:59), one after the other as the reference
    makes them (scene.cpp:60): a call starts where the call before ended up (past the int8 attempt, or straight at
    the filtered passes) and takes the whole way again every sixteenth time.  Pairs whose lists flood and pairs
    whose lists do not alternate in blocks, so that every guess is wrong some of the time: the lists are the
    restatement's every time."""
    n = 1100
    sets = []
    for kind, seed in (("sift", 41), ("gauss", 42)):
        d, p, _ = synth.image_set(2, n, seed=seed, kind=kind)
        exp, m1, m2 = oracle.match(d[:n], p[:n], d[n:], p[n:], threshold=0.5, max_dist_sq=1e8)
        sets.append((d, p, exp, m1, m2))
    order = [0] * 5 + [1] * 3 + [0] * 20 + [1] * 7 + [0, 1] * 6
    for k, which in enumerate(order):
        d, p, exp, m1, m2 = sets[which]
        got = ctx.match_pair(d[:n], p[:n], d[n:], p[n:], threshold=0.5, max_dist_sq=1e8, return_best=True)
        assert np.array_equal(got[0], exp), (k, which)
        assert np.array_equal(got[1], m1) and np.array_equal(got[2], m2), (k, which)


def test_calls_remember_that_the_six_bit_pass_floods(ctx, oracle):
    """One-chunk calls on descriptors whose unrelated cosines (0.88) are inside the 6-bit margin of the threshold: the
    first call tries the 6-bit pass, floods it and goes on under the int8 bound; the following calls with the same
    options on the same result object start with the int8 pass.  Every call gives the same lists, which are the
    restatement's."""
    rng = np.random.default_rng(19)
    n, n_img = 1100, 4
    d, p = _mixed_set(rng, n, n_img, 0.6)
    pl = synth.exhaustive_pairs(n_img)
    s = ctx.imageset(np.full(n_img, n, np.int32), d, p)
    res = s.match(pl, threshold=0.95, max_dist_sq=1e8)
    off, pairs = res.offsets(), res.pairs()
    for _ in range(3):
        res = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res)
        assert np.array_equal(res.offsets(), off) and np.array_equal(res.pairs(), pairs)
    other = s.match(pl, threshold=0.9, max_dist_sq=1e8, result=res)  # other options: the 6-bit pass is tried again
    exp9, _, _ = oracle.match(d[:n], p[:n], d[n:2 * n], p[n:2 * n], threshold=0.9, max_dist_sq=1e8)
    assert np.array_equal(other.pairs()[other.offsets()[0]:other.offsets()[1]], exp9)
    exp, _, _ = oracle.match(d[:n], p[:n], d[n:2 * n], p[n:2 * n], threshold=0.95, max_dist_sq=1e8)
    assert np.array_equal(pairs[off[0]:off[1]], exp)
    other.close()
    s.close()


def _aligned_rows(rng, n):
    """Rows of 32 non-zero elements out of 128, the same multiset of values in every row (so all rows have one norm,
    32.4, and the 6-bit table one scale, 1/32): scaled, every value lies 0.05-0.15 above a representable one where the
    step is 0.5, and rounds down."""
    levels = np.array([4.14, 4.64, 5.14, 5.64, 6.14, 6.64, 7.14], np.float32)
    values = np.resize(levels, 32)
    rows = np.zeros((n, 128), np.float32)
    for i in range(n):
        rows[i, rng.choice(128, 32, replace=False)] = rng.permutation(values)
    return rows


def test_six_bit_margin_holds_when_rounding_errors_line_up(ctx, oracle):
    """Descriptors made so that every non-zero element rounds down in the 6-bit table: the copy of a descriptor is
    short by 2-3 % along the descriptor itself, so the approximate cosine of two near-equal descriptors is short by
    nearly the whole margin, while their true cosines straddle the threshold (image B's rows are image A's with up
    to two elements moved and up to six pairs of values swapped).  The lists must still be the restatement's."""
    rng = np.random.default_rng(23)
    n = 1500
    A = _aligned_rows(rng, n)
    B = A.copy()
    for i in range(n):
        nz = np.flatnonzero(B[i])
        for _ in range(int(rng.integers(0, 3))):  # an element moved to an empty place
            src = rng.choice(np.flatnonzero(B[i]))
            dst = rng.choice(np.flatnonzero(B[i] == 0))
            B[i, dst], B[i, src] = B[i, src], 0.0
        for _ in range(int(rng.integers(0, 7))):  # two values swapped
            j, k = rng.choice(np.flatnonzero(B[i]), 2, replace=False)
            B[i, j], B[i, k] = B[i, k], B[i, j]
    B = B[rng.permutation(n)]
    pA, pB = rng.uniform(0, 1000, (n, 2)), rng.uniform(0, 1000, (n, 2))
    s = ctx.imageset(np.array([n, n], np.int32), np.concatenate([A, B]), np.concatenate([pA, pB]))
    for thr in (0.95, 0.965, 0.98):
        res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=thr, max_dist_sq=1e8)
        off, pairs = res.offsets(), res.pairs()
        exp, _, _ = oracle.match(A, pA, B, pB, threshold=thr, max_dist_sq=1e8)
        rev, _, _ = oracle.match(B, pB, A, pA, threshold=thr, max_dist_sq=1e8)
        assert 50 < len(exp) < n - 50, (thr, len(exp))  # (the threshold cuts through the cosines)
        assert np.array_equal(pairs[off[0]:off[1]], exp) and np.array_equal(pairs[off[1]:off[2]], rev), thr
        assert res.stats_ex()[1:] == (0, 0, 0)  # (and no later pass was needed: the 6-bit lists held them)
        res.close()
    s.close()


def test_int8_margin_holds_when_rounding_errors_line_up(ctx, oracle):
    """The same for the int8 table (the first pass where there is no 6-bit table): 31 of a row's 32 non-zero elements
    are a whole number plus 0.49 once scaled (the 32nd is the largest, 127), so every one rounds down by 0.49 and the
    int32 sum of two near-equal rows is short by 80-90 % of the bound's margin, which is made of exactly these
    worst cases (half a unit per element against the other row's L1 norm).  Image B's rows are image A's with up to
    two elements moved and up to six pairs of values swapped: true cosines on both sides of the thresholds."""
    rng = np.random.default_rng(29)
    n = 1500
    values = np.concatenate([[127.0], np.resize(np.arange(24, 124, 3), 31) + 0.49]).astype(np.float32)
    A = np.zeros((n, 128), np.float32)
    for i in range(n):
        A[i, rng.choice(128, 32, replace=False)] = rng.permutation(values)
    B = A.copy()
    for i in range(n):
        for _ in range(int(rng.integers(0, 3))):
            nz = np.flatnonzero(B[i])
            src = rng.choice(nz[B[i, nz] < 127])
            dst = rng.choice(np.flatnonzero(B[i] == 0))
            B[i, dst], B[i, src] = B[i, src], 0.0
        for _ in range(int(rng.integers(0, 7))):
            nz = np.flatnonzero(B[i])
            j, k = rng.choice(nz[B[i, nz] < 127], 2, replace=False)
            B[i, j], B[i, k] = B[i, k], B[i, j]
    B = B[rng.permutation(n)]
    pA, pB = rng.uniform(0, 1000, (n, 2)), rng.uniform(0, 1000, (n, 2))
    ctx.set_option("match_six_bit", 0)
    try:
        s = ctx.imageset(np.array([n, n], np.int32), np.concatenate([A, B]), np.concatenate([pA, pB]))
        for thr in (0.95, 0.97):
            res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=thr, max_dist_sq=1e8)
            off, pairs = res.offsets(), res.pairs()
            exp, _, _ = oracle.match(A, pA, B, pB, threshold=thr, max_dist_sq=1e8)
            rev, _, _ = oracle.match(B, pB, A, pA, threshold=thr, max_dist_sq=1e8)
            assert 50 < len(exp) < n - 50, (thr, len(exp))
            assert np.array_equal(pairs[off[0]:off[1]], exp) and np.array_equal(pairs[off[1]:off[2]], rev), thr
            assert res.stats_ex()[1:] == (0, 0, 0)  # (the int8 lists held them: no later pass)
            res.close()
        s.close()
    finally:
        ctx.set_option("match_six_bit", 1)
