"""The matcher's two bounds (int8 MFMA, the default; fp16 MFMA) and the chain of passes behind
them: strict int8 -> strict fp16 -> filtered passes -> cell by cell.  Whatever route a pair takes,
the result is the CPU restatement's, bit for bit (reference submodules/match/match.cpp:24-94)."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu

FP16_BOUND, INT8_BOUND = 0, 1


@pytest.fixture(params=[FP16_BOUND, INT8_BOUND], ids=["fp16", "int8"])
def bound(ctx, request):
    ctx.set_option("sim_variant", request.param)
    yield request.param
    ctx.set_option("sim_variant", INT8_BOUND)


def run(ctx, oracle, dA, pA, dB, pB, thr, gate=1e8):
    counts = np.array([len(dA), len(dB)], np.int32)
    s = ctx.imageset(counts, np.concatenate([dA, dB]), np.concatenate([pA, pB]))
    res = s.match(np.array([[0, 1]], np.int32), threshold=thr, max_dist_sq=gate)
    exp, _, _ = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate)
    assert np.array_equal(res.pairs(), exp)
    st = res.stats_ex()
    s.close()
    return st, len(exp)


def images(nA, nB, kind="sift", seed=3, shared=0.3):
    n = max(nA, nB)
    d, p, _ = synth.image_set(2, n, shared_frac=shared, seed=seed, kind=kind)
    return d[:nA].copy(), p[:nA].copy(), d[n:n + nB].copy(), p[n:n + nB].copy()


@pytest.mark.parametrize("kind,thr", [("sift", 0.95), ("sift", 0.8), ("gauss", 0.9), ("gauss", 0.3), ("gauss", 0.0)])
@pytest.mark.parametrize("nA,nB", [(1500, 1100), (1, 300), (257, 513)])
def test_both_bounds(ctx, oracle, bound, kind, thr, nA, nB):
    dA, pA, dB, pB = images(nA, nB, kind=kind, seed=17)
    run(ctx, oracle, dA, pA, dB, pB, thr)
    run(ctx, oracle, dA, pA, dB, pB, thr, 150.0 * 150.0)


def test_int8_bound_is_selective_on_sift_like_data(ctx, oracle):
    """On the benchmark's descriptor distribution the wide int8 margin admits nothing but true matches."""
    ctx.set_option("sim_variant", INT8_BOUND)
    dA, pA, dB, pB = images(3000, 3000, seed=23)
    (cand, cells, filtered, _), n = run(ctx, oracle, dA, pA, dB, pB, 0.95)
    assert cells == 0 and filtered == 0 and n <= cand <= n + 20


def test_int8_flood_is_redone_with_the_fp16_bound(ctx, oracle):
    """Unrelated descriptors sit just below the threshold: inside the int8 margin (a few 1e-2) but outside
    the fp16 one (1.5e-3).  The int8 pass floods its lists, the fp16 strict pass redoes the pair, and the
    filtered passes are not needed."""
    ctx.set_option("sim_variant", INT8_BOUND)
    rng = np.random.default_rng(7)
    n = 1200
    base = synth.sift_like(rng, 1)[0]
    # every descriptor = common part + small individual part: cosines between unrelated ones ~0.93
    dA = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n) * 0.42, 0, 255).astype(np.float32)
    dB = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n) * 0.42, 0, 255).astype(np.float32)
    dB[:300] = dA[:300]
    pA = rng.uniform(0, 3000, (n, 2))
    pB = rng.uniform(0, 3000, (n, 2))
    cosines = (dA[:50] @ dB[500:550].T) / np.outer(np.linalg.norm(dA[:50], axis=1), np.linalg.norm(dB[500:550], axis=1))
    assert 0.91 < np.median(cosines) < 0.94, np.median(cosines)
    ctx.set_option("cand_capacity", 65536)
    try:
        (cand, cells, filtered, redone), m = run(ctx, oracle, dA, pA, dB, pB, 0.95)
    finally:
        ctx.set_option("cand_capacity", 524288)
    assert m >= 300 and cells == 0 and filtered == 0 and redone == 1
    assert cand < 65536


def test_byte_table_equals_float_table(ctx, oracle):
    """sfm_imageset_create_u8 (bytes in) against the float table holding the same values and against the
    restatement; a table with one non-integer element takes the fp32 exact pass and agrees too."""
    ctx.set_option("sim_variant", INT8_BOUND)
    counts = np.array([900, 0, 1300, 700], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=91)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    assert np.array_equal(desc, desc.astype(np.uint8).astype(np.float32))
    pl = np.array([[0, 2], [2, 3], [3, 0], [1, 2], [2, 2]], np.int32)
    sf = ctx.imageset(counts, desc, pts)
    sb = ctx.imageset(counts, desc.astype(np.uint8), pts)
    desc2 = desc.copy()
    desc2[5, 7] += 0.5  # no longer byte-valued: fp32 rows in the exact pass
    sx = ctx.imageset(counts, desc2, pts)
    rf, rb, rx = (s.match(pl, threshold=0.9, max_dist_sq=300.0 ** 2) for s in (sf, sb, sx))
    assert np.array_equal(rf.offsets(), rb.offsets()) and np.array_equal(rf.pairs(), rb.pairs())
    starts = np.concatenate([[0], np.cumsum(counts)])
    off, pairs = rf.offsets(), rf.pairs()
    offx, pairsx = rx.offsets(), rx.pairs()
    for k, (a, b) in enumerate(pl):
        A, B = slice(starts[a], starts[a + 1]), slice(starts[b], starts[b + 1])
        exp, _, _ = oracle.match(desc[A], pts[A], desc[B], pts[B], threshold=0.9, max_dist_sq=300.0 ** 2)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp)
        exp2, _, _ = oracle.match(desc2[A], pts[A], desc2[B], pts[B], threshold=0.9, max_dist_sq=300.0 ** 2)
        assert np.array_equal(pairsx[offx[k]:offx[k + 1]], exp2)
    for s in (sf, sb, sx):
        s.close()


@pytest.mark.parametrize("thr", [0.5, 0.0])
def test_filtered_passes_at_benchmark_size(ctx, oracle, thr):
    """8192 x 8192 with a loose threshold: every stage of the chain at full size (256 slices per pair,
    both orientations of the filtered pass), against the restatement."""
    ctx.set_option("sim_variant", INT8_BOUND)
    d, p, _ = synth.image_set(2, 8192, seed=13)
    (cand, cells, filtered, _), n = run(ctx, oracle, d[:8192], p[:8192], d[8192:], p[8192:], thr, 500.0 ** 2)
    assert filtered == 1 and cells == 0 and n > 2000 and cand < 8192 * 64
