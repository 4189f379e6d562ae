"""Randomised sweep of the matcher against the CPU restatement (reference submodules/match/match.cpp:24-94):
sizes, descriptor kinds, thresholds, pixel gates, metrics, candidate capacities and both bounds drawn from a
seeded generator.  Bar: bit-exact pair lists and bit-exact matches1 / matches2."""
import os

import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu


def _case(rng):
    nA, nB = (int(v) for v in rng.choice([1, 7, 63, 64, 65, 200, 255, 256, 257, 511, 700, 1025, 1500], 2))
    kind = rng.choice(["sift", "sift_floor", "gauss", "mixed", "dupes"])
    dim = int(rng.choice([128, 128, 128, 96, 17]))
    n = max(nA, nB)
    d, p, _ = synth.image_set(2, n, dim=dim, shared_frac=float(rng.uniform(0.0, 0.8)), seed=int(rng.integers(1 << 30)),
                              kind="gauss" if kind == "gauss" else "sift")
    dA, pA, dB, pB = d[:nA].copy(), p[:nA].copy(), d[n:n + nB].copy(), p[n:n + nB].copy()
    if kind == "sift_floor":
        dA, dB = np.maximum(dA, 12), np.maximum(dB, 12)
    elif kind == "mixed":  # some rows scaled, one zero row, one huge row
        dA[::3] *= 0.37
        dB[::5] *= 41.0
        dA[rng.integers(nA)] = 0
        dB[rng.integers(nB)] *= 1e6
    elif kind == "dupes":  # exact duplicates: ties everywhere
        dA[rng.integers(0, nA, nA // 2 + 1)] = dA[0]
        dB[rng.integers(0, nB, nB // 2 + 1)] = dA[0]
    thr = float(rng.choice([0.95, 0.9, 0.8, 0.5, 0.2, 0.0, -1.0, 0.999, 1.0]))
    gate = float(rng.choice([1e8, 1e8, 200.0 ** 2, 40.0 ** 2]))
    metric = int(rng.choice([0, 0, 0, 1]))
    if metric == 1:
        thr = float(rng.choice([0.0, 1e4, -5.0, 3e5]))
    return dA, pA, dB, pB, thr, gate, metric


@pytest.mark.parametrize("seed", range(int(os.environ.get("SFM_FUZZ_CASES", "48"))))
def test_random_case(ctx, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    dA, pA, dB, pB, thr, gate, metric = _case(rng)
    ctx.set_option("sim_variant", int(rng.choice([0, 1, 1])))
    ctx.set_option("cand_capacity", int(rng.choice([524288, 524288, 16384, 1024, 0])))
    try:
        got, g1, g2 = ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric, return_best=True)
    finally:
        ctx.set_option("sim_variant", 1)
        ctx.set_option("cand_capacity", 524288)
    exp, e1, e2 = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric)
    assert np.array_equal(g1, e1), "matches1 differs"
    assert np.array_equal(g2, e2), "matches2 differs"
    assert np.array_equal(got, exp), "pair list differs"


@pytest.mark.parametrize("seed", range(int(os.environ.get("SFM_FUZZ_BATCHES", "12"))))
def test_random_batch(ctx, oracle, seed):
    """Ragged image sets through the batch entry points: random counts (zeros included), random pair lists
    with repeats and self pairs, several chunks, byte and float tables."""
    rng = np.random.default_rng(5000 + seed)
    n_img = int(rng.integers(2, 7))
    counts = rng.choice([0, 1, 40, 256, 257, 600, 1100], n_img).astype(np.int32)
    n = max(int(counts.max()), 1)
    d, p, _ = synth.image_set(n_img, n, shared_frac=float(rng.uniform(0.1, 0.7)), seed=int(rng.integers(1 << 30)))
    kind = str(rng.choice(["sift", "sift", "common part", "gauss"]))
    if kind == "common part":  # unrelated cosines near the threshold: inside the margin of the 6-bit bound, of the int8 bound, or of neither
        base = synth.sift_like(rng, 1)[0]
        keep = rng.random(len(d)) < 0.3  # (rows kept from the usual set still match across images)
        mixed = np.floor(np.clip(base[None, :] * 0.9 + synth.sift_like(rng, len(d)) * float(rng.choice([0.42, 0.6, 0.8])), 0, 255)).astype(np.float32)  # (whole numbers: the table may go up as bytes)
        d = np.where(keep[:, None], d, mixed)
    elif kind == "gauss":
        d, p, _ = synth.image_set(n_img, n, shared_frac=float(rng.uniform(0.1, 0.7)), seed=int(rng.integers(1 << 30)), kind="gauss")
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)] + [np.zeros((0, 128), np.float32)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)] + [np.zeros((0, 2))])
    pl = rng.integers(0, n_img, (int(rng.integers(1, 25)), 2)).astype(np.int32)
    thr = float(rng.choice([0.95, 0.9, 0.6]))
    gate = float(rng.choice([1e8, 120.0 ** 2]))
    as_bytes = bool(rng.integers(2)) and kind != "gauss"
    ctx.set_option("match_six_bit", int(rng.choice([1, 1, 0])))  # (0: no 6-bit table, the int8 pass comes first)
    ctx.set_option("pairs_per_launch", int(rng.choice([1, 3, 2048])))
    ctx.set_option("match_list_observe", int(rng.choice([0, 1, 2, 2])))  # (2: calls of nine chunks and more size their lists as they go)
    try:
        s = ctx.imageset(counts, desc.astype(np.uint8) if as_bytes else desc, pts)
        res = s.match(pl, threshold=thr, max_dist_sq=gate)
        off, pairs = res.offsets(), res.pairs()
        if rng.random() < 0.5:  # the same call again on the same result object (its buffers are kept)
            res = s.match(pl, threshold=thr, max_dist_sq=gate, result=res)
            assert np.array_equal(off, res.offsets()) and np.array_equal(pairs, res.pairs())
    finally:
        ctx.set_option("pairs_per_launch", 2048)
        ctx.set_option("match_list_observe", 1)
        ctx.set_option("match_six_bit", 1)
    starts = np.concatenate([[0], np.cumsum(counts)])
    assert off[0] == 0 and off[-1] == len(pairs)
    for k, (a, b) in enumerate(pl):
        A, B = slice(starts[a], starts[a + 1]), slice(starts[b], starts[b + 1])
        exp, _, _ = oracle.match(desc[A], pts[A], desc[B], pts[B], threshold=thr, max_dist_sq=gate)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"pair {k} ({a},{b})"
    s.close()
