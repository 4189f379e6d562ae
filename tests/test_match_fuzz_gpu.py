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
