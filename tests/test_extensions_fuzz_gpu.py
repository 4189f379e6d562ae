"""Randomised sweeps of the stages added around the reference's matcher and RANSAC — the second-best
ratio test, the binary matcher and the scale chain — against the CPU restatements.  Sizes, thresholds,
gates, capacities, chunking and kernel choices come from a seeded generator.  Bar: bit-exact."""
import os

import numpy as np
import pytest

from sfm_amd import capi, synth
from chain_cases import sequence

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("SFM_FUZZ_CASES", "48"))


@pytest.mark.parametrize("seed", range(N_CASES // 2))
def test_random_ratio_case(ctx, oracle, seed):
    rng = np.random.default_rng(31000 + seed)
    nA, nB = (int(v) for v in rng.choice([1, 9, 64, 255, 257, 600, 1100], 2))
    n = max(nA, nB)
    d, p, _ = synth.image_set(2, n, shared_frac=float(rng.uniform(0.2, 0.9)), seed=int(rng.integers(1 << 30)))
    dA, pA, dB, pB = d[:nA].copy(), p[:nA].copy(), d[n:n + nB].copy(), p[n:n + nB].copy()
    kind = rng.choice(["plain", "twins", "dupes", "float"])
    if kind == "twins":  # near copies inside B: second bests close to the bests
        k = rng.integers(0, nB, nB // 3 + 1)
        dB[k] = np.clip(dB[rng.integers(0, nB, len(k))] + rng.integers(-3, 4, (len(k), 128)), 0, 255)
    elif kind == "dupes":
        dB[rng.integers(0, nB, nB // 4 + 1)] = dB[0]
    elif kind == "float":
        dA = dA * 0.37 + 0.11
        dB = dB * 1.9 + 0.05
    thr = float(rng.choice([0.95, 0.9, 0.7, 0.3, 0.0]))
    gate = float(rng.choice([1e8, 300.0 ** 2, 60.0 ** 2]))
    ratio = float(rng.choice([0.5, 0.7, 0.8, 0.9, 0.99, 1.0]))
    ctx.set_option("sim_variant", int(rng.choice([0, 1, 1])))
    ctx.set_option("cand_capacity", int(rng.choice([524288, 524288, 8192, 512, 0])))
    try:
        s = ctx.imageset(np.array([nA, nB], np.int32), np.concatenate([dA, dB]), np.concatenate([pA, pB]))
        res = s.match(np.array([[0, 1], [1, 0]], np.int32), threshold=thr, max_dist_sq=gate, ratio=ratio)
        off, pairs = res.offsets(), res.pairs()
        res.close()
        s.close()
    finally:
        ctx.set_option("sim_variant", 1)
        ctx.set_option("cand_capacity", 524288)
    exp, _, _ = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, ratio=ratio)
    assert np.array_equal(pairs[off[0]:off[1]], exp), "A->B list differs"
    exp, _, _ = oracle.match(dB, pB, dA, pA, threshold=thr, max_dist_sq=gate, ratio=ratio)
    assert np.array_equal(pairs[off[1]:off[2]], exp), "B->A list differs"


@pytest.mark.parametrize("seed", range(N_CASES // 2))
def test_random_hamming_case(ctx, oracle, seed):
    rng = np.random.default_rng(32000 + seed)
    nbytes = int(rng.choice([16, 32, 32, 32, 64]))
    n_img = int(rng.integers(2, 5))
    counts = rng.choice([0, 1, 63, 256, 257, 700, 1300], n_img).astype(np.int32)
    pool = rng.integers(0, 256, size=(int(counts.max()) + 50, nbytes), dtype=np.uint8)
    if rng.random() < 0.3:
        pool &= 0x11  # few distinct bits: equal distances everywhere
    descs, ptss = [], []
    for c in counts:
        pick = rng.choice(len(pool), size=int(c), replace=False)
        d = pool[pick] ^ np.packbits(rng.random((int(c), nbytes * 8)) < float(rng.choice([0.0, 0.03, 0.1])), axis=1)
        descs.append(d.astype(np.uint8))
        ptss.append(rng.uniform(0, 500, size=(int(c), 2)))
    pl = rng.integers(0, n_img, (int(rng.integers(1, 9)), 2)).astype(np.int32)
    maxd = int(rng.choice([0, 5, nbytes, 2 * nbytes, 4 * nbytes - 1, 4 * nbytes, 8 * nbytes]))
    gate = float(rng.choice([1e8, 150.0 ** 2]))
    ctx.set_option("hamming_mfma", int(rng.integers(2)))
    ctx.set_option("cand_capacity", int(rng.choice([524288, 4096, 256])))
    ctx.set_option("pairs_per_launch", int(rng.choice([1, 3, 2048])))
    try:
        s = capi.BinarySet(ctx, counts, np.concatenate(descs + [np.zeros((0, nbytes), np.uint8)]),
                           np.concatenate(ptss + [np.zeros((0, 2))]))
        res = s.match(pl, max_distance=maxd, max_dist_sq=gate)
        off, pairs = res.offsets(), res.pairs()
        res.close()
        s.close()
    finally:
        ctx.set_option("hamming_mfma", 1)
        ctx.set_option("cand_capacity", 524288)
        ctx.set_option("pairs_per_launch", 2048)
    for k, (a, b) in enumerate(pl):
        exp, _, _ = oracle.match_hamming(descs[a], ptss[a], descs[b], ptss[b], max_distance=maxd, max_dist_sq=gate)
        assert np.array_equal(pairs[off[k]:off[k + 1]], exp), f"pair {k} ({a}, {b}) of {len(pl)}"


@pytest.mark.parametrize("seed", range(max(N_CASES // 4, 4)))
def test_random_chain_case(ctx, oracle, seed):
    rng = np.random.default_rng(33000 + seed)
    n_steps = int(rng.choice([1, 2, 3, 9, 30]))
    n_tracks = int(rng.choice([1, 7, 200, 3000]))
    off, ids, pts, poses = sequence(int(rng.integers(1 << 30)), n_steps, n_tracks, visible=float(rng.uniform(0.3, 1.0)),
                                    shuffle=bool(rng.integers(2)), behind=float(rng.choice([0.0, 0.1, 0.5])),
                                    far=float(rng.choice([0.0, 0.2])))
    cap = float(rng.choice([15.0, 0.8, 1e-3]))
    start = np.eye(4)
    start[:3, 3] = rng.normal(size=3)
    ctx.set_option("chain_table_entries", int(rng.choice([1 << 26, 2, 3 * max(n_tracks, 1) + 1])))
    try:
        got = ctx.scale_chain(off, ids, pts, poses, start_pose=start, max_translation=cap, n_tracks=n_tracks)
    finally:
        ctx.set_option("chain_table_entries", 1 << 26)
    ids2, pts2 = ids.copy(), pts.copy()
    for s in range(n_steps):  # the restatement walks ascending id lists
        o = np.argsort(ids[off[s]:off[s + 1]], kind="stable")
        ids2[off[s]:off[s + 1]], pts2[off[s]:off[s + 1]] = ids[off[s]:off[s + 1]][o], pts[off[s]:off[s + 1]][o]
    exp = oracle.scale_chain(off, ids2, pts2, poses, start_pose=start, max_translation=cap, n_tracks=n_tracks)
    for k in ("extrinsics", "points3d", "counts", "scale"):
        assert np.array_equal(got[k], exp[k]), k
