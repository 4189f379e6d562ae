"""Generates tests/golden/*.npz: small input/expected-output vectors for the matcher and the
RANSAC path, produced by the CPU restatement in oracle/ (the reference ships no fixtures and
cannot be built here, so these pin the restatement against regressions, not against the
reference).  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_api as oracle  # noqa: E402
from sfm_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def matcher_cases():
    cases = {}

    def add(name, dA, pA, dB, pB, thr, gate, metric=0):
        pairs, m1, m2 = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric)
        cases[name] = dict(descA=dA, ptsA=pA, descB=dB, ptsB=pB, threshold=np.float32(thr),
                           max_dist_sq=np.float32(gate), metric=np.int32(metric), pairs=pairs, matches1=m1,
                           matches2=m2)

    d, p, _ = synth.image_set(2, 300, seed=101)
    add("sift_257x300_main_options", d[:257], p[:257], d[300:], p[300:], 0.95, 100.0 * 100.0)
    add("sift_300x257_default_gate", d[300:], p[300:], d[:257], p[:257], 0.95, 1e8)
    g, q, _ = synth.image_set(2, 120, seed=102, kind="gauss")
    add("gauss_100x120_default_options", g[:100], q[:100], g[120:], q[120:], 0.5, 1e8)
    add("gauss_100x120_dot_metric", g[:100], q[:100], g[120:], q[120:], 20.0, 1e8, metric=1)
    t, r, _ = synth.image_set(2, 64, seed=103)
    tA, tB = t[:64].copy(), t[64:].copy()
    tA[10] = tA[40] = tA[7]
    tB[5] = tB[33] = tB[60] = tA[7]     # ties: lowest index must win on both sides
    tA[3] = 0                           # zero descriptor: NaN similarity never matches
    tB[8] = 0
    add("ties_and_zero_descriptor_64x64", tA, r[:64], tB, r[64:], 0.9, 1e8)
    return cases


def ransac_cases():
    cases = {}
    for name, n, H, ratio, adaptive in [("n64_h64", 64, 64, 0.6, True), ("n500_h128_15pct", 500, 128, 0.15, True),
                                        ("n8_h8", 8, 8, 1.0, False)]:
        tv = synth.two_view(n, inlier_ratio=ratio, seed=200 + n)
        S = synth.sample_table(n, H, seed=300 + n)
        r = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=adaptive)
        pose12, cand = oracle.fit_hypotheses(tv["x"], tv["y"], S)
        cases[name] = dict(x=tv["x"], y=tv["y"], K=tv["K"], samples=S, max_sq_err=1.0, success_prob=0.999,
                           adaptive=np.int32(adaptive), inliers=r["inliers"], best_h=np.int32(r["best_h"]),
                           pose=r["pose"], counts=r["counts"], toterr=r["toterr"],
                           iterations=np.int32(r["iterations"]), pose12=pose12, candidate=cand)
    return cases


def loss_cases():
    K = synth.default_K()
    pose = np.eye(4)
    pose[:3, :3] = synth._rodrigues(np.array([0.2, 1.0, -0.3]), 0.15)
    pose[:3, 3] = [0.8, 0.1, 0.59160798]
    rows = []
    for obs1, obs2, pt in [
        ([0.1, -0.05], [0.12, -0.04], [0.5, -0.25, 5.0]),
        ([0.0, 0.0], [0.0, 0.0], [0.0, 0.0, 0.0]),           # lambda1 = 0: 0/0 in camera 1 -> NaN propagates
        ([0.3, 0.2], [0.1, 0.1], [-0.6, -0.4, -2.0]),         # negative depth is not rejected
        ([0.3, 0.2], [0.1, 0.1], [np.inf, np.inf, np.inf]),   # point at infinity
    ]:
        v = oracle.loss(K, obs1, obs2, pt, pose)
        rows.append(dict(obs1=obs1, obs2=obs2, pt=pt, loss=v))
    return dict(K=K, pose=pose, obs1=np.array([r["obs1"] for r in rows]), obs2=np.array([r["obs2"] for r in rows]),
                pt=np.array([r["pt"] for r in rows]), loss=np.array([r["loss"] for r in rows]))


def ratio_cases():
    """matchTwoSided plus the second-best ratio test (an extension: the reference has none)."""
    sys.path.insert(0, os.path.dirname(OUT))
    from test_match_ratio_gpu import crowded_pair
    cases = {}
    for name, seed, n, integer, thr, gate, ratio in [("crowded_400_r080", 5, 400, True, 0.9, 400.0 ** 2, 0.8),
                                                     ("crowded_300_float_r095", 6, 300, False, 0.9, 1e8, 0.95)]:
        A, pA, B, pB = crowded_pair(seed, n=n, integer=integer)
        pairs, m1, m2 = oracle.match(A, pA, B, pB, threshold=thr, max_dist_sq=gate, ratio=ratio)
        cases[name] = dict(descA=A, ptsA=pA, descB=B, ptsB=pB, threshold=np.float32(thr), max_dist_sq=np.float32(gate),
                           ratio=np.float32(ratio), pairs=pairs, matches1=m1, matches2=m2)
    return cases


def l2_cases():
    """Mutual nearest neighbours under the squared Euclidean distance, with and without the second-best test (an
    extension: the reference's metrics are cosine and dot)."""
    sys.path.insert(0, os.path.dirname(OUT))
    from test_match_l2_gpu import scaled_sift
    cases = {}
    for name, n, seed, spread, T, gate, ratio in [("sift_400_spread20", 400, 21, 0.2, 250.0 ** 2, 1e8, 0.0),
                                                  ("sift_300_spread50_r080_gate", 300, 22, 0.5, 260.0 ** 2, 150.0 ** 2, 0.8)]:
        A, pA, B, pB = scaled_sift(n, seed, spread)
        pairs, m1, m2 = oracle.match_l2(A, pA, B, pB, T, max_dist_sq=gate, ratio=ratio)
        cases[name] = dict(descA=A, ptsA=pA, descB=B, ptsB=pB, max_distance_squared=np.float32(T), max_dist_sq=np.float32(gate),
                           ratio=np.float32(ratio), pairs=pairs, matches1=m1, matches2=m2)
    return cases


def hamming_cases():
    """Binary descriptors under the Hamming distance (an extension: the reference has none)."""
    sys.path.insert(0, os.path.dirname(OUT))
    from test_hamming_gpu import binary_views
    cases = {}
    for name, seed, n, nbytes, maxd, gate in [("256bit_500", 1, 500, 32, 48, 1e8), ("128bit_300_gate", 2, 300, 16, 30, 80.0 ** 2),
                                              ("512bit_200", 3, 200, 64, 110, 1e8)]:
        descs, ptss, _ = binary_views(seed, 2, n, nbytes)
        pairs, m1, m2 = oracle.match_hamming(descs[0], ptss[0], descs[1], ptss[1], max_distance=maxd, max_dist_sq=gate)
        cases[name] = dict(descA=descs[0], ptsA=ptss[0], descB=descs[1], ptsB=ptss[1], max_distance=np.int32(maxd),
                           max_dist_sq=np.float32(gate), pairs=pairs, matches1=m1, matches2=m2)
    return cases


def chain_cases():
    """Scale chaining and running point average, scene.cpp:301-421."""
    sys.path.insert(0, os.path.dirname(OUT))
    from chain_cases import sequence
    cases = {}
    for name, seed, steps, tracks, cap in [("12_steps_400_tracks", 7, 12, 400, 15.0), ("5_steps_capped", 8, 5, 150, 0.6)]:
        off, ids, pts, poses = sequence(seed, steps, tracks, behind=0.1, far=0.05)
        start = np.eye(4)
        start[:3, 3] = [0.5, -1.0, 2.0]
        r = oracle.scale_chain(off, ids, pts, poses, start_pose=start, max_translation=cap, n_tracks=tracks)
        cases[name] = dict(offsets=off, track_ids=ids, points=pts, poses=poses, start=start, max_translation=cap,
                           n_tracks=np.int32(tracks), **r)
    return cases


def refit_cases():
    """eightPointAlgorithm on all points, eightPoint.cpp:57-223."""
    cases = {}
    for name, n, seed in [("n40", 40, 11), ("n600", 600, 12)]:
        tv = synth.two_view(n, inlier_ratio=1.0, seed=seed, noise_px=0.2)
        pose, points = oracle.eight_point(tv["x"], tv["y"])[:2]
        cases[name] = dict(x=tv["x"], y=tv["y"], pose=pose, points=points)
    return cases


def main():
    oracle.build()
    groups = [("match_", matcher_cases), ("ransac_", ransac_cases), ("ratio_", ratio_cases), ("l2_", l2_cases), ("hamming_", hamming_cases),
              ("chain_", chain_cases), ("refit_", refit_cases)]
    for prefix, make in groups:
        for name, c in make().items():
            path = os.path.join(OUT, f"{prefix}{name}.npz")
            if not os.path.exists(path) or "--all" in sys.argv:  # committed vectors stay as they are
                np.savez_compressed(path, **c)
    if not os.path.exists(os.path.join(OUT, "loss_cases.npz")) or "--all" in sys.argv:
        np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **loss_cases())
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
