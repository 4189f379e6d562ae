"""Parity of the HIP RANSAC path (through the C-ABI) with the CPU restatement of
solve::RANSAC + eightPointAlgorithmFromSubset + the default loss (reference
submodules/solve/RANSAC.cpp:146-270, eightPoint.cpp:225-372, solve.hpp:85-95,119-124).
Bar: identical inlier sets, identical per-hypothesis inlier counts, identical winner,
models within 1e-6 (observed: bit-identical poses and error sums)."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu


def assert_same_run(got, exp, pose_tol=1e-6):
    assert np.array_equal(got["counts"], exp["counts"]), "per-hypothesis inlier counts differ"
    assert got["best_h"] == exp["best_h"]
    assert got["iterations"] == exp["iterations"]
    assert np.array_equal(got["inliers"], exp["inliers"]), "inlier sets differ"
    assert np.allclose(got["toterr"], exp["toterr"], rtol=1e-12, atol=0), "error sums differ"
    assert np.abs(got["pose"] - exp["pose"]).max() <= pose_tol


@pytest.mark.parametrize("n,H", [(8, 16), (64, 256), (2000, 1024), (333, 100)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_ransac_pair_matches_restatement(ctx, oracle, n, H, adaptive):
    tv = synth.two_view(n, inlier_ratio=0.5 if n < 100 else 0.3, seed=100 + n)
    S = synth.sample_table(n, H, seed=n)
    got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=adaptive)
    exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=adaptive)
    assert_same_run(got, exp)


def test_ransac_stress_ratio(ctx, oracle):
    # 15 % inliers (the RANSAC stress shape, scaled down): the adaptive cap never applies
    tv = synth.two_view(3000, inlier_ratio=0.15, seed=5)
    S = synth.sample_table(3000, 2048, seed=6)
    got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    assert_same_run(got, exp)
    assert got["iterations"] == 2048


def test_benchmark_shape_against_restatement(ctx, oracle):
    """One pair of the benchmark's RANSAC shape at full size (20 000 correspondences, 15 % inliers, 4096
    sample rows, the benchmark's generator): per-row inlier counts and error sums (RANSAC.cpp:213-221), the
    winner and its inlier list are those of the CPU restatement, bit for bit, with the pre-test on and off."""
    tv = synth.two_view(20000, inlier_ratio=0.15, seed=[0, 0])
    S = synth.sample_table(20000, 4096, seed=[7, 0, 0])
    exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    assert exp["iterations"] == 4096
    for flt in (1, 0):
        ctx.set_option("ransac_filter", flt)
        try:
            got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        finally:
            ctx.set_option("ransac_filter", 1)
        assert_same_run(got, exp, pose_tol=0.0)
        assert np.array_equal(got["toterr"].view(np.uint64), exp["toterr"].view(np.uint64))
    # the batch entry point (slices of rows, no per-row outputs) keeps the same winner
    b = ctx.ransac_batch(np.array([0, 20000], np.int64), tv["x"], tv["y"], tv["K"], S, 4096, max_sq_err=1.0,
                         success_prob=0.999, adaptive=True)
    best = b.best()
    assert best["best_h"][0] == exp["best_h"] and best["count"][0] == len(exp["inliers"])
    assert np.array_equal(b.inliers()[1], exp["inliers"])
    b.close()


def test_poses_bit_identical_to_restatement(ctx, oracle):
    tv = synth.two_view(500, inlier_ratio=0.6, seed=9)
    S = synth.sample_table(500, 512, seed=10)
    got = ctx.eight_point_fit(tv["x"], tv["y"], S)
    exp, _ = oracle.fit_hypotheses(tv["x"], tv["y"], S)
    assert np.abs(got - exp).max() <= 1e-9
    same = np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    print("poses bit-identical:", same)
    # every pose is a rotation and a unit translation
    R = got[:, :9].reshape(-1, 3, 3)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-9)
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-9)
    assert np.allclose(np.linalg.norm(got[:, 9:], axis=1), 1.0, atol=1e-9)


def test_sampson_loss(ctx, oracle):
    tv = synth.two_view(800, inlier_ratio=0.4, seed=3)
    S = synth.sample_table(800, 256, seed=4)
    got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, adaptive=False, loss=capi.LOSS_SAMPSON)
    exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, adaptive=False, loss_kind=1)
    assert_same_run(got, exp)


def test_loss_special_values(ctx, oracle):
    """Degenerate correspondences: identical points (parallel rays, det = 0 -> NaN/inf depth),
    a point at the epipole; the loss's NaN rules (std::max keeps loss1 when loss2 is NaN) decide."""
    tv = synth.two_view(200, inlier_ratio=0.7, seed=13)
    x, y = tv["x"].copy(), tv["y"].copy()
    x[5] = y[5] = [0.0, 0.0]
    x[6] = [0.1, 0.2]
    y[6] = [0.1, 0.2]
    x[7] = [1e8, -1e8]
    y[8] = [np.inf, 0.0]
    x[9] = [np.nan, 0.0]
    S = synth.sample_table(200, 200, seed=2)
    S = S[~np.isin(S, [7, 8, 9]).any(axis=1)]
    got = ctx.ransac_pair(x, y, tv["K"], S, max_sq_err=1.0, adaptive=False)
    exp = oracle.ransac(x, y, tv["K"], S, max_sq_err=1.0, adaptive=False)
    assert_same_run(got, exp)


def test_guards(ctx):
    tv = synth.two_view(7, inlier_ratio=1.0, seed=1)
    S = np.zeros((4, 8), np.int32)
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S)
    assert len(r["inliers"]) == 0 and r["best_h"] == -1          # RANSAC.cpp:148-149
    r = ctx.ransac_pair(tv["x"], tv["y"][:5], tv["K"], S)
    assert len(r["inliers"]) == 0
    tv = synth.two_view(20, inlier_ratio=1.0, seed=1)
    with pytest.raises(capi.SfmError) as e:
        ctx.ransac_pair(tv["x"], tv["y"], tv["K"], np.zeros((4, 9), np.int32), min_n=9)
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_no_hypothesis_with_inliers(ctx, oracle):
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-0.5, 0.5, (50, 2)), rng.uniform(-0.5, 0.5, (50, 2))
    S = synth.sample_table(50, 32, seed=1)
    got = ctx.ransac_pair(x, y, synth.default_K(), S, max_sq_err=1e-12, adaptive=True)
    exp = oracle.ransac(x, y, synth.default_K(), S, max_sq_err=1e-12, adaptive=True)
    assert_same_run(got, exp)


def test_batch_ragged(ctx, oracle):
    sizes = [500, 7, 64, 1500, 8, 0, 900]
    H = 192
    xs, ys, Ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        if n >= 8:
            tv = synth.two_view(n, inlier_ratio=0.4, seed=40 + k)
            xs.append(tv["x"]); ys.append(tv["y"])
            Ss.append(synth.sample_table(n, H, seed=50 + k))
        else:
            xs.append(np.zeros((n, 2))); ys.append(np.zeros((n, 2)))
            Ss.append(np.zeros((H, 8), np.int32))
        off.append(off[-1] + n)
    K = synth.default_K()
    b = ctx.ransac_batch(off, np.concatenate(xs), np.concatenate(ys), K, np.concatenate(Ss), H, max_sq_err=1.0,
                         success_prob=0.99, adaptive=True)
    best = b.best()
    ioff, inl = b.inliers()
    for k, n in enumerate(sizes):
        exp = oracle.ransac(xs[k], ys[k], K, Ss[k], max_sq_err=1.0, success_prob=0.99, adaptive=True)
        assert best["best_h"][k] == exp["best_h"], k
        assert best["count"][k] == len(exp["inliers"])
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], exp["inliers"])
        if n >= 8:
            assert best["iterations"][k] == exp["iterations"]
            assert np.abs(best["pose12"][k, :9].reshape(3, 3) - exp["pose"][:3, :3]).max() <= 1e-6
            assert np.abs(best["pose12"][k, 9:] - exp["pose"][:3, 3]).max() <= 1e-6
    b.close()


def test_full_size_recovers_planted_pose(ctx):
    """20 000 correspondences x 4096 hypotheses (the RANSAC stress shape) at 60 % inliers, where
    all-inlier samples exist: properties instead of the oracle.  The winner's inliers are planted
    ones, the pose is the planted pose up to the accuracy of an 8-point fit, the list is ascending,
    and a batch of copies of the pair gives the same answer for every copy."""
    n, H = 20000, 4096
    tv = synth.two_view(n, inlier_ratio=0.6, seed=77)
    S = synth.sample_table(n, H, seed=78)
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=4.0, success_prob=0.999, adaptive=False)
    inl = r["inliers"]
    truth = tv["inlier_mask"]
    assert len(inl) > 0.2 * truth.sum()
    assert truth[inl].mean() > 0.9
    assert np.all(np.diff(inl) > 0)
    assert len(inl) == r["counts"].max() == r["counts"][r["best_h"]]
    R, T = r["pose"][:3, :3], r["pose"][:3, 3]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ tv["R"].T) - 1) / 2, -1, 1)))
    assert ang < 5.0 and abs(T @ tv["T"]) > 0.9
    off = np.arange(4) * n
    b = ctx.ransac_batch(off, np.tile(tv["x"], (3, 1)), np.tile(tv["y"], (3, 1)), tv["K"], np.tile(S, (3, 1)), H,
                         max_sq_err=4.0, success_prob=0.999, adaptive=False)
    best = b.best()
    ioff, binl = b.inliers()
    for k in range(3):
        assert best["best_h"][k] == r["best_h"] and np.array_equal(binl[ioff[k]:ioff[k + 1]], inl)
    b.close()


def test_stress_shape_is_deterministic(ctx):
    """The 15 % / 20 000 / 4096 shape itself: no all-inlier sample is expected, so only
    run-to-run identity and the selection rule (max count, then min error, then first) are checked."""
    n, H = 20000, 4096
    tv = synth.two_view(n, inlier_ratio=0.15, seed=5)
    S = synth.sample_table(n, H, seed=6)
    a = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    b = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    assert a["iterations"] == H  # w^8 < 1e-6: the cap of RANSAC.cpp:248 never applies
    assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["toterr"], b["toterr"])
    assert np.array_equal(a["inliers"], b["inliers"]) and a["best_h"] == b["best_h"]
    c = a["counts"]
    top = np.flatnonzero(c == c.max())
    assert a["best_h"] == top[np.argmin(a["toterr"][top])]
    assert len(a["inliers"]) == c.max()


def test_pretest_never_changes_a_result(ctx, oracle):
    """The scoring kernel's epipolar pre-test only skips evaluations that cannot be inliers:
    with it on and off, per-hypothesis counts and error sums are identical, also for
    correspondences at the epipole, far outside the image, and with a loose threshold."""
    tv = synth.two_view(3000, inlier_ratio=0.5, seed=91, noise_px=1.0)
    x, y = tv["x"].copy(), tv["y"].copy()
    S = synth.sample_table(3000, 512, seed=92)
    # put some observations on / next to the epipoles of the planted pose and far away
    R, T = tv["R"], tv["T"]
    e1 = -R.T @ T
    e2 = T
    for k in range(20):
        x[k] = e1[:2] / e1[2] + (k % 5) * 1e-7
        y[20 + k] = e2[:2] / e2[2] + (k % 7) * 1e-9
    x[50] = [1e9, -1e9]
    y[51] = [1e7, 2e7]
    x[52] = y[52] = [0.0, 0.0]
    for thr in (1.0, 1e4, 1e-6):
        res = []
        for flt in (1, 0):
            ctx.set_option("ransac_filter", flt)
            res.append(ctx.ransac_pair(x, y, tv["K"], S, max_sq_err=thr, adaptive=False))
        ctx.set_option("ransac_filter", 1)
        assert np.array_equal(res[0]["counts"], res[1]["counts"])
        assert np.array_equal(res[0]["toterr"].view(np.uint64), res[1]["toterr"].view(np.uint64))
        assert np.array_equal(res[0]["inliers"], res[1]["inliers"])
        exp = oracle.ransac(x, y, tv["K"], S, max_sq_err=thr, adaptive=False)
        assert np.array_equal(res[0]["counts"], exp["counts"]) and np.array_equal(res[0]["inliers"], exp["inliers"])


@pytest.mark.parametrize("thr", [1.0, 25.0, 1e-2])
def test_pretest_at_the_skip_boundary(ctx, oracle, thr):
    """Observations displaced from their epipolar line by 0.8 ... 1.3 times sqrt(maxSquaredError), under
    hypotheses fitted to noise-free samples (so nearly the planted pose): the cells right at the threshold and
    right at the skip distance.  Counts, error sums and inliers with the pre-test on are those with it off
    and those of the CPU restatement."""
    n = 2400
    tv = synth.two_view(n, inlier_ratio=1.0, seed=301, noise_px=0.0)
    x, y, K = tv["x"].copy(), tv["y"].copy(), tv["K"]
    R, T = tv["R"], tv["T"]
    Tx = np.array([[0, -T[2], T[1]], [T[2], 0, -T[0]], [-T[1], T[0], 0]])
    E = Tx @ R
    fx, fy = K[0, 0], K[1, 1]
    x1h = np.concatenate([x, np.ones((n, 1))], axis=1)
    m = x1h @ E.T  # epipolar lines in normalised camera-2 coordinates
    # unit normal of the line in pixels, mapped back to normalised coordinates
    npx = np.stack([m[:, 0] / fx, m[:, 1] / fy], axis=1)
    npx /= np.linalg.norm(npx, axis=1, keepdims=True)
    d = np.linspace(0.8, 1.3, n - 400) * np.sqrt(thr) * np.where(np.arange(n - 400) % 2, 1.0, -1.0)
    y[400:, 0] += d * npx[400:, 0] / fx
    y[400:, 1] += d * npx[400:, 1] / fy
    S = synth.sample_table(400, 256, seed=302)  # samples among the untouched correspondences only
    res = []
    for flt in (1, 0):
        ctx.set_option("ransac_filter", flt)
        res.append(ctx.ransac_pair(x, y, K, S, max_sq_err=thr, adaptive=False))
    ctx.set_option("ransac_filter", 1)
    assert np.array_equal(res[0]["counts"], res[1]["counts"])
    assert np.array_equal(res[0]["toterr"].view(np.uint64), res[1]["toterr"].view(np.uint64))
    assert np.array_equal(res[0]["inliers"], res[1]["inliers"])
    exp = oracle.ransac(x, y, K, S, max_sq_err=thr, adaptive=False)
    assert np.array_equal(res[0]["counts"], exp["counts"]) and np.array_equal(res[0]["inliers"], exp["inliers"])
    # the band straddles the threshold: some displaced observations are inliers of the best row, most are not
    assert 400 < res[0]["counts"].max() < n


def test_pretest_is_off_for_unusual_intrinsics(ctx, oracle):
    tv = synth.two_view(600, inlier_ratio=0.5, seed=93)
    K = tv["K"].copy()
    K[2] = [1e-5, -2e-5, 1.0]  # last row not (0, 0, 1): observations are not plain pixels
    S = synth.sample_table(600, 128, seed=94)
    got = ctx.ransac_pair(tv["x"], tv["y"], K, S, max_sq_err=2.0, adaptive=False)
    exp = oracle.ransac(tv["x"], tv["y"], K, S, max_sq_err=2.0, adaptive=False)
    assert_same_run(got, exp)


@pytest.mark.parametrize("case", ["skew", "fx_ne_fy", "tiny_focal", "huge_focal", "tight_threshold", "negative_fx",
                                  "zero_threshold", "huge_threshold"])
def test_pretest_with_other_intrinsics(ctx, oracle, case):
    """The pre-test works in pixel metric with fx and fy folded into the hypothesis; whatever K is, counts
    and error sums stay those of evaluating every cell."""
    tv = synth.two_view(900, inlier_ratio=0.4, seed=190)
    K = tv["K"].copy()
    thr = 1.0
    if case == "skew":
        K[0, 1] = 3.5
    elif case == "fx_ne_fy":
        K[0, 0], K[1, 1] = 0.6 * K[0, 0], 1.7 * K[1, 1]
    elif case == "tiny_focal":
        K[0, 0] = K[1, 1] = 2.0
        K[0, 2] = K[1, 2] = 0.5
        thr = 1e-6
    elif case == "huge_focal":
        K[0, 0] = K[1, 1] = 4e5
    elif case == "tight_threshold":
        thr = 1e-4
    elif case == "negative_fx":
        K[0, 0] = -K[0, 0]
    elif case == "zero_threshold":
        thr = 0.0
    elif case == "huge_threshold":
        thr = 1e30
    S = synth.sample_table(900, 256, seed=191)
    got = ctx.ransac_pair(tv["x"], tv["y"], K, S, max_sq_err=thr, adaptive=False)
    exp = oracle.ransac(tv["x"], tv["y"], K, S, max_sq_err=thr, adaptive=False)
    assert_same_run(got, exp)


@pytest.mark.parametrize("first_slice", [64, 256, 1024])
def test_sliced_adaptive_batch_equals_one_pass(ctx, oracle, first_slice):
    """With the adaptive cap on, a batch scores its sample rows in growing slices and pairs whose in-order
    scan has stopped skip the later slices.  Winner, inlier list, error sum and iteration count must be
    those of scoring every row (RANSAC.cpp:182-258 is an in-order scan, rows past the stop are never read)."""
    H = 3000
    cases = [(300, 0.9, 0.2), (500, 0.5, 0.3), (2000, 0.15, 0.5), (8, 1.0, 0.0), (5, 1.0, 0.0), (900, 0.7, 0.4)]
    xs, ys, Ss, off = [], [], [], [0]
    for k, (n, ratio, noise) in enumerate(cases):
        tv = synth.two_view(max(n, 8), inlier_ratio=ratio, seed=100 + k, noise_px=noise)
        xs.append(tv["x"][:n]); ys.append(tv["y"][:n])
        Ss.append(synth.sample_table(max(n, 8), H, seed=200 + k) % max(n, 1))
        off.append(off[-1] + n)
    K = synth.default_K()
    x, y, S = np.concatenate(xs), np.concatenate(ys), np.stack(Ss)
    S[3] = np.tile(np.arange(8), (H, 1))  # the 8-point pair: every row is the same sample
    ctx.set_option("hyps_per_launch", first_slice)
    try:
        b = ctx.ransac_batch(np.array(off), x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        best = b.best()
        ioff, inl = b.inliers()
    finally:
        ctx.set_option("hyps_per_launch", 0)
    stopped_early = 0
    for k, (n, _, _) in enumerate(cases):
        e = oracle.ransac(x[off[k]:off[k + 1]], y[off[k]:off[k + 1]], K, S[k], max_sq_err=1.0, success_prob=0.999, adaptive=True)
        assert best["best_h"][k] == e["best_h"], k
        assert best["iterations"][k] == e["iterations"], k
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], e["inliers"]), k
        if e["best_h"] >= 0:
            assert best["toterr"][k] == e["toterr"][e["best_h"]], k
        stopped_early += e["iterations"] < first_slice
    assert stopped_early >= 2  # some pairs never see the later slices, others run all rows


@pytest.mark.parametrize("loss", [capi.LOSS_REPROJECTION, capi.LOSS_SAMPSON])
def test_split_scoring_is_bit_identical(ctx, oracle, loss):
    """Small launches cut every pair's correspondences over several blocks and add the inliers' errors up
    in a second kernel, in correspondence order: per-row counts and error sums must equal those of the
    one-block-per-row-group path and of the restatement, bit for bit (RANSAC.cpp:207-221)."""
    n, H = 3001, 700  # neither a multiple of the tile sizes
    tv = synth.two_view(n, inlier_ratio=0.6, seed=77, noise_px=0.3)
    S = synth.sample_table(n, H, seed=78)
    kw = dict(max_sq_err=1.0, success_prob=0.999, loss=loss)
    out = {}
    for split in (1, 0):
        ctx.set_option("ransac_split", split)
        try:
            out[split] = (ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, adaptive=False, **kw),
                          ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, adaptive=True, per_row=False, **kw))
        finally:
            ctx.set_option("ransac_split", 1)
    e = oracle.ransac(tv["x"], tv["y"], tv["K"], S, adaptive=False, loss_kind=loss, max_sq_err=kw["max_sq_err"],
                      success_prob=0.999)
    ea = oracle.ransac(tv["x"], tv["y"], tv["K"], S, adaptive=True, loss_kind=loss, max_sq_err=kw["max_sq_err"],
                       success_prob=0.999)
    assert e["counts"].max() > 100
    for split in (1, 0):
        full, adaptive = out[split]
        assert np.array_equal(full["counts"], e["counts"]) and np.array_equal(full["toterr"], e["toterr"]), split
        assert full["best_h"] == e["best_h"] and np.array_equal(full["inliers"], e["inliers"]), split
        assert adaptive["best_h"] == ea["best_h"] and adaptive["iterations"] == ea["iterations"], split
        assert np.array_equal(adaptive["inliers"], ea["inliers"]) and adaptive["counts"] is None, split


def test_repeated_pair_calls_reuse_their_buffers(ctx, oracle):
    """The single-pair entry points keep their device buffers in the context: a large call followed by
    smaller and different ones must not see stale rows."""
    for n, H, seed in [(1500, 512, 1), (40, 16, 2), (700, 2048, 3), (8, 1, 4)]:
        tv = synth.two_view(n, inlier_ratio=0.5, seed=seed)
        S = synth.sample_table(n, H, seed=seed + 10)
        got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        e = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        it = e["iterations"]
        assert got["best_h"] == e["best_h"] and np.array_equal(got["inliers"], e["inliers"])
        assert np.array_equal(got["counts"][:it], e["counts"][:it]) and np.array_equal(got["toterr"][:it], e["toterr"][:it])
        pose12 = ctx.eight_point_fit(tv["x"], tv["y"], S[:4])
        ref12, _ = oracle.fit_hypotheses(tv["x"], tv["y"], S[:4])
        assert np.array_equal(pose12, ref12)


def test_library_drawn_samples_and_time_cap(ctx, oracle):
    """samples = NULL: the library draws the table on the device from the context's seed with the generator
    documented in the header; the result equals the restatement's on the restated table.  A time limit that
    has already passed stops the scan after the first slice of rows (RANSAC.cpp:189-196)."""
    tv = synth.two_view(2500, inlier_ratio=0.35, seed=71)
    H = 4096
    ctx.set_option("ransac_seed", 99)
    try:
        got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], H, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
        S = oracle.sample_table(2500, H, 99, 0)
        exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        assert got["best_h"] == exp["best_h"] and got["iterations"] == exp["iterations"]
        assert np.array_equal(got["inliers"], exp["inliers"]) and np.array_equal(got["pose"], exp["pose"])
        ctx.set_option("ransac_seed", 0)  # clock-seeded, as the reference: still a valid run
        r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], H, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
        assert r["best_h"] >= 0 and r["iterations"] > 0 and len(r["inliers"]) >= 8
        ctx.set_option("ransac_seed", 99)
        ctx.set_option("ransac_max_time_ms", 1)
        ctx.set_option("hyps_per_launch", 64)  # slices of 64, 256, 1024, ... rows
        import time
        time.sleep(0.01)
        capped = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], H, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
        assert capped["iterations"] <= H
        first = oracle.ransac(tv["x"], tv["y"], tv["K"], S[:capped["iterations"]], max_sq_err=1.0, success_prob=0.999, adaptive=True)
        assert np.array_equal(capped["inliers"], first["inliers"])  # the best of the rows that were reached
    finally:
        ctx.set_option("ransac_seed", 0)
        ctx.set_option("ransac_max_time_ms", 0)
        ctx.set_option("hyps_per_launch", 0)


def test_eight_point_from_subset(ctx, oracle):
    """The whole of eightPointAlgorithmFromSubset (eightPoint.cpp:225-372): pose from 8 correspondences, all
    correspondences triangulated under it."""
    tv = synth.two_view(300, inlier_ratio=0.8, seed=19, noise_px=0.2)
    rows = synth.sample_table(300, 1, seed=20)[0]
    pose, pts = ctx.eight_point_from_subset(tv["x"][rows], tv["y"][rows], tv["x"], tv["y"])
    epose, epts, _, _ = oracle.eight_point_from_subset(tv["x"][rows], tv["y"][rows], tv["x"], tv["y"])
    assert np.array_equal(pose, epose) and np.array_equal(pts, epts)
    with pytest.raises(capi.SfmError) as e:
        ctx.eight_point_from_subset(tv["x"][:7], tv["y"][:7], tv["x"], tv["y"])
    assert e.value.code == capi.ERR_INVALID_ARG and "at least 8" in str(e.value)


@pytest.mark.parametrize("ratio,thr,n,H", [(0.15, 1.0, 5000, 1024), (0.5, 1.0, 3000, 512), (0.9, 4.0, 2100, 300),
                                           (0.3, 1e-2, 1999, 257), (0.6, 1e4, 1500, 256), (1.0, 1.0, 777, 130)])
def test_matrix_core_pretest_is_bit_identical(ctx, oracle, ratio, thr, n, H):
    """Unsplit scoring launches run the pre-test on the matrix cores: per-row counts, error sums and the inlier
    list are those of the packed fp32 test, of no pre-test at all, and of the CPU restatement -- also with
    observations at the epipoles, far outside the image and at the origin."""
    tv = synth.two_view(n, inlier_ratio=ratio, seed=1000 + n, noise_px=0.7)
    x, y = tv["x"].copy(), tv["y"].copy()
    S = synth.sample_table(n, H, seed=77 + H)
    R, T = tv["R"], tv["T"]
    e1, e2 = -R.T @ T, T
    for k in range(10):
        x[100 + k] = e1[:2] / e1[2] + (k % 5) * 1e-7
        y[120 + k] = e2[:2] / e2[2] + (k % 7) * 1e-9
    x[50] = [1e9, -1e9]
    y[51] = [1e7, 2e7]
    x[52] = y[52] = [0.0, 0.0]
    x[53] = [40.0, -3.0]  # ordinary pixels, but products beyond what the fp16 operands carry
    y[53] = [-2.0, 30.0]
    res = {}
    ctx.set_option("ransac_split", 0)
    try:
        # mfma: the correspondence side of the test prepared once per call; mfma_blocks: made by every block
        for name, flt, mf, gib in (("mfma", 1, 1, 64), ("mfma_blocks", 1, 1, 0), ("packed", 1, 0, 64), ("off", 0, 0, 64)):
            ctx.set_option("ransac_filter", flt)
            ctx.set_option("ransac_mfma", mf)
            ctx.set_option("ransac_mfma_ops_gib", gib)
            res[name] = ctx.ransac_pair(x, y, tv["K"], S, max_sq_err=thr, adaptive=False)
    finally:
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_filter", 1)
        ctx.set_option("ransac_mfma", 1)
        ctx.set_option("ransac_mfma_ops_gib", 64)
    for other in ("mfma_blocks", "packed", "off"):
        assert np.array_equal(res["mfma"]["counts"], res[other]["counts"]), other
        assert np.array_equal(res["mfma"]["toterr"].view(np.uint64), res[other]["toterr"].view(np.uint64)), other
        assert np.array_equal(res["mfma"]["inliers"], res[other]["inliers"]), other
    exp = oracle.ransac(x, y, tv["K"], S, max_sq_err=thr, adaptive=False)
    assert np.array_equal(res["mfma"]["counts"], exp["counts"]) and np.array_equal(res["mfma"]["inliers"], exp["inliers"])
    assert np.array_equal(res["mfma"]["toterr"].view(np.uint64), np.asarray(exp["toterr"]).view(np.uint64))


def test_matrix_core_pretest_over_a_ragged_batch(ctx, oracle):
    """A batch of pairs of unequal sizes (not multiples of the 16-correspondence steps or the 64-correspondence
    chunks; more pairs than one group of eight, fewer than two, so that both block orders of the kernel run; an
    empty pair and one below eight correspondences): prepared operands, operands made by the blocks and the packed
    fp32 test give the same rows, and the winners are the restatement's."""
    H = 520
    sizes = [1000, 17, 0, 4097, 64, 5, 1023, 2500, 129, 3333, 16]
    xs, ys, Ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        tv = synth.two_view(max(n, 8), inlier_ratio=(0.15, 0.5, 0.8)[k % 3], seed=300 + k, noise_px=0.5)
        xs.append(tv["x"][:n]); ys.append(tv["y"][:n])
        Ss.append(synth.sample_table(max(n, 8), H, seed=400 + k) % max(n, 1))
        off.append(off[-1] + n)
    K = synth.default_K()
    x, y, S = np.concatenate(xs), np.concatenate(ys), np.stack(Ss)
    res = {}
    ctx.set_option("ransac_split", 0)
    try:
        for name, mf, gib in (("prepared", 1, 64), ("blocks", 1, 0), ("packed", 0, 64)):
            ctx.set_option("ransac_mfma", mf)
            ctx.set_option("ransac_mfma_ops_gib", gib)
            b = ctx.ransac_batch(np.array(off), x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=False)
            res[name] = (b.best(), b.inliers())
            b.close()
    finally:
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_mfma", 1)
        ctx.set_option("ransac_mfma_ops_gib", 64)
    for other in ("blocks", "packed"):
        for key in ("best_h", "count", "iterations"):
            assert np.array_equal(res["prepared"][0][key], res[other][0][key]), (other, key)
        assert np.array_equal(res["prepared"][0]["toterr"].view(np.uint64), res[other][0]["toterr"].view(np.uint64)), other
        assert np.array_equal(res["prepared"][1][0], res[other][1][0]) and np.array_equal(res["prepared"][1][1], res[other][1][1]), other
    best, (ioff, inl) = res["prepared"]
    for k, n in enumerate(sizes):
        if n < 8:
            continue
        e = oracle.ransac(x[off[k]:off[k + 1]], y[off[k]:off[k + 1]], K, S[k], max_sq_err=1.0, success_prob=0.999, adaptive=False)
        assert best["best_h"][k] == e["best_h"], k
        if e["best_h"] >= 0:
            assert best["count"][k] == e["counts"][e["best_h"]], k
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], e["inliers"]), k


@pytest.mark.parametrize("ratio,thr,n,H", [(0.15, 1.0, 5000, 1024), (0.5, 1.0, 3000, 512), (0.9, 4.0, 2100, 300), (0.3, 1e-2, 1999, 257),
                                           (0.6, 1e4, 1500, 256), (1.0, 1.0, 777, 130)])
def test_matrix_core_pretest_with_the_sampson_distance(ctx, oracle, ratio, thr, n, H):
    """Sampson distance (extension): unsplit launches run a pre-test on the matrix cores too -- no inlier when
    num^2 >= c^2 (|(E x1h)_01|^2 + 1.001 |x2h|^2) / f^2.  Per-row counts, error sums and the inlier list equal those
    with the test off and the restatement's, also with observations at the epipoles, far outside the image and at
    the origin."""
    tv = synth.two_view(n, inlier_ratio=ratio, seed=2000 + n, noise_px=0.7)
    x, y = tv["x"].copy(), tv["y"].copy()
    S = synth.sample_table(n, H, seed=177 + H)
    R, T = tv["R"], tv["T"]
    e1, e2 = -R.T @ T, T
    for k in range(10):
        x[100 + k] = e1[:2] / e1[2] + (k % 5) * 1e-7
        y[120 + k] = e2[:2] / e2[2] + (k % 7) * 1e-9
    x[50] = [1e9, -1e9]
    y[51] = [1e7, 2e7]
    x[52] = y[52] = [0.0, 0.0]
    x[53] = [40.0, -3.0]
    y[53] = [-2.0, 30.0]
    res = {}
    ctx.set_option("ransac_split", 0)
    try:
        for name, mf, gib in (("mfma", 1, 64), ("mfma_blocks", 1, 0), ("off", 0, 64)):
            ctx.set_option("ransac_mfma", mf)
            ctx.set_option("ransac_mfma_ops_gib", gib)
            res[name] = ctx.ransac_pair(x, y, tv["K"], S, max_sq_err=thr, adaptive=False, loss=capi.LOSS_SAMPSON)
    finally:
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_mfma", 1)
        ctx.set_option("ransac_mfma_ops_gib", 64)
    for other in ("mfma_blocks", "off"):
        assert np.array_equal(res["mfma"]["counts"], res[other]["counts"]), other
        assert np.array_equal(res["mfma"]["toterr"].view(np.uint64), res[other]["toterr"].view(np.uint64)), other
        assert np.array_equal(res["mfma"]["inliers"], res[other]["inliers"]), other
    exp = oracle.ransac(x, y, tv["K"], S, max_sq_err=thr, adaptive=False, loss_kind=capi.LOSS_SAMPSON)
    assert np.array_equal(res["mfma"]["counts"], exp["counts"]) and np.array_equal(res["mfma"]["inliers"], exp["inliers"])
    assert np.array_equal(res["mfma"]["toterr"].view(np.uint64), np.asarray(exp["toterr"]).view(np.uint64))


@pytest.mark.parametrize("n_calc", [9, 12, 40, 300])
def test_eight_point_from_subset_with_more_sample_points(ctx, oracle, n_calc):
    """More than eight sample points (eightPoint.cpp:234-255 takes calcPoints1.size() rows): the pose is that of
    the n x 9 system, every correspondence is triangulated under it."""
    tv = synth.two_view(300, inlier_ratio=1.0, seed=23, noise_px=0.2)
    pose, pts = ctx.eight_point_from_subset(tv["x"][:n_calc], tv["y"][:n_calc], tv["x"], tv["y"])
    epose, epts, _, _ = oracle.eight_point_from_subset(tv["x"][:n_calc], tv["y"][:n_calc], tv["x"], tv["y"])
    assert np.abs(pose - epose).max() < 1e-8
    assert np.abs(pts - epts).max() <= 1e-6 * np.abs(epts).max()


def test_stress_batch_against_restatement(ctx, oracle):
    """The benchmark's RANSAC configuration as a batch: 64 pairs x 20 000 correspondences x 4096 sample rows,
    15 % inliers, adaptive cap on, the matrix-core pre-test reading its prepared operand table -- the slicing,
    the XCD placement and the table are the batch's own.  Four pairs sampled over the batch: winner, inlier
    count, error sum, iterations, pose and inlier list equal those of the CPU restatement (RANSAC.cpp:213-258)
    run on that pair alone."""
    P, n, H = 64, 20000, 4096
    xs, ys, Ss = [], [], []
    for k in range(P):
        tv = synth.two_view(n, inlier_ratio=0.15, seed=[3, k % 8])
        xs.append(tv["x"]); ys.append(tv["y"])
        Ss.append(synth.sample_table(n, H, seed=[4, k]))
    K = synth.default_K()
    off = np.arange(P + 1, dtype=np.int64) * n
    b = ctx.ransac_batch(off, np.concatenate(xs), np.concatenate(ys), K, np.concatenate(Ss), H, max_sq_err=1.0,
                         success_prob=0.999, adaptive=True)
    best = b.best()
    ioff, inl = b.inliers()
    for k in (0, 21, 42, 63):
        e = oracle.ransac(xs[k], ys[k], K, Ss[k], max_sq_err=1.0, success_prob=0.999, adaptive=True)
        assert best["best_h"][k] == e["best_h"] and best["count"][k] == len(e["inliers"]), k
        assert best["iterations"][k] == e["iterations"]
        assert best["toterr"][k] == e["toterr"][e["best_h"]]
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], e["inliers"])
        pose12 = np.concatenate([e["pose"][:3, :3].reshape(-1), e["pose"][:3, 3]])
        assert np.array_equal(best["pose12"][k].view(np.uint64), pose12.view(np.uint64))
    b.close()


def _heavy_batch(sizes, H, seed, ratios=(0.8, 0.6, 0.9)):
    xs, ys, Ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        tv = synth.two_view(n, inlier_ratio=ratios[k % len(ratios)], seed=[seed, k], noise_px=0.4)
        xs.append(tv["x"]); ys.append(tv["y"])
        Ss.append(synth.sample_table(n, H, seed=[seed + 1, k]))
        off.append(off[-1] + n)
    return xs, ys, Ss, np.array(off, np.int64)


def _assert_winners(best, ioff, inl, xs, ys, Ss, K, oracle, adaptive, loss=capi.LOSS_REPROJECTION, thr=1.0):
    for k in range(len(xs)):
        kw = dict(loss_kind=loss) if loss != capi.LOSS_REPROJECTION else {}
        e = oracle.ransac(xs[k], ys[k], K, Ss[k], max_sq_err=thr, success_prob=0.999, adaptive=adaptive, **kw)
        assert best["best_h"][k] == e["best_h"], k
        assert best["iterations"][k] == e["iterations"], k
        assert best["count"][k] == len(e["inliers"]), k
        assert best["toterr"][k] == e["toterr"][e["best_h"]], k
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], e["inliers"]), k


@pytest.mark.parametrize("adaptive", [True, False])
@pytest.mark.parametrize("defer_mib", [2048, 3, 0])
def test_rows_handed_over_are_the_restatement_s(ctx, oracle, adaptive, defer_mib):
    """Pairs with most correspondences inliers: many sample rows hold thousands of inliers, and the scoring launch
    hands them over to the cell-by-cell kernels (with 3 MiB the buffer holds 64 such rows at a time, so most of
    them stay with the scoring kernel; with 0 none is handed over).  Pairs of unequal sizes, one too short to be
    searched for such rows; slices of 128 rows so that later slices have a count to beat and rows are passed
    over.  Winner, count, error sum, iterations and inlier list are the CPU restatement's (RANSAC.cpp:207-258)."""
    sizes = [5000, 4500, 3000, 5555, 4100, 4700, 5000, 4321, 6000]  # (more than one group of eight)
    H = 640
    xs, ys, Ss, off = _heavy_batch(sizes, H, 700)
    K = synth.default_K()
    ctx.set_option("ransac_split", 0)
    ctx.set_option("ransac_defer_mib", defer_mib)
    ctx.set_option("hyps_per_launch", 128)
    try:
        b = ctx.ransac_batch(off, np.concatenate(xs), np.concatenate(ys), K, np.concatenate(Ss), H, max_sq_err=1.0,
                             success_prob=0.999, adaptive=adaptive)
        best = b.best()
        ioff, inl = b.inliers()
        cands, passed, handed = b.handed_over()
        b.close()
    finally:
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_defer_mib", 2048)
        ctx.set_option("hyps_per_launch", 0)
    print("candidates", cands, "passed over", passed, "handed over", handed)
    if defer_mib == 0:
        assert cands == 0
    else:
        assert handed >= 64 and passed > 0 and passed + handed <= cands
        assert (passed + handed < cands) == (defer_mib == 3)  # the small buffer leaves rows with the scoring kernel
    assert (best["count"] * 8 > np.array(sizes)).sum() >= 6  # rows with more than an eighth of the cells inliers did occur
    _assert_winners(best, ioff, inl, xs, ys, Ss, K, oracle, adaptive)


def test_rows_handed_over_with_the_sampson_distance(ctx, oracle):
    """The same with the Sampson distance as the loss: no packed test, so no row is passed over, but rows are
    handed over all the same."""
    sizes = [4200, 5000, 4600]
    H = 384
    xs, ys, Ss, off = _heavy_batch(sizes, H, 720)
    K = synth.default_K()
    ctx.set_option("ransac_split", 0)
    try:
        b = ctx.ransac_batch(off, np.concatenate(xs), np.concatenate(ys), K, np.concatenate(Ss), H, max_sq_err=1.0,
                             success_prob=0.999, adaptive=True, loss=capi.LOSS_SAMPSON)
        best = b.best()
        ioff, inl = b.inliers()
        cands, passed, handed = b.handed_over()
        b.close()
    finally:
        ctx.set_option("ransac_split", 1)
    assert handed > 0 and passed == 0
    _assert_winners(best, ioff, inl, xs, ys, Ss, K, oracle, True, loss=capi.LOSS_SAMPSON)


def test_rows_handed_over_keep_every_row_s_score(ctx, oracle):
    """A single pair scored with per-row outputs (one slice, nothing to beat): rows handed over get their exact
    count and error sum from the cell-by-cell kernels, every other row from the scoring kernel."""
    n, H = 6000, 512
    tv = synth.two_view(n, inlier_ratio=0.85, seed=[730, 0], noise_px=0.3)
    S = synth.sample_table(n, H, seed=[731, 0])
    ctx.set_option("ransac_split", 0)
    try:
        got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    finally:
        ctx.set_option("ransac_split", 1)
    exp = oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    assert (exp["counts"] > n // 8).sum() > 5
    assert np.array_equal(got["counts"], exp["counts"])
    assert np.array_equal(got["toterr"].view(np.uint64), exp["toterr"].view(np.uint64))
    assert_same_run(got, exp)
