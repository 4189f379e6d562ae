"""Signs and order of Eigen's JacobiSVD in the fit and the refit (reference
submodules/solve/eightPoint.cpp:85-86,93-96,254-255,262-265): the four (R, T) candidates are listed in
an order that depends on them (:127-128, :296-297) and a tied cheirality vote keeps the first (:173-177,
:332-336), so the kernels follow the restated algorithm, not just any SVD.

Fit (one hypothesis per lane): poses bit-identical to the CPU restatement, tied rows included.
Refit (one block per pair, sums over rows in the block's order): poses within 1e-8, and never another
candidate than the restatement's where the vote is not tied."""
import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_fit_is_bit_identical_on_tied_rows(ctx, oracle):
    """The RANSAC stress shape: a fifth of the rows have a tied top vote, where only the candidate order decides."""
    tv = synth.two_view(20000, inlier_ratio=0.15, seed=[0, 0])
    S = synth.sample_table(20000, 4096, seed=[7, 0, 0])
    exp, cand, votes = oracle.fit_hypotheses_votes(tv["x"], tv["y"], S)
    tied = (votes == votes.max(axis=1)[:, None]).sum(axis=1) > 1
    assert 0.15 < tied.mean() < 0.3
    got = ctx.eight_point_fit(tv["x"], tv["y"], S)
    assert np.array_equal(bits(got), bits(exp))
    assert np.array_equal(bits(got[tied]), bits(exp[tied]))


@pytest.mark.parametrize("fx,ratio", [(400.0, 0.5), (3000.0, 0.8), (12000.0, 0.3), (150.0, 0.6)])
def test_fit_is_bit_identical_for_other_cameras(ctx, oracle, fx, ratio):
    """A short focal length gives normalised coordinates above 1, so the 8x9 system is divided by its largest
    entry (JacobiSVD scales by it) before the factorisation; a long one leaves the column of ones the largest."""
    K = np.array([[fx, 0.0, 2000.0], [0.0, 1.1 * fx, 1500.0], [0.0, 0.0, 1.0]])
    tv = synth.two_view(2000, inlier_ratio=ratio, seed=int(fx), K=K)
    S = synth.sample_table(2000, 1024, seed=int(fx) + 1)
    exp, _ = oracle.fit_hypotheses(tv["x"], tv["y"], S)
    got = ctx.eight_point_fit(tv["x"], tv["y"], S)
    assert np.array_equal(bits(got), bits(exp))
    if fx < 1000:
        assert np.abs(tv["x"]).max() > 1.0


def test_fit_is_bit_identical_on_degenerate_samples(ctx, oracle):
    """Samples that repeat a point, lie on a line or in a plane, or sit at the origin: rank-deficient systems,
    vanishing columns and pivots that tie.  No NaN is required to match a NaN's payload: rows are compared
    where the restatement is finite, and must be non-finite where it is not."""
    rng = np.random.default_rng(5)
    tv = synth.two_view(64, inlier_ratio=1.0, seed=3)
    x, y = tv["x"].copy(), tv["y"].copy()
    x[8:16] = x[0]                        # one point eight times
    y[8:16] = y[0]
    t = np.linspace(-0.3, 0.3, 8)
    x[16:24] = np.stack([t, 2 * t], axis=1)  # collinear in both images
    y[16:24] = np.stack([t + 0.01, 2 * t], axis=1)
    x[24:32] = 0.0                        # the principal point, in both images
    y[24:32] = 0.0
    x[32:40] = y[32:40] = rng.uniform(-0.5, 0.5, (8, 2))  # no motion at all
    S = np.stack([np.arange(8 * k, 8 * k + 8) for k in range(1, 5)] + [rng.permutation(64)[:8] for _ in range(60)]).astype(np.int32)
    exp, _ = oracle.fit_hypotheses(x, y, S)
    got = ctx.eight_point_fit(x, y, S)
    fin = np.isfinite(exp).all(axis=1)
    assert np.array_equal(bits(got[fin]), bits(exp[fin]))
    assert not np.isfinite(got[~fin]).all(axis=1).any()


def scene(n, seed, planar=0.0, baseline=1.0, noise_px=0.3):
    """n inliers of a scene whose points lie within `planar` (relative depth spread; 0 = generic) of a plane,
    seen from two cameras `baseline` apart (in units of the depth / 5)."""
    rng = np.random.default_rng(seed)
    K = synth.default_K()
    Kinv = np.linalg.inv(K)
    R = synth._rodrigues(rng.standard_normal(3), np.deg2rad(rng.uniform(2.0, 8.0)))
    T = rng.standard_normal(3)
    T *= baseline / np.linalg.norm(T)
    u = np.stack([rng.uniform(200, 3800, n), rng.uniform(200, 2800, n), np.ones(n)], axis=1)
    if planar > 0:
        nrm = np.array([0.1, -0.2, 1.0])
        rays = u @ Kinv.T
        z = 5.0 / (rays @ nrm) * (1.0 + planar * rng.standard_normal(n))
    else:
        z = rng.uniform(2.0, 10.0, n)
    X = (u @ Kinv.T) * z[:, None]
    X2 = X @ R.T + T
    p2 = (X2 / X2[:, 2:3]) @ K.T
    px1 = u[:, :2] + noise_px * rng.standard_normal((n, 2))
    px2 = p2[:, :2] + noise_px * rng.standard_normal((n, 2))
    norm = lambda px: np.ascontiguousarray((np.concatenate([px, np.ones((n, 1))], axis=1) @ Kinv.T)[:, :2])
    return norm(px1), norm(px2), R, T / np.linalg.norm(T)


@pytest.mark.parametrize("n", [8, 9, 10, 11, 17, 64, 255, 256, 257, 1000, 5000])
def test_refit_point_counts(ctx, oracle, n):
    """Eight points take the Q of A^T's factorisation, nine the sweeps on A itself, ten and more the pivoted
    QR of A first (JacobiSVD's three cases); block sizes around 256 cover rows per thread."""
    x, y, _, _ = scene(n, 100 + n)
    pose, pts = ctx.eight_point_algorithm(x, y)
    epose, epts, _, _ = oracle.eight_point(x, y)
    assert np.abs(pose - epose).max() < 1e-8
    assert np.abs(pts - epts).max() <= 1e-6 * np.abs(epts).max()


@pytest.mark.parametrize("kind,kw", [("planar 1e-2", dict(planar=1e-2)), ("planar 1e-3", dict(planar=1e-3)),
                                     ("planar 1e-4", dict(planar=1e-4)),
                                     ("baseline 1 %", dict(baseline=0.05)), ("baseline 0.1 %", dict(baseline=0.005)),
                                     ("noise-free", dict(noise_px=0.0)), ("noise-free planar", dict(noise_px=0.0, planar=1e-3))])
def test_refit_on_ill_conditioned_scenes(ctx, oracle, kind, kw):
    """Scenes on which the n x 9 system has a second small singular value (points near a plane) or a null vector
    that noise moves a lot (a baseline of 1 % and 0.1 % of the depth): the null vector is reached through the
    QR factor of A, not through A^T A, so the kernel and the restatement still agree to 1e-8."""
    for seed in range(6):
        x, y, _, _ = scene(600, 7000 + seed, **kw)
        pose, pts = ctx.eight_point_algorithm(x, y)
        epose, epts, _, _ = oracle.eight_point(x, y)
        assert np.abs(pose - epose).max() < 1e-8, (kind, seed, np.abs(pose - epose).max())
        # the rotation is a rotation and the translation a unit vector whatever the conditioning
        R = pose[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
        assert abs(np.linalg.norm(pose[:3, 3]) - 1) < 1e-9


def test_refit_recovers_a_noise_free_near_planar_pose(ctx):
    """Noise-free points within 1e-3 of a plane: sigma_7 of the system is ~1e-3 of sigma_0, so a route through
    A^T A (condition number squared) would lose the pose at the 1e-9 level; through the QR factor it is
    recovered to 1e-11."""
    worst = 0.0
    for seed in range(8):
        x, y, R, T = scene(600, 7100 + seed, noise_px=0.0, planar=1e-3)
        pose, _ = ctx.eight_point_algorithm(x, y)
        worst = max(worst, np.abs(pose[:3, :3] - R).max(), np.abs(pose[:3, 3] - T).max())
    print("largest pose error on noise-free near-planar scenes:", worst)
    assert worst < 1e-10


def test_refit_sign_follows_the_restatement(ctx, oracle):
    """The sign of the null vector orders the candidates; where it matters (a tied vote) the kernel must
    take the restatement's candidate.  Few points make ties common: 9 to 12 points, many seeds."""
    ties = 0
    for seed in range(120):
        n = 9 + seed % 4
        x, y, _, _ = scene(n, 9000 + seed, noise_px=3.0)
        # spoil two points so that votes split
        rng = np.random.default_rng(seed)
        y[:2] = rng.uniform(-0.5, 0.5, (2, 2))
        epose, _, idx, votes = oracle.eight_point(x, y)
        ties += int((votes == votes.max()).sum() > 1)
        pose, _ = ctx.eight_point_algorithm(x, y)
        assert np.abs(pose - epose).max() < 1e-7, (seed, n, votes)
    assert ties >= 3
