"""Scale chaining + running point average on the device (sfm_scale_chain, sfm_scale_chain_refit)
against the CPU restatement of scene.cpp:301-421.  The kernel performs the restatement's operations
in its order and selects the median exactly, so the bar is bit-exact on every output."""
import numpy as np
import pytest

from sfm_amd import capi, synth, tracks
from chain_cases import sequence

pytestmark = pytest.mark.gpu

KEYS = ("extrinsics", "points3d", "counts", "scale")


def same(got, exp, what=""):
    for k in KEYS:
        assert np.array_equal(got[k], exp[k]), f"{what}: {k} differs (max {np.abs(got[k] - exp[k]).max()})"


def sorted_steps(offsets, ids, pts):
    """The reference's lists ascend by track id; the restatement walks them that way."""
    ids2, pts2 = ids.copy(), pts.copy()
    for s in range(len(offsets) - 1):
        b, e = offsets[s], offsets[s + 1]
        o = np.argsort(ids[b:e], kind="stable")
        ids2[b:e], pts2[b:e] = ids[b:e][o], pts[b:e][o]
    return ids2, pts2


@pytest.mark.parametrize("seed,n_steps,n_tracks", [(1, 26, 1500), (2, 8, 300), (3, 2, 40), (4, 1, 64), (5, 40, 9000),
                                                   (6, 5, 3)])
def test_chain_matches_restatement(ctx, oracle, seed, n_steps, n_tracks):
    off, ids, pts, poses = sequence(seed, n_steps, n_tracks)
    got = ctx.scale_chain(off, ids, pts, poses, n_tracks=n_tracks)
    exp = oracle.scale_chain(off, ids, pts, poses, n_tracks=n_tracks)
    same(got, exp, f"seed {seed}")
    if n_steps > 2 and n_tracks >= 300:
        assert len(set(np.round(got["scale"], 12))) > 2, "the scale never moved: the case tests nothing"
        assert (got["counts"] > 1).sum() > n_tracks // 4


@pytest.mark.parametrize("entries", [2, 3 * 700, 5 * 700 + 13])
def test_chain_in_windows(ctx, oracle, entries):
    """A (step, track) table too small for the sequence: the steps go through in windows of 1, 2, 4."""
    off, ids, pts, poses = sequence(41, 11, 700, behind=0.1)
    exp = oracle.scale_chain(off, ids, pts, poses, n_tracks=700)
    ctx.set_option("chain_table_entries", entries)
    try:
        got = ctx.scale_chain(off, ids, pts, poses, n_tracks=700)
    finally:
        ctx.set_option("chain_table_entries", 1 << 26)
    same(got, exp, f"table of {entries}")


def test_chain_recovers_planted_scale(ctx):
    """The steps come at their own scales (|T| = 1 each); chaining brings them back to the first step's."""
    off, ids, pts, poses = sequence(11, 12, 2000, behind=0.0, noise=0.0)
    got = ctx.scale_chain(off, ids, pts, poses, n_tracks=2000)
    seen = got["counts"] > 2
    assert seen.sum() > 500
    # rebuild the cloud in units of the first step: world / |t_0|; the first camera is the world frame
    rng = np.random.default_rng(11)
    world = rng.uniform([-6, -4, 6], [6, 4, 18], size=(2000, 3))
    P, W = got["points3d"][seen], world[seen]
    c = (P * W).sum() / (W * W).sum()
    assert 0.2 < c < 5 and np.allclose(P, c * W, atol=1e-6), "points are not a single scaled copy of the planted cloud"


def test_chain_any_id_order(ctx, oracle):
    off, ids, pts, poses = sequence(21, 10, 800, shuffle=True)
    got = ctx.scale_chain(off, ids, pts, poses, n_tracks=800)
    ids2, pts2 = sorted_steps(off, ids, pts)
    exp = oracle.scale_chain(off, ids2, pts2, poses, n_tracks=800)
    same(got, exp, "shuffled ids")


def test_chain_gates(ctx, oracle):
    """Far points (outside the ratio gate and the 50-unit gate of the mean), points behind the camera, a
    start pose, and a translation cap that bites."""
    off, ids, pts, poses = sequence(31, 12, 600, behind=0.2, far=0.15)
    start = np.eye(4)
    start[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    start[:3, 3] = [3, -2, 7]
    for cap in (15.0, 0.5):
        got = ctx.scale_chain(off, ids, pts, poses, start_pose=start, max_translation=cap, n_tracks=600)
        exp = oracle.scale_chain(off, ids, pts, poses, start_pose=start, max_translation=cap, n_tracks=600)
        same(got, exp, f"cap {cap}")
    assert np.array_equal(got["extrinsics"][0], start)
    step = np.linalg.inv(got["extrinsics"][:-1]) @ got["extrinsics"][1:]
    assert np.linalg.norm(np.linalg.inv(step)[:, :3, 3], axis=1).max() <= 0.5 + 1e-9


def test_chain_few_ratios_keep_scale(ctx, oracle):
    """Fewer than five shared tracks: the scale stays (scene.cpp:340)."""
    rng = np.random.default_rng(5)
    off = np.array([0, 6, 10, 14], np.int64)
    ids = np.array([0, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 9], np.int32)
    pts = rng.uniform(1, 5, size=(14, 3))
    poses = np.stack([np.eye(4)] * 3)
    poses[:, 0, 3] = 1.0
    got = ctx.scale_chain(off, ids, pts, poses, n_tracks=10)
    exp = oracle.scale_chain(off, ids, pts, poses, n_tracks=10)
    same(got, exp, "few ratios")
    assert np.all(got["scale"] == 1.0)


def test_chain_empty_and_errors(ctx, oracle):
    got = ctx.scale_chain(np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros((0, 4, 4)), n_tracks=4)
    assert got["extrinsics"].shape == (1, 4, 4) and np.array_equal(got["extrinsics"][0], np.eye(4))
    assert np.all(got["counts"] == 0) and np.all(got["points3d"] == 0)
    # steps without points still move the camera
    off = np.array([0, 0, 0], np.int64)
    poses = np.stack([np.eye(4)] * 2)
    poses[:, 2, 3] = 1.0
    got = ctx.scale_chain(off, np.zeros(0, np.int32), np.zeros((0, 3)), poses, n_tracks=0)
    exp = oracle.scale_chain(off, np.zeros(0, np.int32), np.zeros((0, 3)), poses, n_tracks=0)
    assert np.array_equal(got["extrinsics"], exp["extrinsics"]) and got["extrinsics"][2][2, 3] == -2.0
    with pytest.raises(capi.SfmError, match="twice"):
        ctx.scale_chain(np.array([0, 3], np.int64), np.array([4, 2, 4], np.int32), np.ones((3, 3)), np.eye(4)[None],
                        n_tracks=8)
    with pytest.raises(capi.SfmError, match="outside"):
        ctx.scale_chain(np.array([0, 2], np.int64), np.array([1, 8], np.int32), np.ones((2, 3)), np.eye(4)[None],
                        n_tracks=8)
    # the call after a refused one works
    ok = ctx.scale_chain(np.array([0, 2], np.int64), np.array([1, 7], np.int32), np.ones((2, 3)), np.eye(4)[None],
                         n_tracks=8)
    assert list(ok["counts"]) == [0, 1, 0, 0, 0, 0, 0, 1]
    with pytest.raises(capi.SfmError, match="max_translation"):
        ctx.scale_chain(np.array([0, 1], np.int64), np.array([0], np.int32), np.ones((1, 3)), np.eye(4)[None],
                        max_translation=0.0, n_tracks=1)
    with pytest.raises(capi.SfmError, match="ascending"):
        ctx.scale_chain(np.array([0, 2, 1], np.int64), np.array([0, 1], np.int32), np.ones((2, 3)),
                        np.stack([np.eye(4)] * 2), n_tracks=2)


def test_chain_on_refitted_batch(ctx, oracle):
    """matching -> RANSAC -> refit -> chain without the two-view solutions leaving the device, against the
    restatement fed with the fetched refit (so any difference is the chain's)."""
    n_frames, n_feat, n_hyp = 9, 900, 512
    desc, pts, counts, _ = synth.frame_sequence(n_frames, n_feat, seed=7)
    K = synth.default_K()
    pl = np.stack([np.arange(n_frames - 1), np.arange(1, n_frames)], axis=1).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    res = s.match(pl, threshold=0.95, max_dist_sq=100.0 * 100.0)
    off, pairs = res.offsets(), res.pairs()
    corr = capi.Correspondences(ctx, s, res, K)
    batch = corr.ransac(K, n_hyp, 123, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    ioff, inl = batch.inliers()
    rpose, rused, rpts = batch.refit(int(off[-1]))
    # track ids by the reference's rule (scene.cpp:80-120), vectorised and spelled out
    corr_ids, n_tracks = tracks.assign_track_ids(pl, off, pairs, counts)
    track_of, n_chk = [dict() for _ in range(n_frames)], 0
    for k, (a, b) in enumerate(pl):
        for r in range(off[k], off[k + 1]):
            i, j = pairs[r]
            t = track_of[a].get(int(i))
            if t is None:
                t, n_chk = n_chk, n_chk + 1
                track_of[a][int(i)] = t
            track_of[b][int(j)] = t
            assert corr_ids[r] == t
    assert n_chk == n_tracks
    got = batch.scale_chain(corr_ids, n_tracks=n_tracks)
    # the same steps for the restatement: the refit's point set of every pair, ascending by track id
    e_off, e_ids, e_pts = [0], [], []
    for k in range(len(pl)):
        n_inl = ioff[k + 1] - ioff[k]
        rows = inl[ioff[k]:ioff[k + 1]] if n_inl >= 8 else np.arange(off[k + 1] - off[k])
        assert rused[k] == len(rows)
        ids_k = corr_ids[off[k] + rows]
        o = np.argsort(ids_k, kind="stable")
        e_ids.append(ids_k[o])
        e_pts.append(rpts[off[k]:off[k] + rused[k]][o])
        e_off.append(e_off[-1] + len(rows))
    exp = oracle.scale_chain(np.array(e_off, np.int64), np.concatenate(e_ids), np.concatenate(e_pts), rpose,
                             n_tracks=n_tracks)
    same(got, exp, "refit chain")
    assert (got["counts"] > 1).sum() > 200 and len(set(np.round(got["scale"], 9))) > 3
    # a batch that has not been refitted is refused
    batch2 = corr.ransac(K, 64, 5, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    with pytest.raises(capi.SfmError, match="refit"):
        batch2.scale_chain(corr_ids, n_tracks=n_tracks)


@pytest.mark.parametrize("seed,n_steps,n_tracks,entries", [(11, 26, 1500, 1 << 26), (12, 9, 300, 1 << 26), (13, 11, 700, 3 * 700)])
def test_chain_with_colours_matches_restatement(ctx, oracle, seed, n_steps, n_tracks, entries):
    """The running colour average beside the point mean (scene.cpp:377, :394-407): per channel in float, clamped, cut
    off towards zero, updated exactly when the mean is.  Bit-exact, also when the steps go through in windows
    (the colours are carried from window to window like the means)."""
    off, ids, pts, poses = sequence(seed, n_steps, n_tracks, behind=0.1)
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, 256, (len(ids), 3), dtype=np.uint8)
    cols[rng.random(len(ids)) < 0.1] = 255
    cols[rng.random(len(ids)) < 0.1] = 0
    exp = oracle.scale_chain_colors(off, ids, pts, cols, poses, n_tracks=n_tracks)
    ctx.set_option("chain_table_entries", entries)
    try:
        got = ctx.scale_chain_colors(off, ids, pts, cols, poses, n_tracks=n_tracks)
    finally:
        ctx.set_option("chain_table_entries", 1 << 26)
    same(got, exp, f"seed {seed}")
    assert np.array_equal(got["colors"], exp["colors"])
    assert (got["counts"] > 2).sum() > n_tracks // 8 and got["colors"].any()
    # the geometry does not depend on the colours
    same(got, ctx.scale_chain(off, ids, pts, poses, n_tracks=n_tracks), "with and without colours")


def test_pixel_bilinear_matches_restatement(ctx, oracle):
    """util::getPixelBilinearUchar (util.cpp:40-78) for many positions of one image: inside, on pixel centres, on the
    borders and outside (the first neighbour is clamped into the image, the weights are not: the reference
    extrapolates), values that round half-way."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    px = np.concatenate([rng.uniform([-5, -5], [136, 102], (5000, 2)),
                         np.stack([rng.integers(0, 131, 500), rng.integers(0, 97, 500)], axis=1).astype(np.float64),
                         np.stack([rng.integers(0, 130, 500) + 0.5, rng.integers(0, 96, 500) + 0.5], axis=1),
                         np.array([[0.0, 0.0], [130.0, 96.0], [129.999, 95.999], [-1e9, 1e9], [1e300, -1e300]])])
    got = ctx.pixel_bilinear_u8(img, px)
    exp = oracle.pixel_bilinear_u8(img, px[:-1])  # (the cast of 1e300 to int is undefined on the CPU)
    assert np.array_equal(got[:-1], exp)
    with pytest.raises(capi.SfmError):
        ctx.pixel_bilinear_u8(np.zeros((1, 5, 3), np.uint8), px[:3])
