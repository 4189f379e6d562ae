"""Synthetic inputs of the benchmark shapes: SIFT-like descriptor sets with planted matches and
two-view correspondence sets with a planted relative pose.  Everything is drawn from
numpy's counter-based Philox generator with explicit seeds.
"""
from __future__ import annotations

import numpy as np

IMAGE_W, IMAGE_H = 4000.0, 3000.0


def default_K() -> np.ndarray:
    """Phone-like intrinsics: fx = fy = 3000 px, principal point at the centre of 4000x3000."""
    return np.array([[3000.0, 0.0, IMAGE_W / 2], [0.0, 3000.0, IMAGE_H / 2], [0.0, 0.0, 1.0]])


def _rng(seed) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(seed))


def sift_like(rng: np.random.Generator, n: int, dim: int = 128) -> np.ndarray:
    """Integer-valued 0..255 float32 descriptors built the way SIFT quantises them:
    non-negative vector -> L2 normalise -> clamp at 0.2 -> renormalise -> *512 -> round -> saturate."""
    v = rng.gamma(0.6, 1.0, size=(n, dim)).astype(np.float32)
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
    np.minimum(v, 0.2, out=v)
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
    return np.clip(np.rint(v * 512.0), 0, 255).astype(np.float32)


def perturb_sift(rng: np.random.Generator, base: np.ndarray, amp: int = 3) -> np.ndarray:
    """A re-detection of the same features: small integer noise, still 0..255."""
    noise = rng.integers(-amp, amp + 1, size=base.shape).astype(np.float32)
    return np.clip(base + noise, 0, 255).astype(np.float32)


def gaussian_desc(rng: np.random.Generator, n: int, dim: int = 128) -> np.ndarray:
    return rng.standard_normal((n, dim)).astype(np.float32)


def image_set(n_images: int, n_feat: int, dim: int = 128, shared_frac: float = 0.3, seed: int = 1234,
              kind: str = "sift", max_shift: float = 50.0):
    """n_images images of n_feat features.  A fraction shared_frac of every image's features are
    re-detections of one common pool (so every pair of images has ~shared_frac*n_feat true
    matches, displaced by at most max_shift px); the rest are unique to the image.
    Returns (desc [n_images*n_feat, dim] float32, pts [n_images*n_feat, 2] float64, counts)."""
    n_shared = int(round(shared_frac * n_feat))
    rng = _rng(seed)
    pool_desc = sift_like(rng, n_shared, dim) if kind == "sift" else gaussian_desc(rng, n_shared, dim)
    pool_pts = np.stack([rng.uniform(max_shift, IMAGE_W - max_shift, n_shared),
                         rng.uniform(max_shift, IMAGE_H - max_shift, n_shared)], axis=1)
    desc = np.empty((n_images * n_feat, dim), np.float32)
    pts = np.empty((n_images * n_feat, 2), np.float64)
    for im in range(n_images):
        r = _rng([seed, im + 1])
        d = np.empty((n_feat, dim), np.float32)
        p = np.empty((n_feat, 2), np.float64)
        if kind == "sift":
            d[:n_shared] = perturb_sift(r, pool_desc)
            d[n_shared:] = sift_like(r, n_feat - n_shared, dim)
        else:
            d[:n_shared] = pool_desc + 0.05 * r.standard_normal(pool_desc.shape).astype(np.float32)
            d[n_shared:] = gaussian_desc(r, n_feat - n_shared, dim)
        ang = r.uniform(0, 2 * np.pi, n_shared)
        rad = r.uniform(0, max_shift / 2, n_shared)
        p[:n_shared] = pool_pts + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
        p[n_shared:, 0] = r.uniform(0, IMAGE_W, n_feat - n_shared)
        p[n_shared:, 1] = r.uniform(0, IMAGE_H, n_feat - n_shared)
        perm = r.permutation(n_feat)
        desc[im * n_feat:(im + 1) * n_feat] = d[perm]
        pts[im * n_feat:(im + 1) * n_feat] = p[perm]
    counts = np.full(n_images, n_feat, np.int32)
    return desc, pts, counts


def exhaustive_pairs(n_images: int) -> np.ndarray:
    """Upper-triangular pair list, j fastest (consecutive pairs share image i)."""
    i, j = np.triu_indices(n_images, k=1)
    return np.stack([i, j], axis=1).astype(np.int32)


def _rodrigues(axis, angle):
    axis = axis / np.linalg.norm(axis)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)


def two_view(n_corr: int, inlier_ratio: float = 0.15, seed: int = 7, noise_px: float = 0.5, K=None,
             max_rot_deg: float = 20.0, depth=(2.0, 10.0)):
    """Correspondences of one image pair: inliers from a random relative pose (rotation up to
    max_rot_deg, unit baseline, depths `depth` in baselines, Gaussian pixel noise), outliers uniform in
    both images.  Returns dict(x, y [n,2] K^-1-normalised, R, T, inlier_mask, K)."""
    K = default_K() if K is None else np.asarray(K, np.float64)
    Kinv = np.linalg.inv(K)
    rng = _rng(seed)
    n_in = int(round(inlier_ratio * n_corr))
    R = _rodrigues(rng.standard_normal(3), np.deg2rad(rng.uniform(2.0, max_rot_deg)))
    T = rng.standard_normal(3)
    T /= np.linalg.norm(T)
    px1 = np.empty((n_corr, 2))
    px2 = np.empty((n_corr, 2))
    # inliers: sample pixels in view 1 and a depth, project into view 2, keep those inside
    got = 0
    while got < n_in:
        m = (n_in - got) * 2 + 16
        u = np.stack([rng.uniform(0, IMAGE_W, m), rng.uniform(0, IMAGE_H, m), np.ones(m)], axis=1)
        z = rng.uniform(depth[0], depth[1], m)
        X = (u @ Kinv.T) * z[:, None]
        X2 = X @ R.T + T
        ok = X2[:, 2] > 0.1
        p2 = (X2 / X2[:, 2:3]) @ K.T
        ok &= (p2[:, 0] >= 0) & (p2[:, 0] < IMAGE_W) & (p2[:, 1] >= 0) & (p2[:, 1] < IMAGE_H)
        k = min(int(ok.sum()), n_in - got)
        px1[got:got + k] = u[ok][:k, :2]
        px2[got:got + k] = p2[ok][:k, :2]
        got += k
    px1[:n_in] += noise_px * rng.standard_normal((n_in, 2))
    px2[:n_in] += noise_px * rng.standard_normal((n_in, 2))
    n_out = n_corr - n_in
    px1[n_in:] = np.stack([rng.uniform(0, IMAGE_W, n_out), rng.uniform(0, IMAGE_H, n_out)], axis=1)
    px2[n_in:] = np.stack([rng.uniform(0, IMAGE_W, n_out), rng.uniform(0, IMAGE_H, n_out)], axis=1)
    perm = rng.permutation(n_corr)
    px1, px2 = px1[perm], px2[perm]
    mask = np.zeros(n_corr, bool)
    mask[:n_in] = True
    mask = mask[perm]

    def normalise(px):  # scene.cpp:495-501
        h = np.concatenate([px, np.ones((len(px), 1))], axis=1) @ Kinv.T
        return np.ascontiguousarray(h[:, :2])

    return dict(x=normalise(px1), y=normalise(px2), R=R, T=T, inlier_mask=mask, K=K, px1=px1, px2=px2)


def sample_table(n_corr: int, n_hyp: int, seed: int = 11, k: int = 8) -> np.ndarray:
    """n_hyp rows of k distinct indices drawn uniformly from range(n_corr)."""
    rng = _rng(seed)
    if n_corr < k:
        raise ValueError("need at least k correspondences")
    if n_corr <= 256:
        return np.argsort(rng.random((n_hyp, n_corr)), axis=1)[:, :k].astype(np.int32)
    out = rng.integers(0, n_corr, size=(n_hyp, k))
    while True:  # redraw the (rare) rows that repeat an index
        srt = np.sort(out, axis=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if not bad.any():
            return out.astype(np.int32)
        out[bad] = rng.integers(0, n_corr, size=(int(bad.sum()), k))


def frame_sequence(n_frames: int, n_feat: int, carry: float = 0.6, seed: int = 2024, K=None, noise_px: float = 0.3,
                   step: float = 0.08, rot_deg: float = 0.4):
    """A camera moving past a static cloud: n_frames images of n_feat features, about carry*n_feat of
    them re-detections of scene points that are also visible in the neighbouring frames (pixel shift
    between neighbours well under 100 px), the rest clutter unique to the frame.  Stand-in for the
    reference's default run (27 frames, consecutive pairs).
    Returns (desc, pts, counts, poses) with poses[k] = (R, T) world -> camera k."""
    K = default_K() if K is None else np.asarray(K, np.float64)
    Kinv = np.linalg.inv(K)
    rng = _rng(seed)
    n_scene = int(n_feat * carry * 1.6)
    # scene points in the frustum of the middle camera, depths 4..10
    u = np.stack([rng.uniform(-400, IMAGE_W + 400, n_scene), rng.uniform(-300, IMAGE_H + 300, n_scene),
                  np.ones(n_scene)], axis=1)
    X = (u @ Kinv.T) * rng.uniform(4.0, 10.0, n_scene)[:, None]
    scene_desc = sift_like(rng, n_scene)
    axis = np.array([0.1, 1.0, 0.05])
    descs, ptss, poses = [], [], []
    for k in range(n_frames):
        t = k - (n_frames - 1) / 2
        R = _rodrigues(axis, np.deg2rad(rot_deg * t))
        T = np.array([-step * t, 0.01 * t, 0.02 * t])
        poses.append((R, T))
        r = _rng([seed, k + 1])
        Xc = X @ R.T + T
        p = (Xc / Xc[:, 2:3]) @ K.T
        vis = (Xc[:, 2] > 0.5) & (p[:, 0] >= 0) & (p[:, 0] < IMAGE_W) & (p[:, 1] >= 0) & (p[:, 1] < IMAGE_H)
        idx = np.flatnonzero(vis)
        keep = idx[r.random(len(idx)) < 0.9][: int(n_feat * carry)]  # detection misses some
        n_clutter = n_feat - len(keep)
        d = np.concatenate([perturb_sift(r, scene_desc[keep]), sift_like(r, n_clutter)])
        q = np.concatenate([p[keep, :2] + noise_px * r.standard_normal((len(keep), 2)),
                            np.stack([r.uniform(0, IMAGE_W, n_clutter), r.uniform(0, IMAGE_H, n_clutter)], axis=1)])
        perm = r.permutation(n_feat)
        descs.append(d[perm])
        ptss.append(q[perm])
    return (np.concatenate(descs).astype(np.float32), np.concatenate(ptss).astype(np.float64),
            np.full(n_frames, n_feat, np.int32), poses)
