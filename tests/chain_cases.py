"""Synthetic two-view solutions for the scale chain: a camera moving past a point cloud, each step
bringing its own (arbitrary) scale, the way consecutive eight-point solutions do."""
import numpy as np


def small_rotation(rng, deg=3.0):
    w = rng.normal(size=3)
    w *= np.radians(deg) * rng.random() / np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    th = np.linalg.norm(w)
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def sequence(seed, n_steps, n_tracks, visible=0.6, shuffle=False, behind=0.05, noise=1e-3, far=0.0):
    """Returns offsets [S+1], track ids, points [total,3] (in the older camera's frame of each step, at the
    step's own scale), poses [S,4,4] (unit translation, as the eight-point solution gives)."""
    rng = np.random.default_rng(seed)
    world = rng.uniform([-6, -4, 6], [6, 4, 18], size=(n_tracks, 3))
    cam = np.eye(4)  # world -> camera of the older frame
    offsets, ids, pts, poses = [0], [], [], []
    for s in range(n_steps):
        R = small_rotation(rng)
        t = rng.normal(size=3) * 0.4 + np.array([0.5, 0, 0.1])
        step_scale = np.linalg.norm(t)  # the two-view solution has |T| = 1
        pose = np.eye(4)
        pose[:3, :3] = R
        pose[:3, 3] = t / step_scale
        n = int(rng.integers(int(visible * n_tracks * 0.7), int(visible * n_tracks) + 1)) if n_tracks else 0
        sel = np.sort(rng.choice(n_tracks, size=n, replace=False)) if n else np.zeros(0, int)
        if shuffle:
            sel = rng.permutation(sel)
        local = (cam[:3, :3] @ world[sel].T).T + cam[:3, 3]
        local = local / step_scale + rng.normal(size=local.shape) * noise
        flip = rng.random(n) < behind
        local[flip, 2] *= -1
        if far:
            local[rng.random(n) < far] *= 400.0  # beyond both the ratio gate (100) and the mean's gate (50)
        ids.append(sel.astype(np.int32))
        pts.append(local)
        poses.append(pose)
        offsets.append(offsets[-1] + n)
        full = np.eye(4)
        full[:3, :3] = R
        full[:3, 3] = t
        cam = full @ cam
    ids = np.concatenate(ids) if ids else np.zeros(0, np.int32)
    pts = np.concatenate(pts) if pts else np.zeros((0, 3))
    return np.array(offsets, np.int64), ids, pts, np.array(poses).reshape(n_steps, 4, 4)


def numpy_chain(offsets, ids, pts, poses, start=None, max_translation=15.0, n_tracks=0):
    """Independent restatement of scene.cpp:301-421 in numpy (agrees with the C restatement to rounding)."""
    acc = np.eye(4) if start is None else np.array(start, float)
    scale, P3, cnt, ext, scales = 1.0, np.zeros((n_tracks, 3)), np.zeros(n_tracks, np.int32), [acc.copy()], []
    S = len(offsets) - 1
    for s in range(S):
        b, e = offsets[s], offsets[s + 1]
        if s > 0:
            b0, e0 = offsets[s - 1], offsets[s]
            old = {int(t): k for k, t in zip(range(b0, e0), ids[b0:e0])}
            rat = []
            for k in range(b, e):
                j = old.get(int(ids[k]))
                if j is None:
                    continue
                d12 = np.linalg.norm((poses[s - 1] @ np.append(pts[j], 1))[:3])
                d23 = np.linalg.norm(pts[k])
                if d23 > 0.1 and d12 > 0.1 and d23 < 100 and d12 < 100:
                    rat.append(d12 / d23)
            if len(rat) >= 5:
                rel = sorted(rat)[len(rat) // 2]
                if 0.1 <= rel < 10:
                    scale *= rel
        scales.append(scale)
        V = poses[s].copy()
        V[:3, 3] *= scale
        L = np.linalg.norm(V[:3, 3])
        if L > max_translation:
            V[:3, 3] *= max_translation / L
        for k in range(b, e):
            t = ids[k]
            if pts[k, 2] < 0 or t < 0 or t >= n_tracks:
                continue
            g = (acc @ np.append(scale * pts[k], 1))[:3]
            if cnt[t] == 0:
                P3[t], cnt[t] = g, 1
            elif np.linalg.norm(P3[t] - g) < 50:
                P3[t] = P3[t] + (g - P3[t]) / (cnt[t] + 1)
                cnt[t] += 1
        acc = acc @ np.linalg.inv(V)
        ext.append(acc.copy())
    return dict(extrinsics=np.array(ext), points3d=P3, counts=cnt, scale=np.array(scales))
