"""ctypes binding of oracle/libsfm_oracle.so (the CPU restatement).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsfm_oracle.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the restatement with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "sfm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.sfm_oracle_num_threads.restype = C.c_int
    L.sfm_oracle_set_threads.argtypes = [C.c_int]
    L.sfm_oracle_similarity.restype = C.c_float
    L.sfm_oracle_similarity.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
    L.sfm_oracle_match.restype = C.c_int
    L.sfm_oracle_match.argtypes = [_f32p, _f64p, C.c_int, _f32p, _f64p, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_float, _i32p, _i32p, _i32p]
    L.sfm_oracle_match_ratio.restype = C.c_int
    L.sfm_oracle_match_ratio.argtypes = [_f32p, _f64p, C.c_int, _f32p, _f64p, C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_float, _i32p, _i32p, _i32p]
    L.sfm_oracle_match_l2.restype = C.c_int
    L.sfm_oracle_match_l2.argtypes = [_f32p, _f64p, C.c_int, _f32p, _f64p, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_float, _i32p, _i32p, _i32p]
    L.sfm_oracle_match_hamming.restype = C.c_int
    L.sfm_oracle_match_hamming.argtypes = [_u8p, _f64p, C.c_int, _u8p, _f64p, C.c_int, C.c_int, C.c_int, C.c_float,
                                           _i32p, _i32p, _i32p]
    L.sfm_oracle_eight_point_from_subset.restype = C.c_int
    L.sfm_oracle_eight_point_from_subset.argtypes = [_f64p, _f64p, C.c_int, _f64p, _f64p, C.c_int,
                                                     _f64p, _f64p, _i32p]
    L.sfm_oracle_loss.restype = C.c_double
    L.sfm_oracle_loss.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p, C.c_int]
    L.sfm_oracle_svd3.argtypes = [_f64p, _f64p, _f64p, _f64p]
    L.sfm_oracle_null_vector.argtypes = [_f64p, C.c_int, _f64p]
    L.sfm_oracle_triangulate.argtypes = [_f64p, _f64p, _f64p, _f64p]
    L.sfm_oracle_triangulate_normal_equations.argtypes = [_f64p, _f64p, _f64p, _f64p]
    L.sfm_oracle_score_cells.argtypes = [_f64p, _f64p, C.c_int, _f64p, _f64p, C.c_int, C.c_int, _f64p, _f64p]
    L.sfm_oracle_score_cells.restype = None
    L.sfm_oracle_adaptive_max_iter.restype = C.c_int
    L.sfm_oracle_adaptive_max_iter.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
    L.sfm_oracle_ransac.restype = C.c_int
    L.sfm_oracle_ransac.argtypes = [_f64p, _f64p, C.c_int, _f64p, _i32p, C.c_int, C.c_int,
                                    C.c_double, C.c_double, C.c_int, C.c_int, _i32p,
                                    C.POINTER(C.c_int), _f64p, _i32p, _f64p, C.POINTER(C.c_int)]
    L.sfm_oracle_fit_hypotheses.argtypes = [_f64p, _f64p, _i32p, C.c_int, _f64p, _i32p]
    L.sfm_oracle_fit_hypotheses_votes.argtypes = [_f64p, _f64p, _i32p, C.c_int, _f64p, _i32p, _i32p]
    L.sfm_oracle_jacobi_svd_tall.argtypes = [_f64p, C.c_int, C.c_int, _f64p, _f64p, _f64p]
    L.sfm_oracle_normalize_points.argtypes = [_f64p, _f64p, C.c_int, _f64p]
    L.sfm_oracle_sample_table.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.c_int, _i32p]
    L.sfm_oracle_scale_chain.argtypes = [C.c_int, _i64p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int,
                                         _f64p, _f64p, _i32p, _f64p]
    L.sfm_oracle_scale_chain.restype = C.c_int
    L.sfm_oracle_sample_table.restype = None
    _lib = L
    return L


def num_threads() -> int:
    return int(lib().sfm_oracle_num_threads())


def set_threads(n: int) -> None:
    lib().sfm_oracle_set_threads(int(n))


COSINE, DOT = 0, 1
LOSS_REPROJECTION, LOSS_SAMPSON = 0, 1


def similarity(a, b, metric=COSINE) -> float:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().sfm_oracle_similarity(a, b, a.size, metric))


def match(descA, ptsA, descB, ptsB, threshold=0.5, max_dist_sq=1e8, metric=COSINE, ratio=None):
    """matchTwoSided. Returns (pairs[k,2] int32, matches1[nA], matches2[nB]).  ratio: the extension with
    Lowe's second-best test (cosine metric; no counterpart in the reference)."""
    descA = np.ascontiguousarray(descA, np.float32).reshape(len(ptsA), -1) if len(ptsA) else \
        np.zeros((0, np.asarray(descA).shape[-1] if np.asarray(descA).ndim == 2 else 128), np.float32)
    descB = np.ascontiguousarray(descB, np.float32).reshape(len(ptsB), -1) if len(ptsB) else \
        np.zeros((0, descA.shape[1]), np.float32)
    ptsA = np.ascontiguousarray(ptsA, np.float64).reshape(-1, 2)
    ptsB = np.ascontiguousarray(ptsB, np.float64).reshape(-1, 2)
    nA, nB = descA.shape[0], descB.shape[0]
    dim = descA.shape[1] if nA else descB.shape[1]
    m1 = np.full(max(nA, 1), -1, np.int32)
    m2 = np.full(max(nB, 1), -1, np.int32)
    pairs = np.zeros((max(min(nA, nB), 1), 2), np.int32)
    pad = lambda a, shape, dt: a if a.size else np.zeros(shape, dt)
    if ratio is not None:
        assert metric == COSINE
        cnt = lib().sfm_oracle_match_ratio(pad(descA, (1, dim), np.float32), pad(ptsA, (1, 2), np.float64), nA,
                                           pad(descB, (1, dim), np.float32), pad(ptsB, (1, 2), np.float64), nB,
                                           dim, np.float32(threshold), np.float32(max_dist_sq), np.float32(ratio),
                                           m1, m2, pairs.reshape(-1))
    else:
        cnt = lib().sfm_oracle_match(pad(descA, (1, dim), np.float32), pad(ptsA, (1, 2), np.float64), nA,
                                     pad(descB, (1, dim), np.float32), pad(ptsB, (1, 2), np.float64), nB,
                                     dim, metric, np.float32(threshold), np.float32(max_dist_sq),
                                     m1, m2, pairs.reshape(-1))
    if cnt < 0:
        raise MemoryError("oracle similarity matrix allocation failed")
    return pairs[:cnt].copy(), m1[:nA].copy(), m2[:nB].copy()


def eight_point_from_subset(calc1, calc2, all1, all2):
    calc1 = np.ascontiguousarray(calc1, np.float64).reshape(-1, 2)
    calc2 = np.ascontiguousarray(calc2, np.float64).reshape(-1, 2)
    all1 = np.ascontiguousarray(all1, np.float64).reshape(-1, 2)
    all2 = np.ascontiguousarray(all2, np.float64).reshape(-1, 2)
    pose = np.zeros(16, np.float64)
    pts = np.zeros((max(len(all1), 1), 3), np.float64)
    counts = np.zeros(4, np.int32)
    idx = lib().sfm_oracle_eight_point_from_subset(calc1, calc2, len(calc1), all1, all2, len(all1),
                                                   pose, pts.reshape(-1), counts)
    if idx < 0:
        raise ValueError("EightPointFromSubset: at least 8 correspondences required")
    return pose.reshape(4, 4, order="F").copy(), pts[:len(all1)].copy(), idx, counts


def eight_point(points1, points2):
    """eightPointAlgorithm(points1, points2) (eightPoint.cpp:57-223): pose (4x4), points, candidate, counts."""
    return eight_point_from_subset(points1, points2, points1, points2)


def loss(K, obs1, obs2, pt, pose, kind=LOSS_REPROJECTION) -> float:
    K = np.asfortranarray(K, np.float64)
    pose = np.asfortranarray(pose, np.float64)
    return float(lib().sfm_oracle_loss(K.reshape(-1, order="F").copy(),
                                       np.ascontiguousarray(obs1, np.float64),
                                       np.ascontiguousarray(obs2, np.float64),
                                       np.ascontiguousarray(pt, np.float64),
                                       pose.reshape(-1, order="F").copy(), kind))


def svd3(E):
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    U = np.zeros((3, 3)); S = np.zeros(3); V = np.zeros((3, 3))
    lib().sfm_oracle_svd3(E.reshape(-1), U.reshape(-1), S, V.reshape(-1))
    return U, S, V


def null_vector(A):
    A = np.ascontiguousarray(A, np.float64)
    q = np.zeros(9)
    lib().sfm_oracle_null_vector(A.reshape(-1), A.shape[0], q)
    return q


def triangulate(a, b, T):
    out = np.zeros(2)
    lib().sfm_oracle_triangulate(np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64),
                                 np.ascontiguousarray(T, np.float64), out)
    return out


def triangulate_normal_equations(a, b, T):
    """The 3x2 least squares through the normal equations (not the reference's route; measurement only)."""
    out = np.zeros(2)
    lib().sfm_oracle_triangulate_normal_equations(np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64),
                                                  np.ascontiguousarray(T, np.float64), out)
    return out


def score_cells(x, y, K, pose12, method=0):
    """Depth before the float cast and default loss of every (hypothesis, correspondence) cell.
    method 0: the QR solve (what the restatement uses), 1: normal equations."""
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
    pose12 = np.ascontiguousarray(pose12, np.float64).reshape(-1, 12)
    K = np.asfortranarray(K, np.float64).reshape(-1, order="F").copy()
    H, n = len(pose12), len(x)
    lam = np.zeros((H, n))
    err = np.zeros((H, n))
    lib().sfm_oracle_score_cells(x, y, n, K, pose12, H, int(method), lam.reshape(-1), err.reshape(-1))
    return lam, err


def emulate_shuffle(on: bool) -> None:
    """Timed CPU baseline only: ransac() also pays for the reference's per-iteration std::shuffle of all
    correspondence indices (RANSAC.cpp:201).  Results are unchanged."""
    lib().sfm_oracle_emulate_shuffle.argtypes = [C.c_int]
    lib().sfm_oracle_emulate_shuffle(int(bool(on)))


def adaptive_max_iter(n_inliers, n_total, min_n, success_prob) -> int:
    return int(lib().sfm_oracle_adaptive_max_iter(n_inliers, n_total, min_n, success_prob))


def ransac(x, y, K, samples, max_sq_err=10.0, success_prob=0.99, adaptive=True, min_n=8,
           loss_kind=LOSS_REPROJECTION):
    """Returns dict(inliers, best_h, pose(4x4), counts[H], toterr[H], iterations)."""
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
    if len(x) != len(y):
        return dict(inliers=np.zeros(0, np.int32), best_h=-1, pose=np.eye(4), counts=None, toterr=None,
                    iterations=0)
    K = np.asfortranarray(K, np.float64).reshape(-1, order="F").copy()
    samples = np.ascontiguousarray(samples, np.int32).reshape(-1, min_n)
    H = samples.shape[0]
    n = len(x)
    inl = np.zeros(max(n, 1), np.int32)
    pose = np.eye(4).reshape(-1, order="F").copy()
    counts = np.zeros(max(H, 1), np.int32)
    toterr = np.zeros(max(H, 1), np.float64)
    best = C.c_int(-1)
    iters = C.c_int(0)
    pad = lambda a: a if a.size else np.zeros((1, 2))
    cnt = lib().sfm_oracle_ransac(pad(x), pad(y), n, K, samples if samples.size else np.zeros((1, min_n), np.int32),
                                  H, min_n, max_sq_err, success_prob, int(bool(adaptive)), loss_kind, inl,
                                  C.byref(best), pose, counts, toterr, C.byref(iters))
    if cnt < 0:
        raise ValueError("minN outside 8..64")
    return dict(inliers=inl[:cnt].copy(), best_h=int(best.value), pose=pose.reshape(4, 4, order="F").copy(),
                counts=counts[:H].copy(), toterr=toterr[:H].copy(), iterations=int(iters.value))


def fit_hypotheses(x, y, samples):
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
    samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 8)
    H = samples.shape[0]
    pose12 = np.zeros((H, 12))
    cand = np.zeros(H, np.int32)
    lib().sfm_oracle_fit_hypotheses(x, y, samples, H, pose12.reshape(-1), cand)
    return pose12, cand


def fit_hypotheses_votes(x, y, samples):
    """fit_hypotheses plus the four cheirality counts of every row (eightPoint.cpp:299-337), candidate order
    (R1,T), (R1,-T), (R2,T), (R2,-T)."""
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
    samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 8)
    H = samples.shape[0]
    pose12 = np.zeros((H, 12))
    cand = np.zeros(H, np.int32)
    votes = np.zeros((H, 4), np.int32)
    lib().sfm_oracle_fit_hypotheses_votes(x, y, samples, H, pose12.reshape(-1), cand, votes.reshape(-1))
    return pose12, cand, votes


def jacobi_svd_tall(A):
    """The restated JacobiSVD(A, ComputeThinU | ComputeThinV) for a matrix with more rows than columns."""
    A = np.ascontiguousarray(A, np.float64)
    r, c = A.shape
    assert r > c
    U = np.zeros((r, c)); S = np.zeros(c); V = np.zeros((c, c))
    lib().sfm_oracle_jacobi_svd_tall(A.reshape(-1), r, c, U.reshape(-1), S, V.reshape(-1))
    return U, S, V


def normalize_points(K, pixels):
    Kinv = np.linalg.inv(np.asarray(K, np.float64))
    pixels = np.ascontiguousarray(pixels, np.float64).reshape(-1, 2)
    out = np.zeros_like(pixels)
    lib().sfm_oracle_normalize_points(np.asfortranarray(Kinv).reshape(-1, order="F").copy(), pixels,
                                      len(pixels), out)
    return out


def sample_table(n: int, n_hyp: int, seed: int, pair_index: int) -> np.ndarray:
    """The device library's counter-based sample table for one pair (see include/sfm_amd.h)."""
    out = np.zeros((max(n_hyp, 1), 8), np.int32)
    lib().sfm_oracle_sample_table(int(n), int(n_hyp), int(seed), int(pair_index), out)
    return out[:n_hyp]


def scale_chain(offsets, track_ids, points, poses, start_pose=None, max_translation=15.0, n_tracks=0):
    """Scale chaining + running point average (scene.cpp:301-421); same layout as the device call."""
    offsets = np.ascontiguousarray(offsets, np.int64)
    S = len(offsets) - 1
    pad = lambda a, shape, dt: a if a.size else np.zeros(shape, dt)
    track_ids = pad(np.ascontiguousarray(track_ids, np.int32), 1, np.int32)
    points = pad(np.ascontiguousarray(points, np.float64).reshape(-1, 3), (1, 3), np.float64)
    poses_c = pad(np.ascontiguousarray(np.asarray(poses, np.float64).reshape(S, 4, 4).transpose(0, 2, 1)), (1, 4, 4),
                  np.float64)
    start = np.eye(4) if start_pose is None else np.asarray(start_pose, np.float64)
    start_c = np.ascontiguousarray(start.T)
    extr = np.zeros((S + 1, 16))
    pts = np.zeros((max(n_tracks, 1), 3))
    cnt = np.zeros(max(n_tracks, 1), np.int32)
    scale = np.zeros(max(S, 1))
    rc = lib().sfm_oracle_scale_chain(S, offsets, track_ids, points.reshape(-1), poses_c.reshape(-1),
                                      start_c.reshape(-1), float(max_translation), int(n_tracks), extr.reshape(-1),
                                      pts.reshape(-1), cnt, scale)
    if rc != 0:
        raise MemoryError("scale chain")
    return dict(extrinsics=extr.reshape(S + 1, 4, 4).transpose(0, 2, 1).copy(), points3d=pts[:n_tracks],
                counts=cnt[:n_tracks], scale=scale[:S])


def scale_chain_colors(offsets, track_ids, points, colors, poses, start_pose=None, max_translation=15.0, n_tracks=0):
    """scale_chain with the running colour average of scene.cpp:394-407; colors uint8 [total, 3]."""
    offsets = np.ascontiguousarray(offsets, np.int64)
    S = len(offsets) - 1
    pad = lambda a, shape, dt: a if a.size else np.zeros(shape, dt)
    track_ids = pad(np.ascontiguousarray(track_ids, np.int32), 1, np.int32)
    points = pad(np.ascontiguousarray(points, np.float64).reshape(-1, 3), (1, 3), np.float64)
    colors = pad(np.ascontiguousarray(colors, np.uint8).reshape(-1, 3), (1, 3), np.uint8)
    poses_c = pad(np.ascontiguousarray(np.asarray(poses, np.float64).reshape(S, 4, 4).transpose(0, 2, 1)), (1, 4, 4),
                  np.float64)
    start = np.eye(4) if start_pose is None else np.asarray(start_pose, np.float64)
    start_c = np.ascontiguousarray(start.T)
    extr = np.zeros((S + 1, 16))
    pts = np.zeros((max(n_tracks, 1), 3))
    cols = np.zeros((max(n_tracks, 1), 3), np.uint8)
    cnt = np.zeros(max(n_tracks, 1), np.int32)
    scale = np.zeros(max(S, 1))
    f = lib().sfm_oracle_scale_chain_colors
    f.restype = C.c_int
    vp = C.c_void_p
    f.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, C.c_double, C.c_int, vp, vp, vp, vp, vp]
    P = lambda a: a.ctypes.data_as(vp)
    rc = f(S, P(offsets), P(track_ids), P(points), P(colors), P(poses_c), P(start_c), float(max_translation), int(n_tracks),
           P(extr), P(pts), P(cols), P(cnt), P(scale))
    if rc != 0:
        raise MemoryError("scale chain")
    return dict(extrinsics=extr.reshape(S + 1, 4, 4).transpose(0, 2, 1).copy(), points3d=pts[:n_tracks],
                colors=cols[:n_tracks], counts=cnt[:n_tracks], scale=scale[:S])


def pixel_bilinear_u8(image_bgr, pixels):
    """util::getPixelBilinearUchar (util.cpp:40-78) at every (u, v) of pixels: RGB uint8 [n, 3]."""
    img = np.ascontiguousarray(image_bgr, np.uint8)
    rows, cols = img.shape[:2]
    px = np.ascontiguousarray(pixels, np.float64).reshape(-1, 2)
    out = np.zeros((max(len(px), 1), 3), np.uint8)
    f = lib().sfm_oracle_pixel_bilinear_u8
    f.restype = None
    vp = C.c_void_p
    f.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    f(img.ctypes.data_as(vp), rows, cols, px.ctypes.data_as(vp), len(px), out.ctypes.data_as(vp))
    return out[:len(px)]


def match_hamming(descA, ptsA, descB, ptsB, max_distance=64, max_dist_sq=1e8):
    """Mutual nearest neighbours of binary descriptors (uint8 [n, bytes]) under the Hamming distance.
    Returns (pairs[k,2] int32, matches1[nA], matches2[nB])."""
    descA = np.ascontiguousarray(descA, np.uint8)
    descB = np.ascontiguousarray(descB, np.uint8)
    nbytes = descA.shape[1] if descA.ndim == 2 and descA.shape[0] else descB.shape[1]
    ptsA = np.ascontiguousarray(ptsA, np.float64).reshape(-1, 2)
    ptsB = np.ascontiguousarray(ptsB, np.float64).reshape(-1, 2)
    nA, nB = len(ptsA), len(ptsB)
    m1 = np.full(max(nA, 1), -1, np.int32)
    m2 = np.full(max(nB, 1), -1, np.int32)
    pairs = np.zeros((max(min(nA, nB), 1), 2), np.int32)
    pad = lambda a, shape, dt: a if a.size else np.zeros(shape, dt)
    cnt = lib().sfm_oracle_match_hamming(pad(descA, (1, nbytes), np.uint8), pad(ptsA, (1, 2), np.float64), nA,
                                         pad(descB, (1, nbytes), np.uint8), pad(ptsB, (1, 2), np.float64), nB,
                                         nbytes, int(max_distance), np.float32(max_dist_sq), m1, m2, pairs.reshape(-1))
    if cnt < 0:
        raise MemoryError("oracle allocation failed")
    return pairs[:cnt].copy(), m1[:nA].copy(), m2[:nB].copy()


def match_l2(descA, ptsA, descB, ptsB, max_dist_sq_desc, max_dist_sq=1e8, ratio=0.0):
    """Mutual nearest neighbours under the squared Euclidean distance (extension; no counterpart in the
    reference).  Returns (pairs[k,2] int32, matches1[nA], matches2[nB])."""
    ptsA = np.ascontiguousarray(ptsA, np.float64).reshape(-1, 2)
    ptsB = np.ascontiguousarray(ptsB, np.float64).reshape(-1, 2)
    nA, nB = len(ptsA), len(ptsB)
    descA = np.ascontiguousarray(descA, np.float32).reshape(nA, -1) if nA else np.zeros((0, 128), np.float32)
    descB = np.ascontiguousarray(descB, np.float32).reshape(nB, -1) if nB else np.zeros((0, descA.shape[1]), np.float32)
    dim = descA.shape[1] if nA else descB.shape[1]
    m1 = np.full(max(nA, 1), -1, np.int32)
    m2 = np.full(max(nB, 1), -1, np.int32)
    pairs = np.zeros((max(min(nA, nB), 1), 2), np.int32)
    pad = lambda a, shape, dt: a if a.size else np.zeros(shape, dt)
    cnt = lib().sfm_oracle_match_l2(pad(descA, (1, dim), np.float32), pad(ptsA, (1, 2), np.float64), nA,
                                    pad(descB, (1, dim), np.float32), pad(ptsB, (1, 2), np.float64), nB, dim,
                                    np.float32(max_dist_sq_desc), np.float32(max_dist_sq), np.float32(ratio), m1, m2,
                                    pairs.reshape(-1))
    if cnt < 0:
        raise MemoryError("oracle allocation failed")
    return pairs[:cnt].copy(), m1[:nA].copy(), m2[:nB].copy()
