"""RANSAC stress shape (64 pairs x 20000 correspondences x 4096 rows, 15 % inliers) with the reprojection loss and
with the Sampson distance (extension): hypotheses scored per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

ctx = capi.Context(0)
P, n, H = 64, 20000, 4096
K = synth.default_K()
xs, ys = [], []
for k in range(8):
    tv = synth.two_view(n, inlier_ratio=0.15, seed=[3, k])
    xs.append(tv["x"]); ys.append(tv["y"])
x = np.concatenate([xs[k % 8] for k in range(P)]); y = np.concatenate([ys[k % 8] for k in range(P)])
off = np.arange(P + 1, dtype=np.int64) * n
S = np.stack([synth.sample_table(n, H, seed=100 + k) for k in range(P)])
for name, loss in (("reprojection", capi.LOSS_REPROJECTION), ("sampson", capi.LOSS_SAMPSON)):
    b = ctx.ransac_batch(off, x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=True, loss=loss)
    ctx.synchronize()
    t0 = time.perf_counter()
    b = ctx.ransac_batch(off, x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=True, loss=loss, batch=b)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(name, round(P * H / dt / 1e6, 1), "M hyps/s (incl. upload of x, y and the sample table)", float(np.mean(b.best()["count"])))
    b.close()
