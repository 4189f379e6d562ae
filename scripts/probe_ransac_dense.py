"""RANSAC scoring where most correspondences are inliers of most hypotheses (the reference's default run):
pre-test + queue against plain evaluation of everything."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
ctx = capi.Context(0)
P, n, H = 26, 4300, 8192
xs, ys, Ss = [], [], []
for k in range(P):
    tv = synth.two_view(n, inlier_ratio=0.82, seed=k, noise_px=0.3)
    xs.append(tv["x"]); ys.append(tv["y"]); Ss.append(synth.sample_table(n, H, seed=100 + k))
x, y, S = np.concatenate(xs), np.concatenate(ys), np.stack(Ss)
off = np.arange(P + 1, dtype=np.int64) * n
for filt in (1, 0, 1, 0):
    ctx.set_option("ransac_filter", filt)
    b = None
    for it in range(3):
        t = time.perf_counter()
        b = ctx.ransac_batch(off, x, y, synth.default_K(), S, H, max_sq_err=1.0, success_prob=0.999, adaptive=False, batch=b)
        ctx.synchronize()
        dt = time.perf_counter() - t
    print(f"filter {filt}: {dt*1e3:.2f} ms, score kernel {ctx.last_kernel_ms(1)[0]:.2f} ms, mean inliers {b.best()['count'].mean():.0f}")
