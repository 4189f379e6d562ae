"""The matcher on descriptors whose unrelated cosines sit inside the int8 margin (the benchmark's "hard" row):
256 pairs of 40 images x 8192, threshold 0.95; wall time of one call and the pass counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

n, n_img = 8192, 40
ctx = capi.Context(0)
rng = np.random.default_rng(7)
base = synth.sift_like(rng, 1)[0]
hard = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n_img * n) * 0.42, 0, 255).astype(np.float32)
shared = int(0.3 * n)
for im in range(1, n_img):
    hard[im * n:im * n + shared] = hard[:shared]
pts = rng.uniform(0, 4000, (n_img * n, 2))
s = ctx.imageset(np.full(n_img, n, np.int32), hard, pts)
pl = np.ascontiguousarray(synth.exhaustive_pairs(n_img)[:256]).astype(np.int32)
res = s.match(pl, threshold=0.95, max_dist_sq=1e8)
ctx.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    res = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res)
    ctx.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"{best * 1e3:.3f} ms per call, {len(pl) / best:.0f} pairs/s, stats {res.stats_ex()}, matches {res.total()}")
