"""Batch calls on descriptors whose unrelated cosines (about 0.88) lie inside the 6-bit margin of a 0.95 threshold and
outside the int8 margin: the first call pays for a 6-bit pass that floods and an int8 pass that redoes every pair, the
calls after it start with the int8 pass.  python scripts/probe_six_flood.py [images] [weight of the individual part]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
n = 8192
rng = np.random.default_rng(7)
base = synth.sift_like(rng, 1)[0]
d = np.floor(np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n_img * n) * w, 0, 255)).astype(np.float32)
shared = int(0.3 * n)
for im in range(1, n_img):
    d[im * n:im * n + shared] = d[:shared]
pts = rng.uniform(0, 4000, (n_img * n, 2))
ctx = capi.Context(0)
s = ctx.imageset(np.full(n_img, n, np.int32), d, pts)
pl = synth.exhaustive_pairs(n_img)
res = None
for it in range(4):
    t = time.perf_counter()
    res = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"call {it}: {len(pl) / dt / 1e3:.1f} k pairs/s, stats {res.stats_ex()}, matches {res.total()}", flush=True)
