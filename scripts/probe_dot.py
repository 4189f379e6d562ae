import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
ctx = capi.Context(0)
n_img, n = 24, 8192
d, p, c = synth.image_set(n_img, n, seed=2026)
dn = d / np.linalg.norm(d, axis=1, keepdims=True)   # unit-norm descriptors: dotProduct == cosine up to rounding
pl = np.ascontiguousarray(synth.exhaustive_pairs(n_img)).astype(np.int32)
for name, tab, metric, thr in (("cosine", d, 0, 0.95), ("dot on unit rows", dn.astype(np.float32), 1, 0.95), ("dot on raw rows", d, 1, 0.95 * 512 * 512 / 4)):
    s = ctx.imageset(c, tab, p)
    res = s.match(pl, threshold=thr, max_dist_sq=1e8, metric=metric)
    ctx.synchronize()
    t0 = time.perf_counter()
    res = s.match(pl, threshold=thr, max_dist_sq=1e8, metric=metric, result=res)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(name, round(len(pl) / dt), "pairs/s", res.total(), res.stats_ex())
    res.close(); s.close()
