"""Leaves device objects open at interpreter exit; the process must still end with status 0."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
desc, pts, counts, _ = synth.frame_sequence(3, 500, seed=1)
s = ctx.imageset(counts, desc, pts)
res = s.match(np.array([[0, 1], [1, 2]], np.int32), threshold=0.95, max_dist_sq=1e4)
corr = capi.Correspondences(ctx, s, res, synth.default_K())
batch = corr.ransac(synth.default_K(), 64, 1, max_sq_err=1.0, success_prob=0.999, adaptive=True)
batch.refit(int(res.offsets()[-1]))
print("matches", res.total())
