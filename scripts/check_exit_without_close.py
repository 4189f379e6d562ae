"""Destroys the context first and leaves every other device object to the interpreter's exit; the process
must still end with status 0 (run it under rocprofv3 --kernel-trace too: an object destroyed after its
context used to abort in the profiler's stream table)."""
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
n = res.total()
ctx.close()
import atexit
atexit.unregister(capi._close_all)
print("matches", n, "after the context:", res.total())
