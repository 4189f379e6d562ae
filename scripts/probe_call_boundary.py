"""Host-side timing of one batch matcher call at the headline shape: time until the call returns, time until the
device is done, and the device-side span of the similarity kernels."""
import sys, time
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
sys.path.insert(0, ".")
from sfm_amd import capi, synth

ctx = capi.Context(0)
n_images = 200
desc, pts, counts = synth.image_set(n_images, 8192, seed=2026)
pl = synth.exhaustive_pairs(n_images)
s = ctx.imageset(counts, desc, pts)
res = [None, None]
for k in range(6):
    ctx.synchronize()
    t0 = time.perf_counter()
    res[k & 1] = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res[k & 1])
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    ms, nl = ctx.last_kernel_ms(0)
    print("call %d: returned after %.2f ms, device done after %.2f ms, similarity kernels %.2f ms in %d launches, rest %.2f ms"
          % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3, ms, nl, (t2 - t0) * 1e3 - ms))
