"""How long the host spends inside one batch call of the headline shape (the call only enqueues) against the time the
device needs for it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d, p, c = synth.image_set(n_img, 8192, seed=2026)
ctx = capi.Context(0)
s = ctx.imageset(c, d, p)
pl = synth.exhaustive_pairs(n_img)
res = None
for it in range(4):
    t0 = time.perf_counter()
    res = s.match(pl, threshold=0.95, max_dist_sq=1e8, result=res)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"call {it}: host inside the call {1e3 * (t1 - t0):.1f} ms, device done after {1e3 * (t2 - t0):.1f} ms", flush=True)
