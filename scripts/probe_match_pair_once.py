"""Twenty sfm_match_pair calls at 8192 x 8192 (host buffers in, list out) for a kernel trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
n = 8192
d, p, c = synth.image_set(2, n, seed=5)
dA, pA, dB, pB = d[:n], p[:n], d[n:], p[n:]
ctx.match_pair(dA, pA, dB, pB, threshold=0.95, max_dist_sq=1e4)
t = time.perf_counter()
for _ in range(20):
    m = ctx.match_pair(dA, pA, dB, pB, threshold=0.95, max_dist_sq=1e4)
print(f"{(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call, {len(m)} matches")
ctx.close()
