"""Single-pair matcher calls (8192 x 8192, threshold 0.95, host tables in, list out) for a kernel trace, and their
wall time; argument "pinned" keeps the tables in page-locked memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

pinned = "pinned" in sys.argv[1:]
ctx = capi.Context(0)
d, p, _ = synth.image_set(2, 8192, seed=2026)
tabs = [d[:8192], p[:8192], d[8192:], p[8192:]]
if pinned:
    tabs = [ctx.pinned_like(a) for a in tabs]
for _ in range(20):
    m = ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    for _ in range(20):
        m = ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8)
    best = min(best, (time.perf_counter() - t) / 20 * 1e3)
print(f"{best:.3f} ms per call ({'page-locked' if pinned else 'pageable'} tables), {len(m)} matches")
del tabs
ctx.close()
