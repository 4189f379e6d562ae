"""A/B of the single-pair matcher's upload: both images on one stream (option match_overlap 0) or the second image on
a stream of its own (1), page-locked tables, alternating rounds on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
d, p, _ = synth.image_set(2, 8192, seed=2026)
tabs = [ctx.pinned_like(a) for a in (d[:8192], p[:8192], d[8192:], p[8192:])]
for _ in range(30):
    ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8)
best = {0: 1e9, 1: 1e9}
for rnd in range(8):
    for mode in (0, 1):
        ctx.set_option("match_overlap", mode)
        for _ in range(5):
            ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8)
        t = time.perf_counter()
        for _ in range(40):
            ctx.match_pair(*tabs, threshold=0.95, max_dist_sq=1e8)
        ms = (time.perf_counter() - t) / 40 * 1e3
        best[mode] = min(best[mode], ms)
        print(f"round {rnd} match_overlap {mode}: {ms:.3f} ms per call")
print(f"best: one stream {best[0]:.3f} ms, two streams {best[1]:.3f} ms")
del tabs
ctx.close()
