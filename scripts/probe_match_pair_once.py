"""Single-pair matcher calls (8192 x 8192, threshold 0.95, host tables in, list out) for a kernel trace, and their
wall time; argument "pinned" keeps the tables in page-locked memory, "reuse" alternates (A, B) and (B, A) with the
option match_pair_reuse, so that image A of every call is the image B the call before left on the device; "loose"
takes the reference's default threshold 0.5 (match.hpp:59)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

pinned = "pinned" in sys.argv[1:]
reuse = "reuse" in sys.argv[1:]
thr = 0.5 if "loose" in sys.argv[1:] else 0.95
ctx = capi.Context(0)
ctx.set_option("match_pair_reuse", int(reuse))
d, p, _ = synth.image_set(2, 8192, seed=2026)
tabs = [d[:8192], p[:8192], d[8192:], p[8192:]]
if pinned:
    tabs = [ctx.pinned_like(a) for a in tabs]
def call(k):
    t = tabs[2:] + tabs[:2] if reuse and (k & 1) else tabs
    return ctx.match_pair(*t, threshold=thr, max_dist_sq=1e8)
for k in range(20):
    m = call(k)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    for k in range(20):
        m = call(k)
    best = min(best, (time.perf_counter() - t) / 20 * 1e3)
print(f"{best:.3f} ms per call ({'page-locked' if pinned else 'pageable'} tables{', shared frame kept' if reuse else ''}), {len(m)} matches")
del tabs
ctx.close()
