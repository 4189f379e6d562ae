"""One sfm_match_pair call (8192 x 8192, page-locked tables) at the tight threshold of the benchmark and at the
reference's default threshold 0.5 (match.hpp:59): milliseconds per call, mean of 30, best of three rounds."""
import json
import time

import numpy as np

from sfm_amd import capi, synth

ctx = capi.Context(0)
d, p, _ = synth.image_set(2, 8192, seed=2026)
dA, pA, dB, pB = (ctx.pinned_like(a) for a in (d[:8192], p[:8192], d[8192:], p[8192:]))
out = {}
for thr in (0.95, 0.5):
    for _ in range(10):
        ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=1e8)
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            m = ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=1e8)
        best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
    out[str(thr)] = {"ms_per_call": round(best, 3), "matches": int(len(m))}
print(json.dumps(out))
