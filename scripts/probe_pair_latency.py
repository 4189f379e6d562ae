"""Latency of the per-pair entry points (what the reference's call sites would see)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
ctx = capi.Context(0)
for n in (2000, 8192):
    d, p, c = synth.image_set(2, n, seed=5)
    dA, pA, dB, pB = d[:n], p[:n], d[n:], p[n:]
    for it in range(6):
        t = time.perf_counter()
        m = ctx.match_pair(dA, pA, dB, pB, threshold=0.95, max_dist_sq=1e4)
        dt = time.perf_counter() - t
    print(f"match_pair {n}x{n}: {dt*1e3:.2f} ms, {len(m)} matches", flush=True)
tv = synth.two_view(4000, inlier_ratio=0.7, seed=1)
S = synth.sample_table(4000, 8192, seed=2)
for it in range(5):
    t = time.perf_counter()
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    dt = time.perf_counter() - t
print(f"ransac_pair 4000 corr x 8192 hyps: {dt*1e3:.2f} ms")
