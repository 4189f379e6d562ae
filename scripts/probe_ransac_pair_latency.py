"""Latency of one solve::RANSAC call through the C-ABI (host buffers in, inliers out) on a pair the
size of the default run's: ~4300 correspondences, 8192 hypotheses, adaptive cap on."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
for ratio in (0.95, 0.5, 0.15):
    tv = synth.two_view(4300, inlier_ratio=ratio, seed=[3, int(ratio * 100)])
    S = synth.sample_table(4300, 8192, seed=[9])
    for split in (1, 0):
        ctx.set_option("ransac_split", split)
        for per_row in (False, True):
            r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True,
                                per_row=per_row)
            t0 = time.perf_counter()
            for _ in range(10):
                r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True,
                                    per_row=per_row)
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"inlier ratio {ratio}: split {split} per-row outputs {per_row}: {ms:.3f} ms, iterations "
                  f"{r['iterations']}, inliers {len(r['inliers'])}", flush=True)
ctx.close()
