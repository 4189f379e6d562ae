"""Twenty single-pair RANSAC calls (4300 correspondences, 8192 hypotheses, adaptive, sample table drawn by the
library) for a kernel trace, and their wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
tv = synth.two_view(4300, inlier_ratio=0.95, seed=[3, 95])
ctx.set_option("ransac_seed", 7)
for k in range(3):
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], 8192, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
t = time.perf_counter()
for _ in range(20):
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], 8192, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
print(f"{(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call;", r["iterations"], "iterations,", len(r["inliers"]), "inliers")
ctx.close()
