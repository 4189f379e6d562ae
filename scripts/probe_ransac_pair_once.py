"""Twenty single-pair RANSAC calls (4300 correspondences, 8192 hypotheses, adaptive) for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth

ctx = capi.Context(0)
tv = synth.two_view(4300, inlier_ratio=0.95, seed=[3, 95])
S = synth.sample_table(4300, 8192, seed=[9])
for _ in range(20):
    r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
print(r["iterations"], len(r["inliers"]))
ctx.close()
