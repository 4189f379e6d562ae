"""Does the single-pair RANSAC call's latency depend on what else the process holds (torch, a large batch)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1:  # torch first (its HIP runtime must be the one the process loads)
    import torch
    torch.cuda.set_device(0)
    a = torch.zeros(10, device="cuda:0"); torch.cuda.synchronize()
from sfm_amd import capi, synth

def timeit(ctx, tv, tag):
    for k in range(3):
        r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], 8192, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
    t = time.perf_counter()
    for _ in range(20):
        r = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], 8192, max_sq_err=1.0, success_prob=0.999, adaptive=True, per_row=False)
    print(f"{tag}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call;", r["iterations"], "iterations,", len(r["inliers"]), "inliers", flush=True)

ctx = capi.Context(0)
ctx.set_option("ransac_seed", 7)
tv = synth.two_view(4051, inlier_ratio=0.8, seed=[3, 80], noise_px=0.3)
timeit(ctx, tv, "plain")
timeit(ctx, tv, "again")
big = synth.two_view(20000, inlier_ratio=0.15, seed=1)
S = synth.sample_table(20000, 4096, seed=2)
off = np.arange(65, dtype=np.int64) * 20000
b = ctx.ransac_batch(off, np.tile(big["x"], (64, 1)), np.tile(big["y"], (64, 1)), big["K"], np.tile(S, (64, 1)), 4096, max_sq_err=1.0, adaptive=True)
ctx.synchronize()
timeit(ctx, tv, "after a batch")
ctx.close()
