"""RANSAC batches at inlier ratios where a model is found (the rows of bench.py's where_a_model_is_found, without
torch): python scripts/probe_ransac_regimes.py [ratio] [pairs] [filter] [adaptive] [reps] [first slice]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

ratio = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
filt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
adaptive = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
first = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # rows of the first slice (0: the library's choice)
n, H, distinct = 20000, 4096, 16
ctx = capi.Context(0)
xs, ys = [], []
for k in range(distinct):
    tv = synth.two_view(n, inlier_ratio=ratio, seed=[50 + int(ratio * 10), k])
    xs.append(tv["x"]); ys.append(tv["y"])
x = np.concatenate([xs[k % distinct] for k in range(P)])
y = np.concatenate([ys[k % distinct] for k in range(P)])
S = np.concatenate([synth.sample_table(n, H, seed=[9, k]) for k in range(P)])
off = np.arange(P + 1, dtype=np.int64) * n
ctx.set_option("ransac_filter", filt)
if first:
    ctx.set_option("hyps_per_launch", first)
b = None
for it in range(reps):
    t = time.perf_counter()
    b = ctx.ransac_batch(off, x, y, synth.default_K(), S, H, max_sq_err=1.0, success_prob=0.999, adaptive=bool(adaptive), batch=b)
    ctx.synchronize()
    dt = time.perf_counter() - t
    ms, launches = ctx.last_kernel_ms(1)
    print(f"ratio {ratio} pairs {P} filter {filt} adaptive {adaptive}: call {dt * 1e3:.1f} ms (uploads inside), score kernel {ms:.2f} ms in {launches} launches, "
          f"{P * H / (ms * 1e-3) / 1e6:.1f} M hyps/s by the kernel, mean best {b.best()['count'].mean():.0f}, mean iterations {b.best()['iterations'].mean():.0f}", flush=True)
it = b.best()["iterations"]
print("iterations percentiles 5/25/50/75/95/max:", [int(np.percentile(it, q)) for q in (5, 25, 50, 75, 95, 100)], "share above 1280:", float((it > 1280).mean()), "share at 4096:", float((it >= 4096).mean()))
