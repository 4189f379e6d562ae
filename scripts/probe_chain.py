"""Times the scale chain on the default-run stand-in (host-array entry point and the device-chained one)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth, tracks

ctx = capi.Context(0)
frames, feats, H = 27, 8192, 8192
desc, pts, counts, _ = synth.frame_sequence(frames, feats, seed=4242)
K = synth.default_K()
pl = np.stack([np.arange(frames - 1), np.arange(1, frames)], axis=1).astype(np.int32)
s = ctx.imageset(counts, desc, pts)
res = s.match(pl, threshold=0.95, max_dist_sq=1e4)
corr = capi.Correspondences(ctx, s, res, K)
batch = corr.ransac(K, H, 1, max_sq_err=1.0, success_prob=0.999, adaptive=True)
off = res.offsets()
pose, used, rp = batch.refit(int(off[-1]))
ids, nt = tracks.assign_track_ids(pl, off, res.pairs(), counts)
for name, fn in (("refit-chained", lambda: batch.scale_chain(ids, n_tracks=nt)),):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
# host-array variant on the fetched refit
ioff, inl = batch.inliers()
e_off, e_ids, e_pts = [0], [], []
for k in range(len(pl)):
    rows = inl[ioff[k]:ioff[k + 1]] if ioff[k + 1] - ioff[k] >= 8 else np.arange(off[k + 1] - off[k])
    e_ids.append(ids[off[k] + rows]); e_pts.append(rp[off[k]:off[k] + used[k]]); e_off.append(e_off[-1] + len(rows))
e_off = np.array(e_off, np.int64); e_ids = np.concatenate(e_ids); e_pts = np.concatenate(e_pts)
ctx.scale_chain(e_off, e_ids, e_pts, pose, n_tracks=nt)
t0 = time.perf_counter()
for _ in range(20):
    ctx.scale_chain(e_off, e_ids, e_pts, pose, n_tracks=nt)
print("host arrays", (time.perf_counter() - t0) / 20 * 1e3, "ms")
for o in (batch, corr, res, s):
    o.close()
ctx.close()
