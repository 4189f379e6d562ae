"""Work shared by the multi-GPU tests (one process per GPU and one process with several devices): the
inputs, one rank's share of the work, and the root's checks."""
import numpy as np

from sfm_amd import capi, synth


def image_set():
    counts = np.array([300, 0, 513, 1, 777, 640, 256, 255], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=31)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    pl = synth.exhaustive_pairs(len(counts))
    pl = np.concatenate([pl] * 8).astype(np.int32)  # 224 pairs: several blocks of 64 per rank, a short last one
    return counts, desc, pts, pl


def ransac_inputs(n_pairs=150, seed=5):
    """A ragged batch: pair p has 40 + 13 p % 400 correspondences, a few below 8."""
    sizes = [(40 + 13 * p) % 400 for p in range(n_pairs)]
    sizes[3] = 5
    sizes[70] = 0
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    xs, ys, S = [], [], []
    H = 64
    K = synth.default_K()
    for p, n in enumerate(sizes):
        if n == 0:
            xs.append(np.zeros((0, 2))); ys.append(np.zeros((0, 2)))
        else:
            tv = synth.two_view(max(n, 8), inlier_ratio=0.6, seed=[seed, p])
            xs.append(tv["x"][:n]); ys.append(tv["y"][:n])
        S.append(synth.sample_table(max(n, 8), H, seed=[seed, p, 1]) % max(n, 1))
    return off, np.concatenate(xs), np.concatenate(ys), K, np.stack(S).astype(np.int32), H


def ransac_share(ctx, owned, off, x, y, K, S, H):
    sizes = np.diff(off)[owned]
    loff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    lx = np.concatenate([x[off[i]:off[i + 1]] for i in owned] + [np.zeros((0, 2))])
    ly = np.concatenate([y[off[i]:off[i + 1]] for i in owned] + [np.zeros((0, 2))])
    return ctx.ransac_batch(loff, lx, ly, K, S[owned], H, max_sq_err=1.0, success_prob=0.999, adaptive=True)


def check_ransac(got, ref_batch, n_pairs):
    best = ref_batch.best()
    off, inl = ref_batch.inliers()
    assert np.array_equal(got["best_h"], best["best_h"]) and np.array_equal(got["count"], best["count"])
    assert np.array_equal(got["iterations"], best["iterations"])
    assert np.array_equal(got["toterr"].view(np.uint64), best["toterr"].view(np.uint64))
    assert np.array_equal(got["pose12"].view(np.uint64), best["pose12"].view(np.uint64))
    assert np.array_equal(got["inlier_offsets"], off) and np.array_equal(got["inliers"], inl)
    assert len(got["best_h"]) == n_pairs


def run_rank(ctx, comm, rank, world):
    """One process per GPU: this rank's share of the matcher's and the RANSAC work, both gathered to rank 0,
    which checks them against the whole work done on its own device."""
    counts, desc, pts, pl = image_set()
    s = ctx.imageset(counts, desc, pts)
    owned = capi.shard_indices(len(pl), rank, world)
    res = s.match(pl[owned], threshold=0.95, max_dist_sq=150.0 ** 2)
    g = comm.gather(res, len(pl))
    # a second call into the same objects with no accessor in between: the library orders the overwrite
    # behind the gather that still reads the lists
    res = s.match(pl[owned], threshold=0.9, max_dist_sq=1e8, result=res)
    g = comm.gather(res, len(pl), gathered=g)
    assert g.wait_timeout(60000)
    off, x, y, K, S, H = ransac_inputs()
    P = len(off) - 1
    rowned = capi.shard_indices(P, rank, world)
    b = ransac_share(ctx, rowned, off, x, y, K, S, H)
    gr = comm.gather_ransac(b, P)
    assert gr.wait_timeout(60000)
    if rank == 0:
        ref = s.match(pl, threshold=0.9, max_dist_sq=1e8)
        roff, rpairs = ref.offsets(), ref.pairs()
        starts, cnt = g.layout()
        allp = g.pairs()
        assert g.total() == len(rpairs) and int(g.per_rank_totals().sum()) == len(rpairs)
        assert np.array_equal(cnt, np.diff(roff))
        for k in range(len(pl)):
            assert np.array_equal(allp[starts[k]:starts[k] + cnt[k]], rpairs[roff[k]:roff[k + 1]]), k
        refb = ctx.ransac_batch(off, x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        check_ransac(gr.ransac(), refb, P)
        refb.close()
        ref.close()
    for o in (gr, g, b, res, s):
        o.close()
