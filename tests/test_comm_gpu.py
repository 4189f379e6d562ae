"""The library's multi-GPU layer on the one GPU of the test box: communicators of one rank (both ways of
creating them), the gather of a match result (counts by all-gather, lists into the rank-major buffer) and
the one-process driver.  With one rank the rank-major buffer is the local CSR, so everything is checked
against sfm_match_batch's own result, and that against the CPU restatement."""
import numpy as np
import pytest

from sfm_amd import capi, shard, synth

pytestmark = pytest.mark.gpu


def image_set():
    counts = np.array([300, 0, 513, 1, 777, 640], np.int32)
    n = int(counts.max())
    d, p, _ = synth.image_set(len(counts), n, seed=31)
    desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)])
    pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)])
    pl = synth.exhaustive_pairs(len(counts))
    pl = np.concatenate([pl] * 6).astype(np.int32)  # 90 pairs: more than one block of 64
    return counts, desc, pts, pl


def check_against_local(g, res, pl):
    off, pairs = res.offsets(), res.pairs()
    starts, cnt = g.layout()
    assert g.total() == len(pairs) and g.per_rank_totals().tolist() == [len(pairs)]
    assert np.array_equal(starts, off[:-1]) and np.array_equal(cnt, np.diff(off))
    assert np.array_equal(g.pairs(), pairs)


def test_one_rank_gather(ctx, oracle):
    counts, desc, pts, pl = image_set()
    s = ctx.imageset(counts, desc, pts)
    comm = shard.library_comm(ctx, 0, 1)
    assert (comm.rank, comm.world) == (0, 1)
    owned = shard.partition_pairs(len(pl), 0, 1)
    assert np.array_equal(owned, np.arange(len(pl)))
    res = s.match(pl[owned], threshold=0.95, max_dist_sq=150.0 ** 2)
    g = shard.library_gather(comm, res, len(pl))
    check_against_local(g, res, pl)
    res = s.match(pl[owned][::-1].copy(), threshold=0.9, max_dist_sq=1e8, result=res)  # reuse of both objects
    g = shard.library_gather(comm, res, len(pl), gathered=g)
    check_against_local(g, res, pl)
    starts = np.concatenate([[0], np.cumsum(counts)])
    gs, gc = g.layout()
    allp = g.pairs()
    for k, (a, b) in enumerate(pl[::-1]):
        exp, _, _ = oracle.match(desc[starts[a]:starts[a + 1]], pts[starts[a]:starts[a + 1]],
                                 desc[starts[b]:starts[b + 1]], pts[starts[b]:starts[b + 1]], threshold=0.9, max_dist_sq=1e8)
        assert np.array_equal(allp[gs[k]:gs[k] + gc[k]], exp)
    with pytest.raises(capi.SfmError):  # a result over the wrong share
        comm.gather(res, len(pl) + 1)
    g.close()
    comm.close()
    res.close()
    s.close()


def test_one_process_driver(ctx):
    counts, desc, pts, pl = image_set()
    s = ctx.imageset(counts, desc, pts)
    comms = capi.Comm.local([ctx])
    results, gathered = capi.match_batch_multi(comms, [s], pl, threshold=0.95, max_dist_sq=150.0 ** 2)
    check_against_local(gathered[0], results[0], pl)
    ref = s.match(pl, threshold=0.95, max_dist_sq=150.0 ** 2)
    assert np.array_equal(ref.pairs(), gathered[0].pairs())
    results, gathered = capi.match_batch_multi(comms, [s], pl[:10], threshold=0.95, results=results, gathered=gathered)
    check_against_local(gathered[0], results[0], pl[:10])
    with pytest.raises(capi.SfmError):
        capi.Comm.local([ctx, ctx])  # one device listed twice
    for o in gathered + results + comms + [ref, s]:
        o.close()
