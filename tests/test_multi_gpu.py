"""The multi-GPU layer (sfm_amd/csrc/comm.hip; no counterpart in the reference, which is one process that
handles one pair per call, submodules/scene.cpp:60,261).

With one rank (always run): the RANSAC gather, calls that reuse their objects back to back, float and binary
batches alternating on one result, the time limit.  With two or more visible devices (skipped below): one
process that drives two devices, and two processes with one device each -- the receive loop of the root, the
sends of the other ranks and the rank-major layout with world > 1."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from sfm_amd import capi, synth

import multi_cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def second_context():
    """A context on device 1, or None on a box with one GPU."""
    try:
        return capi.Context(1)
    except capi.SfmError as e:
        if e.code in (capi.ERR_NO_DEVICE, capi.ERR_INVALID_ARG, capi.ERR_HIP):
            return None
        raise


def test_ransac_gather_with_one_rank(ctx):
    off, x, y, K, S, H = multi_cases.ransac_inputs()
    P = len(off) - 1
    comm = capi.Comm.local([ctx])[0]
    b = multi_cases.ransac_share(ctx, np.arange(P), off, x, y, K, S, H)
    g = comm.gather_ransac(b, P)
    multi_cases.check_ransac(g.ransac(), b, P)
    assert g.total() == int(b.inlier_offsets()[-1])
    # reuse of batch and gathered object, no accessor in between
    b2 = ctx.ransac_batch(off[:51], x[:off[50]], y[:off[50]], K, S[:50], H, max_sq_err=4.0, adaptive=False, batch=b)
    g = comm.gather_ransac(b2, 50, gathered=g)
    b2 = ctx.ransac_batch(off[:51], x[:off[50]], y[:off[50]], K, S[:50], H, max_sq_err=1.0, adaptive=False, batch=b2)
    g = comm.gather_ransac(b2, 50, gathered=g)
    multi_cases.check_ransac(g.ransac(), b2, 50)
    with pytest.raises(capi.SfmError):
        comm.gather_ransac(b2, 51)           # a batch over the wrong share
    with pytest.raises(capi.SfmError):
        g.pairs()                             # not a matcher's object
    g.close(); b2.close(); comm.close()


def test_back_to_back_batches_into_one_result(ctx):
    """sfm_match_batch_multi twice in a row with no accessor in between: the second call's kernels overwrite the
    lists the first call's gather reads on the communicator's stream; the library orders them."""
    counts, desc, pts, pl = multi_cases.image_set()
    s = ctx.imageset(counts, desc, pts)
    comms = capi.Comm.local([ctx])
    results = gathered = None
    for rep in range(3):
        results, gathered = capi.match_batch_multi(comms, [s], pl, threshold=0.95, max_dist_sq=150.0 ** 2,
                                                   results=results, gathered=gathered)
        results, gathered = capi.match_batch_multi(comms, [s], pl[::-1].copy(), threshold=0.9, max_dist_sq=1e8,
                                                   results=results, gathered=gathered)
    ref = s.match(pl[::-1].copy(), threshold=0.9, max_dist_sq=1e8)
    assert np.array_equal(gathered[0].pairs(), ref.pairs())
    assert gathered[0].wait_timeout(1000)
    for o in gathered + results + comms + [ref, s]:
        o.close()


def test_float_and_binary_batches_alternate_on_one_result(ctx, oracle):
    """One result object used by sfm_match_batch, then sfm_match_batch_hamming, then gathered, and back: every
    call marks where its lists are complete, so the gather never reads lists that are still being written."""
    counts, desc, pts, pl = multi_cases.image_set()
    s = ctx.imageset(counts, desc, pts)
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 256, (int(counts.sum()), 32), dtype=np.uint8)
    bs = capi.BinarySet(ctx, counts, bits, pts)
    comm = capi.Comm.local([ctx])[0]
    res = s.match(pl, threshold=0.95, max_dist_sq=150.0 ** 2)
    g = None
    for rep in range(3):
        res = bs.match(pl, max_distance=110, max_dist_sq=1e8, result=res)
        g = comm.gather(res, len(pl), gathered=g)
        hp = g.pairs()
        res = s.match(pl, threshold=0.95, max_dist_sq=150.0 ** 2, result=res)
        g = comm.gather(res, len(pl), gathered=g)
        fp = g.pairs()
    ref_h = bs.match(pl, max_distance=110, max_dist_sq=1e8)
    ref_f = s.match(pl, threshold=0.95, max_dist_sq=150.0 ** 2)
    assert np.array_equal(hp, ref_h.pairs()) and np.array_equal(fp, ref_f.pairs())
    for o in (g, res, ref_h, ref_f, comm, bs, s):
        o.close()


def test_objects_without_a_gather_are_refused(ctx):
    comm = capi.Comm.local([ctx])[0]
    g = capi.Gathered(comm)
    assert g.wait_timeout(10)  # nothing in flight
    comm.close()


def test_two_devices_in_one_process(ctx):
    ctx1 = second_context()
    if ctx1 is None:
        pytest.skip("needs two visible GPUs")
    try:
        ctxs = [ctx, ctx1]
        for c in ctxs:
            c.set_option("comm_timeout_ms", 60000)
        comms = capi.Comm.local(ctxs)
        counts, desc, pts, pl = multi_cases.image_set()
        sets = [c.imageset(counts, desc, pts) for c in ctxs]
        results = gathered = None
        for root in (0, 1):
            results, gathered = capi.match_batch_multi(comms, sets, pl, threshold=0.95, max_dist_sq=150.0 ** 2, root=root,
                                                       results=results, gathered=gathered)
            results, gathered = capi.match_batch_multi(comms, sets, pl, threshold=0.9, max_dist_sq=1e8, root=root,
                                                       results=results, gathered=gathered)
            ref = sets[0].match(pl, threshold=0.9, max_dist_sq=1e8)
            roff, rpairs = ref.offsets(), ref.pairs()
            g = gathered[root]
            starts, cnt = g.layout()
            allp = g.pairs()
            assert g.total() == len(rpairs) and np.array_equal(cnt, np.diff(roff))
            assert np.array_equal(g.per_rank_totals(), [int(cnt[capi.shard_indices(len(pl), r, 2)].sum()) for r in range(2)])
            for k in range(len(pl)):
                assert np.array_equal(allp[starts[k]:starts[k] + cnt[k]], rpairs[roff[k]:roff[k + 1]]), k
            ref.close()
        # RANSAC: every device runs its share, the winners and inlier lists are gathered to device 1
        off, x, y, K, S, H = multi_cases.ransac_inputs()
        P = len(off) - 1
        batches = [multi_cases.ransac_share(c, capi.shard_indices(P, r, 2), off, x, y, K, S, H) for r, c in enumerate(ctxs)]
        gr = capi.ransac_gather_local(comms, batches, P, root=1)
        refb = ctx.ransac_batch(off, x, y, K, S, H, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        multi_cases.check_ransac(gr[1].ransac(), refb, P)
        for o in gr + batches + [refb] + gathered + results + sets + comms:
            o.close()
    finally:
        ctx1.close()


def test_two_processes_one_device_each(ctx):
    ctx1 = second_context()
    if ctx1 is None:
        pytest.skip("needs two visible GPUs")
    ctx1.close()
    with tempfile.TemporaryDirectory() as tmp:
        id_file = os.path.join(tmp, "comm_id")
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_rank_child.py"), str(r), "2", id_file],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append(out)
        for r, (p, out) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r}: OK" in out, out[-3000:]
