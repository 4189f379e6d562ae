"""Batch matcher on descriptor lengths other than 128 (the exact pass has a kernel of its own for those) and with a
loose threshold under an effective pixel gate: pairs per second on 24 images x 8192, 276 pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

ctx = capi.Context(0)
pl = np.ascontiguousarray(synth.exhaustive_pairs(24)).astype(np.int32)
for dim, thr, gate in ((128, 0.95, 1e8), (64, 0.95, 1e8), (96, 0.95, 1e8), (17, 0.95, 1e8), (128, 0.5, 300.0 ** 2), (128, 0.5, 1e8)):
    d, p, c = synth.image_set(24, 8192, dim=dim, seed=2026)
    s = ctx.imageset(c, d, p)
    res = s.match(pl, threshold=thr, max_dist_sq=gate)
    ctx.synchronize()
    t0 = time.perf_counter()
    res = s.match(pl, threshold=thr, max_dist_sq=gate, result=res)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"dim {dim} threshold {thr} gate {gate:g}: {len(pl) / dt:.0f} pairs/s, matches {res.total()}, stats {res.stats_ex()}")
    res.close(); s.close()
