"""Timing probe of the matcher's similarity-kernel variants on n images x 8192 features
(exhaustive pairs), variants interleaved in one process."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
variants = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
d, p, c = synth.image_set(n_img, n_feat, seed=5)
ctx = capi.Context(0)
s = ctx.imageset(c, d, p)
pl = synth.exhaustive_pairs(n_img)
flops = 2.0 * n_feat * n_feat * 128 * len(pl)
res = None
best = {}
ref_total = None
for rd in range(rounds + 1):
    for v in variants:
        ctx.set_option("sim_variant", v)
        t = time.time()
        res = s.match(pl, threshold=thr, result=res)
        ctx.synchronize()
        dt = time.time() - t
        ms, nl = ctx.last_kernel_ms(0)
        tot = res.total()
        if ref_total is None:
            ref_total = tot
        assert tot == ref_total or v >= 2, (v, tot, ref_total)
        if rd > 0:
            b = best.setdefault(v, [1e9, 1e9, []])
            b[0] = min(b[0], ms); b[1] = min(b[1], dt * 1e3); b[2].append(ms)
for v in variants:
    ms, wall, allms = best[v]
    print(f"variant {v}: sim kernel min {ms:.3f} ms med {np.median(allms):.3f} -> {flops/ms/1e9:.0f} Tops/s ({flops/ms/1e9/25:.1f}% of 2.5 P, {flops/ms/1e9/50:.1f}% of 5 P); "
          f"wall {wall:.2f} ms -> {len(pl)/wall*1e3:.0f} pairs/s; matches {ref_total}", flush=True)
