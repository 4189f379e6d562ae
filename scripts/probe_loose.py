import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
n_img, n_feat = int(sys.argv[1]), int(sys.argv[2])
d, p, c = synth.image_set(n_img, n_feat, seed=5)
ctx = capi.Context(0)
s = ctx.imageset(c, d, p)
pl = synth.exhaustive_pairs(n_img)
for thr in [float(x) for x in sys.argv[3].split(",")]:
    for cap in [262144, 524288]:
        ctx.set_option("cand_capacity", cap)
        res = None
        for it in range(2):
            t = time.time(); res = s.match(pl, threshold=thr, result=res); ctx.synchronize(); dt = time.time() - t
        cand, ovf = res.stats()
        print(f"thr {thr} cap {cap}: {len(pl)/dt:.0f} pairs/s, {dt/len(pl)*1e3:.2f} ms/pair, cand/pair {cand/len(pl):.0f}, cell-by-cell pairs {ovf}, matches/pair {res.total()/len(pl):.0f}", flush=True)
