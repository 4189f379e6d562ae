"""One matcher call at a loose threshold (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth
n_img, n_feat, thr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
d, p, c = synth.image_set(n_img, n_feat, seed=5)
ctx = capi.Context(0)
s = ctx.imageset(c, d, p)
pl = synth.exhaustive_pairs(n_img)
res = None
for it in range(3):
    res = s.match(pl, threshold=thr, result=res)
    ctx.synchronize()
print(res.stats(), res.total())
