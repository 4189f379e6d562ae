"""Quick timing probe of the matcher on n images x 8192 features (exhaustive pairs)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
t = time.time()
d, p, c = synth.image_set(n_img, n_feat, seed=5)
print(f"gen {time.time()-t:.2f}s", flush=True)
ctx = capi.Context(0)
t = time.time()
s = ctx.imageset(c, d, p)
print(f"ingest {time.time()-t:.3f}s, {s.device_bytes()/1e6:.0f} MB", flush=True)
pl = synth.exhaustive_pairs(n_img)
res = None
for it in range(4):
    t = time.time()
    res = s.match(pl, threshold=thr, result=res)
    ctx.synchronize()
    dt = time.time() - t
    ms, nl = ctx.last_kernel_ms(0)
    cand, ovf = res.stats()
    flops = 2.0 * n_feat * n_feat * 128 * len(pl)
    print(f"iter {it}: {len(pl)} pairs in {dt*1e3:.2f} ms -> {len(pl)/dt:.0f} pairs/s; sim kernel {ms:.2f} ms "
          f"({nl} launches) -> {flops/ms/1e9:.1f} TFLOP/s; matches {res.total()} cand {cand} ovf {ovf}", flush=True)
