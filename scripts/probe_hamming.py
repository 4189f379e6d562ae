"""Throughput of the binary matcher: exhaustive pairs over images x 8192 descriptors of 256 bits."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
n = 8192
pool = rng.integers(0, 256, size=(2 * n, 32), dtype=np.uint8)
desc, pts = [], []
for _ in range(n_img):
    pick = rng.choice(len(pool), size=n, replace=False)
    d = pool[pick] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1)
    desc.append(d)
    pts.append(rng.uniform(0, 4000, size=(n, 2)))
ctx = capi.Context(0)
s = capi.BinarySet(ctx, np.full(n_img, n, np.int32), np.concatenate(desc), np.concatenate(pts))
pl = np.array([(a, b) for a in range(n_img) for b in range(a + 1, n_img)], np.int32)
res = s.match(pl, max_distance=64)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    res = s.match(pl, max_distance=64, result=res)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 3
ms, launches = ctx.last_kernel_ms(0)
cells = len(pl) * n * n
print(f"{len(pl)} pairs: {len(pl) / dt:.0f} pairs/s, {dt * 1e3:.1f} ms per batch, kernel {ms:.1f} ms in {launches} launches, "
      f"{cells / (ms * 1e-3) / 1e12:.2f} T cells/s, matches per pair {res.total() / len(pl):.0f}")
res.close(); s.close(); ctx.close()
