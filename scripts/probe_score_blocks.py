"""Per-block start/end times of the unsplit score kernel.  Needs a measurement build of the library:
(cd sfm_amd/csrc && touch ransac.hip && make EXTRA=-DSFM_DEBUG_TIMES); rebuild without EXTRA afterwards."""
import argparse, ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init(); torch.zeros(1, device='cuda:0')
import bench
from sfm_amd import capi

args = argparse.Namespace(ransac_pairs=64, ransac_corr=20000, ransac_hyps=4096, steps=1)
ctx = capi.Context(0)
out = bench.bench_ransac(ctx, args, 0, 1, None, 0)
print(out["value"], out["score_kernel_ms_per_step"])
L = capi.lib()
n = 1024
buf = np.zeros(3 * n, dtype=np.uint64)
L.sfm_debug_score_times.argtypes = [C.c_void_p, C.c_int]
rc = L.sfm_debug_score_times(buf.ctypes.data, 3 * n)
t0 = buf[0::3].astype(np.int64); t1 = buf[1::3].astype(np.int64); hw = buf[2::3]
base = t0.min()
dur = (t1 - t0) / 100.0  # us at 100 MHz
print("rc", rc, "start spread us", (t0.max() - base) / 100.0, "kernel us", (t1.max() - base) / 100.0)
print("block dur us: min %.0f p10 %.0f med %.0f p90 %.0f max %.0f mean %.0f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), dur.mean()))
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
u, c = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "blocks per CU min/max", c.min(), c.max(), "hist", np.bincount(c))
# duration against blocks on the CU
for k in np.unique(c):
    sel = np.isin(key, u[c == k])
    print("  CUs holding", k, "blocks: mean block dur", dur[sel].mean(), "end max", ((t1[sel].max() - base) / 100.0))
late = t0 - base > 10000
print("blocks starting later than 100 us:", int(late.sum()))
np.save("gpurun_out/score_blocks.npy", buf)
ctx.close()
wave = hwid & 0xF; simd = (hwid >> 4) & 3
print("wave slots used by thread 0's wave:", np.bincount(wave), "simd", np.bincount(simd))
for w in np.unique(wave):
    print("  slot", w, "mean dur", dur[wave == w].mean())
