"""Waves of the matrix-core scoring kernel: cells evaluated, drains, running times (measurement build only).
Build the library with the counters into a file of its own and point SFM_AMD_LIB at it:
  cd sfm_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
     -DSFM_DEBUG_WAVES -c ransac.hip -o /tmp/ransac_dbg.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsfm_amd_dbg.so \
     context.o match.o /tmp/ransac_dbg.o pipeline.o chain.o hamming.o comm.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  SFM_AMD_LIB=/tmp/libsfm_amd_dbg.so python scripts/probe_ransac_waves.py [ratio] [adaptive] [first slice]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
ratio = float(sys.argv[1]); adaptive = int(sys.argv[2]); first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
P, n, H, distinct = 256, 20000, 4096, 16
ctx = capi.Context(0)
L = C.CDLL(capi.LIB_PATH)
xs, ys = [], []
for k in range(distinct):
    tv = synth.two_view(n, inlier_ratio=ratio, seed=[50 + int(ratio * 10), k]); xs.append(tv["x"]); ys.append(tv["y"])
x = np.concatenate([xs[k % distinct] for k in range(P)]); y = np.concatenate([ys[k % distinct] for k in range(P)])
S = np.concatenate([synth.sample_table(n, H, seed=[9, k]) for k in range(P)])
off = np.arange(P + 1, dtype=np.int64) * n
if first: ctx.set_option("hyps_per_launch", first)
out = (C.c_ulonglong * 8)()
b = None
for it in range(2):
    b = ctx.ransac_batch(off, x, y, synth.default_K(), S, H, max_sq_err=1.0, success_prob=0.999, adaptive=bool(adaptive), batch=b)
    ctx.synchronize()
    ms, launches = ctx.last_kernel_ms(1)
    L.sfm_debug_wave_stats(out)
    ev, tsum, tmax, waves, drains, second, evmax = [int(v) for v in out[:7]]
    print(f"ratio {ratio} adaptive {adaptive} first {first}: kernel {ms:.2f} ms in {launches} launches; waves {waves}, evaluations {ev} ({ev / max(waves,1):.0f} per wave, max {evmax}), drains {drains} ({ev/max(drains,1):.1f} entries each), "
          f"wave time mean {tsum / max(waves,1) / 100:.0f} us max {tmax / 100:.0f} us, second sweeps {second}; evaluations/s if alone {ev / (ms * 1e-3) / 1e9:.1f} G", flush=True)
