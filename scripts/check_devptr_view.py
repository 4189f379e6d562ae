"""Zero-copy torch views of the library's device-resident CSR (the path bench.py takes at N > 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_amd import capi, synth, shard
d, p, c = synth.image_set(4, 2000, seed=5)
ctx = capi.Context(0)
s = ctx.imageset(c, d, p)
pl = synth.exhaustive_pairs(4)
res = s.match(pl, threshold=0.95)
total = res.total()
optr, pptr = res.device_ptrs()
dev = torch.device("cuda:0")
off_t = torch.as_tensor(shard.DevicePtrView(optr, (len(pl) + 1,), "<i8"), device=dev)
pay_t = torch.as_tensor(shard.DevicePtrView(pptr, (max(total, 1), 2), "<i4"), device=dev)[:total]
assert off_t.data_ptr() == optr and pay_t.data_ptr() == pptr, "not zero-copy"
assert np.array_equal(off_t.cpu().numpy(), res.offsets())
assert np.array_equal(pay_t.cpu().numpy(), res.pairs())
cnt = off_t[1:] - off_t[:-1]
out = shard.gather_csr(cnt, pay_t, len(pl), 0, 1)
print("ok", total, out[0][-1])
