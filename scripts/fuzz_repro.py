"""Re-run one case of tests/test_match_fuzz_gpu.py and show where it differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sfm_amd import capi
from oracle import oracle_api as oracle
from test_match_fuzz_gpu import _case
seed = int(sys.argv[1])
oracle.build()
ctx = capi.Context(0)
rng = np.random.default_rng(9000 + seed)
dA, pA, dB, pB, thr, gate, metric = _case(rng)
variant = int(rng.choice([0, 1, 1])); cap = int(rng.choice([524288, 524288, 16384, 1024, 0]))
print("nA,nB,dim", dA.shape, dB.shape, "thr", thr, "gate", gate, "metric", metric, "variant", variant, "cap", cap)
exp, e1, e2 = oracle.match(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric)
for v in ([variant] if len(sys.argv) < 3 else [0, 1]):
    for c in ([cap] if len(sys.argv) < 3 else [524288, 16384, 1024, 0]):
        ctx.set_option("sim_variant", v); ctx.set_option("cand_capacity", c)
        got, g1, g2 = ctx.match_pair(dA, pA, dB, pB, threshold=thr, max_dist_sq=gate, metric=metric, return_best=True)
        bad1 = np.flatnonzero(g1 != e1); bad2 = np.flatnonzero(g2 != e2)
        print("variant", v, "cap", c, "m1 diffs", len(bad1), "m2 diffs", len(bad2), "pairs equal", np.array_equal(got, exp))
        for i in bad1[:5]:
            print("  row", i, "got", g1[i], "exp", e1[i], "sim got", oracle.similarity(dA[i], dB[g1[i]], metric) if g1[i] >= 0 else None,
                  "sim exp", oracle.similarity(dA[i], dB[e1[i]], metric) if e1[i] >= 0 else None, "normA", np.linalg.norm(dA[i]))
        for j in bad2[:5]:
            print("  col", j, "got", g2[j], "exp", e2[j], "sim got", oracle.similarity(dA[g2[j]], dB[j], metric) if g2[j] >= 0 else None,
                  "sim exp", oracle.similarity(dA[e2[j]], dB[j], metric) if e2[j] >= 0 else None, "normB", np.linalg.norm(dB[j]))
