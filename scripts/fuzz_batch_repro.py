"""Replays one case of tests/test_match_fuzz_gpu.py::test_random_batch (argument: its index) and says which pairs
differ from the restatement with the 6-bit first pass on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import capi, synth
from oracle import oracle_api as oracle

seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
n_img = int(rng.integers(2, 7))
counts = rng.choice([0, 1, 40, 256, 257, 600, 1100], n_img).astype(np.int32)
n = max(int(counts.max()), 1)
d, p, _ = synth.image_set(n_img, n, shared_frac=float(rng.uniform(0.1, 0.7)), seed=int(rng.integers(1 << 30)))
kind = str(rng.choice(["sift", "sift", "common part", "gauss"]))
w = None
if kind == "common part":
    base = synth.sift_like(rng, 1)[0]
    keep = rng.random(len(d)) < 0.3
    w = float(rng.choice([0.42, 0.6, 0.8]))
    mixed = np.floor(np.clip(base[None, :] * 0.9 + synth.sift_like(rng, len(d)) * w, 0, 255)).astype(np.float32)
    d = np.where(keep[:, None], d, mixed)
elif kind == "gauss":
    d, p, _ = synth.image_set(n_img, n, shared_frac=float(rng.uniform(0.1, 0.7)), seed=int(rng.integers(1 << 30)), kind="gauss")
desc = np.concatenate([d[i * n:i * n + c] for i, c in enumerate(counts)] + [np.zeros((0, 128), np.float32)])
pts = np.concatenate([p[i * n:i * n + c] for i, c in enumerate(counts)] + [np.zeros((0, 2))])
pl = rng.integers(0, n_img, (int(rng.integers(1, 25)), 2)).astype(np.int32)
thr = float(rng.choice([0.95, 0.9, 0.6]))
gate = float(rng.choice([1e8, 120.0 ** 2]))
as_bytes = bool(rng.integers(2)) and kind != "gauss"
six = int(rng.choice([1, 1, 0]))
per = int(rng.choice([1, 3, 2048]))
obs = int(rng.choice([0, 1, 2, 2]))
print(f"kind {kind} weight {w} counts {counts.tolist()} pairs {len(pl)} thr {thr} gate {gate} bytes {as_bytes} six {six} per launch {per} observe {obs}")
ctx = capi.Context(0)
starts = np.concatenate([[0], np.cumsum(counts)])
for six_now in (six, 1 - six):
    for per_now in (per, 2048):
        ctx.set_option("match_six_bit", six_now)
        ctx.set_option("pairs_per_launch", per_now)
        ctx.set_option("match_list_observe", obs)
        s = ctx.imageset(counts, desc.astype(np.uint8) if as_bytes else desc, pts)
        res = s.match(pl, threshold=thr, max_dist_sq=gate)
        off, pairs = res.offsets(), res.pairs()
        bad = []
        for k, (a, b) in enumerate(pl):
            A, B = slice(starts[a], starts[a + 1]), slice(starts[b], starts[b + 1])
            exp, _, _ = oracle.match(desc[A], pts[A], desc[B], pts[B], threshold=thr, max_dist_sq=gate)
            got = pairs[off[k]:off[k + 1]]
            if not np.array_equal(got, exp):
                gs, es = set(map(tuple, got)), set(map(tuple, exp))
                bad.append((k, int(a), int(b), len(got), len(exp), sorted(es - gs)[:3], sorted(gs - es)[:3]))
        print(f"six {six_now} per launch {per_now}: stats {res.stats_ex()} differing pairs {bad[:4]} ({len(bad)})")
        res.close(); s.close()

# which of the later passes disagree: the same call under other options, on the first pair only
pl1 = pl[:1]
a, b = pl1[0]
A, B = slice(starts[a], starts[a + 1]), slice(starts[b], starts[b + 1])
exp, e1, e2 = oracle.match(desc[A], pts[A], desc[B], pts[B], threshold=thr, max_dist_sq=gate)
ctx.set_option("pairs_per_launch", 2048)
for bytes_now in (True, False):
    for two in (1, 0):
        for probe in (1, 0):
            for variant in (1, 0):
                ctx.set_option("match_two_phase", two); ctx.set_option("match_flood_probe", probe); ctx.set_option("sim_variant", variant)
                s = ctx.imageset(counts, desc.astype(np.uint8) if bytes_now else desc, pts)
                res = s.match(pl1, threshold=thr, max_dist_sq=gate)
                got = res.pairs()
                print(f"bytes {bytes_now} two_phase {two} probe {probe} variant {variant}: {'same' if np.array_equal(got, exp) else 'DIFFERENT'} {res.stats_ex()}")
                res.close(); s.close()
dA, dB = desc[A].astype(np.float32), desc[B].astype(np.float32)
def cos32(x, y):
    dot = np.float32(0); nx = np.float32(0); ny = np.float32(0)
    for k in range(len(x)):
        dot = np.float32(dot + np.float32(x[k] * y[k])); nx = np.float32(nx + np.float32(x[k] * x[k])); ny = np.float32(ny + np.float32(y[k] * y[k]))
    return dot / (np.sqrt(nx) * np.sqrt(ny))
for (i, j) in [(113, 736), (113, 355)]:
    print(i, j, repr(cos32(dA[i], dB[j])), "exact", float(dA[i].astype(np.float64) @ dB[j].astype(np.float64) / np.linalg.norm(dA[i].astype(np.float64)) / np.linalg.norm(dB[j].astype(np.float64))))
print("oracle best of row 113:", e1[113], "; best of columns 736, 355:", e2[736], e2[355])
