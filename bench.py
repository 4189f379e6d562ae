#!/usr/bin/env python3
"""Benchmark of the hot path: exhaustive-pair descriptor matching at 8192 features/image
(the metric's configuration: 200 synthetic images x 8192 x 128-d float descriptors on one GPU),
plus the RANSAC stress shape as a secondary figure.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed.run launches N of them).  A step is one pass of the
matcher over this rank's pairs with the image table already resident in HBM; at N > 1 the pair
list grows with N (weak scaling: 19 900 pairs per GPU), every rank holds the whole table, and the
step ends with the gather of the match lists to rank 0.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak of MI355X
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA peak: the 16x16x64 int8 form takes the cycles of the bf16 16x16x32 form at twice the K
BOUNDS = {"int8": (1, "i8", PEAK_I8_TOPS), "fp16": (0, "f16", PEAK_F16_TFLOPS)}  # sim_variant, dtype, peak
PEAK_F64_VEC_TFLOPS = 78.6  # fp64 vector peak
RANSAC_BYTES_PER_PAIR = 1_082_368  # 20 000 corr * 32 B + 4096 hyp * (96 + 12) B
RANSAC_FLOPS_PER_EVAL = 92


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=200, help="images at N=1 (pairs per GPU = images*(images-1)/2)")
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--threshold", type=float, default=0.95)
    ap.add_argument("--ratio", type=float, default=None,
                    help="also apply Lowe's second-best ratio test (extension; not the reference's matcher)")
    ap.add_argument("--ransac-pairs", type=int, default=512, help="pairs per RANSAC batch (the stress shape has 10 000)")
    ap.add_argument("--ransac-corr", type=int, default=20000)
    ap.add_argument("--ransac-hyps", type=int, default=4096)
    ap.add_argument("--bound", choices=sorted(BOUNDS), default="int8",
                    help="arithmetic of the matcher's similarity bound (the exact re-scoring is fp32 either way)")
    ap.add_argument("--first-slice", type=int, default=0,
                    help="rows of the first slice of adaptive RANSAC batches (0 = the library's default)")
    ap.add_argument("--pairs-per-launch", type=int, default=0,
                    help="image pairs per matcher chunk (0 = the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--no-extensions", action="store_true", help="skip the modes the reference has no counterpart for")
    ap.add_argument("--cpu-pairs", type=int, default=24, help="8192x8192 pairs the CPU baseline matches (stops after 25 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    from sfm_amd import capi, shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    ctx = capi.Context(local_rank)
    sim_variant, bound_dtype, bound_peak = BOUNDS[args.bound]
    ctx.set_option("sim_variant", sim_variant)
    if "SFM_MATCH_OVERLAP" in os.environ:  # A/B switch of the matcher's two-stream chunk overlap
        ctx.set_option("match_overlap", int(os.environ["SFM_MATCH_OVERLAP"]))
    if args.first_slice > 0:
        ctx.set_option("hyps_per_launch", args.first_slice)
    if args.pairs_per_launch > 0:
        ctx.set_option("pairs_per_launch", args.pairs_per_launch)

    # ---- workload: weak scaling, pairs_per_gpu fixed ---------------------------------------
    pairs_per_gpu = args.images * (args.images - 1) // 2
    total_pairs = pairs_per_gpu * world
    n_images = args.images if world == 1 else shard.images_for_pairs(total_pairs)
    t0 = time.time()
    desc, pts, counts = synth.image_set(n_images, args.features, seed=2026)
    pair_list = synth.exhaustive_pairs(n_images)[:total_pairs]
    owned = shard.partition_pairs(total_pairs, rank, world)
    my_pairs = np.ascontiguousarray(pair_list[owned])
    imgset = ctx.imageset(counts, desc, pts)
    gen_s = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    # Two result objects: at N > 1 the gather of step k - 1 (RCCL, torch's streams) runs while the
    # library's stream matches step k.  Every gather is finished before the closing barrier.
    results = [None, None]
    gathered = None
    res = None

    def launch(k):
        results[k & 1] = imgset.match(my_pairs, threshold=args.threshold, max_dist_sq=1e8, result=results[k & 1],
                                      ratio=args.ratio)
        return results[k & 1]

    def gather(r):
        nonlocal gathered
        total = r.total()  # cached when its step finished
        optr, pptr = r.device_ptrs()
        dev = torch.device(f"cuda:{local_rank}")
        off_t = torch.as_tensor(shard.DevicePtrView(optr, (len(my_pairs) + 1,), "<i8"), device=dev)
        cnt_t = off_t[1:] - off_t[:-1]
        pay_t = torch.as_tensor(shard.DevicePtrView(pptr, (max(total, 1), 2), "<i4"), device=dev)[:total]
        gathered = shard.gather_csr(cnt_t, pay_t, total_pairs, rank, world, ordered=False)

    def run(n_steps):
        """n_steps matcher passes, each followed (at N > 1) by the gather of its match lists to rank 0."""
        nonlocal res
        ms, nl = 0.0, 0
        prev = None
        for k in range(n_steps):
            res = launch(k)
            if world > 1 and prev is not None:
                gather(prev)
            ctx.synchronize()
            if world > 1:
                torch.cuda.synchronize()  # the gather reads prev's buffers, which step k + 1 overwrites
                res.total()               # cached: gather() must not wait on the library's stream later
            a, b = ctx.last_kernel_ms(0)
            ms, nl = ms + a, nl + b
            prev = res
        if world > 1 and prev is not None:
            gather(prev)
            torch.cuda.synchronize()
        return ms, nl

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, kernel_launches = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_matches = res.total()
    cand, ovf = res.stats()
    value = total_pairs * args.steps / elapsed
    flops_per_pair = 2.0 * args.features * args.features * 128
    achieved_tflops = flops_per_pair * len(my_pairs) * args.steps / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0

    out = {
        "metric": "image pairs matched/sec @8k feats/img",
        "value": round(value, 2),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": bound_dtype,
        "dtype_note": ("int8-in/int32-acc" if args.bound == "int8" else "fp16-in/fp32-acc") +
                      " MFMA bounds every cell; cells that can pass are re-scored in the "
                      "reference's fp32 operation order, so match lists are bit-exact",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {n_images} images x {args.features} x 128-d float SIFT-like descriptors, "
                        f"exhaustive pairs ({total_pairs} pairs, {pairs_per_gpu}/GPU), threshold {args.threshold}"
                        + (f", second-best ratio test {args.ratio}" if args.ratio else ""),
            "images": n_images, "features_per_image": args.features, "dim": 128,
            "pairs_total": total_pairs, "pairs_per_gpu": pairs_per_gpu, "parallelism": f"pair-list dp{world}",
            "matches_rank0": int(total_matches), "candidates_rank0": int(cand), "overflow_rank0": int(ovf),
            "setup_s": round(gen_s, 2),
        },
        "roofline": {
            "bound": "mfma", "kernel": "match_sim_kernel",
            "achieved": round(achieved_tflops, 2), "peak": bound_peak, "unit": "TFLOP/s",
            "unit_note": "algorithmic 2*N*M*D operations per pair; with the int8 bound they are integer multiply-adds",
            "frac": round(achieved_tflops / bound_peak, 4),
            "traffic": measured_traffic(len(my_pairs) * args.steps / max(kernel_launches, 1), args.bound),
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, profiles/r1_pmc_traffic*.json)",
            "avg_launch_ms": round(kernel_ms / max(kernel_launches, 1), 4), "launches": kernel_launches,
            "flops_per_pair": flops_per_pair,
        },
    }

    # ---- secondary: RANSAC stress shape ------------------------------------------------------
    if not args.no_ransac:
        out["ransac"] = bench_ransac(ctx, args, rank, world, dist, local_rank)

    # ---- secondary: the stages chained on the device (default-run stand-in) --------------------
    if not args.no_pipeline and world == 1:
        out["pipeline"] = bench_pipeline(ctx, args)
    if not args.no_extensions and world == 1:
        out["extensions"] = bench_extensions(ctx, args)

    # ---- CPU baseline (rank 0, N = 1) ----------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, desc, pts, out.get("ransac"))

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(pairs_per_launch, bound="int8"):
    """HBM bytes per launch of the similarity kernel from the committed PMC pass (FETCH_SIZE and
    WRITE_SIZE are collected by rocprofv3 in separate runs of this script, not inline), scaled to
    this run's pairs per launch; None if the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic_i8.json" if bound == "int8" else "r1_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return round(t["hbm_bytes_per_pair"] * pairs_per_launch)


def bench_extensions(ctx, args):
    """Modes with no counterpart in the reference (checked against this repository's CPU restatement only):
    binary descriptors under the Hamming distance, 40 images x features x 256 bit, exhaustive pairs."""
    from sfm_amd import capi
    rng = np.random.default_rng(0)
    n, n_img = args.features, 40
    pool = rng.integers(0, 256, size=(2 * n, 32), dtype=np.uint8)
    desc, pts = [], []
    for _ in range(n_img):
        pick = rng.choice(len(pool), size=n, replace=False)
        desc.append(pool[pick] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1))
        pts.append(rng.uniform(0, 4000, size=(n, 2)))
    s = capi.BinarySet(ctx, np.full(n_img, n, np.int32), np.concatenate(desc), np.concatenate(pts))
    pl = np.array([(a, b) for a in range(n_img) for b in range(a + 1, n_img)], np.int32)
    res = s.match(pl, max_distance=64)
    ctx.synchronize()
    reps = max(1, args.steps)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = s.match(pl, max_distance=64, result=res)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, launches = ctx.last_kernel_ms(0)
    out = {"hamming": {
        "workload": f"{n_img} images x {n} x 256-bit descriptors, exhaustive pairs ({len(pl)}), max distance 64",
        "pairs_per_s": round(len(pl) / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
        "sign_kernel_ms": round(ms, 3), "sign_kernel_POPs": round(len(pl) * n * n * 512 / (ms * 1e-3) / 1e15, 3),
        "dtype": "i8 signs on MFMA, popcount re-score", "matches_per_pair": round(res.total() / len(pl), 1)}}
    res.close()
    s.close()
    return out


def bench_pipeline(ctx, args):
    """Stand-in for the reference's default run (main.cpp:49-58: 27 frames, consecutive pairs, threshold
    0.95, pixel gate 100^2, 8192 hypotheses, maxSquaredError 1, successProb 0.999) with everything
    resident on the device: match -> normalised correspondences -> RANSAC -> refit on the inliers -> scale chain and
    point average (its track ids come from the host, as in the reference's Scene)."""
    from sfm_amd import capi, synth, tracks
    frames, feats, H = 27, args.features, 8192
    desc, pts, counts, _ = synth.frame_sequence(frames, feats, seed=4242)
    K = synth.default_K()
    pl = np.stack([np.arange(frames - 1), np.arange(1, frames)], axis=1).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    res = corr = batch = None
    ids = n_tracks = None
    stage_ms = np.zeros(5)
    reps = max(1, args.steps)
    for it in range(reps + 1):
        t = [time.perf_counter()]
        res = s.match(pl, threshold=0.95, max_dist_sq=100.0 * 100.0, result=res)
        ctx.synchronize(); t.append(time.perf_counter())
        corr = capi.Correspondences(ctx, s, res, K, reuse=corr)
        ctx.synchronize(); t.append(time.perf_counter())
        batch = corr.ransac(K, H, 20261111, max_sq_err=1.0, success_prob=0.999, adaptive=True, batch=batch)
        ctx.synchronize(); t.append(time.perf_counter())
        lib_rc = capi.lib().sfm_ransac_batch_refit(batch._h)
        ctx.check(lib_rc)
        ctx.synchronize(); t.append(time.perf_counter())
        if ids is None:  # host bookkeeping of the reference's Scene, outside the timed stages
            ids, n_tracks = tracks.assign_track_ids(pl, res.offsets(), res.pairs(), counts)
            t[-1] = time.perf_counter()
        chain = batch.scale_chain(ids, n_tracks=n_tracks)
        t.append(time.perf_counter())
        if it > 0:  # first round warms up allocations
            stage_ms += np.diff(t) * 1e3
    stage_ms /= reps
    best = batch.best()
    total = float(stage_ms.sum())
    out = {
        "workload": f"{frames} synthetic frames x {feats} features, {frames - 1} consecutive pairs, {H} hypotheses per pair",
        "ms": {"match": round(stage_ms[0], 3), "correspondences": round(stage_ms[1], 3),
               "ransac": round(stage_ms[2], 3), "refit": round(stage_ms[3], 3), "chain": round(stage_ms[4], 3), "total": round(total, 3)},
        "pairs_per_s": round((frames - 1) / (total * 1e-3), 1),
        "matches_per_pair": round(res.total() / (frames - 1), 1),
        "mean_inliers": float(np.mean(best["count"])),
        "tracks": int(n_tracks), "points_averaged": int((chain["counts"] > 1).sum()),
    }
    s.close()
    return out


def bench_ransac(ctx, args, rank, world, dist, local_rank):
    """hypotheses scored/s on ransac_pairs pairs/GPU x ransac_corr correspondences (15 % inliers)
    x ransac_hyps hypotheses, inputs resident on the device."""
    import torch
    from sfm_amd import synth
    P, n, H = args.ransac_pairs, args.ransac_corr, args.ransac_hyps
    xs, ys, Ss = [], [], []
    for k in range(P):
        tv = synth.two_view(n, inlier_ratio=0.15, seed=[rank, k])
        xs.append(tv["x"])
        ys.append(tv["y"])
        Ss.append(synth.sample_table(n, H, seed=[7, rank, k]))
    dev = torch.device(f"cuda:{local_rank}")
    xd = torch.from_numpy(np.concatenate(xs)).to(dev)
    yd = torch.from_numpy(np.concatenate(ys)).to(dev)
    sd = torch.from_numpy(np.concatenate(Ss)).to(dev)
    torch.cuda.synchronize()
    off = np.arange(P + 1, dtype=np.int64) * n
    K = synth.default_K()
    batch = None

    def step():
        nonlocal batch
        batch = ctx.ransac_batch(off, None, None, K, None, H, max_sq_err=1.0, success_prob=0.999, adaptive=True,
                                 batch=batch, device_ptrs=(xd.data_ptr(), yd.data_ptr(), sd.data_ptr()))
        ctx.synchronize()

    step()
    if dist is not None:
        dist.barrier()
    steps = max(1, args.steps)
    kms = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        kms += ctx.last_kernel_ms(1)[0]
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    hyps = P * H * steps * world
    pairs_s = P * steps * world / el
    evals_s = P * H * n * steps / (kms * 1e-3) if kms > 0 else 0.0
    best = batch.best()
    return {
        "metric": "RANSAC hyps scored/sec", "value": round(hyps / el, 1), "unit": "hyps/s",
        "pairs_per_s": round(pairs_s, 2), "pairs_per_gpu": P, "correspondences": n, "hypotheses": H,
        "inlier_ratio": 0.15, "dtype": "f64",
        "hbm_GBps_algorithmic": round(pairs_s * RANSAC_BYTES_PER_PAIR / 1e9, 3),
        "score_kernel_ms_per_step": round(kms / steps, 3),
        "score_kernel_algorithmic_fp64_TFLOPs": round(evals_s * RANSAC_FLOPS_PER_EVAL / 1e12, 3),
        "fp64_vector_peak_TFLOPs": PEAK_F64_VEC_TFLOPS,
        "flops_note": "algorithmic = every (hypothesis, correspondence) evaluation at 92 fp64 flops; the kernel proves "
                      "most of them outliers with a packed-fp32 epipolar-distance test and evaluates only the rest, "
                      "so this figure can exceed the fp64 vector peak; results are bit-identical to evaluating all",
        "mean_best_inliers": float(np.mean(best["count"])),
    }


def cpu_baseline(args, desc, pts, ransac_out):
    """The CPU restatement of the reference path (oracle/, OpenMP over all host threads the
    environment grants) on a bounded sample of the same workload."""
    from oracle import oracle_api as oracle
    oracle.build()
    threads = oracle.num_threads()
    n = args.features
    t0 = time.perf_counter()
    done = 0
    for k in range(args.cpu_pairs):
        a, b = 0, k + 1
        oracle.match(desc[a * n:(a + 1) * n], pts[a * n:(a + 1) * n], desc[b * n:(b + 1) * n], pts[b * n:(b + 1) * n],
                     threshold=args.threshold, max_dist_sq=1e8)
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    el = time.perf_counter() - t0
    out = {"value": round(done / el, 4), "unit": "pairs/s", "cores": threads, "kind": "port",
           "sample": f"{done} of the benchmark's {n}x{n} image pairs, matrix-materialising matchTwoSided, "
                     f"gcc -O2 -fopenmp, {threads} threads"}
    if ransac_out is not None:
        from sfm_amd import synth
        nc, Hc, Pc = args.ransac_corr, args.ransac_hyps, 8
        t0 = time.perf_counter()
        done = 0
        for k in range(Pc):
            tv = synth.two_view(nc, inlier_ratio=0.15, seed=[0, k])
            S = synth.sample_table(nc, Hc, seed=[3, k])
            t1 = time.perf_counter()
            oracle.ransac(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
            done += 1
            if time.perf_counter() - t0 > 15.0:
                break
        el = time.perf_counter() - t0
        out["ransac"] = {"value": round(done * Hc / el, 2), "unit": "hyps/s", "cores": threads,
                         "sample": f"{done} pairs x {nc} correspondences x {Hc} hypotheses (input generation included in the time: <5 %)"}
    return out


if __name__ == "__main__":
    main()
