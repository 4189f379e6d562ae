#!/usr/bin/env python3
"""Benchmark of the hot path: exhaustive-pair descriptor matching at 8192 features/image
(the metric's configuration: 200 synthetic images x 8192 x 128-d float descriptors on one GPU),
plus the RANSAC stress shape as a secondary figure.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed.run launches N of them).  A step is one pass of the
matcher over this rank's pairs with the image table already resident in HBM; at N > 1 the pair
list grows with N (weak scaling: 19 900 pairs per GPU), every rank holds the whole table, and the
step ends with the gather of the match lists to rank 0 (the library's RCCL gather).  Rank 0 prints
one JSON line.

    --strong --images 1000                    a fixed list (1000 images, 499 500 pairs) divided over the N GPUs
    --strong --images 100 --features 32768    100 images of 32 768 features, 4950 pairs divided over the N GPUs

Besides the headline the line carries, at N = 1: the same matcher in its other regimes (`regimes`:
fp16 bound, the reference's default threshold 0.5, Gaussian float descriptors, a distribution whose
unrelated descriptors sit just under the threshold), the RANSAC stress shape as one batch of 10 000
pairs plus rows at inlier ratios where RANSAC finds its model, the default-run stand-in and the CPU
restatement timed on a bounded sample.
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
PEAK_F6_TFLOPS = 10000.0   # dense 6-bit / 4-bit MFMA peak: the 16x16x128 form with e2m3 operands takes those cycles at four times the K
# sim_variant, match_six_bit, dtype, peak: what the first bound pass of every pair computes in
BOUNDS = {"e2m3": (1, 1, "f6 (e2m3)", PEAK_F6_TFLOPS), "int8": (1, 0, "i8", PEAK_I8_TOPS), "fp16": (0, 0, "f16", PEAK_F16_TFLOPS)}
DTYPE_NOTES = {"e2m3": "e2m3-in/fp32-acc", "int8": "int8-in/int32-acc", "fp16": "fp16-in/fp32-acc"}
RANSAC_BYTES_PER_PAIR = 1_082_368  # 20 000 corr * 32 B + 4096 hyp * (96 + 12) B
# Vector-ALU issue peak: 256 CUs x 4 SIMDs, one wave64 instruction per 4 cycles each, at the 2.4 GHz behind the
# data sheet's vector figures (157.3 TFLOP/s fp32 = 1024 SIMDs x 16 lanes x 2 (packed) x 2 (fma) x 2.4 GHz).
PEAK_VALU_GINST = 1024 * 2.4 / 4
PEAK_FP64_TFLOPS = 78.6  # vector FP64, MI355X data sheet


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
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the pair list of --images is fixed and divided over the N GPUs")
    ap.add_argument("--gather", choices=["library", "torch"], default="library",
                    help="N > 1: gather the match lists with the library's RCCL calls or with torch.distributed")
    ap.add_argument("--ransac-pairs", type=int, default=10000, help="pairs of the RANSAC batch (the stress shape has 10 000)")
    ap.add_argument("--ransac-distinct", type=int, default=500,
                    help="correspondence sets generated on the host; the batch repeats them (separate copies in HBM, "
                         "every pair with a sample table of its own)")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="with one rank (launched through torch.distributed.run): run the N > 1 code path anyway, "
                         "process group, communicator and gather included")
    ap.add_argument("--no-regimes", action="store_true", help="skip the matcher's other regimes and the extra RANSAC rows")
    ap.add_argument("--ransac-corr", type=int, default=20000)
    ap.add_argument("--ransac-hyps", type=int, default=4096)
    ap.add_argument("--bound", choices=sorted(BOUNDS), default="e2m3",
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

    # Standard output carries the one JSON line and nothing else: libraries that write to file descriptor 1
    # (RCCL prints a version banner there when a communicator is created) go to standard error instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from sfm_amd import capi, shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.rehearse_gather  # the N > 1 code path
    if multi:
        import torch.distributed as dist
        if world == 1:  # --rehearse-gather outside a launcher: a group of one
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
                os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    ctx = capi.Context(local_rank)
    sim_variant, six_bit, bound_dtype, bound_peak = BOUNDS[args.bound]
    if "SFM_SIM_VARIANT" in os.environ:  # the library's timing-only ablations of the similarity kernel (wrong results)
        sim_variant = int(os.environ["SFM_SIM_VARIANT"])
    ctx.set_option("sim_variant", sim_variant)
    ctx.set_option("match_six_bit", six_bit)
    if "SFM_MATCH_OVERLAP" in os.environ:  # A/B switch of the matcher's two-stream chunk overlap
        ctx.set_option("match_overlap", int(os.environ["SFM_MATCH_OVERLAP"]))
    if "SFM_MATCH_LIST_OBSERVE" in os.environ:  # A/B switch: match lists sized as the call goes / for the worst case
        ctx.set_option("match_list_observe", int(os.environ["SFM_MATCH_LIST_OBSERVE"]))
    if "SFM_RANSAC_MFMA" in os.environ:  # A/B switch: the RANSAC pre-test on the matrix cores (1) or the vector ALU (0)
        ctx.set_option("ransac_mfma", int(os.environ["SFM_RANSAC_MFMA"]))
    if "SFM_RANSAC_MFMA_OPS_GIB" in os.environ:  # A/B switch: 0 = every block makes the matrix-core operands of its correspondences
        ctx.set_option("ransac_mfma_ops_gib", int(os.environ["SFM_RANSAC_MFMA_OPS_GIB"]))
    if "SFM_MATCH_COL_SPLIT" in os.environ:  # A/B switch: column parts per row block of small matcher launches
        ctx.set_option("match_col_split", int(os.environ["SFM_MATCH_COL_SPLIT"]))
    if args.first_slice > 0:
        ctx.set_option("hyps_per_launch", args.first_slice)
    if args.pairs_per_launch > 0:
        ctx.set_option("pairs_per_launch", args.pairs_per_launch)

    # ---- workload: weak scaling (pairs per GPU fixed) unless --strong (pair list fixed) ------------
    if args.strong:
        total_pairs = args.images * (args.images - 1) // 2
        pairs_per_gpu = (total_pairs + world - 1) // world
        n_images = args.images
    else:
        pairs_per_gpu = args.images * (args.images - 1) // 2
        total_pairs = pairs_per_gpu * world
        n_images = args.images if world == 1 else shard.images_for_pairs(total_pairs)
    t0 = time.time()
    desc, pts, counts = synth.image_set(n_images, args.features, seed=2026)
    pair_list = synth.exhaustive_pairs(n_images)[:total_pairs]
    owned = shard.partition_pairs(total_pairs, rank, world)
    my_pairs = np.ascontiguousarray(pair_list[owned])
    if args.strong:
        pairs_per_gpu = len(my_pairs)
    imgset = ctx.imageset(counts, desc, pts)
    gen_s = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    # Two result objects: at N > 1 the gather of step k - 1 (RCCL, on the communicator's own stream) runs
    # while the library's stream matches step k.  Every gather is finished before the closing barrier.
    results = [None, None]
    gathers = [None, None]
    gathered = None
    res = None
    gather_mode = args.gather if multi else "none"
    comm = None
    if multi and gather_mode == "library":
        try:
            comm = shard.library_comm(ctx, rank, world)
        except capi.SfmError as e:  # an error, not a hang: the torch.distributed stand-by takes over on every rank
            print(f"[rank {rank}] library communicator failed ({e}); gathering with torch.distributed", file=sys.stderr)
            comm = None
        ok = torch.tensor([1 if comm is not None else 0], device=f"cuda:{local_rank}")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
            gather_mode = "torch (library communicator failed)"

    def launch(k):
        slot = (k & 1) if multi else 0  # two result objects only where a gather still reads the one of the step before
        if gathers[slot] is not None:
            gathers[slot].wait()  # the gather that read this result object two steps ago
        results[slot] = imgset.match(my_pairs, threshold=args.threshold, max_dist_sq=1e8, result=results[slot],
                                     ratio=args.ratio)
        return results[slot]

    def gather(r, k):
        nonlocal gathered
        if comm is not None:
            gathers[k & 1] = gathered = shard.library_gather(comm, r, total_pairs, gathered=gathers[k & 1])
            return
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
        # Steps follow each other without a wait on the host: the library queues step k + 1 behind step k (and the
        # gather of step k - 1 beside it), as a caller that streams batches would.  Only the torch stand-by for the
        # gather needs the host between steps.
        for k in range(n_steps):
            res = launch(k)
            if multi and prev is not None:
                gather(prev, k - 1)
            if multi and comm is None:
                ctx.synchronize()
                torch.cuda.synchronize()  # the gather reads prev's buffers, which step k + 1 overwrites
                res.total()               # cached: gather() must not wait on the library's stream later
            prev = res
        ctx.synchronize()
        ms, nl = ctx.last_kernel_ms(0)  # summed over the steps ("span_accumulate")
        if multi and prev is not None:
            gather(prev, n_steps - 1)
            for g in gathers:
                if g is not None:
                    g.wait()
            torch.cuda.synchronize()
        return ms, nl

    ctx.set_option("span_accumulate", 1)
    if multi:  # both result objects have seen a call before the timed region (the first call on one sizes its lists)
        launch(1)
        ctx.synchronize()
        ctx.last_kernel_ms(0)
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, kernel_launches = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_option("span_accumulate", 0)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_matches = res.total()
    cand, ovf = res.stats()
    res_bytes = res.device_bytes()
    gathered_total = None
    if rank == 0 and multi and gathered is not None:
        gathered_total = int(gathered.total()) if comm is not None else int(gathered[2].shape[0])
    gather_check = None
    if multi:
        gather_check = check_gather(torch, dist, ctx, comm, gathered, imgset, pair_list, total_pairs, total_matches, rank,
                                    local_rank, args)
    # the timed region leaves every step's lists on the device (a device-resident pipeline reads them there);
    # what fetching one step's lists to the host costs is measured once, outside the timed region
    t0 = time.perf_counter()
    fetched = res.pairs()
    fetch_ms = (time.perf_counter() - t0) * 1e3
    fetch_bytes = int(fetched.nbytes)
    del fetched
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
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": bound_dtype,
        "dtype_note": DTYPE_NOTES[args.bound] +
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
            "gather": gather_mode, "matches_gathered_on_rank0": gathered_total, "gather_check": gather_check,
            "lists_left_on_device": True,
            "fetching_one_steps_lists_rank0": {"ms": round(fetch_ms, 2), "bytes": fetch_bytes,
                                               "note": "sfm_match_result_pairs into pageable host memory, once, after the timed region"},
            "setup_s": round(gen_s, 2),
            "device_bytes_rank0": {"image_set": imgset.device_bytes(), "matcher_work_buffers_and_lists": res_bytes},
        },
        "roofline": {
            "bound": "mfma", "kernel": "match_sim_kernel",
            "achieved": round(achieved_tflops, 2), "peak": bound_peak, "unit": "TFLOP/s",
            "unit_note": "algorithmic 2*N*M*D operations per pair, priced against the dense matrix-core peak of the bound's operand type "
                         "(e2m3 10 000, int8 5 000, fp16 2 500 TFLOP/s)",
            "frac": round(achieved_tflops / bound_peak, 4),
            "traffic": measured_traffic(len(my_pairs) * args.steps / max(kernel_launches, 1), args.bound),
            "traffic_unit": "HBM bytes per launch",
            "traffic_source": "static, not measured in this run: bytes per pair from the committed rocprofv3 PMC passes "
                              "(profiles/r3_pmc_traffic_f6.json; r3_pmc_traffic_i8.json and r1_pmc_traffic.json for the int8 and fp16 bounds: separate FETCH_SIZE / WRITE_SIZE "
                              "runs of this script, scripts/pmc_traffic.sh) "
                              "times this run's pairs per launch",
            "avg_launch_ms": round(kernel_ms / max(kernel_launches, 1), 4), "launches": kernel_launches,
            "flops_per_pair": flops_per_pair,
        },
    }

    # ---- secondary: RANSAC stress shape ------------------------------------------------------
    if not args.no_ransac:
        out["ransac"] = bench_ransac(ctx, args, rank, world, dist, local_rank, comm if multi else None)

    # ---- secondary: the stages chained on the device (default-run stand-in) --------------------
    if not args.no_pipeline and world == 1:
        out["pipeline"] = bench_pipeline(ctx, args)
    if not args.no_extensions and world == 1:
        out["extensions"] = bench_extensions(ctx, args)
    if not args.no_regimes and world == 1:
        out["regimes"] = bench_regimes(ctx, args, imgset, pair_list, flops_per_pair)
        ctx.set_option("sim_variant", sim_variant)

    # ---- CPU baseline (rank 0, N = 1) ----------------------------------------------------------
    pipe_inputs = out.get("pipeline", {}).pop("_cpu_inputs", None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, desc, pts, out.get("ransac"), pipe_inputs)

    # the figures a reader of the line's tail needs, once more at its very end
    out["summary"] = {"pairs_per_s": out["value"], "n_gpus": world, "match_sim_kernel_frac_of_mfma_peak": out["roofline"]["frac"],
                      "ransac_hyps_per_s": out.get("ransac", {}).get("value"),
                      "cpu_pairs_per_s": out.get("cpu_baseline", {}).get("value"),
                      "cpu_ransac_hyps_per_s": out.get("cpu_baseline", {}).get("ransac", {}).get("value")}
    for r in out.get("ransac", {}).get("where_a_model_is_found", []):  # batches of 256 pairs where a model exists, pre-test on
        if r.get("pre_test"):
            tag = f"ransac_at_{int(round(100 * r['inlier_ratio']))}pct_inliers"
            out["summary"][tag] = {"pairs_per_s_scan_stopped_by_the_cap": r["adaptive_stop"]["pairs_per_s"],
                                   "hyps_per_s_every_row_scored": r["all_rows_scored"]["hyps_per_s"]}
    for r in out.get("regimes", []):
        if r.get("threshold") == 0.5 and "pairs_per_s" in r and "pairs_per_s_at_the_reference_default_threshold" not in out["summary"]:
            out["summary"]["pairs_per_s_at_the_reference_default_threshold"] = r["pairs_per_s"]
    if rank == 0:
        sys.stdout.flush()
        line = (json.dumps(out) + "\n").encode()
        while line:
            line = line[os.write(json_fd, line):]
    if comm is not None:
        for g in gathers:
            if g is not None:
                g.close()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def check_gather(torch, dist, ctx, comm, gathered, imgset, pair_list, total_pairs, local_total, rank, local_rank, args):
    """N > 1 (or --rehearse-gather): the gathered lists against what the ranks hold.  The total on the root must
    be the sum of the ranks' totals, and eight pairs sampled over the whole list, matched again on rank 0 alone,
    must equal their gathered lists.  A mismatch ends the run with an error; a failure of the check itself is
    reported on the line."""
    dev = torch.device(f"cuda:{local_rank}")
    t = torch.tensor([int(local_total)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    want = int(t.item())
    if rank != 0 or gathered is None:
        return None
    try:
        if comm is not None:
            got_total = int(gathered.total())
            starts, cnt = gathered.layout()
            allp = gathered.pairs()
        else:
            starts, cnt = np.asarray(gathered[0]), np.asarray(gathered[1])  # shard.gather_csr(ordered=False)
            allp = gathered[2].cpu().numpy()
            got_total = int(allp.shape[0])
        sample = np.unique(np.linspace(0, total_pairs - 1, 8).astype(np.int64))
        ref = imgset.match(np.ascontiguousarray(pair_list[sample]), threshold=args.threshold, max_dist_sq=1e8, ratio=args.ratio)
        roff, rp = ref.offsets(), ref.pairs()
        ref.close()
        same = all(np.array_equal(allp[starts[i]:starts[i] + cnt[i]], rp[roff[k]:roff[k + 1]]) for k, i in enumerate(sample))
    except Exception as e:  # the check, not the gather, failed
        return {"error": f"{type(e).__name__}: {e}"}
    if got_total != want or not same:
        raise SystemExit(f"gather check failed: {got_total} matches on rank 0, the ranks hold {want}; sampled lists equal: {same}")
    return {"sum_of_rank_totals": want, "sampled_pairs_equal_to_single_gpu_lists": int(len(sample))}


def measured_traffic(pairs_per_launch, bound="e2m3"):
    """HBM bytes per launch of the similarity kernel from the committed PMC pass (FETCH_SIZE and
    WRITE_SIZE are collected by rocprofv3 in separate runs of this script, not inline), scaled to
    this run's pairs per launch; None if the profile is absent."""
    path = os.path.join(ROOT, "profiles", {"e2m3": "r3_pmc_traffic_f6.json", "int8": "r3_pmc_traffic_i8.json"}.get(bound, "r1_pmc_traffic.json"))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return round(t["hbm_bytes_per_pair"] * pairs_per_launch)


def bench_regimes(ctx, args, imgset, pair_list, flops_per_pair):
    """The matcher outside the headline's best case.  The headline's descriptors let the int8 bound admit
    nothing but true matches (nor does the e2m3 bound that now comes first); these rows show what the same entry point costs when it does not:
    the fp16 bound on the same data, the reference's default threshold (match.hpp:59: 0.5, where most cells
    pass and the row-best / column-best filter decides), Gaussian float descriptors, and a distribution whose
    unrelated descriptors sit at cosine 0.93, inside the e2m3 and int8 margins of a 0.95 threshold (every pair is redone
    with the fp16 bound).  `frac` prices the strict bound kernel alone against its MFMA peak; `end_to_end_frac`
    prices the whole call (all passes) against the peak of the bound the row starts with."""
    from sfm_amd import capi, synth
    n = args.features

    def row(label, s, pl, thr, variant, note, metric=0):
        sv, six, dtype, peak = BOUNDS[variant]
        ctx.set_option("sim_variant", sv)
        ctx.set_option("match_six_bit", six)
        res = s.match(pl, threshold=thr, max_dist_sq=1e8, metric=metric)
        ctx.synchronize()
        t0 = time.perf_counter()
        res = s.match(pl, threshold=thr, max_dist_sq=1e8, result=res, metric=metric)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.set_option("match_six_bit", 1)  # (image sets made between the rows get the 6-bit table)
        ms, launches = ctx.last_kernel_ms(0)
        cand, cells, filtered, redone = res.stats_ex()
        out = {"workload": label, "threshold": thr, "bound": variant, "pairs": len(pl), "pairs_per_s": round(len(pl) / dt, 1),
               "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": launches,
               # (a strict pass whose lists flood stops early and hands the pair on: its time prices nothing then)
               "frac": round(flops_per_pair * len(pl) / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 and not (filtered or redone or cells) else None,
               "end_to_end_frac": round(flops_per_pair * len(pl) / dt / 1e12 / peak, 4),
               "matches_per_pair": round(res.total() / len(pl), 1), "candidates_per_pair": round(cand / len(pl), 1),
               "pairs_scored_cell_by_cell": cells, "pairs_through_filtered_passes": filtered,
               "pairs_redone_with_fp16_bound": redone, "note": note}
        res.close()
        return out

    def row_metric(label, s, pl, thr, variant, metric, note):
        out = row(label, s, pl, thr, variant, note, metric=metric)
        out["metric"] = "dot" if metric == capi.METRIC_DOT else "cosine"
        return out

    rows = []
    head = np.ascontiguousarray(pair_list[:min(len(pair_list), 4096)])
    rows.append(row("the headline's images", imgset, head, args.threshold, "int8",
                    "int8 bound as the first pass instead of e2m3: half the MFMA rate, a margin of 2-4e-2 instead of 0.1"))
    rows.append(row("the headline's images", imgset, head, args.threshold, "fp16",
                    "fp16 bound as the first pass: a quarter of the e2m3 rate, a margin of 1.5e-3"))
    loose = np.ascontiguousarray(pair_list[:min(len(pair_list), 1024)])
    rows.append(row("the headline's images", imgset, loose, 0.5, "e2m3",
                    "the reference's default threshold: nearly every cell passes, the strict lists overflow, one sweep finds the maxima of rows and columns and one collects the cells within reach of them"))
    n_img = 40
    pl = synth.exhaustive_pairs(n_img)
    d, p, c = synth.image_set(n_img, n, seed=2027, kind="gauss")
    s = ctx.imageset(c, d, p)
    rows.append(row(f"{n_img} images x {n} Gaussian float descriptors (sigma 1; shared features re-detected with noise 0.05)",
                    s, pl, args.threshold, "e2m3", "general floats: exact pass reads the fp32 table"))
    rows.append(row(f"{n_img} images x {n} Gaussian float descriptors", s, pl, 0.5, "e2m3",
                    "unrelated Gaussian descriptors have cosine ~ N(0, 1/128): 0.5 is still selective"))
    s.close()
    # the reference's other metric on descriptors of unit norm, where it equals the cosine up to rounding
    d, p, c = synth.image_set(24, n, seed=2026)
    unit = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    s = ctx.imageset(c, unit, p)
    pl24 = synth.exhaustive_pairs(24)
    rows.append(row_metric("24 of the headline's images, rows scaled to unit norm, dotProduct (match.hpp:18-27)", s, pl24, args.threshold,
                           "e2m3", capi.METRIC_DOT, "a.b > T implies a cosine above T / (maxA maxB): strict passes under that bound, "
                                                    "the exact pass scores the dot product"))
    s.close()
    rng = np.random.default_rng(7)
    base = synth.sift_like(rng, 1)[0]
    hard = np.clip(base[None, :] * 0.9 + synth.sift_like(rng, n_img * n) * 0.42, 0, 255).astype(np.float32)
    shared = int(0.3 * n)
    for im in range(1, n_img):  # 30 % of every image re-detects image 0's features
        hard[im * n:im * n + shared] = hard[:shared]
    pts = rng.uniform(0, 4000, (n_img * n, 2))
    s = ctx.imageset(np.full(n_img, n, np.int32), hard, pts)
    rows.append(row(f"{n_img} images x {n} SIFT-like descriptors with a common part: unrelated cosines ~0.93", s,
                    pl, args.threshold, "e2m3",
                    "the e2m3 and int8 margins admit the unrelated cells, lists flood, the fp16 bound redoes every pair"))
    s.close()
    return rows


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

    # L2 mode: SIFT-like descriptors whose norms differ by up to 10 % (so that the L2 order is not the cosine order),
    # mutual nearest neighbours under |a|^2 + |b|^2 - 2 a.b below 250^2, with and without the second-best test
    from sfm_amd import synth
    n_img = 40
    d, p, _ = synth.image_set(n_img, n, seed=77)
    d = np.floor(d * rng.uniform(0.9, 1.0, size=(len(d), 1)).astype(np.float32)).astype(np.float32)
    s = ctx.imageset(np.full(n_img, n, np.int32), d, p)
    pl = synth.exhaustive_pairs(n_img).astype(np.int32)
    out["l2"] = {"workload": f"{n_img} images x {n} x 128-d SIFT-like descriptors with norms spread over 10 %, exhaustive pairs "
                             f"({len(pl)}), squared distance below 250^2", "rows": []}
    for ratio in (0.0, 0.8):
        res = s.match(pl, threshold=250.0 ** 2, metric=capi.METRIC_L2, ratio=ratio)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = s.match(pl, threshold=250.0 ** 2, metric=capi.METRIC_L2, ratio=ratio, result=res)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ms, launches = ctx.last_kernel_ms(0)
        st = res.stats_ex()
        out["l2"]["rows"].append({
            "second_best_ratio": ratio, "pairs_per_s": round(len(pl) / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
            "sim_kernel_ms": round(ms, 3), "matches_per_pair": round(res.total() / len(pl), 1),
            "candidates_per_pair": round(st[0] / len(pl), 1), "pairs_scored_cell_by_cell": st[1],
            "pairs_through_filtered_passes": st[2], "pairs_redone_with_fp16_bound": st[3]})
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
    rounds = []
    reps, warm = max(20, args.steps), 5  # a round is ~1.5 ms: the first rounds of a process run slow (clocks, runtime pools)
    for it in range(reps + warm):
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
        if it >= warm:
            rounds.append(np.diff(t) * 1e3)
    stage_ms = np.median(np.array(rounds), axis=0)  # per stage, over the timed rounds
    best = batch.best()
    total = float(stage_ms.sum())
    # The reference's own call pattern (scene.cpp:60, :261): one pair per call, host buffers in, lists out.
    starts = np.concatenate([[0], np.cumsum(counts)])
    f0, f1 = slice(starts[0], starts[1]), slice(starts[1], starts[2])
    cx, cy = corr.fetch()
    coff = corr.offsets()
    x0, y0 = cx[coff[0]:coff[1]], cy[coff[0]:coff[1]]
    def per_call_ms(fn, calls=30, rounds=3):
        """Steady state: the first dozen calls of a process run at half speed (clocks, runtime pools), and a box that
        has just run a long kernel may take a round to settle: ten calls untimed, then the best of three rounds."""
        for _ in range(10):
            fn()
        best_ms, last = float("inf"), None
        for _ in range(rounds):
            t0 = time.perf_counter()
            for _ in range(calls):
                last = fn()
            best_ms = min(best_ms, (time.perf_counter() - t0) / calls * 1e3)
        return best_ms, last

    dA, pA, dB, pB = desc[f0], pts[f0], desc[f1], pts[f1]
    match_call_ms, _ = per_call_ms(lambda: ctx.match_pair(dA, pA, dB, pB, threshold=0.95, max_dist_sq=100.0 * 100.0))
    # the same tables in page-locked memory (sfm_host_alloc), where an adaptor that packs Keypoints can keep them
    qA, rA, qB, rB = (ctx.pinned_like(a) for a in (dA, pA, dB, pB))
    match_pinned_ms, _ = per_call_ms(lambda: ctx.match_pair(qA, rA, qB, rB, threshold=0.95, max_dist_sq=100.0 * 100.0))
    # a video's order (scene.cpp:60: image A of a call is image B of the call before) with the caller's promise that
    # a table seen again at the same address has not changed: that image comes from the device
    ctx.set_option("match_pair_reuse", 1)
    turn = [0]
    def shared_frame_call():
        turn[0] ^= 1
        return (ctx.match_pair(qA, rA, qB, rB, threshold=0.95, max_dist_sq=100.0 * 100.0) if turn[0] else
                ctx.match_pair(qB, rB, qA, rA, threshold=0.95, max_dist_sq=100.0 * 100.0))
    match_shared_ms, _ = per_call_ms(shared_frame_call)
    ctx.set_option("match_pair_reuse", 0)
    # the same call with MATCHING_OPTIONS' defaults (match.hpp:59-60: threshold 0.5, gate 10000^2): most cells clear
    # the threshold, the strict lists flood, and the two-phase sweeps decide
    match_default_ms, _ = per_call_ms(lambda: ctx.match_pair(qA, rA, qB, rB, threshold=0.5, max_dist_sq=1e8))
    # and as bytes (sfm_match_pair_u8): the synthetic SIFT-like tables hold integers in [0, 255], as OpenCV's do
    match_u8_ms = None
    if dA.size and dB.size and np.array_equal(dA, dA.astype(np.uint8)) and np.array_equal(dB, dB.astype(np.uint8)):
        uA, uB = ctx.pinned_like(dA.astype(np.uint8)), ctx.pinned_like(dB.astype(np.uint8))
        match_u8_ms, lst8 = per_call_ms(lambda: ctx.match_pair(uA, rA, uB, rB, threshold=0.95, max_dist_sq=100.0 * 100.0))
        lst32 = ctx.match_pair(qA, rA, qB, rB, threshold=0.95, max_dist_sq=100.0 * 100.0)
        assert np.array_equal(lst8, lst32), "byte tables and float tables must give one list"
        del uA, uB
    del qA, rA, qB, rB
    ctx.set_option("ransac_seed", 20261111)
    ransac_call_ms, rr = per_call_ms(lambda: ctx.ransac_pair(x0, y0, K, H, max_sq_err=1.0, success_prob=0.999, adaptive=True,
                                                           per_row=False))
    ctx.set_option("ransac_seed", 0)
    out = {
        "workload": f"{frames} synthetic frames x {feats} features, {frames - 1} consecutive pairs, {H} hypotheses per pair",
        "timing": f"median per stage over {reps} rounds after {warm} untimed ones",
        "ms": {"match": round(stage_ms[0], 3), "correspondences": round(stage_ms[1], 3),
               "ransac": round(stage_ms[2], 3), "refit": round(stage_ms[3], 3), "chain": round(stage_ms[4], 3), "total": round(total, 3)},
        "pairs_per_s": round((frames - 1) / (total * 1e-3), 1),
        "matches_per_pair": round(res.total() / (frames - 1), 1),
        "mean_inliers": float(np.mean(best["count"])),
        "tracks": int(n_tracks), "points_averaged": int((chain["counts"] > 1).sum()),
        "one_pair_per_call": {
            "what": "the reference's call pattern: sfm_match_pair / sfm_ransac_pair once per frame pair, host buffers in "
                    "(pageable unless said otherwise), lists out; upload, ingest and download inside the time; mean of 30 "
                    "calls, best of three rounds after ten untimed calls",
            "match_pair_ms": round(match_call_ms, 3), "match_pair_ms_tables_in_page_locked_memory": round(match_pinned_ms, 3),
            "match_pair_ms_byte_tables_in_page_locked_memory": None if match_u8_ms is None else round(match_u8_ms, 3),
            "match_pair_ms_reference_default_options_page_locked": round(match_default_ms, 3),
            "match_pair_ms_shared_frame_kept_on_device_page_locked": round(match_shared_ms, 3),
            "features": int(counts[0]),
            "ransac_pair_ms": round(ransac_call_ms, 3), "correspondences": int(len(x0)), "hypotheses": H,
            "ransac_iterations": int(rr["iterations"]), "ransac_inliers": int(len(rr["inliers"])),
        },
    }
    out["_cpu_inputs"] = (desc, pts, counts, pl, cx, cy, coff, K, H)
    s.close()
    return out


def device_sample_tables(torch, dev, pairs, hyps, n_corr, seed):
    """[pairs * hyps, 8] int32 on the device: rows of 8 distinct indices of range(n_corr), uniform (rows that
    repeat an index are redrawn), every pair with rows of its own."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    t = torch.randint(0, n_corr, (pairs * hyps, 8), generator=gen, device=dev, dtype=torch.int32)
    while True:
        srt = torch.sort(t, dim=1).values
        bad = (srt[:, 1:] == srt[:, :-1]).any(dim=1)
        nbad = int(bad.sum().item())
        if nbad == 0:
            return t.contiguous()
        t[bad] = torch.randint(0, n_corr, (nbad, 8), generator=gen, device=dev, dtype=torch.int32)


def ransac_inputs(torch, dev, pairs, distinct, n_corr, hyps, ratio, seed):
    """Correspondences of `pairs` image pairs on the device: `distinct` sets generated on the host (synth.two_view)
    and repeated (separate copies in HBM), and a sample table per pair drawn on the device."""
    from sfm_amd import synth
    distinct = max(1, min(distinct, pairs))
    xs, ys = [], []
    for k in range(distinct):
        tv = synth.two_view(n_corr, inlier_ratio=ratio, seed=[seed, k])
        xs.append(tv["x"])
        ys.append(tv["y"])
    reps = (pairs + distinct - 1) // distinct
    xd = torch.from_numpy(np.stack(xs)).to(dev).repeat(reps, 1, 1)[:pairs].reshape(-1, 2).contiguous()
    yd = torch.from_numpy(np.stack(ys)).to(dev).repeat(reps, 1, 1)[:pairs].reshape(-1, 2).contiguous()
    sd = device_sample_tables(torch, dev, pairs, hyps, n_corr, 1000 + seed)
    torch.cuda.synchronize()
    return xd, yd, sd


def ransac_valu_profile(mfma):
    """Vector-ALU instructions the scoring kernel executes per wave and group of correspondences on the stress
    shape, from the committed counter pass of the kernel that runs (static; None if the file is absent).
    Returns (instructions, correspondences per group, extra fields, source)."""
    if mfma:
        name = next((f for f in ("r3_pmc_sq_ransac_mfma.json", "r2e_pmc_sq_ransac_mfma.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
        if name:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            for k, v in t.items():
                if isinstance(v, dict) and "ransac_score_mfma_kernel" in k:
                    extra = {"kernel": k.replace("void ", "").replace("(RansacArgs)", ""), "tree_of_the_counter_pass": v.get("tree", "an earlier tree"),
                             "vector_instructions_per_wave_per_16_correspondences": v["vector_instructions_per_wave_step"],
                             "of_which_mfma": v["mfma_instructions_per_wave_step"],
                             "matrix_pipe_busy_fraction_of_simd_cycles_under_counters": round(v["mfma_busy_fraction_of_simd_cycles"], 4),
                             "valu_busy_fraction_of_simd_cycles_under_counters": round(v["valu_busy_fraction_of_simd_cycles"], 4),
                             "clock_GHz_under_counters": round(v["clock_GHz_under_load"], 3)}
                    return v["vector_instructions_per_wave_step"], 16, extra, "profiles/" + name
        return None, None, None, None
    for name in ("r2_pmc_sq_ransac.json", "r1_final4_pmc_sq_ransac.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                t = json.load(f)
            for k, v in t.items():
                if isinstance(v, dict) and "ransac_score_kernel<0, false, true>" in k:
                    per4 = v.get("valu_instructions_per_wave_per_four_correspondences")
                    return per4, 4, {"kernel": "ransac_score_kernel", "valu_instructions_per_wave_per_four_correspondences": per4}, "profiles/" + name
    return None, None, None, None


def exact_path_profile():
    """Vector instructions per wave and (hypothesis, correspondence) cell of the scoring kernel that evaluates
    every cell (pre-test off), from the committed counter pass (static; None if the file is absent)."""
    path = os.path.join(ROOT, "profiles", "r3_pmc_sq_ransac_exact_a.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    for k, v in t.items():
        if isinstance(v, dict) and "ransac_score_kernel<0, false, false>" in k:
            cells = v["SQ_WAVES"] * 20000.0  # the pass ran 128 pairs x 4096 rows x 20 000 correspondences
            return {"valu_instructions_per_wave_and_cell": round(v["SQ_INSTS_VALU"] / cells, 1),
                    "valu_busy_fraction_of_simd_cycles_under_counters": round(4.0 * v["SQ_INSTS_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
                    "clock_GHz_under_counters": round(v["clock_GHz_under_load"], 3),
                    "source": "static, not measured in this run: SQ_INSTS_VALU of scripts/pmc_ransac_regimes.sh 0.8 0 "
                              "(profiles/r3_pmc_sq_ransac_exact_a.json) scaled to this run's cells"}
    return None


def bench_ransac(ctx, args, rank, world, dist, local_rank, comm=None):
    """hypotheses scored/s on one batch of ransac_pairs pairs/GPU x ransac_corr correspondences (15 % inliers)
    x ransac_hyps hypotheses, inputs resident on the device; at N = 1 also rows at inlier ratios where RANSAC
    finds its model, with the pre-test on and off."""
    import torch
    from sfm_amd import synth
    P, n, H = args.ransac_pairs, args.ransac_corr, args.ransac_hyps
    dev = torch.device(f"cuda:{local_rank}")
    K = synth.default_K()

    from sfm_amd import capi
    gathered_info = {}

    def measure(pairs, distinct, ratio, adaptive, filt, steps, seed, loss=0, gather_total=0):
        xd, yd, sd = ransac_inputs(torch, dev, pairs, distinct, n, H, ratio, seed)
        off = np.arange(pairs + 1, dtype=np.int64) * n
        ctx.set_option("ransac_filter", filt)
        batch = None
        g = None

        def step():
            nonlocal batch, g
            batch = ctx.ransac_batch(off, None, None, K, None, H, max_sq_err=1.0, success_prob=0.999, adaptive=adaptive,
                                     batch=batch, device_ptrs=(xd.data_ptr(), yd.data_ptr(), sd.data_ptr()), loss=loss)
            if gather_total:  # winners and inlier lists of every rank to rank 0 (RCCL), inside the step
                g = comm.gather_ransac(batch, gather_total, gathered=g)
                g.wait()
            ctx.synchronize()

        step()
        if dist is not None:
            dist.barrier()
        kms, launches = 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            a, b = ctx.last_kernel_ms(1)
            kms, launches = kms + a, launches + b
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        best = batch.best()
        best["rows_with_many_inliers"] = dict(zip(("candidates", "passed_over", "handed_over"), batch.handed_over()))
        if g is not None:
            if rank == 0:  # what arrived: every pair of the whole list, and this rank's own share unchanged
                r = g.ransac()
                mine = capi.shard_indices(gather_total, rank, world)
                ok = (len(r["best_h"]) == gather_total and np.array_equal(r["best_h"][mine], best["best_h"])
                      and np.array_equal(r["count"][mine], best["count"]) and int(r["inlier_offsets"][-1]) == int(g.total()))
                if not ok:
                    raise SystemExit("RANSAC gather check failed: rank 0's own results differ from their gathered copy")
                gathered_info.update({"pairs_gathered_on_rank0": int(gather_total), "inliers_gathered_on_rank0": int(g.total()),
                                      "rank0_share_equal_to_its_batch": True})
            g.close()
        batch.close()
        ctx.set_option("ransac_filter", 1)
        del xd, yd, sd
        return el, kms, launches, best

    steps = max(1, args.steps)
    n_total = P * world
    if comm is not None:  # the rank's share of a list of P pairs per GPU, dealt in blocks of 64 like the matcher's
        P = int(capi.shard_count(n_total, rank, world))
    el, kms, launches, best = measure(P, args.ransac_distinct, 0.15, True, 1, steps, rank, gather_total=n_total if comm is not None else 0)
    hyps = n_total * H * steps
    pairs_s = n_total * steps / el
    mfma_on = int(os.environ.get("SFM_RANSAC_MFMA", "1")) != 0  # the library's default; the stress shape meets its conditions
    per_grp, grp, extra, src = ransac_valu_profile(mfma_on)
    valu = None
    if per_grp and kms > 0:
        insts = (P * H / 64.0) * (n / float(grp)) * per_grp * steps  # wave-instructions of this rank's timed launches
        rate = insts / (kms * 1e-3) / 1e9
        valu = {"bound": "valu issue", "achieved": round(rate, 1), "peak": PEAK_VALU_GINST, "unit": "G wave-instructions/s",
                "frac": round(rate / PEAK_VALU_GINST, 4)}
        valu.update(extra)
        valu["source"] = (f"static, not measured in this run: SQ_INSTS_VALU of the scoring kernel from {src} "
                          "(rocprofv3 --pmc pass of this script on the stress shape) scaled to this run's waves; peak = 1024 SIMDs x "
                          "2.4 GHz / 4 cycles per wave64 instruction (matrix instructions are issued through the same port and "
                          "are counted among the vector instructions)")
    out = {
        "metric": "RANSAC hyps scored/sec", "value": round(hyps / el, 1), "unit": "hyps/s",
        "what_it_measures": "15 % inliers, 4096 eight-point samples: no all-inlier sample exists, so every hypothesis is a "
                            "wrong model and the rate is that of rejecting them (each still scored against all 20 000 "
                            "correspondences: the pre-test proves > 99.9 % of the cells outliers, the rest are evaluated exactly)",
        "pairs_per_s": round(pairs_s, 2), "pairs_per_gpu": P, "pairs_per_batch": P, "distinct_correspondence_sets": min(args.ransac_distinct, P),
        "correspondences": n, "hypotheses": H, "inlier_ratio": 0.15, "dtype": "f64",
        "hbm_GBps_algorithmic": round(pairs_s * RANSAC_BYTES_PER_PAIR / 1e9, 3),
        "hbm_note": "compulsory bytes per pair (1.08 MB) x pairs/s: < 1 % of HBM bandwidth, the stage is bound by vector-ALU issue",
        "score_kernel_ms_per_step": round(kms / steps, 3), "score_kernel_launches_per_step": launches // steps,
        "roofline": valu,
        "mean_best_inliers": float(np.mean(best["count"])),
        "gather": ("library: winners' records and inlier lists of every rank to rank 0 inside the step (sfm_ransac_gather)"
                   if comm is not None else "none"),
    }
    out.update(gathered_info)
    if world == 1 and not args.no_regimes:
        rows = []
        exact = exact_path_profile()
        for ratio in (0.5, 0.8):
            for filt in (1, 0):
                el, kms, launches, best = measure(256, 64, ratio, False, filt, 1, 50 + int(ratio * 10))
                el2, kms2, _, best2 = measure(256, 64, ratio, True, filt, 1, 50 + int(ratio * 10))
                rows.append({"inlier_ratio": ratio, "pre_test": bool(filt), "pairs": 256,
                             "all_rows_scored": {"hyps_per_s": round(256 * H / el, 1), "score_kernel_ms": round(kms, 3),
                                                 "rows_with_many_inliers": best["rows_with_many_inliers"],
                                                 "mean_best_inliers": float(np.mean(best["count"]))},
                             "adaptive_stop": {"pairs_per_s": round(256 / el2, 1), "score_kernel_ms": round(kms2, 3),
                                               "rows_with_many_inliers": best2["rows_with_many_inliers"],
                                               "mean_iterations": float(np.mean(best2["iterations"])),
                                               "mean_best_inliers": float(np.mean(best2["count"]))}})
                if not filt and kms > 0:  # every cell evaluated exactly: the kernel's own roofline
                    evals = 256.0 * H * n / (kms * 1e-3)
                    rows[-1]["roofline"] = {
                        "kernel": "ransac_score_kernel<0, false, false>", "evaluations_per_s": round(evals, 0),
                        "fp64_flops_frac": round(evals * 92.0 / (PEAK_FP64_TFLOPS * 1e12), 4),
                        "fp64_note": "92 algorithmic FP64 operations per evaluation against the 78.6 TFLOP/s vector peak: the "
                                     "evaluation is ten IEEE divisions and four square roots deep, which cost an order of magnitude "
                                     "more instructions than they count operations",
                    }
                    if exact:
                        rate = evals / 64.0 * exact["valu_instructions_per_wave_and_cell"] / 1e9
                        rows[-1]["roofline"].update({
                            "bound": "valu issue", "achieved": round(rate, 1), "peak": PEAK_VALU_GINST, "unit": "G wave-instructions/s",
                            "frac": round(rate / PEAK_VALU_GINST, 4)})
                        rows[-1]["roofline"].update(exact)
        out["where_a_model_is_found"] = rows
        # extension (no counterpart in the reference): the Sampson distance as the loss, on the stress shape
        el, kms, launches, best = measure(512, 64, 0.15, True, 1, 1, 91, loss=1)
        out["sampson_distance_extension"] = {"pairs": 512, "inlier_ratio": 0.15, "hyps_per_s": round(512 * H / el, 1),
                                             "score_kernel_ms": round(kms, 3), "mean_best_inliers": float(np.mean(best["count"]))}
    return out


def cpu_baseline(args, desc, pts, ransac_out, pipe_inputs=None):
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
        oracle.emulate_shuffle(True)  # the reference shuffles all indices once per iteration (RANSAC.cpp:201)
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
        oracle.emulate_shuffle(False)
        out["ransac"] = {"value": round(done * Hc / el, 2), "unit": "hyps/s", "cores": threads, "kind": "port",
                         "sample": f"{done} pairs x {nc} correspondences x {Hc} hypotheses; per hypothesis the full shuffle of "
                                   f"the {nc} indices (RANSAC.cpp:201), the fit, the triangulation of every correspondence by "
                                   "column-pivoted QR and the loss; OpenMP over hypotheses in place of tbb::parallel_for; "
                                   "input generation included in the time (<5 %)"}
    if pipe_inputs is not None:
        # the default-run stand-in's 26 pairs through the restatement: match, then RANSAC on the same
        # correspondences with the library's generator restated for the sample tables
        fdesc, fpts, fcounts, fpl, cx, cy, coff, K, H = pipe_inputs
        fstarts = np.concatenate([[0], np.cumsum(fcounts)])
        t0 = time.perf_counter()
        for a, b in fpl:
            A, B = slice(fstarts[a], fstarts[a + 1]), slice(fstarts[b], fstarts[b + 1])
            oracle.match(fdesc[A], fpts[A], fdesc[B], fpts[B], threshold=0.95, max_dist_sq=100.0 * 100.0)
        tm = time.perf_counter() - t0
        oracle.emulate_shuffle(True)
        t0 = time.perf_counter()
        for k in range(len(fpl)):
            nk = int(coff[k + 1] - coff[k])
            S = oracle.sample_table(nk, H, 20261111, k)
            oracle.ransac(cx[coff[k]:coff[k + 1]], cy[coff[k]:coff[k + 1]], K, S, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        tr = time.perf_counter() - t0
        oracle.emulate_shuffle(False)
        out["default_run_stand_in"] = {"pairs": int(len(fpl)), "match_ms": round(tm * 1e3, 1), "ransac_ms": round(tr * 1e3, 1),
                                       "cores": threads, "kind": "port",
                                       "note": "the restatement scores every row of the table and then replays the in-order scan; "
                                               "the reference's loop stops at the adaptive cap, so its RANSAC time is lower"}
    return out


if __name__ == "__main__":
    main()
