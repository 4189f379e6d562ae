/*
 * sfm_amd.h -- C-ABI of libsfm_amd.so: the MI355X (gfx950) implementation of the
 * SfM pipeline's data-parallel front end (pairwise descriptor matching and
 * RANSAC two-view geometry).
 *
 * Plain C: opaque handles, raw pointers and sizes, int status codes, no
 * exceptions.  Unless a parameter says "device", every pointer is host memory.
 * Matrices are column-major (Eigen's default), so Mat3::data(), Mat4::data()
 * and std::vector<Vec2>::data() of the reference can be passed as they are.
 *
 * Each entry point names the reference interface (file:line in the reference
 * tree) whose body it replaces.
 */
#ifndef SFM_AMD_H
#define SFM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define SFM_OK 0
#define SFM_ERR_INVALID_ARG 1  /* null pointer, negative size, unknown enum     */
#define SFM_ERR_NO_DEVICE 2    /* no HIP device / wrong architecture            */
#define SFM_ERR_OOM 3          /* device or host allocation failed              */
#define SFM_ERR_UNSUPPORTED 4  /* valid request this build has no kernel for    */
#define SFM_ERR_CAPACITY 5     /* caller's output buffer too small              */
#define SFM_ERR_HIP 6          /* a HIP runtime call failed; see sfm_last_error */
#define SFM_ERR_TIMEOUT 7      /* a multi-GPU exchange did not finish in time   */

/* match/match.hpp:18-46: the two similarity metrics the reference provides.
 * (MATCHING_OPTIONS::similarityMetric holding anything else has no kernel.) */
#define SFM_METRIC_COSINE 0
#define SFM_METRIC_DOT 1
#define SFM_METRIC_L2 2 /* extension, through sfm_match_batch_l2 only */

/* match/match.hpp:10-13: matchingAlgorithm; the only value the reference has. */
#define SFM_MATCH_TWO_SIDED 0

/* solve/solve.hpp:119-124: the default RANSAC_OPTIONS::loss; Sampson is an
 * extension with no counterpart in the reference. */
#define SFM_LOSS_REPROJECTION 0
#define SFM_LOSS_SAMPSON 1

typedef struct sfm_ctx sfm_ctx;
typedef struct sfm_imageset sfm_imageset;
typedef struct sfm_match_result sfm_match_result;
typedef struct sfm_ransac_batch sfm_ransac_batch;

/* ---- context -------------------------------------------------------------
 * One context per device per host thread.  Owns a HIP stream and a grow-only
 * device workspace; every call below is asynchronous on that stream unless it
 * returns data to the host.  No counterpart in the reference (it is CPU-only).
 */
int sfm_ctx_create(int device, sfm_ctx **ctx_out);
void sfm_ctx_destroy(sfm_ctx *ctx);
int sfm_ctx_synchronize(sfm_ctx *ctx);
/* Message of the last failure on this context ("" if none). */
const char *sfm_last_error(const sfm_ctx *ctx);
const char *sfm_version(void);

/* Tunables (all optional).  key: "cand_capacity" (candidate slots per image
 * pair, default 524288; pairs that need more are scored cell by cell),
 * "pairs_per_launch" (default 2048), "hyps_per_launch" (rows of the first
 * slice of an adaptive RANSAC batch: rows are scored in slices growing by a
 * factor of four and pairs whose scan has stopped skip the later ones;
 * default 0 = 256 rows when the launch is small enough to be split over
 * correspondences, else sized to fill the device), "ransac_split" (1 = a
 * scoring launch with few blocks cuts every pair's correspondences over several
 * blocks and adds the errors up in order in a second kernel, default; 0 = one
 * block per 256 rows always; results are identical), "sim_variant"
 * (instance of the similarity kernel: 1 = the default, the int8 bound, preceded
 * by the 6-bit bound where "match_six_bit" provides its table; 0 = fp16
 * bound; 2..11 = ablations that give wrong results and exist for timing only),
 * "ransac_filter" (1 = skip evaluations the
 * epipolar-distance pre-test proves to be outliers, default; 0 = evaluate
 * everything; results are identical), "ransac_mfma" (1 = unsplit scoring
 * launches run that pre-test on the matrix cores, default; 0 = on the vector
 * ALU in packed fp32; results are identical), "ransac_mfma_ops_gib" (largest table of prepared matrix-core
 * operands, 128 bytes per correspondence, that a RANSAC call may allocate, in
 * GiB, default 64; 0 or a call that needs more = every block makes the
 * operands of its correspondences as it goes; results are identical),
 * "ransac_defer_mib" (buffer, in MiB, default 2048, for the sample rows with
 * many inliers that a batch scored from prepared operands evaluates cell by cell
 * with kernels of their own, 8 bytes per correspondence of such a row; 0 = every
 * row stays in the scoring kernel; results are identical),
 * "chain_table_entries" (entries of the
 * scale chain's (step, track) table, default 2^26; a sequence that needs more
 * goes through in windows of steps; results are identical), "hamming_mfma"
 * (1 = 256-bit binary descriptors are first matched on the matrix cores as
 * sign vectors, default; 0 = popcount kernel only; results are identical),
 * "ransac_seed" / "ransac_max_time_ms" (single-pair RANSAC: see sfm_ransac_pair),
 * "match_list_observe" (1, default: a batch call of more than 32 chunks does not
 * allocate its match lists for the worst case, min(nA, nB) entries per pair, but
 * keeps the buffer ahead of the running total that the device reports chunk by
 * chunk, growing it at most a few times; 2 = calls of more than 8 chunks do; 0 =
 * always the worst case; results are identical),
 * "match_overlap" (1 = a batch of several chunks runs what follows a chunk's
 * bound passes on a second stream of the context while the next chunk's bound
 * passes start, default; 0 = everything on one stream; results are identical;
 * either way the call's work is ordered before whatever follows it on the
 * context), "match_two_phase" (1 = pairs whose candidate lists overflow and whose
 * pixel gate excludes nothing are redone with two cheap sweeps, the maxima
 * of rows and columns first and a strict sweep against them second,
 * default; 0 = they take the filtered sweep with running row bounds like the
 * pairs with an effective gate; results are identical),
 * "match_flood_probe" (1 = a pair whose lists overflowed under the int8 bound
 * is sampled, 64 x 256 cells, under the fp16 bound, and goes straight to the
 * filtered passes when that bound would overflow them as well, default; 0 =
 * it always tries the fp16 strict pass first; results are identical),
 * "match_six_bit" (1 = image sets of batch calls also get a table of 6-bit
 * floats (e2m3, 128 bytes per row), and the first bound pass of the cosine and
 * L2 metrics runs on it, at twice the int8 rate of the matrix cores; its
 * error bound is wider, and a pair it floods is redone under the int8 bound,
 * default; 0 = the int8 pass comes first; set it before the image set is
 * made; results are identical),
 * "match_dot_strict" (1 = with the dot metric a pair first takes the strict
 * passes under the cosine bound threshold / (largest norm of A x largest
 * norm of B), which is tight for descriptors of unit norm, default; 0 = the
 * bound is scaled by the two norms cell by cell for every pair; results are
 * identical),
 * "match_pair_reuse" (0 = every sfm_match_pair / sfm_match_pair_u8 call uploads
 * both images, default; 1 = the caller promises that a descriptor table and a
 * point table passed again under the same addresses, with the same row count
 * and descriptor length, have not changed since the last call: when image A
 * of a call is image B of the call before - the reference matches frame i
 * against i + 1 and then i + 1 against i + 2, scene.cpp:60 - it is copied from
 * the device instead of uploaded; results are identical when the promise
 * holds),
 * "cand_observe" (1 = a batch call of several chunks runs a first
 * chunk of at most 128 pairs with "cand_capacity" slots per pair, looks at how
 * far the lists filled and sizes the lists of the other chunks from that,
 * remembering the size in the result object for later calls with the same
 * options, default; 0 = every chunk gets "cand_capacity" slots per pair;
 * results are identical), "match_col_split" (0 = a matcher launch too small to fill the
 * device cuts every row block's columns into 2, 4, 8 or 16 parts, one block
 * each, default; 1, 2, 4, 8, 16 = that many parts always; results are
 * identical).
 * "span_accumulate" (1 = sfm_ctx_last_kernel_ms sums over all batch calls
 * since it was last read instead of the last call only; default 0).
 * Returns SFM_ERR_INVALID_ARG for an
 * unknown key or a value out of range. */
int sfm_ctx_set_option(sfm_ctx *ctx, const char *key, int64_t value);

/* Page-locked host memory.  No counterpart in the reference: the adaptor that
 * packs std::vector<Keypoint> into tables (match.hpp:49-53) owns those tables
 * and may as well keep them here; sfm_match_pair and sfm_ransac_pair then
 * upload at the link's rate (about twice the rate from pageable memory, which
 * the runtime has to pin piece by piece).  Any host pointer is accepted by
 * every entry point either way.  sfm_host_free may be called after the
 * context is gone. */
int sfm_host_alloc(sfm_ctx *ctx, size_t bytes, void **ptr_out);
void sfm_host_free(void *ptr);

/* Device time, in milliseconds, that the dominant kernel of the last batch
 * call spent (sum over its launches, measured with HIP events on the
 * context's stream), and how many launches that was.  which: 0 = matcher
 * similarity kernel, 1 = RANSAC scoring kernel. */
int sfm_ctx_last_kernel_ms(sfm_ctx *ctx, int which, double *ms_out, int *launches_out);

/* ---- matching: one image pair ---------------------------------------------
 * Replaces the body of
 *   std::vector<std::tuple<int,int>> SfM::match::match(keypoints1, keypoints2, options)
 *   (submodules/match/match.hpp:70, match.cpp:9-22 -> matchTwoSided, match.cpp:24-94)
 * for options.similarityMetric in {cosineSimilarity, dotProduct}
 * (match.hpp:18-46).  descA: nA x dim floats, row-major (Keypoint::descriptor,
 * SfM.hpp:132, packed); ptsA: nA x 2 doubles (Keypoint::point, SfM.hpp:133).
 * Writes (i, j) index pairs, ascending i, into pairs_out[2*capacity]; the
 * count goes to *count_out.  capacity >= min(nA, nB) always suffices.
 * algorithm other than SFM_MATCH_TWO_SIDED -> SFM_ERR_INVALID_ARG (the
 * reference throws std::invalid_argument, match.cpp:18).  Empty inputs give
 * count 0.  matches1_out[nA] / matches2_out[nB] (optional, may be NULL)
 * receive the per-row / per-column best indices (match.cpp:30-31), -1 = none.
 */
int sfm_match_pair(sfm_ctx *ctx, int algorithm,
                   const float *descA, const double *ptsA, int nA,
                   const float *descB, const double *ptsB, int nB, int dim,
                   int metric, float threshold, float max_distance_px_squared,
                   int32_t *pairs_out, int capacity, int *count_out,
                   int32_t *matches1_out, int32_t *matches2_out);

/* The same call for byte-valued descriptors: descA / descB hold one byte per
 * element (nA x dim, nB x dim).  OpenCV's SIFT delivers integers in [0, 255]
 * that the reference stores as floats (submodules/detect/detectOpenCv.cpp:166-169);
 * an adaptor that sees only such values while it packs the Keypoint vectors
 * (match.hpp:49-53) can hand them over as bytes: a quarter of the upload, and
 * the widened bytes are the same float values, so the results are those of
 * sfm_match_pair on the float tables, bit for bit. */
int sfm_match_pair_u8(sfm_ctx *ctx, int algorithm,
                      const uint8_t *descA, const double *ptsA, int nA,
                      const uint8_t *descB, const double *ptsB, int nB, int dim,
                      int metric, float threshold, float max_distance_px_squared,
                      int32_t *pairs_out, int capacity, int *count_out,
                      int32_t *matches1_out, int32_t *matches2_out);

/* ---- matching: resident image table + pair list ----------------------------
 * The batch form of the same function, for many pairs over one image set
 * (the reference calls match() once per consecutive frame pair,
 * submodules/scene.cpp:60).  The image set is uploaded once and stays in HBM:
 * desc = all descriptors back to back (sum(counts) x dim floats), pts likewise
 * (sum(counts) x 2 doubles), counts[n_images] = keypoints per image.
 */
int sfm_imageset_create(sfm_ctx *ctx, int n_images, const int32_t *counts,
                        const float *desc, const double *pts, int dim,
                        sfm_imageset **set_out);
/* The same from byte-valued descriptors (SIFT as OpenCV quantises it,
 * detectOpenCv.cpp:166-169: integers 0..255), desc[rows][dim] uint8: a quarter
 * of the upload; results are those of the float table holding the same values. */
int sfm_imageset_create_u8(sfm_ctx *ctx, int n_images, const int32_t *counts, const uint8_t *desc,
                           const double *pts, int dim, sfm_imageset **set_out);
/* Same, with desc/pts already in device memory of ctx's device. */
int sfm_imageset_create_device(sfm_ctx *ctx, int n_images, const int32_t *counts,
                               const float *desc_device, const double *pts_device, int dim,
                               sfm_imageset **set_out);
void sfm_imageset_destroy(sfm_imageset *set);
/* Bytes of device memory the set holds. */
int64_t sfm_imageset_device_bytes(const sfm_imageset *set);

/* Match every (a, b) in pair_list[2*n_pairs] (image indices into the set).
 * The result stays on the device as CSR: offsets[n_pairs+1] (int64) and
 * pairs[2*offsets[n_pairs]] (int32 (i, j), ascending i within a pair).  The
 * call is asynchronous; the accessors below synchronise.  A result object may
 * be passed back in through *result_inout to reuse its buffers (pass a
 * pointer to NULL the first time).
 */
int sfm_match_batch(sfm_ctx *ctx, const sfm_imageset *set,
                    const int32_t *pair_list, int n_pairs,
                    int metric, float threshold, float max_distance_px_squared,
                    sfm_match_result **result_inout);
/* Extension with no counterpart in the reference (its matcher has no second-best
 * test; pinned by this library's CPU restatement only): sfm_match_batch with
 * the cosine metric plus Lowe's ratio test on the unit sphere.  Row i whose
 * best column j1 has similarity s1 keeps it only if every other column j
 * inside the pixel gate has (1 - s1) < ratio^2 * (1 - sim(i, j)), evaluated in
 * double on the fp32 similarities (d^2 = 2 - 2 sim, so this is d1 < ratio * d2
 * for the second-nearest d2); rows that fail have no match, then the mutual
 * check of match.cpp:84-91 runs as usual.  ratio in (0, 1]. */
int sfm_match_batch_ratio(sfm_ctx *ctx, const sfm_imageset *set,
                          const int32_t *pair_list, int n_pairs,
                          float threshold, float max_distance_px_squared, float ratio,
                          sfm_match_result **result_inout);
/* Extension with no counterpart in the reference (its metrics are cosine and dot,
 * match.hpp:18-46; pinned by this library's CPU restatement only): the structure
 * of matchTwoSided (match.cpp:24-94) under the squared Euclidean distance, for
 * descriptors that are not unit-norm (where the L2 order differs from the cosine
 * order).  d^2(a, b) = (|a|^2 + |b|^2) - 2 a.b, each of the three an fp32 sum in
 * index order, negative results clamped to 0.  A cell is admissible when
 * d^2 < max_distance_squared and the keypoints lie inside the pixel gate; rows and
 * columns keep their nearest admissible cell (the lowest index among equals);
 * ratio > 0 adds Lowe's test: a row keeps its best at d1 only if every other
 * column inside the gate has d1^2 < ratio^2 d2^2 (in double); then the mutual
 * check, ascending row.  ratio = 0: no test; else ratio in (0, 1]. */
int sfm_match_batch_l2(sfm_ctx *ctx, const sfm_imageset *set,
                       const int32_t *pair_list, int n_pairs,
                       float max_distance_squared, float max_distance_px_squared, float ratio,
                       sfm_match_result **result_inout);
int sfm_match_result_total(sfm_match_result *res, int64_t *total_matches_out);
int sfm_match_result_offsets(sfm_match_result *res, int64_t *offsets_out /* n_pairs+1 */);
int sfm_match_result_pairs(sfm_match_result *res, int32_t *pairs_out /* 2*total */);
/* Device views (valid until the result is reused or destroyed). */
int sfm_match_result_device(sfm_match_result *res, const int64_t **offsets_device,
                            const int32_t **pairs_device);
/* Number of candidate cells the exact pass evaluated / that did not fit the
 * candidate list and were evaluated inside the similarity kernel. */
int sfm_match_result_stats(sfm_match_result *res, int64_t *candidates_out, int64_t *overflow_out);
/* stats4_out[0] candidates re-scored exactly, [1] pairs scored cell by cell,
 * [2] pairs that needed the filtered (row-best / column-best) passes, [3] pairs
 * whose int8 bound admitted too many cells and that the fp16 bound settled. */
int sfm_match_result_stats_ex(sfm_match_result *res, int64_t *stats4_out);
/* Bytes of device memory the result object holds (work buffers and lists; no
 * counterpart in the reference). */
int64_t sfm_match_result_device_bytes(const sfm_match_result *res);
void sfm_match_result_destroy(sfm_match_result *res);

/* ---- RANSAC: one image pair -------------------------------------------------
 * Replaces the body of
 *   std::vector<int> SfM::solve::RANSAC(x, y, K, options, verbose)
 *   (submodules/solve/solve.hpp:137, RANSAC.cpp:146-270)
 * for options.model == eightPointAlgorithmFromSubset (eightPoint.cpp:225-372)
 * and options.loss == the default reprojection loss (solve.hpp:119-124).
 * x, y: n x 2 doubles, K^-1-normalised (scene.cpp:246-247); K: 3x3
 * column-major.  samples[n_hyp][8]: the index rows to fit (the reference draws
 * them with a time-seeded shuffle, RANSAC.cpp:170,201, so any table is a legal
 * run); hypotheses are taken in row order, as one TBB worker would.  samples ==
 * NULL: the library draws the table on the device (the generator documented at
 * sfm_corr_sample_table, pair index 0) from the context option "ransac_seed"
 * (0, the default: seeded from the clock, as the reference).  The context option
 * "ransac_max_time_ms" (default 0 = none) is RANSAC_OPTIONS::maxTimeMs
 * (RANSAC.cpp:189-196): with adaptive != 0 and no per-row outputs the rows are
 * scored in growing slices, and once the device clock passes the limit no further
 * slice is scored.
 * inliers_out receives up to min(n, capacity) entries of which the first
 * *count_out are the result.
 * min_n must be 8 (RANSAC_OPTIONS::minN, solve.hpp:113); other values ->
 * SFM_ERR_UNSUPPORTED.  n < min_n -> count 0 (RANSAC.cpp:148-149).
 * adaptive != 0 applies the maxIter update of RANSAC.cpp:245-258 while
 * scanning the hypotheses in order.
 * Outputs: inliers_out[capacity] ascending indices and *count_out; optional
 * (NULL to skip): *best_hyp_out (row of samples, -1 if none), pose_out[16]
 * (4x4 column-major [R|T] of the winner), counts_out[n_hyp], toterr_out[n_hyp],
 * *iterations_out (hypotheses the in-order scan reached).
 */
int sfm_ransac_pair(sfm_ctx *ctx, const double *x, const double *y, int n, const double *K,
                    const int32_t *samples, int n_hyp, int min_n,
                    double max_squared_error, double success_prob, int adaptive, int loss,
                    int32_t *inliers_out, int capacity, int *count_out,
                    int *best_hyp_out, double *pose_out,
                    int32_t *counts_out, double *toterr_out, int *iterations_out);

/* sfm_ransac_pair for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc, code that holds a capture open on another stream).  No counterpart in the reference.
 * Same arguments, same kernels, same results as sfm_ransac_pair, but all device work space is a fixed array
 * that the library's code object carries (about 3.4 MB per GPU, held until the process ends), so the call
 * performs no hipMalloc, no hipFree and creates no events.  It still copies from and to the caller's host
 * arrays and waits for its stream before it returns, so it cannot itself be recorded into a graph.  Limits: n <= 16384 correspondences and
 * n_hyp <= 8192 sample rows; beyond them the call returns SFM_ERR_CAPACITY and does nothing.  All calls
 * on one GPU share that array: calls that overlap in time on one GPU (two contexts or two host threads)
 * race; successive calls are safe. */
int sfm_ransac_pair_static(sfm_ctx *ctx, const double *x, const double *y, int n, const double *K,
                           const int32_t *samples, int n_hyp, int min_n,
                           double max_squared_error, double success_prob, int adaptive, int loss,
                           int32_t *inliers_out, int capacity, int *count_out,
                           int *best_hyp_out, double *pose_out,
                           int32_t *counts_out, double *toterr_out, int *iterations_out);

/* Fit only: pose12_out[n_hyp][12] = R row-major (9) then T (3) per sample row.
 * Replaces the fit half of eightPointAlgorithmFromSubset (eightPoint.cpp:233-344). */
int sfm_eight_point_fit(sfm_ctx *ctx, const double *x, const double *y, int n,
                        const int32_t *samples, int n_hyp, double *pose12_out);

/* The whole of eightPointAlgorithmFromSubset (eightPoint.cpp:225-372): pose from the
 * n_calc >= 8 sample correspondences (8: the 8x9 system of a RANSAC row; more:
 * the n_calc x 9 system, solved as eightPointAlgorithm solves it), then every one
 * of the n_all correspondences triangulated under it (depth through float, :365).
 * pose16_out column-major [R|T; 0 0 0 1], points_out[n_all][3]; either may be
 * NULL.  n_calc < 8 -> SFM_ERR_INVALID_ARG with the reference's message. */
int sfm_eight_point_from_subset(sfm_ctx *ctx, const double *calc1, const double *calc2, int n_calc,
                                const double *all1, const double *all2, int n_all,
                                double *pose16_out, double *points_out);

/* eightPointAlgorithm(points1, points2), eightPoint.cpp:57-223, on one point
 * set: pose16_out = column-major [R|T; 0 0 0 1], points_out[n][3] (either may be
 * NULL).  n < 8 -> SFM_ERR_INVALID_ARG with the reference's message. */
int sfm_eight_point_algorithm(sfm_ctx *ctx, const double *x, const double *y, int n,
                              double *pose16_out, double *points_out);

/* ---- RANSAC: batch over many pairs ------------------------------------------
 * CSR correspondences: pair p owns x[2*off[p] .. 2*off[p+1]) (same for y);
 * samples[p][n_hyp][8] index into the pair's own range.  All pairs share K and
 * the options.  Results per pair: best hypothesis, its count, its total error,
 * its pose (12 doubles, R row-major then T), and CSR inlier lists.
 */
int sfm_ransac_batch_run(sfm_ctx *ctx, int n_pairs, const int64_t *corr_offsets,
                         const double *x, const double *y, const double *K,
                         const int32_t *samples, int n_hyp,
                         double max_squared_error, double success_prob, int adaptive, int loss,
                         sfm_ransac_batch **batch_inout);
/* Same with x, y, samples already in device memory (offsets stay on the host). */
int sfm_ransac_batch_run_device(sfm_ctx *ctx, int n_pairs, const int64_t *corr_offsets,
                                const double *x_device, const double *y_device, const double *K,
                                const int32_t *samples_device, int n_hyp,
                                double max_squared_error, double success_prob, int adaptive,
                                int loss, sfm_ransac_batch **batch_inout);
int sfm_ransac_batch_best(sfm_ransac_batch *b, int32_t *best_hyp_out, int32_t *best_count_out,
                          double *best_toterr_out, double *pose12_out, int32_t *iterations_out);
int sfm_ransac_batch_inlier_offsets(sfm_ransac_batch *b, int64_t *offsets_out /* n_pairs+1 */);
int sfm_ransac_batch_inliers(sfm_ransac_batch *b, int32_t *inliers_out);
/* Refit on the winner's inliers: eightPointAlgorithm(inliers1, inliers2),
 * eightPoint.cpp:57-223, as Scene::getApproximations calls it after RANSAC
 * (scene.cpp:264-298): the point set of a pair is its inlier list when that has
 * at least 8 entries, else all its correspondences, else nothing (used = 0,
 * pose = identity).  The correspondences of the RANSAC call must still be alive
 * when this runs (device variants read the caller's memory).
 * fetch: pose16_out[n_pairs][16] column-major [R|T; 0 0 0 1]; used_out[n_pairs];
 * points_out[total][3] with pair p's used[p] triangulated points starting at
 * row corr_offsets[p].  Any output may be NULL. */
int sfm_ransac_batch_refit(sfm_ransac_batch *b);
int sfm_ransac_batch_refit_fetch(sfm_ransac_batch *b, double *pose16_out, int32_t *used_out,
                                 double *points_out);
/* How the last run dealt with sample rows that hold many inliers (no counterpart in the
 * reference, whose loop evaluates every correspondence of every row in turn,
 * RANSAC.cpp:207-221): rows found to pass the pre-test on more than a set share of a
 * pair's first correspondences (candidates), of which those that could not reach the
 * best count of the slices before and were left at count 0 (passed over), and those
 * evaluated cell by cell by kernels of their own (handed over); the rest stayed with
 * the scoring kernel.  Results do not depend on any of this.  Any output may be NULL. */
int sfm_ransac_batch_handed_over(sfm_ransac_batch *b, int64_t *candidates_out, int64_t *passed_over_out,
                                 int64_t *handed_over_out);
void sfm_ransac_batch_destroy(sfm_ransac_batch *b);

/* ---- binary descriptors (extension) -----------------------------------------
 * No counterpart in the reference (its descriptors are float vectors,
 * SfM.hpp:132); pinned by this library's CPU restatement only.  The structure
 * of matchTwoSided (match.cpp:24-94) under the Hamming distance: a cell is
 * admissible when popcount(a xor b) <= max_distance and the keypoints lie
 * inside the pixel gate; rows and columns take their nearest admissible cell,
 * the lowest index among equals; mutual pairs, ascending row.
 * desc[sum(counts)][bytes_per_desc] uint8 (16, 32 or 64 bytes: 128, 256 or 512
 * bits), pts[sum(counts)][2] pixels.  The result is an ordinary match result.
 */
typedef struct sfm_binset sfm_binset;
int sfm_binset_create(sfm_ctx *ctx, int n_images, const int32_t *counts, const uint8_t *desc,
                      const double *pts, int bytes_per_desc, sfm_binset **set_out);
void sfm_binset_destroy(sfm_binset *set);
int sfm_match_batch_hamming(sfm_ctx *ctx, const sfm_binset *set, const int32_t *pair_list, int n_pairs,
                            int max_distance, float max_distance_px_squared,
                            sfm_match_result **result_inout);

/* ---- matching -> RANSAC without leaving the device -------------------------
 * Replaces the correspondence assembly of Scene::getApproximations,
 * scene.cpp:216-252: for every match (i, j) of every pair of a match result,
 * x = normalizePoints(pixel of A_i), y = normalizePoints(pixel of B_j)
 * (scene.cpp:495-501: the first two components of Kinv * (u, v, 1); Kinv is the
 * caller's K.inverse(), column-major), in match-list order, in the CSR layout
 * of the batched RANSAC calls (offsets = the match result's offsets).
 */
typedef struct sfm_corr sfm_corr;
int sfm_corr_from_matches(sfm_ctx *ctx, const sfm_imageset *set, sfm_match_result *res,
                          const double *Kinv, sfm_corr **corr_inout);
int sfm_corr_offsets(sfm_corr *c, int64_t *offsets_out /* n_pairs+1 */);
int sfm_corr_device(sfm_corr *c, const double **x_device, const double **y_device);
int sfm_corr_fetch(sfm_corr *c, double *x_out, double *y_out);
/* Sample table samples[pair][n_hyp][8] drawn on the device (replaces the
 * time-seeded shuffle of RANSAC.cpp:170,201 by a counter-based generator, so a
 * run can be repeated): row h of pair p with n correspondences takes, for
 * k = 0..7, position floor(r_k * (n - k) / 2^64) among the indices not taken
 * yet, r_k = mix(s + (k + 1) * 0x9E3779B97F4A7C15), s = mix(seed ^ mix(p * 2^32 + h)),
 * mix = the splitmix64 finaliser.  Pairs with fewer than 8 correspondences get
 * rows of zeros.  samples_out (host, optional) receives a copy. */
int sfm_corr_sample_table(sfm_corr *c, int n_hyp, uint64_t seed, const int32_t **samples_device,
                          int32_t *samples_out);
/* sfm_corr_sample_table + sfm_ransac_batch_run_device on the correspondences
 * (scene.cpp:261 for every pair of the match result). */
int sfm_ransac_batch_run_corr(sfm_ctx *ctx, sfm_corr *c, const double *K, int n_hyp, uint64_t seed,
                              double max_squared_error, double success_prob, int adaptive, int loss,
                              sfm_ransac_batch **batch_inout);
void sfm_corr_destroy(sfm_corr *c);

/* ---- scale chaining and running point average ------------------------------
 * Replaces the tail of Scene::solveForLastAddedFrame, scene.cpp:301-421, for a
 * whole sequence at once (sfm_scale_chain: geometry only; sfm_scale_chain_colors
 * adds the colour average of scene.cpp:377, :394-407).  Step s joins frames s and s + 1 and brings the
 * two-view solution of that pair: pose (column-major 4x4, m_frame23.pose), the
 * triangulated points in the older camera's frame (m_frame23.points) and the
 * track id of each point (m_trackIndices23), distinct inside a step (the
 * reference keeps them ascending; the result does not depend on their order).
 * CSR over steps: step s owns rows offsets[s] .. offsets[s+1].
 *   s > 0: ratios |pose12 * P12| / |P23| over the tracks both steps hold, both
 *          lengths in (0.1, 100); with at least 5 ratios their median (element
 *          n/2 in ascending order, getMedian scene.cpp:486-493) multiplies the
 *          accumulated scale when it lies in [0.1, 10)          (scene.cpp:301-348)
 *   view = pose, translation * scale, clamped to max_translation  (:356-365)
 *   points with z >= 0: global = accPose * (scale * P); first sighting of a
 *          track stores it, later ones within 50 of the mean update the
 *          running mean and count                                 (:367-411)
 *   accPose *= view^-1, appended to the extrinsics                (:421)
 * Outputs (host, any may be NULL): extrinsics_out[n_steps+1][16] (entry 0 is
 * start_pose16, scene.cpp:181), points3d_out[n_tracks][3] (zeros for tracks
 * never seen), counts_out[n_tracks], acc_scale_out[n_steps] (scale used by
 * step s).  A track id outside [0, n_tracks), or one that occurs twice in a
 * step, is refused (SFM_ERR_INVALID_ARG).
 */
int sfm_scale_chain(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const int32_t *track_ids,
                    const double *points, const double *poses16, const double *start_pose16,
                    double max_translation, int n_tracks, double *extrinsics_out,
                    double *points3d_out, int32_t *counts_out, double *acc_scale_out);
/* The same with the colour of every track (scene.cpp:377, :394-407): colors[total][3]
 * is the colour of the observation behind every point (what
 * util::getPixelBilinearUchar returns for it, scene.cpp:377; sfm_pixel_bilinear_u8
 * below computes it); the first sighting of a track stores it, every later one
 * that updates the running mean updates the colour too, per channel in float
 * arithmetic, clamped to [0, 255] and cut off towards zero (:394-407).
 * colors_out[n_tracks][3] (zeros for tracks never seen). */
int sfm_scale_chain_colors(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const int32_t *track_ids,
                           const double *points, const uint8_t *colors, const double *poses16,
                           const double *start_pose16, double max_translation, int n_tracks,
                           double *extrinsics_out, double *points3d_out, uint8_t *colors_out,
                           int32_t *counts_out, double *acc_scale_out);
/* util::getPixelBilinearUchar (submodules/util/util.cpp:40-78) for n positions of
 * one image: bgr[rows][cols][3] bytes as OpenCV holds them, pixels[n][2] = (u, v)
 * (column, row; denormalizePoints of a correspondence, scene.cpp:377), rgb_out[n][3].
 * The four neighbours' weights and the weighted sum in double, rounded to the
 * nearest integer (ties to even) and clamped, as cv::saturate_cast<uchar> does. */
int sfm_pixel_bilinear_u8(sfm_ctx *ctx, const uint8_t *bgr, int rows, int cols, const double *pixels, int n,
                          uint8_t *rgb_out);
/* Same, reading poses and points of the steps from a refitted RANSAC batch
 * without leaving the device (pair p of the batch = step p; its point set is
 * the refit's: the inlier list when that has at least 8 entries, else all
 * correspondences, else none).  corr_track_ids[total]: the track id of every
 * correspondence of the batch (host), distinct inside a pair. */
int sfm_scale_chain_refit(sfm_ransac_batch *b, const int32_t *corr_track_ids,
                          const double *start_pose16, double max_translation, int n_tracks,
                          double *extrinsics_out, double *points3d_out, int32_t *counts_out,
                          double *acc_scale_out);

/* ---- several GPUs of one node ------------------------------------------------
 * No counterpart in the reference (one process, one pair per call,
 * submodules/scene.cpp:60).  Pairs are independent, so a pair list is cut
 * statically -- blocks of 64 consecutive pairs dealt round-robin: pair i belongs
 * to rank (i / 64) % world -- every GPU matches its share against its own copy of
 * the image set, and the only exchange is the gather of the match lists to one
 * rank with RCCL: an all-gather of the per-pair counts, then every rank's list
 * point to point into one buffer on the root that holds rank 0's lists, then
 * rank 1's, ... (each in the order of that rank's pairs).
 */
typedef struct sfm_comm sfm_comm;
typedef struct sfm_gathered sfm_gathered;
/* Host-only helpers of the partition: pairs rank owns; their ascending indices
 * into the whole list; and, from gathered counts[world][max_owned], where every
 * pair's list starts in the rank-major buffer and how long it is. */
int64_t sfm_shard_count(int64_t n_pairs, int rank, int world);
int sfm_shard_indices(int64_t n_pairs, int rank, int world, int64_t *indices_out);
int sfm_shard_layout(int64_t n_pairs, int world, const int32_t *counts, int64_t max_owned,
                     int64_t *starts_out, int32_t *counts_out);
/* Host-only: the rank-major buffers of a RANSAC gather (see sfm_ransac_gather)
 * put into the order of the whole pair list.  records[world][max_owned] of 120
 * bytes {int32 best row, inlier count, iterations, 0; double error sum; double
 * pose12[12]}, counts[world][max_owned], lists = rank 0's inlier lists, then rank
 * 1's, ...  Outputs as sfm_gathered_ransac's; each may be NULL. */
int sfm_shard_unpack_ransac(int64_t n_pairs, int world, const int32_t *counts, int64_t max_owned,
                            const void *records, const int32_t *lists, int32_t *best_hyp_out,
                            int32_t *best_count_out, double *best_toterr_out, int32_t *iterations_out,
                            double *pose12_out, int64_t *inlier_offsets_out, int32_t *inliers_out);
/* One process per GPU: rank 0 makes an id (128 bytes) and hands it to the other
 * ranks by any channel; every rank then creates its communicator (collective). */
int sfm_comm_unique_id(void *id128_out);
int sfm_comm_create(sfm_ctx *ctx, int rank, int world, const void *id128, sfm_comm **comm_out);
/* One process that owns n devices: ctxs[i] (one context per device, distinct
 * devices) becomes rank i. */
int sfm_comm_create_local(int n, sfm_ctx *const *ctxs, sfm_comm **comms_out);
int sfm_comm_rank(const sfm_comm *comm, int *rank_out, int *world_out);
void sfm_comm_destroy(sfm_comm *comm);
/* Gather (collective over the communicator's ranks): local = this rank's match
 * result over the pairs sfm_shard_indices gives it, in that order.  It runs on a
 * stream of the communicator's own and waits only for the batch call that
 * produced `local`, so the next batch (into another result object) may be started
 * before the previous one is gathered; the call blocks the host once, between
 * its two steps, until the counts have arrived.  sfm_gathered_wait blocks until
 * the lists are in place (root) and `local` may be overwritten (every rank).
 * *gathered_inout is reused across calls (pass a pointer to NULL first). */
int sfm_match_gather(sfm_comm *comm, sfm_match_result *local, int64_t n_pairs_total, int root,
                     sfm_gathered **gathered_inout);
/* The same for all n ranks of a local communicator at once (their calls must be
 * issued together). */
int sfm_match_gather_local(int n, sfm_comm *const *comms, sfm_match_result *const *locals,
                           int64_t n_pairs_total, int root, sfm_gathered **gathered_inout);
/* sfm_match_batch over n devices of one process: partition, match (the devices
 * work side by side), gather to rank root.  sets[i] is the image set on rank
 * i's device (the same images on every device). */
int sfm_match_batch_multi(int n, sfm_comm *const *comms, const sfm_imageset *const *sets,
                          const int32_t *pair_list, int64_t n_pairs,
                          int metric, float threshold, float max_distance_px_squared, int root,
                          sfm_match_result **results_inout, sfm_gathered **gathered_inout);
/* Results.  total and per-rank totals (every rank); per pair of the whole list
 * the start and length of its list in the rank-major buffer (every rank); the
 * lists themselves, pairs_out[2 * total] (root only; synchronises); device views
 * (pairs: root only, else NULL; counts[world][max_owned]). */
int sfm_gathered_wait(sfm_gathered *g);
/* The same with a time limit in milliseconds (<= 0: none): SFM_ERR_TIMEOUT when
 * the exchange has not finished by then -- e.g. a rank that never joined it --
 * instead of a process that hangs.  The waits inside the gather calls and the
 * accessors use the context option "comm_timeout_ms" (default 120000, 0: none). */
int sfm_gathered_wait_timeout(sfm_gathered *g, int64_t timeout_ms);
int sfm_gathered_total(sfm_gathered *g, int64_t *total_out, int64_t *per_rank_totals_out);
int sfm_gathered_layout(sfm_gathered *g, int64_t *starts_out, int32_t *counts_out);
int sfm_gathered_pairs(sfm_gathered *g, int32_t *pairs_out);
int sfm_gathered_device(sfm_gathered *g, const int32_t **pairs_device, const int32_t **counts_device);
void sfm_gathered_destroy(sfm_gathered *g);
/* RANSAC results of a sharded pair list (no counterpart in the reference, whose
 * one call per pair is submodules/scene.cpp:261): `local` is the batch this rank
 * ran over the pairs sfm_shard_indices gives it, in that order.  The winners'
 * records (row, inlier count, error sum, iterations, pose) and the ascending
 * inlier lists go to the root the way the match lists do: an all-gather of the
 * per-pair counts, then point-to-point transfers.  sfm_gathered_wait, _total
 * (inliers in all and per rank), _layout and _destroy serve the object as they
 * serve a matcher's; sfm_gathered_ransac (root only; synchronises) returns the
 * results in the order of the whole pair list: best_hyp, best_count, toterr,
 * iterations [n_pairs_total], pose12 [n_pairs_total][12] (R row-major, then T;
 * identity for a pair without a winner), the inlier lists as CSR
 * (inlier_offsets [n_pairs_total + 1], inliers [total]).  Outputs may be NULL. */
int sfm_ransac_gather(sfm_comm *comm, sfm_ransac_batch *local, int64_t n_pairs_total, int root,
                      sfm_gathered **gathered_inout);
int sfm_ransac_gather_local(int n, sfm_comm *const *comms, sfm_ransac_batch *const *locals,
                            int64_t n_pairs_total, int root, sfm_gathered **gathered_inout);
int sfm_gathered_ransac(sfm_gathered *g, int32_t *best_hyp_out, int32_t *best_count_out,
                        double *best_toterr_out, int32_t *iterations_out, double *pose12_out,
                        int64_t *inlier_offsets_out, int32_t *inliers_out);

#ifdef __cplusplus
}
#endif
#endif /* SFM_AMD_H */
