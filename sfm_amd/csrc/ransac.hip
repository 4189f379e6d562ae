// RANSAC two-view geometry on gfx950: batched 8-point hypotheses and their scoring.
//
// What is computed is solve::RANSAC of the reference (submodules/solve/RANSAC.cpp:146-270) with
// options.model = eightPointAlgorithmFromSubset (solve/eightPoint.cpp:225-372) and the default
// options.loss (solve/solve.hpp:85-95,119-124), for an explicit table of 8-index sample rows
// taken in row order:
//   ransac_prep_kernel   per correspondence: values the reference recomputes per hypothesis
//                        (K*obs of solve.hpp:91, x2h.x2h of the 3x2 system's second column)
//   ransac_fit_kernel    one hypothesis per lane: 8x9 null vector, 3x3 SVD, the four (R,T)
//                        candidates and the cheirality vote (eightPoint.cpp:243-344)
//   ransac_score_kernel  one hypothesis per lane, correspondences staged in LDS and read as
//                        broadcasts: triangulate (eightPoint.cpp:349-369, with the float cast of
//                        :365), the two reprojection errors, count and sum inliers in index order
//   ransac_select_kernel in-order scan with the adaptive cap (RANSAC.cpp:237-258)
//   ransac_inlier_*      ascending inlier list of the winner (RANSAC.cpp:213-221)
//
// All fp64 arithmetic below is written operation by operation (this file is compiled with
// contraction off): same products, same sums, same order as the CPU restatement the tests
// compare against, so counts and error sums agree bit for bit.  Eigen's JacobiSVD (the 8x9 system's null
// vector, the 3x3 essential matrix) and colPivHouseholderQr().solve (the 3x2 triangulation system) are
// followed operation by operation, so that signs, pivots and the order of the candidates are Eigen's.
#include "common.h"

#include <cmath>
#include <mutex>

#pragma clang fp contract(off)

#define SCORE_CH 256 // correspondences staged in LDS per step
#define SCORE_QCAP 16 // per-lane queue of correspondences awaiting exact evaluation
#define REC_D 12     // doubles per prepared correspondence (11 used, padded to 96 bytes)
#define REC_N 11     // of which make_record fills this many
// The pre-test skips a cell whose estimated epipolar distance, less the margin that pays for the fp32
// rounding, is at least this many times sqrt(maxSquaredError); a factor of sqrt(1.02) of it belongs to
// that margin (see the scoring kernel), which leaves 1.052.
#define SKIP_FACTOR 1.0625
#define SCORE_CH32 1024 // correspondences staged per step by the instances that run the pre-test

struct RansacArgs {
    const double2 *x;
    const double2 *y;
    const int64_t *corrOff; // [P+1], device
    const int32_t *samples; // [P][H][8]
    double *rec;            // [total][REC_D]
    float *rec32;           // pre-test records in fp32, four correspondences interleaved (see rec32_start)
    double *pose12;         // [P][H][12] R row-major then T
    int *counts;            // [P][H]
    double *toterr;         // [P][H]
    int *bestH;             // [P]
    int *bestCount;         // [P]
    double *bestErr;        // [P]
    int *iterations;        // [P]
    int *inlCount;          // [P]
    const int64_t *inlOff;  // [P+1]
    int32_t *inliers;
    double K[9]; // column-major
    double Ki[4]; // pre-test constants (valid when filter != 0): 1/fx, 1/fy, 1/min(fx,fy), max(fx,fy)^2
    int plainK;   // 1: K = [fx 0 cx; 0 fy cy; 0 0 1] with finite entries (lets an evaluation skip the camera-1 error, see reprojection_loss)
    int filter;   // 1: the epipolar pre-test of the scoring kernel may be used
    int mfma;     // 1: unsplit launches run the pre-test on the matrix cores (ransac_score_mfma_kernel)
    int mfKe;     // and scale the hypothesis side of num by 2^mfKe
    double mfI[2]; // that test's 1/fx, 1/fy: the camera's for the reprojection loss, 2/(fx + fy) twice for the Sampson distance
    // rows with many inliers, handed over for evaluation cell by cell (see ransac_heavy_find_kernel)
    unsigned int *heavyCount; // [0]: candidates found in this slice, [1]: slots claimed (both may run past their limits: claims beyond were refused);
                              // of the whole call: [2] candidates settled, [3] passed over, [4] handed over
    long long *candRow;       // [candMax] pair << 32 | row
    unsigned int *candUb;     // [candMax] cells of the row that pass the packed test: an upper bound of its count
    long long *heavyRow;      // [heavyMax] pair << 32 | row of the slot
    double *heavyErr;         // [heavyMax][heavyStride] error of the cell where it is an inlier's, else -1
    signed char *rowMark;     // [P][H] 0, ROW_HANDED or ROW_PASSED for the rows of the current slice
    int candMax, heavyMax, heavyStride;
    int heavyBound;           // 1: candUb is made for this slice, and rows whose bound is below the pair's best count are passed over
    uint4 *mfOps; // the correspondence side of that test, made once per call (see mfma_ops_start), or NULL: the
                  // scoring kernel makes it from the fp32 records as it goes
    double maxSqErr;
    double successProb;
    int H;
    int nPairs;
    int adaptive;
    int loss;
    // Hypotheses are evaluated in phases [hBegin, hEnd): a pair whose in-order scan has already
    // stopped (RANSAC.cpp:189, :245-258) takes no part in later phases.
    int hBegin, hEnd;
    int *active;  // [P] 1 while the pair's scan has not stopped
    int *maxIter; // [P] the scan's current iteration cap
    // wall-clock cap (RANSAC.cpp:189-196): device clock at the start of the call, and the ticks it may run
    const unsigned long long *t0;
    long long maxTicks; // 0 = no cap
};

// Last row (exclusive) of pair p in the current phase.  After the first phase a pair's scan has an iteration cap
// (RANSAC.cpp:245-258) that can only fall: rows at or past it are never reached by the in-order loop, so they are
// neither fitted nor scored, whatever the phase's common end is.
__device__ __forceinline__ int pair_row_end(const RansacArgs &g, int p)
{
    return (g.hBegin > 0 && g.adaptive) ? min(g.hEnd, g.maxIter[p]) : g.hEnd;
}

// ---------------------------------------------------------------------------------------
// shared device math (mirrors the CPU restatement's evaluation order)
// ---------------------------------------------------------------------------------------
struct Pose {
    double R[3][3];
    double T[3];
};

__device__ __forceinline__ void rot_apply_h(const double R[3][3], double x, double y, double out[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = (R[i][0] * x + R[i][1] * y) + R[i][2];
}

// Least squares of [-a, b] * lambda = T with b = (x2, y2, 1), bb = b.b, the way the reference solves it:
// A_tri.colPivHouseholderQr().solve(T) (eightPoint.cpp:321,363).  Eigen's algorithm (ColPivHouseholderQR::
// computeInPlace / _solve_impl, makeHouseholder, applyHouseholderOnTheLeft) written out for a 3x2 matrix, in
// the operation order of the CPU restatement the tests compare with:
//   factor: the column of the larger norm (the first of equals) is the first pivot; a reflector of
//           (c0, tail) is beta = -sign(c0) sqrt(c0^2 + |tail|^2), essential = tail / (c0 - beta),
//           tau = (beta - c0) / beta, or tau = 0, beta = c0 for a vanishing tail; the second column's norm is
//           downdated (LAPACK xGEQPF) and only decides whether the second pivot counts;
//   solve:  c = Q^T T by the reflectors in turn, back substitution, column permutation.
// The factorisation does not depend on T, and every operation of the solve commutes with negation, so the
// solution for -T is the negated solution for T, bit for bit: the cheirality vote factors once per rotation.
struct Qr32 {
    double beta0, tau0, e01, e02, r01, beta1, tau1, e11;
    int pivots;   // nonzero pivots: 2, or 1 when the second column vanishes against the first
    bool swapped; // b was the first pivot
};

__device__ __forceinline__ void householder_make(double c0, double t1, double t2, bool two, double &beta, double &tau,
                                                 double &e1, double &e2)
{
    const double tailSqNorm = two ? t1 * t1 + t2 * t2 : t1 * t1;
    if (tailSqNorm <= 2.2250738585072014e-308) {
        tau = 0.0;
        beta = c0;
        e1 = 0.0;
        e2 = 0.0;
    } else {
        double b = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0) b = -b;
        e1 = t1 / (c0 - b);
        e2 = two ? t2 / (c0 - b) : 0.0;
        tau = (b - c0) / b;
        beta = b;
    }
}

// n1 = sqrt(bb) is passed in: it belongs to the correspondence, callers hoist it out of their loops over poses.
__device__ __forceinline__ Qr32 qr32_factor(const double a[3], double x2, double y2, double n1in)
{
    const double eps = 2.220446049250313e-16;
    Qr32 F;
    double p0 = -a[0], p1 = -a[1], p2 = -a[2];
    double q0 = x2, q1 = y2, q2 = 1.0;
    const double n0 = sqrt((p0 * p0 + p1 * p1) + p2 * p2);
    double n1 = n1in;
    const double maxn = (n1 > n0) ? n1 : n0;
    F.pivots = 2;
    F.swapped = n1 > n0;
    if (F.swapped) {
        double t = p0; p0 = q0; q0 = t;
        t = p1; p1 = q1; q1 = t;
        t = p2; p2 = q2; q2 = t;
        n1 = n0;
    }
    householder_make(p0, p1, p2, true, F.beta0, F.tau0, F.e01, F.e02);
    if (F.tau0 != 0.0) {
        double tmp = F.e01 * q1 + F.e02 * q2;
        tmp = tmp + q0;
        q0 = q0 - F.tau0 * tmp;
        q1 = q1 - tmp * (F.tau0 * F.e01);
        q2 = q2 - tmp * (F.tau0 * F.e02);
    }
    // The remaining column's norm, downdated.  Eigen multiplies temp by (updated norm / directly computed norm)^2;
    // both are n1 at this point, and x / x is exactly 1 for every finite non-zero x (for an infinite or NaN n1 the
    // factor is NaN and the comparison it feeds is false), so the division is left out.
    double upd = n1;
    if (upd != 0.0) {
        double temp = fabs(q0) / upd;
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const bool direct = (n1 - n1 == 0.0) && temp <= 1.4901161193847656e-08; // (a NaN factor fails the comparison)
        upd = direct ? sqrt(q1 * q1 + q2 * q2) : upd * sqrt(temp);
    }
    // The second pivot counts unless upd^2 < 2 (maxn eps)^2 / 3.  That threshold is ~3.3e-32 maxn^2: unless upd^2 is
    // below 4e-32 maxn^2 (which the exact threshold cannot exceed, roundings included) the answer is known without
    // the threshold's division.
    const double u2 = upd * upd;
    if (u2 < (maxn * maxn) * 4e-32) {
        const double threshold_helper = ((maxn * eps) * (maxn * eps)) / 3.0;
        if (u2 < threshold_helper * 2.0) F.pivots = 1;
    }
    double unused;
    householder_make(q1, q2, 0.0, false, F.beta1, F.tau1, F.e11, unused);
    F.r01 = q0;
    return F;
}

__device__ __forceinline__ void qr32_solve(const Qr32 &F, const double T[3], double &l0, double &l1)
{
    double c0 = T[0], c1 = T[1], c2 = T[2];
    double s1 = 0.0;
    if (F.tau0 != 0.0) {
        double tmp = F.e01 * c1 + F.e02 * c2;
        tmp = tmp + c0;
        c0 = c0 - F.tau0 * tmp;
        c1 = c1 - tmp * (F.tau0 * F.e01);
        c2 = c2 - tmp * (F.tau0 * F.e02);
    }
    if (F.pivots == 2) {
        if (F.tau1 != 0.0) {
            double tmp = F.e11 * c2;
            tmp = tmp + c1;
            c1 = c1 - F.tau1 * tmp;
        }
        if (c1 != 0.0) {
            c1 = c1 / F.beta1;
            c0 = c0 - c1 * F.r01;
        }
        s1 = c1;
    }
    if (c0 != 0.0) c0 = c0 / F.beta0;
    l0 = F.swapped ? s1 : c0;
    l1 = F.swapped ? c0 : s1;
}

__device__ __forceinline__ void triangulate_lambdas(const double a[3], double x2, double y2, double n1,
                                                    const double T[3], double &l0, double &l1)
{
    const Qr32 F = qr32_factor(a, x2, y2, n1);
    qr32_solve(F, T, l0, l1);
}

// B1 of reprojection_loss: what the camera-1 error of a correspondence cannot exceed, or inf.
__device__ __forceinline__ double loss1_bound(const double *K, int plainK, double x1, double y1)
{
    const double sx = (fabs(K[0] * x1) + fabs(K[3] * y1)) + fabs(K[6]);
    const double sy = (fabs(K[1] * x1) + fabs(K[4] * y1)) + fabs(K[7]);
    const bool ordinary = plainK && fabs(x1) <= 1e6 && fabs(y1) <= 1e6 && sx <= 1e18 && sy <= 1e18;
    return ordinary ? (70.0 * 1.232595164407831e-32) * (sx * sx + sy * sy) : INFINITY; // 70 * 2^-106
}

// Default loss (solve.hpp:119-124) of one prepared correspondence (a record of make_record) under pose (R, T),
// or +inf where the correspondence is shown not to be an inlier (error >= thr, or NaN) without finishing it.
// Two shortcuts, neither of which changes a value that is used:
//   * the loss is max(loss1, loss2) by std::max's rule, (loss1 < loss2) ? loss2 : loss1.  If loss2 >= thr the
//     result is loss2, or a loss1 that is not below loss2 (NaN included): not below thr in any case, so the
//     camera-1 error is never looked at;
//   * the triangulated point lies on the ray of the first observation by construction, so loss1 is rounding
//     noise: with K = [fx 0 cx; 0 fy cy; 0 0 1], a finite non-zero depth and ordinary coordinates the third
//     row of K pt is the depth itself, every one of the eight operations behind each of its two terms is off
//     by at most 2^-53 relatively, and loss1 <= 66.1 2^-106 (Sx^2 + Sy^2) with Sx = |fx x1| + |0 y1| + |cx|, Sy
//     likewise.  The record carries B1 = 70 2^-106 (Sx^2 + Sy^2) (inf where the conditions fail); a finite loss2
//     at or above it is the maximum, whatever loss1 is exactly.
// STAGED: r is a prepared record (make_record: both observations in pixels, B1 and the norm of the second ray
// come with it).  Otherwise r = {x1, y1, x2, y2} and everything else is made from them where it is needed, with
// the same expressions: an evaluation that is one of a few survivors of the pre-test then keeps four values
// alive across the factorisation instead of eleven.
template <bool STAGED>
__device__ __forceinline__ double reprojection_loss(const double *K, int plainK, double thr, const Pose &P, const double *r)
{
    const double x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
    double a[3];
    rot_apply_h(P.R, x1, y1, a);
    double l0, l1;
    triangulate_lambdas(a, x2, y2, STAGED ? r[10] : sqrt((x2 * x2 + y2 * y2) + 1.0), P.T, l0, l1);
    const float lambda1 = (float)l0; // eightPoint.cpp:365
    const double l = (double)lambda1;
    const double pt0 = l * x1, pt1 = l * y1, pt2 = l;
    // camera 2, view = [R|T] (solve.hpp:122)
    const double q0 = ((P.R[0][0] * pt0 + P.R[0][1] * pt1) + P.R[0][2] * pt2) + P.T[0];
    const double q1 = ((P.R[1][0] * pt0 + P.R[1][1] * pt1) + P.R[1][2] * pt2) + P.T[1];
    const double q2 = ((P.R[2][0] * pt0 + P.R[2][1] * pt1) + P.R[2][2] * pt2) + P.T[2];
    double p0 = (K[0] * q0 + K[3] * q1) + K[6] * q2;
    double p1 = (K[1] * q0 + K[4] * q1) + K[7] * q2;
    double p2 = (K[2] * q0 + K[5] * q1) + K[8] * q2;
    p0 = p0 / p2;
    p1 = p1 / p2;
    const double o2x = STAGED ? r[7] : (K[0] * x2 + K[3] * y2) + K[6]; // solve.hpp:91
    const double o2y = STAGED ? r[8] : (K[1] * x2 + K[4] * y2) + K[7];
    double dx = p0 - o2x, dy = p1 - o2y;
    const double loss2 = dx * dx + dy * dy;
    if (loss2 >= thr) return INFINITY;
    const double B1 = STAGED ? r[9] : loss1_bound(K, plainK, x1, y1);
    if (loss2 >= B1 && l != 0.0 && fabs(l) < INFINITY) return loss2;
    // camera 1, view = identity (solve.hpp:121)
    p0 = (K[0] * pt0 + K[3] * pt1) + K[6] * pt2;
    p1 = (K[1] * pt0 + K[4] * pt1) + K[7] * pt2;
    p2 = (K[2] * pt0 + K[5] * pt1) + K[8] * pt2;
    p0 = p0 / p2;
    p1 = p1 / p2;
    const double o1x = STAGED ? r[5] : (K[0] * x1 + K[3] * y1) + K[6];
    const double o1y = STAGED ? r[6] : (K[1] * x1 + K[4] * y1) + K[7];
    dx = p0 - o1x;
    dy = p1 - o1y;
    const double loss1 = dx * dx + dy * dy;
    return (loss1 < loss2) ? loss2 : loss1; // std::max(loss1, loss2)
}

// Sampson distance in pixel^2 (extension loss; no counterpart in the reference).
__device__ __forceinline__ void essential_from_pose(const Pose &P, double E[3][3])
{
    const double Tx[3][3] = {{0, -P.T[2], P.T[1]}, {P.T[2], 0, -P.T[0]}, {-P.T[1], P.T[0], 0}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            E[i][j] = (Tx[i][0] * P.R[0][j] + Tx[i][1] * P.R[1][j]) + Tx[i][2] * P.R[2][j];
}

__device__ __forceinline__ double sampson_loss(const double *K, const double E[3][3], double x1, double y1,
                                               double x2, double y2)
{
    double Ex1[3], Etx2[2];
#pragma unroll
    for (int i = 0; i < 3; i++) Ex1[i] = (E[i][0] * x1 + E[i][1] * y1) + E[i][2];
#pragma unroll
    for (int j = 0; j < 2; j++) Etx2[j] = (E[0][j] * x2 + E[1][j] * y2) + E[2][j];
    const double num = (x2 * Ex1[0] + y2 * Ex1[1]) + Ex1[2];
    const double den = (Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1]) + (Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1]);
    const double f = 0.5 * (K[0] + K[4]);
    return (num * num / den) * (f * f);
}

// ---------------------------------------------------------------------------------------
// prep
// ---------------------------------------------------------------------------------------
// First float of pair p's pre-test records (its correspondences start at row `base`): a multiple of 20,
// and regions of consecutive pairs do not overlap (each takes 20 * ceil(n / 4) floats).
__host__ __device__ inline size_t rec32_start(int64_t base, int p) { return (size_t)20 * (size_t)((base + 3 * (int64_t)p + 3) / 4); }

// The values of a correspondence that the reference recomputes per hypothesis: both observations, x2h.x2h,
// and obs = K * (ox, oy, 1) for both (solve.hpp:90-91).  With the pre-test on they are made where they are
// needed (the rare exact evaluations) instead of being stored: 80 bytes per correspondence less to write.
__device__ __forceinline__ void make_record(const RansacArgs &g, int64_t at, double r[REC_N])
{
    const double2 a = g.x[at], b = g.y[at];
    r[0] = a.x;
    r[1] = a.y;
    r[2] = b.x;
    r[3] = b.y;
    r[4] = (b.x * b.x + b.y * b.y) + 1.0;
    r[5] = (g.K[0] * a.x + g.K[3] * a.y) + g.K[6];
    r[6] = (g.K[1] * a.x + g.K[4] * a.y) + g.K[7];
    r[7] = (g.K[0] * b.x + g.K[3] * b.y) + g.K[6];
    r[8] = (g.K[1] * b.x + g.K[4] * b.y) + g.K[7];
    r[9] = loss1_bound(g.K, g.plainK, a.x, a.y);
    r[10] = sqrt(r[4]); // the norm of the second observation's ray: the second column of every 3x2 system of this correspondence
}

__device__ __forceinline__ double mfma_margin(const RansacArgs &g, double x1n2, double x2n2, double amax);

__global__ __launch_bounds__(256) void ransac_prep_kernel(const RansacArgs g, int64_t total)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    double r[REC_D];
    make_record(g, k, r);
    if (!g.filter) { // the instances without the pre-test stage these records in LDS
        double *dst = g.rec + k * REC_D;
#pragma unroll
        for (int i = 0; i < REC_N; i++) dst[i] = r[i];
        dst[REC_N] = 0.0;
        if (!g.mfma) return; // (Sampson distance: unsplit launches run the matrix-core test, split ones the instances above)
    }
    // x1h.x1h for the scoring kernel's pre-test, or +inf for a correspondence it must not skip
    const double big = 1e6;
    const bool tame = fabs(r[5]) <= big && fabs(r[6]) <= big && fabs(r[7]) <= big && fabs(r[8]) <= big;
    r[9] = tame ? (r[0] * r[0] + r[1] * r[1]) + 1.0 : INFINITY; // (from here on r[9] is the pre-test's, not B1)
    // fp32 record of the pre-test: x1, y1, fx x2 / c, fy y2 / c (the observation in pixels off the centre, in
    // units of the skip distance c = SKIP_FACTOR sqrt(maxSquaredError)) and the margin G that pays for the
    // rounding of the fp32 estimate (derived in the scoring kernel), or inf.  Correspondences 4j .. 4j + 3 of a
    // pair share twenty floats, field by field, so that one packed instruction serves two of them.
    int lo = 0, hi = g.nPairs; // the pair that owns correspondence k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.corrOff[mid] <= k) lo = mid; else hi = mid;
    }
    const int64_t base = g.corrOff[lo];
    const int kk = (int)(k - base);
    const int n = (int)(g.corrOff[lo + 1] - base);
    float *d = g.rec32 + rec32_start(base, lo) + (size_t)(kk >> 2) * 20 + (kk & 3);
    const double c = SKIP_FACTOR * sqrt(g.maxSqErr); // the skip distance, folded into the line (see the scoring kernel)
    d[0] = (float)r[0];
    d[4] = (float)r[1];
    d[8] = (float)(r[2] * g.K[0] / c);
    d[12] = (float)(r[3] * g.K[4] / c);
    const double x1n = sqrt(r[9]), x2n = sqrt(r[4]);
    const double D = (1e-6 * c * g.Ki[2]) * x1n + 2e-6 * (x2n * x1n); // Ki[2] = 1 / min(fx, fy)
    double G = tame ? fmax(70.0 * (D * D), 1e-30) : INFINITY; // (kept clear of the denormals)
    if (g.mfma && tame) // the matrix-core test needs a wider margin; the packed test stays valid with it
        G = fmax(G, mfma_margin(g, r[9], r[4], fmax(fmax(fabs(r[0]), fabs(r[1])), 1.0) * fmax(fmax(fabs(r[2]), fabs(r[3])), 1.0)));
    if (g.mfma && tame && g.loss == SFM_LOSS_SAMPSON) {
        // Sampson = num^2 f^2 / (a + b), a = |(E x1h)_01|^2, b = |(E^T x2h)_01|^2 <= |x2h|^2 (E = [T]x R with |T| = 1
        // has norm 1; the hypothesis side checks |T|): no inlier when num^2 >= c^2 (a + 1.001 |x2h|^2) / f^2.  The
        // first term is the q form with 1/f for both focal lengths, the second a constant of the correspondence
        // that rides with the margin (2e-5 of it more for its hi + lo rounding and its share of the accumulator's).
        G += (c * c) * (g.mfI[0] * g.mfI[0]) * (1.001 * r[4]) * (1.0 + 2e-5);
    }
    d[16] = (float)(G * 1.000001);
    if (kk == n - 1) // the rest of the last group is never read as a result
        for (int f = (kk & 3) + 1; f < 4; f++)
#pragma unroll
            for (int e = 0; e < 5; e++) d[4 * e + f - (kk & 3)] = 0.0f;
}

// ---------------------------------------------------------------------------------------
// fit: one hypothesis per lane
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double M[3][3])
{
    return (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0])) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// Eigen's JacobiSVD, which the reference takes its null vector and the factors of the essential matrix from
// (eightPoint.cpp:254-255, :262-265), written out: the order of the four (R, T) candidates, and with it the
// outcome of a tied cheirality vote (:296-337), depends on the signs and the order that algorithm returns.
// Sources followed (Eigen 3.3 / 3.4): JacobiSVD::compute and its ColPivHouseholderQR preconditioner,
// ColPivHouseholderQR::computeInPlace, makeHouseholder, applyHouseholderOnTheLeft, HouseholderSequence::evalTo,
// real_2x2_jacobi_svd, JacobiRotation::makeJacobi, apply_rotation_in_the_plane.  Sums run in index order, as in
// the CPU restatement the tests compare with bit for bit.
#define SFM_DBL_MIN 2.2250738585072014e-308
#define SFM_DBL_EPS 2.220446049250313e-16

struct JRot {
    double c, s;
};

// JacobiRotation::makeJacobi(x, y, z): diagonalises [x y; y z].
__device__ __forceinline__ void make_jacobi(double x, double y, double z, JRot &j)
{
    const double deno = 2.0 * fabs(y);
    if (deno < SFM_DBL_MIN) {
        j.c = 1.0;
        j.s = 0.0;
        return;
    }
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    const double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    j.s = -sign_t * (y / fabs(y)) * fabs(t) * n;
    j.c = n;
}

// real_2x2_jacobi_svd on [m00 m01; m10 m11]: J_left m J_right is diagonal.
__device__ __forceinline__ void real_2x2_jacobi_svd(double m00, double m01, double m10, double m11, JRot &jl, JRot &jr)
{
    JRot rot1;
    const double t = m00 + m11;
    const double d = m10 - m01;
    if (fabs(d) < SFM_DBL_MIN) {
        rot1.s = 0.0;
        rot1.c = 1.0;
    } else {
        const double u = t / d;
        const double tmp = sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    const bool ident = rot1.c == 1.0 && rot1.s == 0.0;
    const double r00 = ident ? m00 : rot1.c * m00 + rot1.s * m10;
    const double r01 = ident ? m01 : rot1.c * m01 + rot1.s * m11;
    const double r11 = ident ? m11 : -rot1.s * m01 + rot1.c * m11;
    make_jacobi(r00, r01, r11, jr);
    const double oc = jr.c, os = -jr.s; // j_left = rot1 * j_right.transpose()
    jl.c = rot1.c * oc - rot1.s * os;
    jl.s = rot1.c * os + rot1.s * oc;
}

// apply_rotation_in_the_plane on one (x, y): (c x + s y, -s x + c y); the identity leaves them untouched.
__device__ __forceinline__ void rotate_pair(double &x, double &y, double c, double s, bool ident)
{
    const double xi = x, yi = y;
    const double nx = c * xi + s * yi;
    const double ny = -s * xi + c * yi;
    x = ident ? xi : nx;
    y = ident ? yi : ny;
}

// One (p, q) visit of a sweep of JacobiSVD<Mat3>: W <- J_left W, U <- U J_left^T, W <- W J_right, V <- V J_right.
template <int P_, int Q_>
__device__ __forceinline__ bool svd3_visit(double W[3][3], double U[3][3], double V[3][3], double &maxDiag)
{
    const double pm = (2.0 * SFM_DBL_EPS) * maxDiag;
    const double threshold = (SFM_DBL_MIN < pm) ? pm : SFM_DBL_MIN;
    if (!(fabs(W[P_][Q_]) > threshold || fabs(W[Q_][P_]) > threshold)) return false;
    JRot jl, jr;
    real_2x2_jacobi_svd(W[P_][P_], W[P_][Q_], W[Q_][P_], W[Q_][Q_], jl, jr);
    const bool il = jl.c == 1.0 && jl.s == 0.0, ir = jr.c == 1.0 && -jr.s == 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) rotate_pair(W[P_][i], W[Q_][i], jl.c, jl.s, il);
#pragma unroll
    for (int i = 0; i < 3; i++) rotate_pair(U[i][P_], U[i][Q_], jl.c, jl.s, il);
#pragma unroll
    for (int i = 0; i < 3; i++) rotate_pair(W[i][P_], W[i][Q_], jr.c, -jr.s, ir);
#pragma unroll
    for (int i = 0; i < 3; i++) rotate_pair(V[i][P_], V[i][Q_], jr.c, -jr.s, ir);
    const double app = fabs(W[P_][P_]), aqq = fabs(W[Q_][Q_]);
    const double mx = (app < aqq) ? aqq : app;
    if (maxDiag < mx) maxDiag = mx;
    return true;
}

__device__ __forceinline__ void svd3_swap_cols(double U[3][3], double S[3], double V[3][3], int a, int b)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double t = U[i][a];
        U[i][a] = U[i][b];
        U[i][b] = t;
        t = V[i][a];
        V[i][a] = V[i][b];
        V[i][b] = t;
    }
    const double t = S[a];
    S[a] = S[b];
    S[b] = t;
}

// JacobiSVD<Mat3>(E, ComputeFullU | ComputeFullV): E = U diag(S) V^T, S descending.
__device__ __forceinline__ void svd3(const double E[3][3], double U[3][3], double S[3], double V[3][3])
{
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (fabs(E[i][j]) > scale) scale = fabs(E[i][j]);
    if (scale == 0.0) scale = 1.0;
    double W[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            W[i][j] = E[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    double maxDiag = fabs(W[0][0]);
    if (fabs(W[1][1]) > maxDiag) maxDiag = fabs(W[1][1]);
    if (fabs(W[2][2]) > maxDiag) maxDiag = fabs(W[2][2]);
    for (int sweep = 0; sweep < 1000; sweep++) {
        bool any = svd3_visit<1, 0>(W, U, V, maxDiag);
        any |= svd3_visit<2, 0>(W, U, V, maxDiag);
        any |= svd3_visit<2, 1>(W, U, V, maxDiag);
        if (!any) break;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double a = W[i][i];
        S[i] = fabs(a);
        if (a < 0.0) {
#pragma unroll
            for (int r = 0; r < 3; r++) U[r][i] = -U[r][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) S[i] = S[i] * scale;
    // selection sort, the first of equal maxima
    {
        int pos = 0;
        double best = S[0];
        if (S[1] > best) { best = S[1]; pos = 1; }
        if (S[2] > best) { pos = 2; }
        if (pos == 1) svd3_swap_cols(U, S, V, 0, 1);
        if (pos == 2) svd3_swap_cols(U, S, V, 0, 2);
        if (S[2] > S[1]) svd3_swap_cols(U, S, V, 1, 2);
    }
}

// applyHouseholderOnTheLeft on one column w[K_..8] of a matrix with nine rows: tmp = essential . w[K_+1..] +
// w[K_]; w[K_] -= tau tmp; w[K_+1..] -= tmp (tau essential).  The essential part is ess[K_+1..8].
template <int K_>
__device__ __forceinline__ void householder_apply9(double w[9], const double ess[9], double tau)
{
    if (tau != 0.0) {
        double tmp = 0.0;
#pragma unroll
        for (int i = K_ + 1; i < 9; i++) tmp = tmp + ess[i] * w[i];
        tmp = tmp + w[K_];
        w[K_] = w[K_] - tau * tmp;
#pragma unroll
        for (int i = K_ + 1; i < 9; i++) w[i] = w[i] - tmp * (tau * ess[i]);
    }
}

// One step of ColPivHouseholderQR::computeInPlace on the 9x8 matrix whose column c is M[c][0..8].
template <int K_>
__device__ __forceinline__ void colpiv_step9x8(double M[8][9], double nu[8], double nd[8], double h[8])
{
    // the column of the largest updated norm, the first of equals, comes to position K_
    int biggest = K_;
    double bn = nu[K_];
#pragma unroll
    for (int j = K_ + 1; j < 8; j++)
        if (nu[j] > bn) {
            bn = nu[j];
            biggest = j;
        }
#pragma unroll
    for (int j = K_ + 1; j < 8; j++) {
        const bool sw = biggest == j;
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const double a = M[K_][r], b = M[j][r];
            M[K_][r] = sw ? b : a;
            M[j][r] = sw ? a : b;
        }
        const double u0 = nu[K_], u1 = nu[j], d0 = nd[K_], d1 = nd[j];
        nu[K_] = sw ? u1 : u0;
        nu[j] = sw ? u0 : u1;
        nd[K_] = sw ? d1 : d0;
        nd[j] = sw ? d0 : d1;
    }
    // makeHouseholderInPlace on M[K_][K_..8]
    double tailSqNorm = 0.0;
#pragma unroll
    for (int i = K_ + 1; i < 9; i++) tailSqNorm = tailSqNorm + M[K_][i] * M[K_][i];
    const double c0 = M[K_][K_];
    if (tailSqNorm <= SFM_DBL_MIN) {
        h[K_] = 0.0;
#pragma unroll
        for (int i = K_ + 1; i < 9; i++) M[K_][i] = 0.0;
    } else {
        double b = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0) b = -b;
        const double den = c0 - b;
#pragma unroll
        for (int i = K_ + 1; i < 9; i++) M[K_][i] = M[K_][i] / den;
        h[K_] = (b - c0) / b;
        M[K_][K_] = b;
    }
#pragma unroll
    for (int j = K_ + 1; j < 8; j++) householder_apply9<K_>(M[j], M[K_], h[K_]);
    // the norms of the remaining columns, downdated as in LAPACK's xGEQPF
#pragma unroll
    for (int j = K_ + 1; j < 8; j++) {
        if (nu[j] != 0.0) {
            double temp = fabs(M[j][K_]) / nu[j];
            temp = (1.0 + temp) * (1.0 - temp);
            temp = temp < 0.0 ? 0.0 : temp;
            const double ratio = nu[j] / nd[j];
            const double temp2 = temp * (ratio * ratio);
            const bool direct = temp2 <= 1.4901161193847656e-08; // sqrt(eps)
            double sq = 0.0;
#pragma unroll
            for (int i = K_ + 1; i < 9; i++) sq = sq + M[j][i] * M[j][i];
            const double root = sqrt(direct ? sq : temp);
            nd[j] = direct ? root : nd[j];
            nu[j] = direct ? root : nu[j] * root;
        }
    }
}

// JacobiSVD(A, ComputeFullV).matrixV().col(8) of the 8x9 system A whose row c is M[c][0..8]: the last column of
// the Q of the column-pivoted Householder QR of (A / scale)^T.  Every loop is unrolled: M lives in registers.
__device__ __forceinline__ void null_vector_8x9(double M[8][9], double q[9])
{
    double scale = 0.0;
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
        for (int r = 0; r < 9; r++)
            if (fabs(M[c][r]) > scale) scale = fabs(M[c][r]);
    if (scale == 0.0) scale = 1.0;
    if (scale != 1.0) { // (x / 1.0 is x)
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int r = 0; r < 9; r++) M[c][r] = M[c][r] / scale;
    }
    double nu[8], nd[8], h[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        double sq = 0.0;
#pragma unroll
        for (int r = 0; r < 9; r++) sq = sq + M[c][r] * M[c][r];
        nd[c] = nu[c] = sqrt(sq);
    }
    colpiv_step9x8<0>(M, nu, nd, h);
    colpiv_step9x8<1>(M, nu, nd, h);
    colpiv_step9x8<2>(M, nu, nd, h);
    colpiv_step9x8<3>(M, nu, nd, h);
    colpiv_step9x8<4>(M, nu, nd, h);
    colpiv_step9x8<5>(M, nu, nd, h);
    colpiv_step9x8<6>(M, nu, nd, h);
    colpiv_step9x8<7>(M, nu, nd, h);
    // HouseholderSequence::evalTo, the column it makes of e_8: the reflectors from the last to the first
#pragma unroll
    for (int r = 0; r < 9; r++) q[r] = 0.0;
    q[8] = 1.0;
    householder_apply9<7>(q, M[7], h[7]);
    householder_apply9<6>(q, M[6], h[6]);
    householder_apply9<5>(q, M[5], h[5]);
    householder_apply9<4>(q, M[4], h[4]);
    householder_apply9<3>(q, M[3], h[3]);
    householder_apply9<2>(q, M[2], h[2]);
    householder_apply9<1>(q, M[1], h[1]);
    householder_apply9<0>(q, M[0], h[0]);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_fit_kernel(const RansacArgs g)
{
    const int p = blockIdx.y;
    const int h = g.hBegin + blockIdx.x * 64 + threadIdx.x;
    const int64_t base = g.corrOff[p];
    const int64_t n = g.corrOff[p + 1] - base;
    if (n < 8 || !g.active[p] || h >= pair_row_end(g, p)) return;
    const int32_t *s = g.samples + ((size_t)p * g.H + h) * 8;
    int idx[8];
    bool valid = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        idx[i] = s[i];
        valid &= idx[i] >= 0 && idx[i] < n;
    }
    double *out = g.pose12 + ((size_t)p * g.H + h) * 12;
    if (!valid) { // a row that names no correspondence scores nothing
#pragma unroll
        for (int i = 0; i < 12; i++) out[i] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }

    // eightPoint.cpp:243-251: row = [x1*x2, y1*x2, x2, x1*y2, y1*y2, y2, x1, y1, 1]
    double M[8][9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double2 a = g.x[base + idx[i]], b = g.y[base + idx[i]];
        M[i][0] = a.x * b.x;
        M[i][1] = a.y * b.x;
        M[i][2] = b.x;
        M[i][3] = a.x * b.y;
        M[i][4] = a.y * b.y;
        M[i][5] = b.y;
        M[i][6] = a.x;
        M[i][7] = a.y;
        M[i][8] = 1.0;
    }
    double q[9];
    null_vector_8x9(M, q);

    // eightPoint.cpp:257-265
    const double E[3][3] = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}};
    double U[3][3], S[3], V[3][3];
    svd3(E, U, S, V);

    // eightPoint.cpp:274-289: U Rz(+90) has columns (u1, -u0, u2); U Rz(-90) has (-u1, u0, u2)
    double R1[3][3], R2[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double w = U[i][2] * V[j][2];
            R1[i][j] = (U[i][1] * V[j][0] - U[i][0] * V[j][1]) + w;
            R2[i][j] = (U[i][0] * V[j][1] - U[i][1] * V[j][0]) + w;
        }
    if (det3(R1) < 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) R1[i][j] = -R1[i][j];
    }
    if (det3(R2) < 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) R2[i][j] = -R2[i][j];
    }
    const double T12[3] = {U[0][2], U[1][2], U[2][2]};

    // eightPoint.cpp:296-337: cheirality vote over the 8 sample points, first strict maximum.  Candidates
    // 2r and 2r + 1 share the rotation and have opposite translations: one factorisation, one solve, and
    // the solution for -T is the negated one.
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; i++) {
        const double2 a = g.x[base + idx[i]], b = g.y[base + idx[i]];
        const double nb = sqrt((b.x * b.x + b.y * b.y) + 1.0);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double av[3];
            rot_apply_h(r ? R2 : R1, a.x, a.y, av);
            double l0, l1;
            triangulate_lambdas(av, b.x, b.y, nb, T12, l0, l1);
            if (l0 > 0 && l1 > 0) cnt[2 * r]++;
            if (-l0 > 0 && -l1 > 0) cnt[2 * r + 1]++;
        }
    }
    int maxIndex = 0, maxPoints = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (cnt[c] > maxPoints) {
            maxIndex = c;
            maxPoints = cnt[c];
        }
    const double sgn = (maxIndex & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) out[3 * i + j] = (maxIndex < 2) ? R1[i][j] : R2[i][j];
        out[9 + i] = sgn * T12[i];
    }
}

// ---------------------------------------------------------------------------------------
// score: one hypothesis per lane, correspondences broadcast from LDS
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_pose(const double *src, Pose &P)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) P.R[i][j] = src[3 * i + j];
        P.T[i] = src[9 + i];
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat(float v)
{
    f32x2 r;
    r.x = v;
    r.y = v;
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// SPLIT: the launch has too few (pair, 256-row) blocks to fill the device, so blockIdx.z cuts every
// pair's correspondences into segments and a block scores its rows on one segment only.  A row's
// inlier count and error sum must still accumulate in correspondence order (the sum is not
// associative), so instead of accumulating, the block writes every cell of
// errBuf[(correspondence, row)]: the error of an inlier, a negative NaN pattern otherwise, and
// ransac_sum_kernel adds the inliers' errors up in order.
// FILTER: the instance that runs the pre-test (reprojection loss, plain intrinsics).  It stages only the
// fp32 records (20 B per correspondence, SCORE_CH32 at a time, prepared once by ransac_prep_kernel) and
// fetches the fp64 record of the rare survivor from global memory; the other instances evaluate every
// cell and stage the fp64 records.
#ifdef SFM_DEBUG_TIMES // measurement build only (make EXTRA=-DSFM_DEBUG_TIMES): start, end and placement of every block
__device__ unsigned long long g_scoreTimes[3 * 8192];
extern "C" int sfm_debug_score_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scoreTimes), n * 8); }
#endif
template <int LOSS, bool SPLIT, bool FILTER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FILTER && !SPLIT ? 4 : 1))) void ransac_score_kernel(const RansacArgs g, double *__restrict__ errBuf)
{
#ifdef SFM_DEBUG_TIMES
    const unsigned long long dbgT0 = wall_clock64();
#endif
    static_assert(!FILTER || LOSS != SFM_LOSS_SAMPSON, "the pre-test bounds the reprojection error");
    constexpr int CH = FILTER ? SCORE_CH32 : SCORE_CH;
    __shared__ __attribute__((aligned(16))) double rec[FILTER ? 2 : SCORE_CH * REC_D];
    // POOL (split launches with the pre-test): a cell is written to errBuf on its own, in no particular order,
    // so the survivors of a wave go to one list that all 64 lanes drain together, 64 entries at a time,
    // whichever rows they belong to.  Rows with many inliers (every cell a survivor) then no longer hold the
    // rest of their wave up.
    constexpr bool POOL = SPLIT && FILTER;
    constexpr int POOL_CAP = 192; // fewer than 64 entries left by a drain + two correspondences x 64 rows
    __shared__ unsigned int pend[(FILTER && !POOL) ? SCORE_QCAP : 1][256]; // per-lane queues of correspondence indices (relative to the pair)
    __shared__ unsigned int poolK[POOL ? 4 : 1][POOL ? POOL_CAP : 1];   // correspondence (relative to the pair)
    __shared__ unsigned char poolL[POOL ? 4 : 1][POOL ? POOL_CAP : 1];  // lane = row of the wave
    __shared__ __attribute__((aligned(16))) float rec32[FILTER ? (SCORE_CH32 / 4 + 2) * 20 : 4]; // + 2: the pipeline reads ahead
    const int p = blockIdx.y;
    const int tid = threadIdx.x;
    const int h = g.hBegin + blockIdx.x * 256 + tid;
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    if (!g.active[p]) return; // the pair's scan stopped in an earlier phase
    const int rowEnd = pair_row_end(g, p);
    if (g.hBegin + (int)blockIdx.x * 256 >= rowEnd) return; // rows the scan cannot reach (the whole block)
    const bool active = h < rowEnd;
    int kBeg = 0, kEnd = n;
    if (SPLIT) {
        const int seg = (((n + (int)gridDim.z - 1) / (int)gridDim.z) + 3) & ~3; // whole groups of the fp32 records
        kBeg = min(n, (int)blockIdx.z * seg);
        kEnd = min(n, kBeg + seg);
        if (kBeg >= kEnd) return;
    }
    const int rows = g.hEnd - g.hBegin;
    double *errCol = SPLIT ? errBuf + (size_t)base * rows + (h - g.hBegin) : nullptr; // + k * rows
    // a slice of fewer than 256 rows leaves whole waves without a row: they only help staging
    const bool waveIdle = SPLIT && g.hBegin + (int)blockIdx.x * 256 + (tid & ~63) >= rowEnd;
    Pose P;
    load_pose(g.pose12 + ((size_t)p * g.H + (active ? h : 0)) * 12, P);
    double E[3][3];
    essential_from_pose(P, E);
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = g.K[i];
    const double thr = g.maxSqErr;
    // fp32 splats for the pre-test
    // rows 0 and 1 of E divided by fx, fy: the line of a correspondence comes out in pixel metric
    f32x2 Ef[9];
    const double skipDist = SKIP_FACTOR * sqrt(thr);
#pragma unroll
    for (int i = 0; i < 9; i++) Ef[i] = splat((float)(E[i / 3][i % 3] * (i < 3 ? g.Ki[0] * skipDist : (i < 6 ? g.Ki[1] * skipDist : 1.0))));

    int count = 0;
    double tot = 0.0;
    // Correspondences that survive the pre-test are not evaluated on the spot (one lane in a few
    // hundred needs it at any time, and the whole wave would walk through the long exact path for
    // it) but queued per lane and evaluated together: every lane then works on an entry of its own.
    // A lane drains its queue in order, so its count and error sum still accumulate in
    // correspondence order (RANSAC.cpp:207-221).
    int npend = 0;
    int kChunk = 0; // first correspondence of the chunk in LDS
    const double none = __longlong_as_double(-1ll); // negative NaN pattern: fails v >= 0
    auto record = [&](int kk, double e) {
        if (SPLIT) { // every (correspondence, row) cell is written exactly once
            if (active) errCol[(size_t)(kChunk + kk) * rows] = (e < thr) ? e : none;
        } else if (e < thr) { // RANSAC.cpp:216-220
            count++;
            tot += e;
        }
    };
    // (the pre-test instance queues indices relative to the pair, kChunk staying 0 for record(), and makes
    // the fp64 record from the observations in global memory; the others index the chunk staged in LDS)
    auto evaluate = [&](int kk) {
        if constexpr (FILTER) { // a survivor of the pre-test: its two observations from global memory
            const double2 o1 = g.x[base + kk], o2 = g.y[base + kk];
            const double xy[4] = {o1.x, o1.y, o2.x, o2.y};
            record(kk, reprojection_loss<false>(K, g.plainK, thr, P, xy));
        } else {
            record(kk, reprojection_loss<true>(K, g.plainK, thr, P, rec + kk * REC_D));
        }
    };
    auto drain = [&]() {
        for (int i = 0; __builtin_amdgcn_ballot_w64(i < npend) != 0; i++)
            if (i < npend) evaluate(pend[i][tid]);
        npend = 0;
    };
    int poolCount = 0; // wave-uniform
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long activeMask = __builtin_amdgcn_ballot_w64(active);
    auto pool_push = [&](int k, unsigned long long skipMask) __attribute__((always_inline)) {
        const unsigned long long want = ~skipMask & activeMask;
        if ((want >> lane) & 1ull) {
            const int at = poolCount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
            poolK[POOL ? wave : 0][POOL ? at : 0] = (unsigned)k;
            poolL[POOL ? wave : 0][POOL ? at : 0] = (unsigned char)lane;
        } else if (active) {
            errCol[(size_t)k * rows] = none;
        }
        poolCount += __builtin_popcountll(want);
    };
    auto pool_drain = [&](int take) { // the last `take` (<= 64) entries, one per lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < take) {
            const int at = poolCount - take + lane;
            const int k = (int)poolK[POOL ? wave : 0][POOL ? at : 0];
            const int row = (int)blockIdx.x * 256 + wave * 64 + (int)poolL[POOL ? wave : 0][POOL ? at : 0]; // relative to hBegin
            Pose Q;
            load_pose(g.pose12 + ((size_t)p * g.H + g.hBegin + row) * 12, Q);
            const double2 o1 = g.x[base + k], o2 = g.y[base + k];
            const double xy[4] = {o1.x, o1.y, o2.x, o2.y};
            const double e = reprojection_loss<false>(K, g.plainK, thr, Q, xy);
            errBuf[((size_t)base + k) * rows + row] = (e < thr) ? e : none;
        }
        poolCount -= take;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const float *rec32g = FILTER ? g.rec32 + rec32_start(base, p) : nullptr;
    // The instruction arbiter of a SIMD favours its oldest wave: of the four blocks a CU holds, which all start
    // together and have the same work, the first would finish a third ahead of the last, and the stragglers
    // cannot fill the vector ALU on their own.  Each wave therefore moves its priority round from chunk to
    // chunk, starting from its slot on the SIMD, so that every wave leads a quarter of the time and all
    // finish together.
    unsigned hwId = 0;
    if (FILTER && !SPLIT) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
    unsigned turn = hwId & 3; // wave slot on the SIMD
    for (int k0 = kBeg; k0 < kEnd; k0 += CH) {
        const int m = min(CH, kEnd - k0);
        kChunk = FILTER ? 0 : k0;
        if (FILTER && !SPLIT) {
            switch (turn++ & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        if (FILTER) { // k0 is a multiple of four: whole twenty-float groups
            const uint4 *src = reinterpret_cast<const uint4 *>(rec32g + (size_t)(k0 >> 2) * 20);
            for (int q = tid; q < ((m + 3) >> 2) * 5; q += 256) reinterpret_cast<uint4 *>(rec32)[q] = src[q];
        } else {
            const double2 *src = reinterpret_cast<const double2 *>(g.rec + (base + k0) * REC_D);
            for (int q = tid; q < m * (REC_D / 2); q += 256) reinterpret_cast<double2 *>(rec)[q] = src[q];
        }
        __syncthreads();
        if (waveIdle) {
        } else if (LOSS == SFM_LOSS_SAMPSON) {
            for (int kk = 0; kk < m; kk++) {
                const double *r = rec + kk * REC_D;
                record(kk, sampson_loss(K, E, r[0], r[1], r[2], r[3]));
            }
        } else if (!FILTER) {
            for (int kk = 0; kk < m; kk++) evaluate(kk);
        } else {
            // Pre-test, in packed fp32 (two correspondences per instruction).  Whatever depth the
            // triangulation picks, the point lies on the ray of x1, so its image in camera 2 lies
            // on the epipolar line of x1; the camera-2 reprojection error is therefore at least
            // the distance from the observation to that line.  With K = [fx 0 cx; 0 fy cy; 0 0 1]
            // and m = E x1h, that distance in pixels is |x2 m0 + y2 m1 + m2| / sqrt((m0/fx)^2 + (m1/fy)^2).
            // The kernel works with M0 = c m0/fx, M1 = c m1/fy (E's rows pre-multiplied; c = SKIP_FACTOR
            // times sqrt(maxSquaredError)) and fx x2 / c, fy y2 / c (staged per correspondence), so that
            // "distance >= c" reads num^2 >= q with num = x2h.m and q = M0^2 + M1^2.
            // One comparison decides, with the rounding of the fp32 estimate priced in: the exact
            // values satisfy |num*| >= |num| - e and sqrt(q*) <= sqrt(q) + d, where
            //   d <= 3.4e-7 c |x1h| / min(fx,fy)   (four roundings of 6e-8 on each term of M0, M1; the rows of
            //                                       E = [T]x R have norm <= 1 for the unit translation of a fitted pose)
            //   e <= 6e-7 |x2h| |x1h|              (the same on num: |x2h.m| <= |x2h| |x1h|),
            // and (a + D)^2 <= (1 + h) a^2 + (1 + 1/h) D^2 for any h > 0.  With h = 0.015 and D = d + e,
            //   num^2 >= 1.02 q + 70 D^2
            // implies |num*| >= sqrt(q*), the other 0.005 paying for the roundings of q, num^2 and the staged
            // constant.  The staged G is 70 D^2 with d, e three times the bounds above, and the factor
            // 1.02 is part of SKIP_FACTOR: a correspondence that passes lies at least
            // SKIP_FACTOR / sqrt(1.02) sqrt(maxSquaredError) from the line, cannot be an inlier of this
            // hypothesis, and its exact evaluation is skipped.  G is inf for an observation that is not an
            // ordinary pixel; NaNs fail the test.
            // Two pairs of correspondences per iteration, their records fetched from LDS up front.
            auto test_pair = [&](int kk, const f32x2 x1, const f32x2 y1, const f32x2 u2, const f32x2 v2, const f32x2 gg)
                                 __attribute__((always_inline)) {
                const f32x2 m0 = pk_fma(Ef[0], x1, pk_fma(Ef[1], y1, Ef[2]));
                const f32x2 m1 = pk_fma(Ef[3], x1, pk_fma(Ef[4], y1, Ef[5]));
                const f32x2 m2 = pk_fma(Ef[6], x1, pk_fma(Ef[7], y1, Ef[8]));
                const f32x2 num = pk_fma(u2, m0, pk_fma(v2, m1, m2));
                const f32x2 q = pk_fma(m0, m0, pk_fma(m1, m1, gg));
                const f32x2 n2 = num * num;
                // The comparisons deliver lane masks; they are combined in scalar registers, and most of the
                // time no lane of the wave has a survivor: one uniform branch around the pushes.
                constexpr int OGE = 3; // ordered >=: false on NaN
                const unsigned long long skipMask0 = __builtin_amdgcn_fcmpf(n2.x, q.x, OGE);
                const unsigned long long skipMask1 = __builtin_amdgcn_fcmpf(n2.y, q.y, OGE);
                if constexpr (POOL) {
                    if (kk < m) pool_push(k0 + kk, skipMask0);
                    if (kk + 1 < m) pool_push(k0 + kk + 1, skipMask1);
                    while (poolCount >= 64) pool_drain(64);
                    return;
                }
                if (!SPLIT && (skipMask0 & skipMask1) == __builtin_amdgcn_ballot_w64(true)) return;
                const bool skip0 = (skipMask0 >> (tid & 63)) & 1ull, skip1 = (skipMask1 >> (tid & 63)) & 1ull;
                if (kk < m) {
                    if (!skip0) pend[npend++][tid] = (unsigned)(k0 + kk);
                    else if (SPLIT && active) errCol[(size_t)(k0 + kk) * rows] = none;
                }
                if (kk + 1 < m) {
                    if (!skip1) pend[npend++][tid] = (unsigned)(k0 + kk + 1);
                    else if (SPLIT && active) errCol[(size_t)(k0 + kk + 1) * rows] = none;
                }
                // room for the next two: drain as soon as any lane of the wave holds more than six
                if (__builtin_amdgcn_ballot_w64(npend > SCORE_QCAP - 2) != 0) drain();
            };
            // Software pipeline over two register sets: the next four records are on their way from LDS
            // while the current four are tested.  (Groups past the chunk's last hold stale or padded values:
            // their results are discarded by the kk < m tests.)
            auto load4 = [&](int kk, float4(&r)[5]) __attribute__((always_inline)) {
                const float4 *rp = reinterpret_cast<const float4 *>(rec32 + (kk >> 2) * 20);
#pragma unroll
                for (int i = 0; i < 5; i++) r[i] = rp[i];
            };
            auto test4 = [&](int kk, const float4(&r)[5]) __attribute__((always_inline)) {
                test_pair(kk, f32x2{r[0].x, r[0].y}, f32x2{r[1].x, r[1].y}, f32x2{r[2].x, r[2].y}, f32x2{r[3].x, r[3].y}, f32x2{r[4].x, r[4].y});
                test_pair(kk + 2, f32x2{r[0].z, r[0].w}, f32x2{r[1].z, r[1].w}, f32x2{r[2].z, r[2].w}, f32x2{r[3].z, r[3].w}, f32x2{r[4].z, r[4].w});
            };
            float4 ra[5], rb[5];
            load4(0, ra);
            for (int kk = 0; kk < m; kk += 8) {
                load4(kk + 4, rb);
                test4(kk, ra);
                load4(kk + 8, ra);
                test4(kk + 4, rb);
            }
        }
        __syncthreads();
    }
    if (POOL) {
        if (!waveIdle && poolCount > 0) pool_drain(poolCount);
    } else if (FILTER) {
        drain(); // what the last pushes left
    }
    if (active && !SPLIT) {
        g.counts[(size_t)p * g.H + h] = count;
        g.toterr[(size_t)p * g.H + h] = tot;
    }
#ifdef SFM_DEBUG_TIMES
    if (tid == 0 && !SPLIT) {
        const int b = (blockIdx.y * gridDim.x + blockIdx.x) & 8191;
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_scoreTimes[3 * b] = dbgT0;
        g_scoreTimes[3 * b + 1] = wall_clock64();
        g_scoreTimes[3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

// ---------------------------------------------------------------------------------------
// scoring with the pre-test on the matrix cores
// ---------------------------------------------------------------------------------------
// The pre-test of ransac_score_kernel compares two forms that are each bilinear in a hypothesis and a
// correspondence:
//   num = x2h^T E x1h                      = sum over nine terms of  (x2h_i x1h_j) . E_ij
//   q   = c^2 ((m0/fx)^2 + (m1/fy)^2)      = sum over six terms of   (monomial of x1h) . (entry of c^2 S),
//         m = E x1h, S = E0^T E0 / fx^2 + E1^T E1 / fy^2 (E0, E1 the first two rows of E),
// and skips the cell when num^2 >= q + G (G the margin of the correspondence).  Here both run as
// v_mfma_f32_16x16x32_f16 on tiles of 16 hypotheses x 16 correspondences.  fp16 keeps 11 bits, so every operand
// v is carried as hi + lo (hi = fp16(v), lo = fp16(v - hi): 22 bits) and a product as hi.hi + hi.lo + lo.hi --
// three of the 32 K-slots per term: 27 for num; 18 for q plus two for G, which rides in q's sum (correspondence
// side G hi/lo, hypothesis side a constant), so that one fused multiply-add per cell, d = num^2 - q', and the
// sign of the smallest d of a lane's four cells decide.  Operands are scaled by powers of two into fp16's range
// (MF_SA, 2^mfKe on the two sides of num; MF_SB and 2^(P-4) for q, P = 2 (10 + mfKe); 2^(P-6) and 64 for G).
//
// Margin.  With u = 2^-22 per operand (hi + lo), one more for the dropped lo.lo product, 27 additions in the
// accumulator priced at 2^-22 each (the rounding of the matrix pipe's adder is not documented: this is four
// times round-to-nearest), 4 . 2^-24 for the fp32 inputs of the products, and sum |a_k e_k| <= |a| |e| =
// sqrt(2) |x1h| |x2h| (E = [T]x R has singular values 1, 1, 0):  |num - num*| <= 1.1e-5 |x1h| |x2h|.  Should
// the pipe flush fp16 denormals (operands below 2^-14 after scaling), each operand is off by up to 2^-14 in
// scaled units: 2 . 2^-14 (sum |e_k| / MF_SA + sum |a_k| / 2^mfKe) <= 5.2e-7 + 3.7e-4 / 2^mfKe |x1h| |x2h|.
// D is the sum of the three; as in ransac_score_kernel, num^2 >= 1.02 q* + 70 D^2 then implies that the exact
// distance is at least the skip distance, and the same accounting for q (21 additions, |monomials| <= |x1h|^2,
// |entries| <= sqrt(2) c^2 (1/fx^2 + 1/fy^2)) adds 1.1 eps_q to the margin.  ransac_prep_kernel stores that G
// when this kernel may run (it exceeds the packed test's margin, which stays valid with it).
//
// A wave holds the operands of 16 T hypotheses in registers and walks the pair's correspondences 16 at a
// time.  The correspondence side is made once per call by ransac_mfma_ops_kernel (PRE; 128 B per correspondence,
// read by the waves straight from global memory, no barrier in the loop: a wave that stops to evaluate its
// survivors holds nobody up -- 17.8 -> 15.3 ms per 1024 pairs of the stress shape against staging through
// LDS with a barrier per chunk) or, when there is no room for that table and in the entry point that allocates
// nothing, by the block from the fp32 records while a chunk of 128 is staged in LDS.  Survivors -- (row, correspondence)
// cells, < 0.1 % -- go to one queue per wave and are evaluated exactly 64 at a time, whichever rows they
// belong to; the queue is first-in first-out and a row's cells enter it in correspondence order, so adding
// the inliers' errors to their rows in queue order reproduces the sequential sum (RANSAC.cpp:207-221).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#define MF_CH 128 // correspondences staged per step (two threads make the operands of one: 256 threads, 2 sides)
#define MF_SA 1024.0f
#define MF_SB 16.0f

__device__ __forceinline__ void split_half(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// The margin of a correspondence for the matrix-core test (see above); inf when its operands would leave
// fp16's range.  x1n2 = x1h.x1h, x2n2 = x2h.x2h, amax = max |x2h_i x1h_j|.
__device__ __forceinline__ double mfma_margin(const RansacArgs &g, double x1n2, double x2n2, double amax)
{
    if (!(amax <= 32.0)) return INFINITY;
    const double c = SKIP_FACTOR * sqrt(g.maxSqErr);
    const double se = ldexp(1.0, g.mfKe), sQ = ldexp(1.0, 2 * (10 + g.mfKe) - 4);
    const double D = (1.1e-5 + 3.7e-4 / se) * sqrt(x1n2 * x2n2) + 5.2e-7;
    const double cfp = 1.4143 * (c * c) * (g.mfI[0] * g.mfI[0] + g.mfI[1] * g.mfI[1]);
    const double epsq = (1.0e-5 * x1n2 + 1.9e-5) * cfp + 3.0e-4 * x1n2 / sQ;
    return (70.0 * (D * D) + 1.1 * epsq) * 1.00001;
}

// Hypothesis side of both forms, 64 halves: [0, 32) for num, [32, 64) for q.  A row that is not in play
// (valid = false) gets zeros: d = 0 for every correspondence, never a survivor.  A pose that is not a unit
// translation and a rotation (entries of E beyond 1, or not finite) cannot be bounded: its row gets zeros for
// num and fp16's largest value on the constant monomial, so that every cell of it survives (d < 0).  No operand
// is ever inf or NaN, so no d is NaN and "d < 0" is the whole test.
__device__ __forceinline__ void mfma_hyp_operands(const RansacArgs &g, const Pose &P, bool valid, _Float16 *out)
{
    double E[3][3];
    essential_from_pose(P, E);
    bool sane = true;
#pragma unroll
    for (int i = 0; i < 9; i++) sane = sane && (fabs(E[i / 3][i % 3]) <= 1.001);
    sane = sane && ((P.T[0] * P.T[0] + P.T[1] * P.T[1]) + P.T[2] * P.T[2] <= 1.0005); // (the Sampson bound needs |E| <= 1)
    const float se = ldexpf(1.0f, g.mfKe);
    const double c = SKIP_FACTOR * sqrt(g.maxSqErr);
    const double sQ = ldexp(c * c, 2 * (10 + g.mfKe) - 4);
    const double ix2 = g.mfI[0] * g.mfI[0], iy2 = g.mfI[1] * g.mfI[1];
    const int ja[6] = {0, 1, 2, 0, 0, 1}, jb[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int k = 0; k < 64; k++) out[k] = (_Float16)0.0f;
    if (!valid) return;
    out[32 + 18] = (_Float16)-64.0f; // (the q side is negated: the matrix cores deliver -q', and d = num^2 + (-q'))
    out[32 + 19] = (_Float16)-64.0f;
    if (!sane) {
        out[32 + 6] = (_Float16)-65504.0f;
        return;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        _Float16 hi, lo;
        split_half((float)E[k / 3][k % 3] * se, hi, lo);
        out[3 * k] = hi;
        out[3 * k + 1] = lo;
        out[3 * k + 2] = hi;
    }
#pragma unroll
    for (int m = 0; m < 6; m++) {
        const double S = (E[0][ja[m]] * E[0][jb[m]]) * ix2 + (E[1][ja[m]] * E[1][jb[m]]) * iy2;
        _Float16 hi, lo;
        split_half((float)(-(m < 3 ? S : 2.0 * S) * sQ), hi, lo);
        out[32 + 3 * m] = hi;
        out[32 + 3 * m + 1] = lo;
        out[32 + 3 * m + 2] = hi;
    }
}

// Correspondence side, from the fp32 record (x1, y1, fx x2 / c, fy y2 / c, G): 32 halves for num (side 0) or
// 32 for q (side 1).  G = inf (not an ordinary pixel, or out of fp16's range): num's operands are zero and
// q' is huge -- the cell survives whatever the hypothesis in play.
__device__ __forceinline__ void mfma_corr_operands(const RansacArgs &g, int side, float x1, float y1, float u2, float v2,
                                                   float G, _Float16 *out)
{
    const bool tameIn = G < INFINITY;
#pragma unroll
    for (int k = 0; k < 32; k++) out[k] = (_Float16)0.0f;
    if (side == 0) {
        const bool tame = tameIn && G * ldexpf(1.0f, 2 * (10 + g.mfKe) - 6) < 65000.0f;
        const float c = (float)(SKIP_FACTOR * sqrt(g.maxSqErr));
        const float x2h[3] = {u2 * (c * (float)g.Ki[0] * MF_SA), v2 * (c * (float)g.Ki[1] * MF_SA), MF_SA};
        const float x1h[3] = {x1, y1, 1.0f};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            _Float16 hi, lo;
            split_half(tame ? x2h[k / 3] * x1h[k % 3] : 0.0f, hi, lo);
            out[3 * k] = hi;
            out[3 * k + 1] = hi;
            out[3 * k + 2] = lo;
        }
    } else {
        const float mono[6] = {x1 * x1, y1 * y1, 1.0f, x1 * y1, x1, y1};
        _Float16 gh, gl;
        const float Gs = G * ldexpf(1.0f, 2 * (10 + g.mfKe) - 6);
        const bool tame = tameIn && Gs < 65000.0f;
        split_half(tame ? Gs : 0.0f, gh, gl);
#pragma unroll
        for (int m = 0; m < 6; m++) {
            _Float16 hi, lo;
            split_half(tame ? mono[m] * MF_SB : 0.0f, hi, lo);
            out[3 * m] = hi;
            out[3 * m + 1] = hi;
            out[3 * m + 2] = lo;
        }
        out[18] = tame ? gh : (_Float16)65504.0f; // (a margin beyond fp16's range saturates the same way)
        out[19] = tame ? gl : (_Float16)0.0f;
    }
}

// The correspondence side made once per call instead of once per block of rows.  A step of 16 correspondences
// takes 2 KB: the 16 x 32 halves of num in fragment order ([K-octet][correspondence][8 halves]: 64 lanes read
// 1 KB without a gap), then those of q.  First step of pair p (its correspondences start at row `base`):
// regions of consecutive pairs do not overlap, each takes ceil(n / 16) steps.
__host__ __device__ inline size_t mfma_ops_start(int64_t base, int p) { return (size_t)((base + 15 * (int64_t)p + 15) / 16); }

// One thread per (correspondence slot, side); slots past the pair's last correspondence get zeros (d = 0).
__global__ __launch_bounds__(256) void ransac_mfma_ops_kernel(const RansacArgs g)
{
    const int p = blockIdx.y;
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    const int slots = ((n + 15) >> 4) << 4;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int kk = t >> 1, side = t & 1;
    if (kk >= slots) return;
    _Float16 ob[32];
    if (kk < n) {
        const float *grp = g.rec32 + rec32_start(base, p) + (size_t)(kk >> 2) * 20 + (kk & 3);
        mfma_corr_operands(g, side, grp[0], grp[4], grp[8], grp[12], grp[16], ob);
    } else {
#pragma unroll
        for (int i = 0; i < 32; i++) ob[i] = (_Float16)0.0f;
    }
    uint4 *dst = g.mfOps + (mfma_ops_start(base, p) + (size_t)(kk >> 4)) * 128 + side * 64 + (kk & 15);
#pragma unroll
    for (int v = 0; v < 4; v++) {
        half8 a;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = ob[8 * v + i];
        *reinterpret_cast<half8 *>(dst + v * 16) = a;
    }
}

#ifdef SFM_DEBUG_WAVES // measurement build only: cells evaluated, waves and their running times (100 MHz ticks), per launch
__device__ unsigned long long g_waveStat[8];
extern "C" int sfm_debug_wave_stats(unsigned long long *out)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_waveStat), sizeof(unsigned long long) * 8);
    unsigned long long zero[8] = {0};
    if (rc == 0) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_waveStat), zero, sizeof zero);
    return rc;
}
#endif
// DEFER: rows with many inliers hold a wave up -- their cells are evaluated 64 at a time by a wave whose other
// rows have long been done, and at the end of a launch a few such waves keep the device waiting.  Such rows are
// dealt with beforehand (ransac_heavy_find_kernel and the kernels after it) and marked (rowMark):
// ROW_HANDED rows are evaluated cell by cell after this kernel (ransac_heavy_eval_kernel, ransac_heavy_sum_kernel),
// ROW_PASSED rows cannot matter to the scan and stay at count 0.  Neither gets operands here, so none of their
// cells passes the test; the sums of ROW_HANDED rows are not written.
#define DEFER_CHECK_STEPS 64 // steps of 16 correspondences over which ransac_heavy_find_kernel counts
#define DEFER_SHARE_FIRST 8    // a row is a candidate when more than one in this many of those cells pass the test: first slice,
#define DEFER_SHARE_LATER 32 // and later slices when a bound can be had (most candidates are then passed over)
#define ROW_HANDED (-1)
#define ROW_PASSED (-2)
// Pairs whose best model so far holds few of their correspondences are not searched for such rows after the first
// slice (no such row has shown yet, and the search would cost more than it saves).
__device__ __forceinline__ bool defer_applies(const RansacArgs &g, int p, int n)
{
    if (g.heavyMax <= 0 || n < 16 * 4 * DEFER_CHECK_STEPS) return false;
    return g.hBegin == 0 || (long long)g.bestCount[p] * DEFER_SHARE_FIRST >= n;
}
template <int T, bool PRE, int LOSS = SFM_LOSS_REPROJECTION, bool DEFER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void ransac_score_mfma_kernel(const RansacArgs g)
{
    // A wave takes 16 T rows (T tiles of the matrix cores) through all correspondences of its pair.  T = 4 reads
    // the correspondence side once per 64 rows; T = 1 and 2 read it four times and twice as often but make four
    // times and twice as many waves of a slice of rows, each with a quarter and a half of the work: they are
    // for slices that would leave most of the device idle otherwise, where a wave's running time decides.
    constexpr int ROWS_W = 16 * T, ROWS_B = 4 * ROWS_W;
    static_assert(T == 1 || T == 2 || T == 4, "lane l keeps the sums of row l of its wave; eight tiles per wave (three waves per SIMD) measured slower");
    constexpr int OWN = 1; // rows whose sums a lane keeps
    // One drain site per step: the queue must take what a step can add (every cell of the wave's T tiles) on
    // top of what a drain leaves (< 64).
    constexpr int QCAP = 64 + T * 256;
    constexpr int OPB_HALVES = 2 * MF_CH * 32; // num side, then q side
    constexpr int SLICE_HALVES = 4 * 64 * 64;  // the hypothesis operands on their way into fragments (64 x 64 halves per wave)
    // per wave, behind the queues: the errors of the 64 entries of a drain, and per row of the wave the lanes
    // whose entries are inliers of that row (one bit per lane)
    constexpr int ACC_BYTES = 64 * 8 + 64 * 8;
    constexpr int FRONT = (PRE ? 0 : OPB_HALVES * 2); // the staged correspondence side, when the blocks make it
    constexpr int ACC_AT0 = FRONT + 4 * QCAP * 4;
    constexpr int ACC_AT = ACC_AT0 >= SLICE_HALVES * 2 ? ACC_AT0 : SLICE_HALVES * 2;
    static_assert(ACC_AT >= SLICE_HALVES * 2 && ACC_AT % 8 == 0, "the row masks are cleared before the set-up uses its slices");
    constexpr int SMEM_BYTES = ACC_AT + 4 * ACC_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    _Float16 *opB = reinterpret_cast<_Float16 *>(smem);
    unsigned int *pool = reinterpret_cast<unsigned int *>(smem + FRONT) + (threadIdx.x >> 6) * QCAP;
    double *accE = reinterpret_cast<double *>(smem + ACC_AT + (threadIdx.x >> 6) * ACC_BYTES);
    unsigned long long *accM = reinterpret_cast<unsigned long long *>(accE + 64);
    accM[threadIdx.x & 63] = 0ull;
#ifdef SFM_DEBUG_WAVES
    const unsigned long long dbgT0 = wall_clock64();
    unsigned long long dbgEvals = 0, dbgDrains = 0, dbgSecond = 0;
    (void)dbgSecond;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    // Blocks go to the eight XCDs in turn.  All row blocks of a pair run on one of them (same residue mod 8),
    // so that its L2 serves the pair's correspondences to all of them; the pairs past the last full group of
    // eight take the plain order.
    int p, rowBlock;
    {
        const int gx = (g.hEnd - g.hBegin + ROWS_B - 1) / ROWS_B, L = (int)blockIdx.x;
        const int full = (g.nPairs >> 3) << 3;
        if (L < full * gx) {
            const int grp = L / (8 * gx), l = L - grp * 8 * gx;
            p = grp * 8 + (l & 7);
            rowBlock = l >> 3;
        } else {
            const int l = L - full * gx;
            p = full + l / gx;
            rowBlock = l - (l / gx) * gx;
        }
    }
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    if (!g.active[p]) return;
    const int rowEnd = pair_row_end(g, p);
    if (g.hBegin + rowBlock * ROWS_B >= rowEnd) return; // rows the scan cannot reach (the whole block)
    const int rowW0 = g.hBegin + rowBlock * ROWS_B + wave * ROWS_W; // the wave's first row
    const int nRowsW = max(0, min(ROWS_W, rowEnd - rowW0));
    const double *poseP = g.pose12 + (size_t)p * g.H * 12;
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = g.K[i];
    const double thr = g.maxSqErr;

    // (DEFER) the row whose sums this lane keeps (row `lane` of the wave) was handed over, or is passed over
    bool ownHanded = false, ownPassed = false;
    if constexpr (DEFER) {
        if (defer_applies(g, p, n) && lane < nRowsW) {
            const int mark = g.rowMark[(size_t)p * g.H + rowW0 + lane];
            ownHanded = mark == ROW_HANDED;
            ownPassed = mark == ROW_PASSED;
        }
    }
    // hypothesis operands: every lane makes those of one row, the wave's slice of LDS turns them into fragments
    half8 An[T], Aq[T];
    {
        _Float16 *slice = opB + wave * (64 * 64);
        const bool valid = lane < nRowsW && !ownHanded && !ownPassed;
        Pose P;
        load_pose(poseP + (size_t)(valid ? rowW0 + lane : g.hBegin) * 12, P);
        _Float16 ops[64];
        mfma_hyp_operands(g, P, valid, ops);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 8; v++) {
            half8 w;
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = ops[8 * v + i];
            *reinterpret_cast<half8 *>(slice + lane * 64 + 8 * v) = w;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < T; tt++) {
            An[tt] = *reinterpret_cast<const half8 *>(slice + (tt * 16 + col) * 64 + quad * 8);
            Aq[tt] = *reinterpret_cast<const half8 *>(slice + (tt * 16 + col) * 64 + 32 + quad * 8);
        }
        __syncthreads();
    }

    int count[OWN];
    double tot[OWN];
#pragma unroll
    for (int i = 0; i < OWN; i++) {
        count[i] = 0;
        tot[i] = 0.0;
    }
    int qHead = 0, qCount = 0; // wave-uniform
    auto drain = [&](int take) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int hl = 0;
        double e = 0.0;
        bool inl = false;
#ifdef SFM_DEBUG_WAVES
        dbgEvals += take;
        dbgDrains++;
#endif
        if (lane < take) {
            int at = qHead + lane;
            if (at >= QCAP) at -= QCAP;
            const unsigned ent = pool[at];
            const int k = (int)(ent & 0xFFFFFFu);
            hl = (int)(ent >> 24);
            Pose Q;
            load_pose(poseP + (size_t)(rowW0 + hl) * 12, Q);
            const double2 o1 = g.x[base + k], o2 = g.y[base + k];
            const double xy[4] = {o1.x, o1.y, o2.x, o2.y};
            if constexpr (LOSS == SFM_LOSS_SAMPSON) {
                double E[3][3];
                essential_from_pose(Q, E);
                e = sampson_loss(K, E, xy[0], xy[1], xy[2], xy[3]);
            } else {
                e = reprojection_loss<false>(K, g.plainK, thr, Q, xy);
            }
            inl = e < thr; // RANSAC.cpp:216
        }
        qHead += take;
        if (qHead >= QCAP) qHead -= QCAP;
        qCount -= take;
        // inliers in queue order, each added to its row (lane hl % 64 keeps the sums of rows hl, hl + 64)
        // Inliers are added to their rows in queue order (lane order here).  Rows are independent of each
        // other: every entry marks its lane in its row's mask, and the lane that keeps a row's sum walks that
        // mask from the lowest lane up, so the rows' chains of additions run side by side (with most cells
        // inliers a drain holds 16 entries of each of 4 rows: 16 rounds instead of 64).
        if (__builtin_amdgcn_ballot_w64(inl) != 0) {
            accE[lane] = e;
            if (inl) atomicOr(reinterpret_cast<unsigned int *>(accM) + 2 * hl + (lane >> 5), 1u << (lane & 31));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < OWN; i++) {
                unsigned long long mk = accM[i * 64 + lane];
                if (mk) accM[i * 64 + lane] = 0ull;
                // Four entries per round: their errors are fetched from LDS together and then added in order (a
                // fetch per addition left the chain waiting on LDS for every entry).  Where fewer than four are
                // left the chain adds -0.0, which changes no sum.
                while (__builtin_amdgcn_ballot_w64(mk != 0ull) != 0) {
                    double ev[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool has = mk != 0ull;
                        const int l = has ? __builtin_ctzll(mk) : 0;
                        mk &= mk - 1ull;
                        const double v = accE[l];
                        ev[j] = has ? v : -0.0;
                        count[i] += has ? 1 : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) tot[i] += ev[j];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    unsigned hwId;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
    unsigned turn = hwId & 3; // the waves of a SIMD take turns at the top priority (see ransac_score_kernel)
    const floatx4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // One step: the wave's 16 T rows against 16 correspondences (operands at opS: 1 KB for num, 1 KB for q;
    // kStep = index of the first of them in the pair).
    auto test_step = [&](const half8 bn, const half8 bq, int kStep) __attribute__((always_inline)) {
        const int k = kStep + col;
        floatx4 nu[T], qq[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            nu[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(An[t], bn, zero, 0, 0, 0);
            qq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[t], bq, zero, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            // d = num^2 - q' per cell; no operand is inf or NaN, so neither is d
            const f32x2 dA = __builtin_elementwise_fma(f32x2{nu[t][0], nu[t][1]}, f32x2{nu[t][0], nu[t][1]}, f32x2{qq[t][0], qq[t][1]});
            const f32x2 dB = __builtin_elementwise_fma(f32x2{nu[t][2], nu[t][3]}, f32x2{nu[t][2], nu[t][3]}, f32x2{qq[t][2], qq[t][3]});
            const float d[4] = {dA.x, dA.y, dB.x, dB.y};
            const float least = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
            if (__builtin_amdgcn_fcmpf(least, 0.0f, 4) == 0) continue; // 4: ordered <
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool hit = d[r] < 0.0f;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask == 0) continue;
                if (hit) {
                    int at = qHead + qCount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (at >= QCAP) at -= QCAP;
                    pool[at] = (unsigned)k | ((unsigned)(t * 16 + quad * 4 + r) << 24);
                }
                qCount += __builtin_popcountll(mask);
            }
        }
        while (qCount >= 64) drain(64);
    };
    auto take_turn = [&]() __attribute__((always_inline)) {
        switch (turn++ & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    if constexpr (PRE) {
        // The prepared operands are in fragment order, so a wave reads its two fragments of a step straight from
        // global memory (1 KB each, without a gap; the 64 waves that walk one pair share them through the caches),
        // one step ahead.  No staging and no barrier: the waves of a block run on their own from here on.
        const int stepsPair = (n + 15) >> 4;
        if (nRowsW > 0 && stepsPair > 0) {
            const half8 *opsL = reinterpret_cast<const half8 *>(g.mfOps + mfma_ops_start(base, p) * 128) + lane;
            half8 bn = opsL[0], bq = opsL[64];
            for (int s = 0; s < stepsPair; s++) {
                const half8 *opN = opsL + (size_t)min(s + 1, stepsPair - 1) * 128;
                const half8 bn2 = opN[0], bq2 = opN[64];
                if ((s & 7) == 0) take_turn();
                test_step(bn, bq, 16 * s);
                bn = bn2;
                bq = bq2;
            }
        }
    } else {
        const float *rec32g = g.rec32 + rec32_start(base, p);
        float recNext[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        auto fetch = [&](int kFirst) __attribute__((always_inline)) {
            const int k = kFirst + (tid & (MF_CH - 1));
            if (k < n) {
                const float *grp = rec32g + (size_t)(k >> 2) * 20 + (k & 3);
#pragma unroll
                for (int f = 0; f < 5; f++) recNext[f] = grp[4 * f];
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < n; k0 += MF_CH) {
            const int m = min(MF_CH, n - k0);
            take_turn();
            { // stage: two threads per correspondence, one side each (zeros past the pair's last: d = 0)
                const int kk = tid & (MF_CH - 1), side = tid >> 7; // (a wave makes one side: no divergence)
                _Float16 ob[32];
                if (kk < m) {
                    mfma_corr_operands(g, side, recNext[0], recNext[1], recNext[2], recNext[3], recNext[4], ob);
                } else {
#pragma unroll
                    for (int i = 0; i < 32; i++) ob[i] = (_Float16)0.0f;
                }
                // fragment order: [step of 16][side][K-octet][correspondence of the step][8 halves] -- the 64 lanes
                // of a fragment read (and the 64 of a write) cover 1 KB without a gap
                _Float16 *dst = opB + (kk >> 4) * 1024 + side * 512 + (kk & 15) * 8;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    half8 a;
#pragma unroll
                    for (int i = 0; i < 8; i++) a[i] = ob[8 * v + i];
                    *reinterpret_cast<half8 *>(dst + v * 128) = a;
                }
            }
            fetch(k0 + MF_CH); // the next chunk's records travel while this one is tested
            __syncthreads();
            if (nRowsW > 0) {
                const int steps = (m + 15) >> 4;
                const _Float16 *opC = opB + lane * 8;
                half8 bn = *reinterpret_cast<const half8 *>(opC), bq = *reinterpret_cast<const half8 *>(opC + 512);
                for (int s = 0; s < steps; s++) {
                    const _Float16 *opN = opC + min(s + 1, steps - 1) * 1024;
                    const half8 bn2 = *reinterpret_cast<const half8 *>(opN), bq2 = *reinterpret_cast<const half8 *>(opN + 512);
                    test_step(bn, bq, k0 + s * 16);
                    bn = bn2;
                    bq = bq2;
                }
            }
            __syncthreads();
        }
    }
    if (qCount > 0) drain(qCount);
#ifdef SFM_DEBUG_WAVES
    if (lane == 0 && nRowsW > 0) {
        const unsigned long long dt = wall_clock64() - dbgT0;
        atomicAdd(&g_waveStat[0], dbgEvals);
        atomicAdd(&g_waveStat[1], dt);
        atomicMax(&g_waveStat[2], dt);
        atomicAdd(&g_waveStat[3], 1ull);
        atomicAdd(&g_waveStat[4], dbgDrains);
        atomicAdd(&g_waveStat[5], dbgSecond);
        atomicMax(&g_waveStat[6], dbgEvals);
    }
#endif
#pragma unroll
    for (int i = 0; i < OWN; i++) {
        const int hl = i * 64 + lane;
        if (hl < nRowsW && !ownHanded) {
            g.counts[(size_t)p * g.H + rowW0 + hl] = count[i];
            g.toterr[(size_t)p * g.H + rowW0 + hl] = tot[i];
        }
    }
}

// Finds the candidates, before ransac_score_mfma_kernel<.., DEFER> runs with the same grid: a wave runs the
// matrix-core test of its 64 rows over the pair's first 16 DEFER_CHECK_STEPS correspondences and only counts, per
// row, the cells that pass.  A row with more than one in DEFER_SHARE_* enters the candidate list; every row of
// the slice is marked 0 (ransac_heavy_assign_kernel marks the candidates it settles).
__global__ __launch_bounds__(256) void ransac_heavy_find_kernel(const RansacArgs g)
{
    constexpr int T = 4, ROWS_W = 64, ROWS_B = 256;
    __shared__ __attribute__((aligned(16))) _Float16 opB[4 * 64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    int p, rowBlock;
    { // the same placement as the scoring kernel's
        const int gx = (g.hEnd - g.hBegin + ROWS_B - 1) / ROWS_B, L = (int)blockIdx.x;
        const int full = (g.nPairs >> 3) << 3;
        if (L < full * gx) {
            const int grp = L / (8 * gx), l = L - grp * 8 * gx;
            p = grp * 8 + (l & 7);
            rowBlock = l >> 3;
        } else {
            const int l = L - full * gx;
            p = full + l / gx;
            rowBlock = l - (l / gx) * gx;
        }
    }
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    if (!g.active[p] || !defer_applies(g, p, n)) return;
    const int rowEnd = pair_row_end(g, p);
    if (g.hBegin + rowBlock * ROWS_B >= rowEnd) return;
    const int rowW0 = g.hBegin + rowBlock * ROWS_B + wave * ROWS_W;
    const int nRowsW = max(0, min(ROWS_W, rowEnd - rowW0));
    half8 An[T], Aq[T];
    {
        _Float16 *slice = opB + wave * (64 * 64);
        const bool valid = lane < nRowsW;
        Pose P;
        load_pose(g.pose12 + ((size_t)p * g.H + (valid ? rowW0 + lane : g.hBegin)) * 12, P);
        _Float16 ops[64];
        mfma_hyp_operands(g, P, valid, ops);
#pragma unroll
        for (int v = 0; v < 8; v++) {
            half8 w;
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = ops[8 * v + i];
            *reinterpret_cast<half8 *>(slice + lane * 64 + 8 * v) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int tt = 0; tt < T; tt++) {
            An[tt] = *reinterpret_cast<const half8 *>(slice + (tt * 16 + col) * 64 + quad * 8);
            Aq[tt] = *reinterpret_cast<const half8 *>(slice + (tt * 16 + col) * 64 + 32 + quad * 8);
        }
    }
    if (nRowsW <= 0) return;
    int ub[T][4]; // cells of the lane's column that passed, per row 16 t + 4 quad + r
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) ub[t][r] = 0;
    const floatx4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const half8 *opsL = reinterpret_cast<const half8 *>(g.mfOps + mfma_ops_start(base, p) * 128) + lane;
    half8 bn = opsL[0], bq = opsL[64];
    for (int s = 0; s < DEFER_CHECK_STEPS; s++) {
        const half8 *opN = opsL + (size_t)min(s + 1, DEFER_CHECK_STEPS - 1) * 128;
        const half8 bn2 = opN[0], bq2 = opN[64];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const floatx4 nu = __builtin_amdgcn_mfma_f32_16x16x32_f16(An[t], bn, zero, 0, 0, 0);
            const floatx4 qq = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[t], bq, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) ub[t][r] += __builtin_fmaf(nu[r], nu[r], qq[r]) < 0.0f ? 1 : 0;
        }
        bn = bn2;
        bq = bq2;
    }
    // Lane (quad, col) speaks for row 16 (col >> 2) + 4 quad + (col & 3) of the wave: its cells are spread over the
    // 16 lanes of the quad, counted under index [col >> 2][col & 3].
    int mine = 0;
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int c = ub[t][r];
            c += __shfl_xor(c, 1);
            c += __shfl_xor(c, 2);
            c += __shfl_xor(c, 4);
            c += __shfl_xor(c, 8);
            if (col == 4 * t + r) mine = c;
        }
    const int spokenRow = (col >> 2) * 16 + quad * 4 + (col & 3);
    const bool inPlay = spokenRow < nRowsW;
    const int share = g.heavyBound ? DEFER_SHARE_LATER : DEFER_SHARE_FIRST;
    const bool heavy = inPlay && mine * share >= 16 * DEFER_CHECK_STEPS;
    const unsigned long long want = __builtin_amdgcn_ballot_w64(heavy);
    if (want != 0ull) {
        const int first = __builtin_ctzll(want);
        unsigned at = 0;
        if (lane == first) at = atomicAdd(&g.heavyCount[0], (unsigned)__builtin_popcountll(want));
        at = __builtin_amdgcn_readlane(at, first);
        const unsigned mineAt = at + __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        if (heavy && mineAt < (unsigned)g.candMax) { // (a candidate beyond the list's end stays with the scoring kernel)
            g.candRow[mineAt] = ((long long)p << 32) | (long long)(unsigned)(rowW0 + spokenRow);
            g.candUb[mineAt] = 0u;
        }
    }
    if (inPlay) g.rowMark[(size_t)p * g.H + rowW0 + spokenRow] = 0;
}

// The cells of a candidate row that pass the packed fp32 test of ransac_score_kernel (a cell that fails it lies
// too far from its epipolar line to be an inlier): their number bounds the row's count from above.  A block
// takes BOUND_GROUP consecutive candidates in turn; those of one pair (the waves that find them list the rows of
// a pair together) share the reads of the pair's records, which are what the kernel waits for.
#define BOUND_GROUP 4
__global__ __launch_bounds__(256) void ransac_heavy_bound_kernel(const RansacArgs g)
{
    __shared__ unsigned int part[4][BOUND_GROUP];
    const int cands = (int)min(g.heavyCount[0], (unsigned)g.candMax);
    const double skipDist = SKIP_FACTOR * sqrt(g.maxSqErr);
    for (int c0 = blockIdx.x * BOUND_GROUP; c0 < cands; c0 += gridDim.x * BOUND_GROUP) {
        int pj[BOUND_GROUP];
        float Ef[BOUND_GROUP][9];
#pragma unroll
        for (int j = 0; j < BOUND_GROUP; j++) {
            pj[j] = -1;
            if (c0 + j < cands) {
                const long long hr = g.candRow[c0 + j];
                pj[j] = (int)(hr >> 32);
                Pose Q;
                load_pose(g.pose12 + ((size_t)pj[j] * g.H + (int)(hr & 0xFFFFFFFFll)) * 12, Q);
                double E[3][3];
                essential_from_pose(Q, E);
#pragma unroll
                for (int i = 0; i < 9; i++)
                    Ef[j][i] = (float)(E[i / 3][i % 3] * (i < 3 ? g.Ki[0] * skipDist : (i < 6 ? g.Ki[1] * skipDist : 1.0)));
            }
        }
        unsigned todo = 0;
#pragma unroll
        for (int j = 0; j < BOUND_GROUP; j++) todo |= (pj[j] >= 0 ? 1u : 0u) << j;
        while (todo) { // the candidates of the first open one's pair together
            int p = -1;
            unsigned grp = 0;
#pragma unroll
            for (int j = 0; j < BOUND_GROUP; j++) {
                if (((todo >> j) & 1u) && p < 0) p = pj[j];
                if (((todo >> j) & 1u) && pj[j] == p) grp |= 1u << j;
            }
            todo &= ~grp;
            const int64_t base = g.corrOff[p];
            const int n = (int)(g.corrOff[p + 1] - base);
            const float *rec = g.rec32 + rec32_start(base, p);
            unsigned int cnt[BOUND_GROUP];
#pragma unroll
            for (int j = 0; j < BOUND_GROUP; j++) cnt[j] = 0;
            for (int k = threadIdx.x; k < n; k += 256) {
                const float *grpRec = rec + (size_t)(k >> 2) * 20 + (k & 3);
                const float x1 = grpRec[0], y1 = grpRec[4], u2 = grpRec[8], v2 = grpRec[12], gg = grpRec[16];
#pragma unroll
                for (int j = 0; j < BOUND_GROUP; j++) {
                    if (!((grp >> j) & 1u)) continue;
                    const float m0 = __builtin_fmaf(Ef[j][0], x1, __builtin_fmaf(Ef[j][1], y1, Ef[j][2]));
                    const float m1 = __builtin_fmaf(Ef[j][3], x1, __builtin_fmaf(Ef[j][4], y1, Ef[j][5]));
                    const float m2 = __builtin_fmaf(Ef[j][6], x1, __builtin_fmaf(Ef[j][7], y1, Ef[j][8]));
                    const float num = __builtin_fmaf(u2, m0, __builtin_fmaf(v2, m1, m2));
                    const float q = __builtin_fmaf(m0, m0, __builtin_fmaf(m1, m1, gg));
                    cnt[j] += !(num * num >= q) ? 1u : 0u; // (a NaN passes)
                }
            }
#pragma unroll
            for (int j = 0; j < BOUND_GROUP; j++) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) cnt[j] += __shfl_xor(cnt[j], d);
                if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][j] = cnt[j];
            }
            __syncthreads();
            if (threadIdx.x < BOUND_GROUP && ((grp >> threadIdx.x) & 1u))
                g.candUb[c0 + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            __syncthreads();
        }
    }
}

// Settles the candidates, a thread each.  A candidate whose bound is below the count of the pair's best model so
// far can neither replace it (RANSAC.cpp:224, more inliers or as many with a smaller error sum) nor lower the
// iteration cap (:245-258): the in-order scan passes over it whatever its exact count is, so it is marked
// ROW_PASSED and none of its cells is evaluated.  The others claim a slot of the error buffer and are marked
// ROW_HANDED; where no slot is left the row keeps its mark of 0 and the scoring kernel takes it.
__global__ __launch_bounds__(256) void ransac_heavy_assign_kernel(const RansacArgs g)
{
    const int cands = (int)min(g.heavyCount[0], (unsigned)g.candMax);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cands; c += gridDim.x * 256) {
        const long long hr = g.candRow[c];
        const int p = (int)(hr >> 32), row = (int)(hr & 0xFFFFFFFFll);
        int mark = 0;
        if (g.heavyBound && g.candUb[c] < (unsigned)g.bestCount[p]) {
            mark = ROW_PASSED;
        } else {
            const unsigned slot = atomicAdd(&g.heavyCount[1], 1u);
            if (slot < (unsigned)g.heavyMax) {
                g.heavyRow[slot] = hr;
                mark = ROW_HANDED;
            }
        }
        if (mark) g.rowMark[(size_t)p * g.H + row] = (signed char)mark;
        // (tallies of the call, for sfm_ransac_batch_handed_over: a wave adds up its lanes)
        const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
        const unsigned long long pa = __builtin_amdgcn_ballot_w64(mark == ROW_PASSED), ha = __builtin_amdgcn_ballot_w64(mark == ROW_HANDED);
        if ((threadIdx.x & 63) == __builtin_ctzll(all)) {
            atomicAdd(&g.heavyCount[2], (unsigned)__builtin_popcountll(all));
            if (pa) atomicAdd(&g.heavyCount[3], (unsigned)__builtin_popcountll(pa));
            if (ha) atomicAdd(&g.heavyCount[4], (unsigned)__builtin_popcountll(ha));
        }
    }
}

// The rows handed over, cell by cell: a block takes 256 correspondences of one such row, a lane one cell (the
// row's pose is the same for the whole block).  Every lane works, whatever the row's share of inliers, and the
// blocks are as many as the device holds, each taking (slot, chunk) items in turn.
template <int LOSS>
__global__ __launch_bounds__(256) void ransac_heavy_eval_kernel(const RansacArgs g)
{
    const int used = (int)min(g.heavyCount[1], (unsigned)g.heavyMax);
    const int chunks = g.heavyStride / 256;
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = g.K[i];
    const double thr = g.maxSqErr;
    for (long long item = blockIdx.x; item < (long long)used * chunks; item += gridDim.x) {
        const int slot = (int)(item / chunks), chunk = (int)(item - (long long)slot * chunks);
        const long long hr = g.heavyRow[slot];
        const int p = (int)(hr >> 32), row = (int)(hr & 0xFFFFFFFFll);
        const int64_t base = g.corrOff[p];
        const int n = (int)(g.corrOff[p + 1] - base);
        if (chunk * 256 >= n) continue;
        const int k = chunk * 256 + (int)threadIdx.x;
        double e = -1.0;
        if (k < n) {
            Pose Q;
            load_pose(g.pose12 + ((size_t)p * g.H + row) * 12, Q);
            const double2 o1 = g.x[base + k], o2 = g.y[base + k];
            const double xy[4] = {o1.x, o1.y, o2.x, o2.y};
            if constexpr (LOSS == SFM_LOSS_SAMPSON) {
                double E[3][3];
                essential_from_pose(Q, E);
                e = sampson_loss(K, E, xy[0], xy[1], xy[2], xy[3]);
            } else {
                e = reprojection_loss<false>(K, g.plainK, thr, Q, xy);
            }
            if (!(e < thr)) e = -1.0; // RANSAC.cpp:216
        }
        g.heavyErr[(size_t)slot * g.heavyStride + k] = e;
    }
}

// Count and error sum of the rows handed over, in correspondence order (RANSAC.cpp:207-221).  The additions of a
// row form one dependent chain of as many links as the pair has correspondences, so what counts is how fast a
// chain advances, not how many run: a wave takes 16 rows and nothing but its own pace -- all 64 lanes fetch tiles of
// 64 correspondences x 16 rows (the next tile is on its way while this one is added up) and put them through the
// wave's LDS transposed, telling inliers from the rest and counting them as they go (a cell that is no inlier goes in
// as -0.0, which changes no sum); lanes 0..15 then add a row each, one LDS read and one addition per cell, sixteen
// reads ahead.  No barrier between waves.
#define HSUM_ROWS 16
__global__ __launch_bounds__(64) void ransac_heavy_sum_kernel(const RansacArgs g)
{
    __shared__ double tile[2][64][HSUM_ROWS + 1];
    __shared__ int sN[HSUM_ROWS], sP[HSUM_ROWS], sRow[HSUM_ROWS], cnts[HSUM_ROWS];
    const int used = (int)min(g.heavyCount[1], (unsigned)g.heavyMax);
    const int slot0 = blockIdx.x * HSUM_ROWS;
    if (slot0 >= used) return;
    const int lane = threadIdx.x;
    if (lane < HSUM_ROWS) {
        const int slot = slot0 + lane;
        int n = 0, p = -1, row = 0;
        if (slot < used) {
            const long long hr = g.heavyRow[slot];
            p = (int)(hr >> 32);
            row = (int)(hr & 0xFFFFFFFFll);
            n = (int)(g.corrOff[p + 1] - g.corrOff[p]);
        }
        sN[lane] = n;
        sP[lane] = p;
        sRow[lane] = row;
    }
    __syncthreads(); // (one wave: orders the LDS writes above before the reads below)
    int nMax = 0, myN[HSUM_ROWS];
#pragma unroll
    for (int j = 0; j < HSUM_ROWS; j++) {
        myN[j] = sN[j];
        nMax = max(nMax, myN[j]);
    }
    const double *src = g.heavyErr + (size_t)slot0 * g.heavyStride + lane;
    constexpr int DEPTH = 3; // tiles on their way: a fetch takes several times as long as a tile's additions
    double regs[DEPTH][HSUM_ROWS];
    auto fetch = [&](double (&r)[HSUM_ROWS], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < HSUM_ROWS; j++) r[j] = (k0 + lane < myN[j]) ? src[(size_t)j * g.heavyStride + k0] : -1.0;
    };
    int cnt[HSUM_ROWS];
#pragma unroll
    for (int j = 0; j < HSUM_ROWS; j++) cnt[j] = 0;
    double tot = 0.0;
    auto consume = [&](double (&r)[HSUM_ROWS], int k0, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < HSUM_ROWS; j++) {
            const bool in = r[j] >= 0.0;
            tile[b][lane][j] = in ? r[j] : -0.0;
            cnt[j] += in ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k0 + DEPTH * 64 < nMax) fetch(r, k0 + DEPTH * 64);
        if (lane < HSUM_ROWS) { // sixteen cells are on their way from LDS while the sixteen before them are added
            double v[16], w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = tile[b][i][lane];
#pragma unroll
            for (int kk = 0; kk < 64; kk += 32) {
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = tile[b][kk + 16 + i][lane];
#pragma unroll
                for (int i = 0; i < 16; i++) tot = tot + v[i];
                if (kk + 32 < 64) {
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = tile[b][kk + 32 + i][lane];
                }
#pragma unroll
                for (int i = 0; i < 16; i++) tot = tot + w[i];
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d * 64 < nMax) fetch(regs[d], d * 64);
    for (int k0 = 0; k0 < nMax; k0 += DEPTH * 64) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (k0 + d * 64 < nMax) consume(regs[d], k0 + d * 64, d & 1);
    }
#pragma unroll
    for (int j = 0; j < HSUM_ROWS; j++) { // the wave's lanes hold 64 correspondences' worth of row j each
        int c = cnt[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0) cnts[j] = c;
    }
    __syncthreads();
    if (lane < HSUM_ROWS && sP[lane] >= 0) {
        g.counts[(size_t)sP[lane] * g.H + sRow[lane]] = cnts[lane];
        g.toterr[(size_t)sP[lane] * g.H + sRow[lane]] = tot;
    }
}

// Second half of a split scoring launch: every row adds up its inliers' errors in correspondence
// order (RANSAC.cpp:207-221) from the (correspondence, row) buffer.  The additions of a row form one
// dependent chain, so the block's job is to keep them fed: R rows per block, all waves fetch tiles of
// SUM_KT correspondences x R rows into registers (two tiles in flight) while the first wave adds up
// the tile in LDS.  R = 64 when there are blocks enough to fill the device, 16 otherwise: a block pulls
// what one CU's memory path delivers, so fewer rows per block means more CUs at work.
#define SUM_THREADS 512
#define SUM_KT 128
template <int R>
__global__ __launch_bounds__(SUM_THREADS) void ransac_sum_kernel(const RansacArgs g, const double *__restrict__ errBuf)
{
    __shared__ double tile[SUM_KT][R];
    const int p = blockIdx.y;
    if (!g.active[p]) return;
    const int tid = threadIdx.x;
    const int r = tid % R, sub = tid / R;   // row of the block, fetch slot
    constexpr int SLOTS = SUM_THREADS / R;  // correspondences fetched side by side
    constexpr int PER = SUM_KT / SLOTS;     // fetches of a thread per tile
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    const int rows = g.hEnd - g.hBegin;
    if (g.hBegin + (int)blockIdx.x * R >= pair_row_end(g, p)) return; // rows nobody scored (the whole block)
    const int hl = blockIdx.x * R + r;
    const bool live = g.hBegin + hl < pair_row_end(g, p);
    const double *col = errBuf + (size_t)base * rows + (live ? hl : 0);
    const double none = __longlong_as_double(-1ll);
    constexpr int DEPTH = 2; // tiles in flight (four measured the same: the additions, not the fetches, set the pace)
    double bufs[DEPTH][PER];
    auto fetch = [&](double (&buf)[PER], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = k0 + sub + i * SLOTS;
            buf[i] = (live && k < n) ? col[(size_t)k * rows] : none;
        }
    };
    // Only the additions of a row depend on each other.  Telling inliers from the rest and counting them is
    // done by the threads that fetch: a cell that is no inlier goes into the tile as -0.0 (x + (-0.0) = x for
    // every x this sum can hold), so the chain is one LDS read and one addition per cell.
    __shared__ int cnts[R];
    if (tid < R) cnts[tid] = 0;
    int cnt = 0;
    double tot = 0.0;
    auto consume = [&](double (&buf)[PER], int kNext) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; i++) { // cells past the end of the pair hold the "none" pattern
            const bool in = buf[i] >= 0;
            tile[sub + i * SLOTS][r] = in ? buf[i] : -0.0;
            cnt += in ? 1 : 0;
        }
        __syncthreads();
        if (kNext < n) fetch(buf, kNext);
        if (tid < R) { // sixteen cells are on their way from LDS while the sixteen before them are added
            double v[16], w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = tile[i][r];
#pragma unroll 1
            for (int kk = 0; kk < SUM_KT; kk += 32) {
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = tile[kk + 16 + i][r];
#pragma unroll
                for (int i = 0; i < 16; i++) tot = tot + v[i];
                if (kk + 32 < SUM_KT) {
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = tile[kk + 32 + i][r];
                }
#pragma unroll
                for (int i = 0; i < 16; i++) tot = tot + w[i];
            }
        }
        __syncthreads();
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d * SUM_KT < n || d == 0) fetch(bufs[d], d * SUM_KT);
    for (int k0 = 0; k0 < n; k0 += DEPTH * SUM_KT) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (k0 + d * SUM_KT < n) consume(bufs[d], k0 + (DEPTH + d) * SUM_KT);
    }
    __syncthreads(); // (cnts is cleared also when the pair has no correspondences)
    if (cnt) atomicAdd(&cnts[r], cnt);
    __syncthreads();
    if (tid < R && live) {
        const int h = g.hBegin + hl;
        g.counts[(size_t)p * g.H + h] = cnts[tid];
        g.toterr[(size_t)p * g.H + h] = tot;
    }
}

// ---------------------------------------------------------------------------------------
// select: what an in-order worker would keep (RANSAC.cpp:182-186, :229-258)
// ---------------------------------------------------------------------------------------
__device__ int safe_cast_to_int(double v) // RANSAC.cpp:14-32
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int)0x80000000;
    return (int)v;
}

// The reference's loop (RANSAC.cpp:182-258) visits the sample rows in order, keeps the row with the
// most inliers (then the smaller error sum, then the earlier row), lowers its iteration cap whenever
// the kept row changes, and stops at the first row whose index has reached the cap.  The cap
// depends only on the largest inlier count seen so far and never rises, so the scan parallelises:
// a prefix maximum of the counts gives every row the cap in force when the loop reaches it, the
// first row at or past its cap is where the loop stops, and the kept row is the lexicographic best
// of the rows before that.  One block per pair; the state is carried from phase to phase.
__device__ __forceinline__ int iteration_cap(const RansacArgs &g, int bestCount, int n)
{
    if (!g.adaptive || bestCount <= 0) return g.H;
    const double inlierRatio = (double)bestCount / (double)n; // RANSAC.cpp:245-258
    const double succesProb = pow(inlierRatio, 8.0);
    if (!(succesProb >= 1e-6)) return g.H;
    return min(g.H, safe_cast_to_int(ceil(log(1. - g.successProb) / log(1 - succesProb))));
}

__global__ __launch_bounds__(256) void ransac_select_kernel(const RansacArgs g)
{
    __shared__ int sMax[256];
    __shared__ int sStop;
    __shared__ int sC[256], sH[256];
    __shared__ double sE[256];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = (int)(g.corrOff[p + 1] - g.corrOff[p]);
    if (n < 8) { // RANSAC.cpp:148-149
        if (tid == 0) {
            g.bestH[p] = -1;
            g.bestCount[p] = 0;
            g.bestErr[p] = INFINITY;
            g.iterations[p] = 0;
            g.active[p] = 0;
        }
        return;
    }
    if (!g.active[p]) return;
    // state at the start of this phase
    int capIn = g.H, bestCount = 0, best = -1;
    double bestError = INFINITY;
    if (g.hBegin > 0) {
        capIn = g.maxIter[p];
        bestCount = g.bestCount[p];
        best = g.bestH[p];
        bestError = g.bestErr[p];
    }
    // (rows at or past the cap the phase began with were not scored: the scan cannot reach them)
    const int hEnd = (g.hBegin > 0 && g.adaptive) ? min(g.hEnd, capIn) : g.hEnd;
    const int rows = max(hEnd - g.hBegin, 0);
    const int seg = (rows + 255) / 256;
    const int q0 = g.hBegin + min(rows, tid * seg), q1 = g.hBegin + min(rows, (tid + 1) * seg);
    const int *cnt = g.counts + (size_t)p * g.H;
    const double *err = g.toterr + (size_t)p * g.H;

    // largest count of this thread's rows, then the exclusive prefix maximum over threads
    int local = 0;
    for (int q = q0; q < q1; q++) local = max(local, cnt[q]);
    sMax[tid] = local;
    if (tid == 0) sStop = hEnd;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = (tid >= d) ? sMax[tid - d] : 0;
        __syncthreads();
        sMax[tid] = max(sMax[tid], v);
        __syncthreads();
    }
    int running = max(bestCount, tid ? sMax[tid - 1] : 0);

    // the first row that the loop does not reach: its index is at or past the cap in force
    int cap = min(capIn, iteration_cap(g, running, n));
    for (int q = q0; q < q1; q++) {
        if (q >= cap) {
            atomicMin(&sStop, q);
            break;
        }
        const int c = cnt[q];
        if (c > running) {
            running = c;
            cap = min(cap, iteration_cap(g, running, n));
        }
    }
    __syncthreads();
    const int stop = sStop;

    // best of the rows before the stop: more inliers, then smaller error sum, then earlier row
    int bc = 0, bh = -1;
    double be = INFINITY;
    for (int q = q0; q < min(q1, stop); q++) {
        const int c = cnt[q];
        const double e = err[q];
        if (c > 0 && (c > bc || (c == bc && e < be))) {
            bc = c;
            be = e;
            bh = q;
        }
    }
    sC[tid] = bc;
    sE[tid] = be;
    sH[tid] = bh;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) { // full ties go to the earlier row, wherever the tree has moved it
            const int c = sC[tid + d];
            const double e = sE[tid + d];
            if (c > 0 && (c > sC[tid] || (c == sC[tid] && (e < sE[tid] || (e == sE[tid] && sH[tid + d] < sH[tid]))))) {
                sC[tid] = c;
                sE[tid] = e;
                sH[tid] = sH[tid + d];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (sC[0] > 0 && (sC[0] > bestCount || (sC[0] == bestCount && sE[0] < bestError))) {
            bestCount = sC[0];
            bestError = sE[0];
            best = sH[0];
        }
        const int capOut = min(capIn, iteration_cap(g, bestCount, n));
        g.bestH[p] = best;
        g.bestCount[p] = bestCount;
        g.bestErr[p] = bestError;
        g.iterations[p] = stop;
        g.maxIter[p] = capOut;
        // the scan goes on iff it consumed this phase without reaching its cap and rows remain, and the
        // caller's time is not up (the reference looks at the clock every 50 iterations and stops the loop)
        const bool timeUp = g.maxTicks > 0 && (long long)(wall_clock64() - *g.t0) > g.maxTicks;
        g.active[p] = (stop == g.hEnd && stop < capOut && g.hEnd < g.H && !timeUp) ? 1 : 0; // (stop < g.hEnd where the cap cut the phase short)
    }
}

// ---------------------------------------------------------------------------------------
// inlier list of the winner, ascending
// ---------------------------------------------------------------------------------------
// Inlier list of every pair's winner, ascending (RANSAC.cpp:213-221), in three small kernels so that a
// pair's correspondences spread over many blocks: flags and per-block counts; per-pair totals and block
// offsets; the indices, written from the flags without evaluating again.  grid.x blocks of 256
// correspondences per pair (the same for every pair: the largest pair decides).
__global__ __launch_bounds__(256) void ransac_inlier_flag_kernel(const RansacArgs g, unsigned char *__restrict__ flag,
                                                                int *__restrict__ blkCnt)
{
    __shared__ int waveCnt[4];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int best = g.bestH[p];
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    const int k = blockIdx.x * 256 + tid;
    bool ok = false;
    if (best >= 0 && k < n) {
        Pose P;
        load_pose(g.pose12 + ((size_t)p * g.H + best) * 12, P);
        double K[9];
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = g.K[i];
        const double2 o1 = g.x[base + k], o2 = g.y[base + k];
        const double xy[4] = {o1.x, o1.y, o2.x, o2.y};
        double e;
        if (g.loss == SFM_LOSS_SAMPSON) {
            double E[3][3];
            essential_from_pose(P, E);
            e = sampson_loss(K, E, xy[0], xy[1], xy[2], xy[3]);
        } else {
            e = reprojection_loss<false>(K, g.plainK, g.maxSqErr, P, xy);
        }
        ok = e < g.maxSqErr;
    }
    if (k < n) flag[base + k] = ok ? 1 : 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
    if (lane == 0) waveCnt[wave] = __builtin_popcountll(mask);
    __syncthreads();
    if (tid == 0) blkCnt[(size_t)p * gridDim.x + blockIdx.x] = waveCnt[0] + waveCnt[1] + waveCnt[2] + waveCnt[3];
}

// One block per pair: inlCount[p] and, in place of the block counts, their exclusive prefix.
__global__ __launch_bounds__(256) void ransac_inlier_count_kernel(const RansacArgs g, int *__restrict__ blkCnt, int nBlk)
{
    __shared__ int part[256];
    const int p = blockIdx.x, tid = threadIdx.x;
    int *cnt = blkCnt + (size_t)p * nBlk;
    const int per = (nBlk + 255) / 256;
    const int lo = min(nBlk, tid * per), hi = min(nBlk, lo + per);
    int s = 0;
    for (int b = lo; b < hi; b++) s += cnt[b];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int b = lo; b < hi; b++) {
        const int c = cnt[b];
        cnt[b] = run;
        run += c;
    }
    if (tid == 255) g.inlCount[p] = part[255];
}

__global__ __launch_bounds__(256) void ransac_inlier_write_kernel(const RansacArgs g, const unsigned char *__restrict__ flag,
                                                                 const int *__restrict__ blkOff)
{
    __shared__ int waveCnt[4];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = g.corrOff[p];
    const int n = (int)(g.corrOff[p + 1] - base);
    const int k = blockIdx.x * 256 + tid;
    if (blockIdx.x * 256 >= n) return;
    const bool ok = k < n && flag[base + k] != 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
    const int prefix = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) waveCnt[wave] = __builtin_popcountll(mask);
    __syncthreads();
    int off = blkOff[(size_t)p * gridDim.x + blockIdx.x];
    for (int w = 0; w < wave; w++) off += waveCnt[w];
    if (ok) g.inliers[g.inlOff[p] + off + prefix] = k;
}

__global__ __launch_bounds__(1024) void ransac_scan_kernel(const int *__restrict__ counts, int n,
                                                           int64_t *__restrict__ offsets)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int k = lo; k < hi; k++) s += counts[k];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    if (tid == 0) offsets[0] = 0;
    for (int k = lo; k < hi; k++) {
        run += counts[k];
        offsets[k + 1] = run;
    }
}

// ---------------------------------------------------------------------------------------
// refit on the inliers: eightPointAlgorithm(points1, points2), eightPoint.cpp:57-223
// ---------------------------------------------------------------------------------------
// One block per pair.  The point set is the winner's inlier list when it has at least 8 entries,
// otherwise all correspondences of the pair (scene.cpp:264-298), otherwise nothing.
//   1. matrixV().col(8) of JacobiSVD(A, ComputeFullV) for the n x 9 matrix A (eightPoint.cpp:66-86), by
//      Eigen's route, so that its sign -- which orders the four candidates -- is Eigen's, and without squaring
//      the condition number: for n > 9 the column-pivoted Householder QR of A / max|A_ij|, then two-sided
//      Jacobi sweeps on the 9 x 9 factor R with V starting from the column permutation; for n = 9 the sweeps
//      on A itself; for n = 8 the last column of the Q of A^T's factorisation (as the fit kernel does).
//      The factorisation keeps no copy of A.  A reflector changes row r of the remaining columns by
//      (r's entry in the pivot column / den) * (tau * tmp_j), with den, tau and tmp_j the same for all rows,
//      so a row's state after k steps follows from its two points and 9 k block-wide scalars kept in LDS:
//      every step, each thread rebuilds its rows' state and adds their products with the pivot column to nine
//      sums (reduced over the block in a fixed order); thread 0 turns the sums into the step's reflector,
//      the row of R and the downdated column norms (recomputed by one more pass where Eigen's rule asks).
//      The sweeps run in one wave on R in LDS (lanes 0..8 own one element of the two rows / columns a
//      rotation touches; every lane works the rotation out for itself from the same four elements).
//      Sums over rows are taken in the block's order, not Eigen's, and tmp_j is formed from the raw sums:
//      agreement with the CPU restatement is to rounding (poses to 1e-8 in the tests), not bit for bit.
//   2. Thread 0: essential matrix -> 3x3 SVD -> the four (R, T) candidates (eightPoint.cpp:88-128).
//   3. All threads: cheirality vote over all n points, first strict maximum (eightPoint.cpp:130-181).
//   4. All threads: triangulation of all n points with the float cast of lambda (eightPoint.cpp:189-220).
struct RefitArgs {
    double *pose16; // [P][16] column-major 4x4
    int *used;      // [P] points used (0: fewer than 8 correspondences, pose = identity)
    double *points; // [total][3]; pair p at corrOff[p]
};

struct RefitQr {
    double R[9][9];    // the factor's rows as they are finished; then the work matrix of the sweeps
    double V[9][9];
    double tmpc[9][9]; // [m][j]: tau_m * tmp of the column at position j in step m
    double invden[9];  // 1 / (c0 - beta) of step m, 0 for a step without a reflector
    double nu[9], nd[9]; // column norms: updated, and as last computed directly
    double rowK[9];
    double G[9];
    double red[4][9];
    double scale;
    int transp[9];
    int direct; // bit j: column j's norm has to be recomputed after this step
};

// A value loaded from LDS at an address that is the same in every lane is uniform to the compiler.  amdclang 22
// turned a select between two integer constants on an fp64 comparison of two such values (the pivot of step 7
// below: 7 or 8) into s_cselect_b32 on SCC without ever moving the v_cmp's result into SCC, so the select read
// whatever the last exec-mask instruction had left there (scripts/scan_scc_select.py finds the pattern in the
// ISA).  Passing an operand through an empty asm makes it a per-lane value and the select a v_cndmask.
__device__ __forceinline__ double per_lane(double v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void refit_block_sum(RefitQr &S, double acc[9], int cnt, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < 9; e++) {
        if (e < cnt) {
            double v = acc[e];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
            if (lane == 0) S.red[wave][e] = v;
        }
    }
    __syncthreads();
    if (tid < cnt) S.G[tid] = (S.red[0][tid] + S.red[1][tid]) + (S.red[2][tid] + S.red[3][tid]);
    __syncthreads();
}

// Row (a, b) of A / scale after the first K_ steps of the factorisation, columns in their current positions.
template <int K_>
__device__ __forceinline__ void refit_row_state(const RefitQr &S, double2 a, double2 b, double cur[9])
{
    cur[0] = a.x * b.x;
    cur[1] = a.y * b.x;
    cur[2] = b.x;
    cur[3] = a.x * b.y;
    cur[4] = a.y * b.y;
    cur[5] = b.y;
    cur[6] = a.x;
    cur[7] = a.y;
    cur[8] = 1.0;
    const double scale = S.scale;
    if (scale != 1.0) {
#pragma unroll
        for (int j = 0; j < 9; j++) cur[j] = cur[j] / scale;
    }
#pragma unroll
    for (int m = 0; m < K_; m++) {
        const int t = S.transp[m];
#pragma unroll
        for (int j = m + 1; j < 9; j++) {
            const bool sw = t == j;
            const double u = cur[m], v = cur[j];
            cur[m] = sw ? v : u;
            cur[j] = sw ? u : v;
        }
        const double ess = cur[m] * S.invden[m];
#pragma unroll
        for (int j = m + 1; j < 9; j++) cur[j] = cur[j] - S.tmpc[m][j] * ess;
    }
}

// Squared norms over rows >= K_ of the columns at positions K_.. after K_ steps -> S.G[j - K_].
template <int K_, typename RowFn>
__device__ __forceinline__ void refit_colsq_pass(RefitQr &S, int n, int tid, RowFn row)
{
    double acc[9];
#pragma unroll
    for (int e = 0; e < 9; e++) acc[e] = 0.0;
    for (int r = tid; r < n; r += 256) {
        if (r < K_) continue;
        double2 a, b;
        row(r, a, b);
        double cur[9];
        refit_row_state<K_>(S, a, b, cur);
#pragma unroll
        for (int j = K_; j < 9; j++) acc[j - K_] += cur[j] * cur[j];
    }
    refit_block_sum(S, acc, 9 - K_, tid);
}

// Step K_ of ColPivHouseholderQR::computeInPlace on the n x 9 matrix.
template <int K_, typename RowFn>
__device__ __forceinline__ void refit_qr_step(RefitQr &S, int n, int tid, RowFn row)
{
    if (tid == 0) { // the column of the largest updated norm, the first of equals, comes to position K_
        int biggest = K_;
        double bn = per_lane(S.nu[K_]);
        for (int j = K_ + 1; j < 9; j++) {
            const double nj = per_lane(S.nu[j]);
            if (nj > bn) {
                bn = nj;
                biggest = j;
            }
        }
        S.transp[K_] = biggest;
        if (biggest != K_) {
            double t = S.nu[K_];
            S.nu[K_] = S.nu[biggest];
            S.nu[biggest] = t;
            t = S.nd[K_];
            S.nd[K_] = S.nd[biggest];
            S.nd[biggest] = t;
            for (int i = 0; i < K_; i++) { // the rows of R above move with their columns
                t = S.R[i][K_];
                S.R[i][K_] = S.R[i][biggest];
                S.R[i][biggest] = t;
            }
        }
    }
    __syncthreads();
    double acc[9];
#pragma unroll
    for (int e = 0; e < 9; e++) acc[e] = 0.0;
    const int tK = S.transp[K_];
    for (int r = tid; r < n; r += 256) {
        if (r < K_) continue;
        double2 a, b;
        row(r, a, b);
        double cur[9];
        refit_row_state<K_>(S, a, b, cur);
#pragma unroll
        for (int j = K_ + 1; j < 9; j++) {
            const bool sw = tK == j;
            const double u = cur[K_], v = cur[j];
            cur[K_] = sw ? v : u;
            cur[j] = sw ? u : v;
        }
        if (r == K_) {
#pragma unroll
            for (int j = K_; j < 9; j++) S.rowK[j] = cur[j];
        } else {
#pragma unroll
            for (int j = K_; j < 9; j++) acc[j - K_] += cur[K_] * cur[j];
        }
    }
    refit_block_sum(S, acc, 9 - K_, tid); // G[0] = |tail|^2, G[j - K_] = tail . column j
    if (tid == 0) {
        const double c0 = S.rowK[K_], tailSqNorm = S.G[0];
        double tau, beta, invden;
        if (tailSqNorm <= SFM_DBL_MIN) {
            tau = 0.0;
            beta = c0;
            invden = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tailSqNorm);
            if (c0 >= 0.0) beta = -beta;
            invden = 1.0 / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        S.invden[K_] = invden;
        for (int j = 0; j < K_; j++) S.R[K_][j] = 0.0;
        S.R[K_][K_] = beta;
        int direct = 0;
        for (int j = K_ + 1; j < 9; j++) {
            const double tmp = S.G[j - K_] * invden + S.rowK[j];
            const double rkj = S.rowK[j] - tau * tmp;
            S.R[K_][j] = rkj;
            S.tmpc[K_][j] = tau * tmp;
            if (S.nu[j] != 0.0) { // the norm downdate of LAPACK's xGEQPF, as Eigen applies it
                double temp = fabs(rkj) / S.nu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                const double ratio = S.nu[j] / S.nd[j];
                const double temp2 = per_lane(temp * (ratio * ratio));
                if (temp2 <= 1.4901161193847656e-08) direct |= 1 << j;
                else S.nu[j] = S.nu[j] * sqrt(temp);
            }
        }
        S.direct = direct;
    }
    __syncthreads();
    if constexpr (K_ < 8) {
        if (S.direct) { // the same for the whole block
            refit_colsq_pass<K_ + 1>(S, n, tid, row);
            if (tid > K_ && tid < 9 && ((S.direct >> tid) & 1)) S.nd[tid] = S.nu[tid] = sqrt(S.G[tid - K_ - 1]);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void ransac_refit_kernel(const RansacArgs g, const RefitArgs rf)
{
    __shared__ RefitQr S;
    __shared__ double qs[9];
    __shared__ double cand[2][9], T12s[3];
    __shared__ int cnt4[4][4];
    __shared__ int chosen;

    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = g.corrOff[p];
    const int nAll = (int)(g.corrOff[p + 1] - base);
    const int nInl = g.inlCount[p];
    const bool useInl = nInl >= 8;
    const int n = useInl ? nInl : nAll;
    const int32_t *list = g.inliers + g.inlOff[p];
    double *pose = rf.pose16 + (size_t)p * 16;
    if (n < 8) {
        if (tid < 16) pose[tid] = ((tid & 3) == (tid >> 2)) ? 1.0 : 0.0;
        if (tid == 0) rf.used[p] = 0;
        return;
    }
    auto row = [&](int r, double2 &a, double2 &b) __attribute__((always_inline)) {
        const int64_t at = base + (useInl ? list[r] : r);
        a = g.x[at];
        b = g.y[at];
    };

    // 1. the right singular vector of the smallest singular value, Eigen's way
    if (n == 8) {
        if (tid == 0) {
            double M[8][9], q[9];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                double2 a, b;
                row(i, a, b);
                M[i][0] = a.x * b.x;
                M[i][1] = a.y * b.x;
                M[i][2] = b.x;
                M[i][3] = a.x * b.y;
                M[i][4] = a.y * b.y;
                M[i][5] = b.y;
                M[i][6] = a.x;
                M[i][7] = a.y;
                M[i][8] = 1.0;
            }
            null_vector_8x9(M, q);
#pragma unroll
            for (int i = 0; i < 9; i++) qs[i] = q[i];
        }
    } else {
        { // scale = the largest |A_ij| (1 for a matrix of zeros)
            double mx = 0.0;
            for (int r = tid; r < n; r += 256) {
                double2 a, b;
                row(r, a, b);
                const double e[9] = {a.x * b.x, a.y * b.x, b.x, a.x * b.y, a.y * b.y, b.y, a.x, a.y, 1.0};
#pragma unroll
                for (int j = 0; j < 9; j++)
                    if (fabs(e[j]) > mx) mx = fabs(e[j]);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double o = __shfl_down(mx, d, 64);
                if (o > mx) mx = o;
            }
            if (lane == 0) S.red[wave][0] = mx;
            __syncthreads();
            if (tid == 0) {
                double m = S.red[0][0];
                for (int w = 1; w < 4; w++)
                    if (S.red[w][0] > m) m = S.red[w][0];
                S.scale = (m == 0.0) ? 1.0 : m;
            }
            __syncthreads();
        }
        if (n == 9) { // square: the sweeps run on A / scale itself, V starts as the identity
            if (tid < 9) {
                double2 a, b;
                row(tid, a, b);
                double cur[9];
                refit_row_state<0>(S, a, b, cur);
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    S.R[tid][j] = cur[j];
                    S.V[tid][j] = (tid == j) ? 1.0 : 0.0;
                }
            }
        } else {
            refit_colsq_pass<0>(S, n, tid, row);
            if (tid < 9) S.nd[tid] = S.nu[tid] = sqrt(S.G[tid]);
            __syncthreads();
            refit_qr_step<0>(S, n, tid, row);
            refit_qr_step<1>(S, n, tid, row);
            refit_qr_step<2>(S, n, tid, row);
            refit_qr_step<3>(S, n, tid, row);
            refit_qr_step<4>(S, n, tid, row);
            refit_qr_step<5>(S, n, tid, row);
            refit_qr_step<6>(S, n, tid, row);
            refit_qr_step<7>(S, n, tid, row);
            refit_qr_step<8>(S, n, tid, row);
            if (tid == 0) { // colsPermutation(): the transpositions applied on the right of the identity
                int perm[9];
                for (int i = 0; i < 9; i++) perm[i] = i;
                for (int k = 0; k < 9; k++) {
                    const int t = perm[k];
                    perm[k] = perm[S.transp[k]];
                    perm[S.transp[k]] = t;
                }
                for (int i = 0; i < 9; i++)
                    for (int j = 0; j < 9; j++) S.V[i][j] = (i == perm[j]) ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        // JacobiSVD's sweeps on the 9 x 9 work matrix (p = 1..8, q = 0..p-1), in one wave: no block barrier
        if (wave == 0) {
            auto wave_sync = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            double maxDiag = 0.0;
            for (int i = 0; i < 9; i++)
                if (fabs(S.R[i][i]) > maxDiag) maxDiag = fabs(S.R[i][i]);
            for (int sweep = 0; sweep < 1000; sweep++) {
                bool finished = true;
                for (int pp = 1; pp < 9; pp++) {
                    for (int qq = 0; qq < pp; qq++) {
                        wave_sync();
                        const double wpq = S.R[pp][qq], wqp = S.R[qq][pp];
                        const double pm = (2.0 * SFM_DBL_EPS) * maxDiag;
                        const double threshold = (SFM_DBL_MIN < pm) ? pm : SFM_DBL_MIN;
                        if (fabs(wpq) > threshold || fabs(wqp) > threshold) { // the same in every lane
                            finished = false;
                            JRot jl, jr;
                            real_2x2_jacobi_svd(S.R[pp][pp], wpq, wqp, S.R[qq][qq], jl, jr);
                            const bool il = jl.c == 1.0 && jl.s == 0.0, ir = jr.c == 1.0 && -jr.s == 0.0;
                            if (lane < 9) { // rows pp, qq of the work matrix
                                double x = S.R[pp][lane], y = S.R[qq][lane];
                                rotate_pair(x, y, jl.c, jl.s, il);
                                S.R[pp][lane] = x;
                                S.R[qq][lane] = y;
                            }
                            wave_sync();
                            if (lane < 9) { // columns pp, qq of the work matrix and of V
                                double x = S.R[lane][pp], y = S.R[lane][qq];
                                rotate_pair(x, y, jr.c, -jr.s, ir);
                                S.R[lane][pp] = x;
                                S.R[lane][qq] = y;
                                x = S.V[lane][pp];
                                y = S.V[lane][qq];
                                rotate_pair(x, y, jr.c, -jr.s, ir);
                                S.V[lane][pp] = x;
                                S.V[lane][qq] = y;
                            }
                            wave_sync();
                            const double app = fabs(S.R[pp][pp]), aqq = fabs(S.R[qq][qq]);
                            const double mx = (app < aqq) ? aqq : app;
                            if (maxDiag < mx) maxDiag = mx;
                        }
                    }
                }
                if (finished) break;
            }
            wave_sync();
            if (lane == 0) { // singular values, descending by selection (the first of equal maxima); column 8
                double sv[9];
                int col[9];
                for (int i = 0; i < 9; i++) {
                    sv[i] = per_lane(fabs(S.R[i][i]) * S.scale);
                    col[i] = i;
                }
                for (int i = 0; i < 9; i++) {
                    int pos = i;
                    for (int k = i + 1; k < 9; k++)
                        if (sv[k] > sv[pos]) pos = k;
                    if (sv[pos] == 0.0) break;
                    if (pos != i) {
                        const double t = sv[i];
                        sv[i] = sv[pos];
                        sv[pos] = t;
                        const int c = col[i];
                        col[i] = col[pos];
                        col[pos] = c;
                    }
                }
                for (int i = 0; i < 9; i++) qs[i] = S.V[i][col[8]];
            }
        }
    }
    __syncthreads();

    // 2. candidates
    if (tid == 0) {
        double q[9];
        for (int i = 0; i < 9; i++) q[i] = qs[i];
        const double E[3][3] = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}};
        double U[3][3], S[3], W[3][3];
        svd3(E, U, S, W);
        double R1[3][3], R2[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double w = U[i][2] * W[j][2];
                R1[i][j] = (U[i][1] * W[j][0] - U[i][0] * W[j][1]) + w;
                R2[i][j] = (U[i][0] * W[j][1] - U[i][1] * W[j][0]) + w;
            }
        const double s1 = det3(R1) < 0 ? -1.0 : 1.0, s2 = det3(R2) < 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                cand[0][3 * i + j] = s1 * R1[i][j];
                cand[1][3 * i + j] = s2 * R2[i][j];
            }
            T12s[i] = U[i][2];
        }
    }
    __syncthreads();

    // 3. cheirality vote
    double Rc[2][3][3], T12[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Rc[0][i][j] = cand[0][3 * i + j];
            Rc[1][i][j] = cand[1][3 * i + j];
        }
        T12[i] = T12s[i];
    }
    int cnt[4] = {0, 0, 0, 0};
    for (int k = tid; k < n; k += 256) {
        const int64_t at = base + (useInl ? list[k] : k);
        const double2 a = g.x[at], b = g.y[at];
        const double nb = sqrt((b.x * b.x + b.y * b.y) + 1.0);
#pragma unroll
        for (int r = 0; r < 2; r++) { // candidates 2r, 2r + 1: same rotation, opposite translations
            double av[3];
            rot_apply_h(Rc[r], a.x, a.y, av);
            double l0, l1;
            triangulate_lambdas(av, b.x, b.y, nb, T12, l0, l1);
            if (l0 > 0 && l1 > 0) cnt[2 * r]++;
            if (-l0 > 0 && -l1 > 0) cnt[2 * r + 1]++;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int v = cnt[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
        if (lane == 0) cnt4[wave][c] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int maxIndex = 0, maxPoints = 0;
        for (int c = 0; c < 4; c++) {
            const int v = cnt4[0][c] + cnt4[1][c] + cnt4[2][c] + cnt4[3][c];
            if (v > maxPoints) {
                maxIndex = c;
                maxPoints = v;
            }
        }
        chosen = maxIndex;
        rf.used[p] = n;
    }
    __syncthreads();
    const int ci = chosen;
    const double sgn = (ci & 1) ? -1.0 : 1.0;
    const double Tc[3] = {sgn * T12[0], sgn * T12[1], sgn * T12[2]};
    if (tid < 16) { // column-major [R | T; 0 0 0 1]
        const int r = tid & 3, c = tid >> 2;
        double v = (r == c) ? 1.0 : 0.0;
        if (r < 3 && c < 3) v = cand[ci >> 1][3 * r + c];
        if (r < 3 && c == 3) v = Tc[r];
        pose[tid] = v;
    }

    // 4. triangulate
    double *pts = rf.points + (size_t)base * 3;
    for (int k = tid; k < n; k += 256) {
        const int64_t at = base + (useInl ? list[k] : k);
        const double2 a = g.x[at], b = g.y[at];
        const double nb = sqrt((b.x * b.x + b.y * b.y) + 1.0);
        double av[3];
        rot_apply_h(Rc[ci >> 1], a.x, a.y, av);
        double l0, l1;
        triangulate_lambdas(av, b.x, b.y, nb, Tc, l0, l1);
        const double lam = (double)(float)l0; // eightPoint.cpp:216
        pts[3 * k + 0] = lam * a.x;
        pts[3 * k + 1] = lam * a.y;
        pts[3 * k + 2] = lam;
    }
}

__global__ void ransac_clock_kernel(unsigned long long *t0) { *t0 = wall_clock64(); }

// eightPoint.cpp:349-369 on its own: every correspondence triangulated under one pose (R row-major, then T),
// depth through float (:365), point = depth * (x1, y1, 1).
__global__ __launch_bounds__(256) void ransac_triangulate_kernel(const double *__restrict__ pose12, const double2 *__restrict__ p1,
                                                                 const double2 *__restrict__ p2, int n, double *__restrict__ points)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Pose P;
    load_pose(pose12, P);
    const double2 a = p1[k], b = p2[k];
    double av[3];
    rot_apply_h(P.R, a.x, a.y, av);
    double l0, l1;
    triangulate_lambdas(av, b.x, b.y, sqrt((b.x * b.x + b.y * b.y) + 1.0), P.T, l0, l1);
    const double lam = (double)(float)l0;
    points[3 * k + 0] = lam * a.x;
    points[3 * k + 1] = lam * a.y;
    points[3 * k + 2] = lam;
}

// Winner of every pair in one dense table (one copy to the host instead of one per pair):
// packed[p] = {bestH, bestCount, iterations, 0 | bestErr | pose12 of the winner, identity if there is none}.
struct PackedBest {
    int bestH, bestCount, iterations, pad;
    double bestErr;
    double pose12[12];
};
__global__ __launch_bounds__(256) void ransac_pack_kernel(const RansacArgs g, PackedBest *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.nPairs) return;
    PackedBest r;
    r.bestH = g.bestH[p];
    r.bestCount = g.bestCount[p];
    r.iterations = g.iterations[p];
    r.pad = 0;
    r.bestErr = g.bestErr[p];
#pragma unroll
    for (int i = 0; i < 12; i++)
        r.pose12[i] = r.bestH >= 0 ? g.pose12[((size_t)p * g.H + r.bestH) * 12 + i] : ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0);
    out[p] = r;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct sfm_ransac_batch {
    sfm_ctx *ctx = nullptr;
    int n_pairs = 0;
    int H = 0;
    int64_t total = 0;
    DevBuf x, y, samples, corrOff, rec, pose12, counts, toterr, bestH, bestCount, bestErr, iterations, inlCount,
        inlOff, inliers, refitPose, refitUsed, refitPoints, active, maxIter, errBuf, rec32, mfOps, inlFlag, inlBlk, clock, packed,
        heavyCount, heavyRow, heavyErr, candRow, candUb, rowMark;
    PinBuf stage;   // single-pair calls: the winner's record and its inlier list in one copy
    PinBuf stageIn; // and the two correspondence arrays on their way in
    RansacArgs args;       // of the last run (device pointers; x, y may belong to the caller)
    bool scored = false;   // the last run went through selection (not fit-only)
    bool refitted = false; // sfm_ransac_batch_refit ran on the last run's result
    std::vector<int64_t> corrOffHost; // the last run's offsets
    ReadFence readFence; // a gather on another stream still reads the winners' records and inlier lists up to this point
    // the rows handed over are evaluated on a stream of their own, beside the scoring kernel that left them out
    hipStream_t side = nullptr;
    hipEvent_t sideGo = nullptr, sideDone = nullptr;
};

extern "C" void sfm_ransac_batch_destroy(sfm_ransac_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    DevBuf *bufs[] = {&b->x, &b->y, &b->samples, &b->corrOff, &b->rec, &b->pose12, &b->counts, &b->toterr,
                      &b->bestH, &b->bestCount, &b->bestErr, &b->iterations, &b->inlCount, &b->inlOff, &b->inliers,
                      &b->refitPose, &b->refitUsed, &b->refitPoints, &b->active, &b->maxIter, &b->errBuf, &b->rec32, &b->mfOps, &b->inlFlag, &b->inlBlk,
                      &b->clock, &b->packed, &b->heavyCount, &b->heavyRow, &b->heavyErr, &b->candRow, &b->candUb, &b->rowMark};
    for (DevBuf *d : bufs) d->release();
    b->stage.release();
    b->stageIn.release();
    b->readFence.destroy();
    if (b->side) (void)hipStreamDestroy(b->side);
    if (b->sideGo) (void)hipEventDestroy(b->sideGo);
    if (b->sideDone) (void)hipEventDestroy(b->sideDone);
    sfm_ctx *owner = b->ctx;
    delete b;
    sfm_ctx_release(owner);
}

// Segments per pair of a scoring launch over `rows` rows: more than one when the launch would leave
// most of the device idle (too few (pair, 256-row) blocks) and its (correspondence, row) error buffer
// stays moderate.
static int split_factor(const sfm_ctx *ctx, int n_pairs, int rows, int64_t total)
{
    if (!ctx->ransac_split || n_pairs <= 0 || rows <= 0) return 1;
    const int64_t blocks = (int64_t)n_pairs * ((rows + 255) / 256);
    const size_t errBytes = (size_t)rows * (size_t)total * sizeof(double);
    if (blocks >= 2 * ctx->n_cus || errBytes > ((size_t)1 << 31)) return 1;
    return (int)std::max<int64_t>(1, std::min<int64_t>((4 * ctx->n_cus + blocks - 1) / blocks, total / n_pairs / 16));
}

// The pre-test bounds the reprojection loss and needs K = [fx 0 cx; 0 fy cy; 0 0 1] with positive focal lengths
// (K column-major: K(r,c) = K[r + 3c]).
static bool pretest_applies(const sfm_ctx *ctx, const double *K, int loss)
{
    const double fx = K[0], fy = K[4];
    return K[2] == 0.0 && K[5] == 0.0 && K[8] == 1.0 && K[1] == 0.0 && K[3] == 0.0 && fx > 0.0 && fy > 0.0 &&
           std::isfinite(fx) && std::isfinite(fy) && std::isfinite(1.0 / fx) && std::isfinite(1.0 / fy) &&
           ctx->ransac_filter != 0 && loss != SFM_LOSS_SAMPSON;
}

// The matrix-core scoring kernel (64 rows per wave); pre: the correspondence side was prepared; defer: rows with
// many inliers were handed over to ransac_heavy_eval_kernel.
template <int LOSS>
static void (*mfma_score_kernel(bool pre, bool defer))(const RansacArgs)
{
    if (pre && defer) return ransac_score_mfma_kernel<4, true, LOSS, true>;
    if (pre) return ransac_score_mfma_kernel<4, true, LOSS>;
    return ransac_score_mfma_kernel<4, false, LOSS>;
}

static int ransac_run(sfm_ctx *ctx, int n_pairs, const int64_t *corr_offsets, const double *x, const double *y,
                      const double *K, const int32_t *samples, int n_hyp, double maxSqErr, double successProb,
                      int adaptive, int loss, bool on_device, bool fit_only, sfm_ransac_batch **batch_inout,
                      bool phased = true, bool no_alloc = false)
{
    // no_alloc: the batch's buffers are fixed slices of the static workspace; nothing here may allocate (no
    // split scoring, whose buffer grows with the launch; no kernel timing, whose events are created on demand).
    if (!ctx || !batch_inout) return SFM_ERR_INVALID_ARG;
    if (n_pairs < 0 || n_hyp < 0 || (n_pairs > 0 && !corr_offsets))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: bad sizes");
    if (!K) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: null K");
    if (loss != SFM_LOSS_REPROJECTION && loss != SFM_LOSS_SAMPSON)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: unknown loss %d", loss);
    if (n_pairs > 65535) SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "ransac: more than 65535 pairs per call");
    int64_t total = 0;
    if (n_pairs > 0) {
        if (corr_offsets[0] != 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: offsets must start at 0");
        for (int p = 0; p < n_pairs; p++) {
            const int64_t n = corr_offsets[p + 1] - corr_offsets[p];
            if (n < 0 || n > 0x7FFFFFFF) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: offsets not ascending at pair %d", p);
        }
        total = corr_offsets[n_pairs];
    }
    if (total > 0 && (!x || !y)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: null correspondences");
    if (n_hyp > 0 && n_pairs > 0 && !samples && on_device) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: null sample table");
    const bool draw_samples = !samples && !on_device && n_hyp > 0 && n_pairs > 0; // the library draws the table itself
    SFM_HIP(ctx, hipSetDevice(ctx->device));

    sfm_ransac_batch *b = *batch_inout;
    if (!b) {
        b = new (std::nothrow) sfm_ransac_batch();
        if (!b) SFM_FAIL(ctx, SFM_ERR_OOM, "ransac: host allocation failed");
        sfm_ctx_retain(ctx);
        b->ctx = ctx;
        *batch_inout = b;
    } else if (b->ctx != ctx) {
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: batch belongs to another context");
    }
    SFM_HIP(ctx, b->readFence.wait_on(ctx->stream)); // a gather of the run before may still read its outputs
    b->n_pairs = n_pairs;
    b->H = n_hyp;
    b->total = total;
    b->scored = false;
    b->refitted = false;
    if (n_pairs > 0) b->corrOffHost.assign(corr_offsets, corr_offsets + n_pairs + 1); // (capacity reserved at set-up for the static batch)
    else b->corrOffHost.assign(1, 0);
    if (!no_alloc) sfm_span_reset(ctx, 1);
    hipStream_t st = ctx->stream;
    const size_t PH = (size_t)(n_pairs ? n_pairs : 1) * (size_t)(n_hyp ? n_hyp : 1);
    const size_t P1 = (size_t)(n_pairs ? n_pairs : 1);

    RansacArgs g;
    memset(&g, 0, sizeof g);
    if (on_device) {
        g.x = reinterpret_cast<const double2 *>(x);
        g.y = reinterpret_cast<const double2 *>(y);
        g.samples = samples;
    } else {
        SFM_HIP(ctx, b->x.reserve((size_t)(total ? total : 1) * sizeof(double2)));
        SFM_HIP(ctx, b->y.reserve((size_t)(total ? total : 1) * sizeof(double2)));
        SFM_HIP(ctx, b->samples.reserve(PH * 8 * sizeof(int32_t)));
        if (total && !no_alloc && (size_t)total * sizeof(double2) <= ((size_t)1 << 20)) {
            // small inputs (one pair of the reference's frame loop): both arrays through one pinned buffer, two
            // copies that are asynchronous for real instead of two staged copies from pageable memory
            const size_t bytes = (size_t)total * sizeof(double2);
            SFM_HIP(ctx, b->stageIn.reserve(2 * bytes));
            SFM_HIP(ctx, hipStreamSynchronize(st)); // the buffer may still feed the previous call's copies
            memcpy(b->stageIn.p, x, bytes);
            memcpy(b->stageIn.as<unsigned char>() + bytes, y, bytes);
            SFM_HIP(ctx, hipMemcpyAsync(b->x.p, b->stageIn.p, bytes, hipMemcpyHostToDevice, st));
            SFM_HIP(ctx, hipMemcpyAsync(b->y.p, b->stageIn.as<unsigned char>() + bytes, bytes, hipMemcpyHostToDevice, st));
        } else if (total) {
            SFM_HIP(ctx, hipMemcpyAsync(b->x.p, x, (size_t)total * sizeof(double2), hipMemcpyHostToDevice, st));
            SFM_HIP(ctx, hipMemcpyAsync(b->y.p, y, (size_t)total * sizeof(double2), hipMemcpyHostToDevice, st));
        }
        if (n_pairs && n_hyp && !draw_samples)
            SFM_HIP(ctx, hipMemcpyAsync(b->samples.p, samples, (size_t)n_pairs * n_hyp * 8 * sizeof(int32_t),
                                        hipMemcpyHostToDevice, st));
        g.x = b->x.as<double2>();
        g.y = b->y.as<double2>();
        g.samples = b->samples.as<int32_t>();
    }
    SFM_HIP(ctx, b->corrOff.reserve((P1 + 1) * sizeof(int64_t)));
    // (with the pre-test on, the fp64 records are made where they are needed and not stored)
    SFM_HIP(ctx, b->rec.reserve(pretest_applies(ctx, K, loss) ? 256 : (size_t)(total ? total : 1) * REC_D * sizeof(double)));
    SFM_HIP(ctx, b->rec32.reserve((rec32_start(total, n_pairs) + 40) * sizeof(float)));
    SFM_HIP(ctx, b->pose12.reserve(PH * 12 * sizeof(double)));
    SFM_HIP(ctx, b->counts.reserve(PH * sizeof(int)));
    SFM_HIP(ctx, b->toterr.reserve(PH * sizeof(double)));
    SFM_HIP(ctx, b->bestH.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->bestCount.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->bestErr.reserve(P1 * sizeof(double)));
    SFM_HIP(ctx, b->iterations.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->inlCount.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->inlOff.reserve((P1 + 1) * sizeof(int64_t)));
    SFM_HIP(ctx, b->inliers.reserve((size_t)(total ? total : 1) * sizeof(int32_t)));
    if (n_pairs == 0) return SFM_OK;
    SFM_HIP(ctx, hipMemcpyAsync(b->corrOff.p, corr_offsets, (size_t)(n_pairs + 1) * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));

    if (draw_samples) {
        uint64_t seed = ctx->ransac_seed;
        if (seed == 0) { // as the reference: from the clock (RANSAC.cpp:170)
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            seed = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
        }
        sfm_launch_sample_table(b->corrOff.as<int64_t>(), n_pairs, n_hyp, seed, b->samples.as<int32_t>(), st);
    }
    SFM_HIP(ctx, b->clock.reserve(sizeof(unsigned long long)));
    SFM_HIP(ctx, b->packed.reserve(P1 * sizeof(PackedBest)));
    g.t0 = b->clock.as<unsigned long long>();
    g.maxTicks = 0;
    if (ctx->ransac_max_time_ms > 0 && !fit_only) { // wall_clock64 ticks at 100 MHz
        g.maxTicks = (long long)ctx->ransac_max_time_ms * 100000ll;
        hipLaunchKernelGGL(ransac_clock_kernel, dim3(1), dim3(1), 0, st, b->clock.as<unsigned long long>());
    }
    g.corrOff = b->corrOff.as<int64_t>();
    g.rec = b->rec.as<double>();
    g.rec32 = b->rec32.as<float>();
    g.nPairs = n_pairs;
    g.pose12 = b->pose12.as<double>();
    g.counts = b->counts.as<int>();
    g.toterr = b->toterr.as<double>();
    g.bestH = b->bestH.as<int>();
    g.bestCount = b->bestCount.as<int>();
    g.bestErr = b->bestErr.as<double>();
    g.iterations = b->iterations.as<int>();
    g.inlCount = b->inlCount.as<int>();
    g.inlOff = b->inlOff.as<int64_t>();
    g.inliers = b->inliers.as<int32_t>();
    for (int i = 0; i < 9; i++) g.K[i] = K[i];
    {
        bool finite = true;
        for (int i = 0; i < 9; i++) finite = finite && std::isfinite(K[i]) && std::fabs(K[i]) <= 1e12;
        g.plainK = (finite && K[1] == 0.0 && K[2] == 0.0 && K[3] == 0.0 && K[5] == 0.0 && K[8] == 1.0) ? 1 : 0;
    }
    bool sampsonTest = false;
    {
        // K column-major: K(r,c) = K[r + 3c].  The pre-test needs K = [fx 0 cx; 0 fy cy; 0 0 1] with
        // positive focal lengths: observations K*(x,y,1) are then pixels, and the epipolar line of a
        // hypothesis keeps a simple form in them (see the scoring kernel).
        const double fx = K[0], fy = K[4];
        const bool ok = pretest_applies(ctx, K, loss);
        // (the Sampson distance has no packed test; its matrix-core test needs the same form of K)
        sampsonTest = loss == SFM_LOSS_SAMPSON && pretest_applies(ctx, K, SFM_LOSS_REPROJECTION) && !no_alloc;
        const bool form = ok || sampsonTest;
        g.filter = ok ? 1 : 0;
        g.Ki[0] = form ? 1.0 / fx : 0.0;
        g.Ki[1] = form ? 1.0 / fy : 0.0;
        g.Ki[2] = form ? 1.0 / std::min(fx, fy) : 0.0;
        g.Ki[3] = form ? std::max(fx, fy) * std::max(fx, fy) : 0.0;          // bound of |m|^2 from the scaled line
        g.mfI[0] = sampsonTest ? 2.0 / (fx + fy) : g.Ki[0];
        g.mfI[1] = sampsonTest ? 2.0 / (fx + fy) : g.Ki[1];
    }
    g.maxSqErr = maxSqErr;
    g.mfma = 0;
    g.mfOps = nullptr;
    g.loss = loss;
    if ((g.filter || sampsonTest) && ctx->ransac_mfma && maxSqErr > 0.0 && std::isfinite(maxSqErr)) {
        // scale of the hypothesis side: the largest entry of the q form, 2 c^2 (1/fx^2 + 1/fy^2) 2^(P-4) with
        // P = 2 (10 + mfKe), must stay inside fp16 (2^15 here)
        const double c2 = SKIP_FACTOR * SKIP_FACTOR * maxSqErr;
        const double cf2 = c2 * (g.mfI[0] * g.mfI[0] + g.mfI[1] * g.mfI[1]);
        const int ke = (int)std::floor((18.0 - std::log2(cf2)) / 2.0 - 10.0);
        bool fits = ke >= -4; // (a threshold of hundreds of focal lengths: the packed test serves)
        for (int p = 0; p < n_pairs && fits; p++) fits = corr_offsets[p + 1] - corr_offsets[p] < (1 << 24); // queue entries
        if (fits) {
            g.mfma = 1;
            g.mfKe = std::min(ke, 10);
        }
    }
    g.successProb = successProb;
    g.H = n_hyp;
    g.adaptive = adaptive;
    g.loss = loss;

    // Phases of hypothesis rows.  With the adaptive cap on, a pair whose in-order scan has stopped
    // needs no further rows, so rows are fitted and scored in growing slices and the scan is resumed
    // after each; the first slice is sized to fill the device (a slice's grid is rows / 256 blocks
    // per pair).  When the caller wants every row's score one phase covers all; without the cap, two (see below)
    // where rows with many inliers can be handed over, else one.
    SFM_HIP(ctx, b->active.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->maxIter.reserve(P1 * sizeof(int)));
    g.active = b->active.as<int>();
    g.maxIter = b->maxIter.as<int>();
    SFM_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)b->active.p, 1, (size_t)n_pairs, st));
    int64_t maxN = 0;
    for (int p = 0; p < n_pairs; p++) maxN = std::max(maxN, corr_offsets[p + 1] - corr_offsets[p]);
    // Will the scoring launches run the test on the matrix cores from prepared operands?  Then rows with many
    // inliers can be handed over to the cell-by-cell kernels (a buffer of one error per cell of such a row, as many
    // rows as the budget holds), and a row that cannot reach the best count of the slices before is passed over:
    // reason enough to score in slices also when no iteration cap is in force.
    const size_t opsBytes = (mfma_ops_start(total, n_pairs) + 1) * 2048;
    const bool opsFit = !no_alloc && ctx->ransac_mfma_ops_gib > 0 && opsBytes <= ((size_t)ctx->ransac_mfma_ops_gib << 30);
    const bool mfmaPath = !fit_only && g.mfma && ((g.filter && loss == SFM_LOSS_REPROJECTION) || sampsonTest) && opsFit &&
                          split_factor(ctx, n_pairs, std::min(n_hyp, 256), total) == 1;
    g.heavyMax = 0;
    g.heavyStride = (int)((maxN + 255) / 256 * 256);
    if (mfmaPath && ctx->ransac_defer_mib > 0 && maxN >= 16 * 4 * DEFER_CHECK_STEPS && n_hyp > 0) {
        const size_t perRow = (size_t)g.heavyStride * sizeof(double);
        int64_t slots = (int64_t)(((size_t)ctx->ransac_defer_mib << 20) / perRow);
        slots = std::min<int64_t>(slots, ((int64_t)n_pairs * n_hyp + 63) / 64 * 64) / 64 * 64;
        slots = std::min<int64_t>(slots, 1 << 20);
        if (slots >= 64 && b->heavyErr.reserve((size_t)slots * perRow) == hipSuccess) {
            const int64_t cands = std::min<int64_t>((int64_t)n_pairs * n_hyp, 1 << 22);
            SFM_HIP(ctx, b->heavyRow.reserve((size_t)slots * sizeof(long long)));
            SFM_HIP(ctx, b->heavyCount.reserve(8 * sizeof(unsigned int)));
            SFM_HIP(ctx, hipMemsetAsync(b->heavyCount.p, 0, 8 * sizeof(unsigned int), st));
            SFM_HIP(ctx, b->candRow.reserve((size_t)cands * sizeof(long long)));
            SFM_HIP(ctx, b->candUb.reserve((size_t)cands * sizeof(unsigned int)));
            SFM_HIP(ctx, b->rowMark.reserve(PH));
            g.rowMark = b->rowMark.as<signed char>();
            if (!b->side) {
                SFM_HIP(ctx, hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
                SFM_HIP(ctx, hipEventCreateWithFlags(&b->sideGo, hipEventDisableTiming));
                SFM_HIP(ctx, hipEventCreateWithFlags(&b->sideDone, hipEventDisableTiming));
            }
            g.heavyMax = (int)slots;
            g.candMax = (int)cands;
            g.heavyErr = b->heavyErr.as<double>();
            g.heavyRow = b->heavyRow.as<long long>();
            g.heavyCount = b->heavyCount.as<unsigned int>();
            g.candRow = b->candRow.as<long long>();
            g.candUb = b->candUb.as<unsigned int>();
        } else if (slots >= 64) {
            (void)hipGetLastError(); // no room: the waves keep their rows
        }
    }
    const bool defer = g.heavyMax > 0;
    int bounds[40]; // slices grow by a factor of four: at most 17 of them below 2^31 rows
    int nBounds = 0;
    bounds[nBounds++] = 0;
    if (phased && !fit_only && (adaptive || defer)) {
        // The first slice is as small as its launch can be made efficient: 256 rows (one block of rows
        // per pair; with most correspondences inliers the cap falls below that at once) when the scoring
        // launch can be split over correspondences with a moderate buffer, otherwise enough rows to fill
        // the device twice over (a CU holds four blocks of the scoring kernel: 8 blocks per CU are two full
        // rounds, and with slices growing by four the later launches are whole rounds too when the pair count
        // divides evenly).  Where rows can be handed over, a quarter of that (half without a cap) and a growth
        // of eight: the first slice has no count to beat yet, so each of its rows with many inliers is
        // evaluated in full, and every slice after it ends with the device running empty once (measured on 256
        // pairs of 20000 correspondences: at 80 % inliers the smaller slice is worth a third of the time, at
        // 15 % the larger one a twentieth).
        // Small batches (measured on 64 and 16 pairs): one that is at most a round of blocks in all takes a block
        // per CU first (64 pairs: 3.6 / 2.6 / 1.3 ms at 80 / 50 / 15 % inliers against 8.5 / 1.5 / 1.0 ms in one
        // slice); one of at most a block per CU stays whole, as before.
        const int64_t blocksInAll = (int64_t)n_pairs * ((n_hyp + 255) / 256);
        const int64_t fill = (!defer || blocksInAll <= ctx->n_cus) ? 8 * (int64_t)ctx->n_cus
                             : blocksInAll <= 4 * (int64_t)ctx->n_cus ? (int64_t)ctx->n_cus
                                                                      : (adaptive ? 2 : 4) * (int64_t)ctx->n_cus;
        int64_t h = 256 * std::max<int64_t>(1, (fill + n_pairs - 1) / n_pairs);
        if (!no_alloc && split_factor(ctx, n_pairs, 256, total) > 1 && (size_t)total * 256 * sizeof(double) <= ((size_t)1 << 30)) h = 256;
        if (ctx->hyps_per_launch > 0) h = 64 * ((ctx->hyps_per_launch + 63) / 64); // caller's first slice
        // (a single pair whose scan has stopped still pays a few microseconds for every later slice's
        // launches: its slices grow by eight, a batch's by four)
        int grow = (n_pairs == 1 || defer) ? 8 : 4;
        if (!adaptive) grow = 1 << 20; // (no cap to wait for: the first slice finds a count to beat, the second takes the rest)
        while (h < n_hyp && nBounds < 38) {
            bounds[nBounds++] = (int)h;
            h *= grow;
        }
    }
    bounds[nBounds++] = n_hyp;
    if (n_hyp > 0) // poses of rows that are never fitted must not be read as garbage
        SFM_HIP(ctx, hipMemsetAsync(b->pose12.p, 0, (size_t)n_pairs * n_hyp * 12 * sizeof(double), st));
    if (!fit_only && total > 0)
        hipLaunchKernelGGL(ransac_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g, total);
    bool opsTried = false;
    for (int ph = 0; ph + 1 < nBounds; ph++) {
        g.hBegin = bounds[ph];
        g.hEnd = bounds[ph + 1];
        const int rows = g.hEnd - g.hBegin;
        if (rows > 0) hipLaunchKernelGGL(ransac_fit_kernel, dim3((rows + 63) / 64, n_pairs), dim3(64), 0, st, g);
        if (fit_only) continue;
        if (rows > 0) {
            const size_t errBytes = (size_t)rows * (size_t)total * sizeof(double);
            const int Z = no_alloc ? 1 : split_factor(ctx, n_pairs, rows, total);
            if (!no_alloc) SFM_HIP(ctx, sfm_span_begin(ctx, 1));
            const dim3 grid((rows + 255) / 256, n_pairs, Z);
            const bool pretest = g.filter != 0;
            bool handOver = false;
            double *eb = nullptr;
            if (Z > 1) {
                SFM_HIP(ctx, b->errBuf.reserve(errBytes));
                eb = b->errBuf.as<double>();
            }
            void (*score)(const RansacArgs, double *);
            if (loss == SFM_LOSS_SAMPSON)
                score = Z > 1 ? ransac_score_kernel<SFM_LOSS_SAMPSON, true, false> : ransac_score_kernel<SFM_LOSS_SAMPSON, false, false>;
            else if (pretest)
                score = Z > 1 ? ransac_score_kernel<SFM_LOSS_REPROJECTION, true, true> : ransac_score_kernel<SFM_LOSS_REPROJECTION, false, true>;
            else
                score = Z > 1 ? ransac_score_kernel<SFM_LOSS_REPROJECTION, true, false>
                              : ransac_score_kernel<SFM_LOSS_REPROJECTION, false, false>;
            if (Z == 1 && g.mfma && ((pretest && loss == SFM_LOSS_REPROJECTION) || sampsonTest))
            {
                // The correspondence side of the test is made once per call (128 B per correspondence) when the
                // table fits the budget; otherwise, and for the entry point that allocates nothing, every block
                // makes it from the fp32 records as it goes.
                if (!opsTried && opsFit) {
                    opsTried = true;
                    if (b->mfOps.reserve(opsBytes) == hipSuccess) {
                        g.mfOps = b->mfOps.as<uint4>();
                        const unsigned gxOps = (unsigned)((((maxN + 15) / 16) * 32 + 255) / 256);
                        if (gxOps > 0) hipLaunchKernelGGL(ransac_mfma_ops_kernel, dim3(gxOps, n_pairs), dim3(256), 0, st, g);
                    } else {
                        (void)hipGetLastError(); // no room: the blocks make their operands
                    }
                }
                const bool pre = g.mfOps != nullptr;
                handOver = pre && defer;
                const dim3 mgrid((unsigned)(((rows + 255) / 256) * n_pairs));
                if (handOver) {
                    // (the bound needs the packed test: the reprojection error under plain intrinsics)
                    g.heavyBound = (ph > 0 && g.filter && loss == SFM_LOSS_REPROJECTION) ? 1 : 0;
                    SFM_HIP(ctx, hipMemsetAsync(g.heavyCount, 0, 2 * sizeof(unsigned int), st));
                    hipLaunchKernelGGL(ransac_heavy_find_kernel, dim3((unsigned)(((rows + 255) / 256) * n_pairs)), dim3(256), 0, st, g);
                    if (g.heavyBound) hipLaunchKernelGGL(ransac_heavy_bound_kernel, dim3((unsigned)(8 * ctx->n_cus)), dim3(256), 0, st, g);
                    hipLaunchKernelGGL(ransac_heavy_assign_kernel, dim3((unsigned)ctx->n_cus), dim3(256), 0, st, g);
                }
                void (*mk)(const RansacArgs) = sampsonTest ? mfma_score_kernel<SFM_LOSS_SAMPSON>(pre, handOver)
                                                           : mfma_score_kernel<SFM_LOSS_REPROJECTION>(pre, handOver);
                if (handOver) { // the rows handed over, beside the scoring kernel
                    SFM_HIP(ctx, hipEventRecord(b->sideGo, st));
                    SFM_HIP(ctx, hipStreamWaitEvent(b->side, b->sideGo, 0));
                    const unsigned evalBlocks = (unsigned)(8 * ctx->n_cus);
                    if (sampsonTest)
                        hipLaunchKernelGGL(ransac_heavy_eval_kernel<SFM_LOSS_SAMPSON>, dim3(evalBlocks), dim3(256), 0, b->side, g);
                    else
                        hipLaunchKernelGGL(ransac_heavy_eval_kernel<SFM_LOSS_REPROJECTION>, dim3(evalBlocks), dim3(256), 0, b->side, g);
                    hipLaunchKernelGGL(ransac_heavy_sum_kernel, dim3((unsigned)(g.heavyMax / HSUM_ROWS)), dim3(64), 0, b->side, g);
                    SFM_HIP(ctx, hipEventRecord(b->sideDone, b->side));
                }
                hipLaunchKernelGGL(mk, mgrid, dim3(256), 0, st, g);
                if (handOver) SFM_HIP(ctx, hipStreamWaitEvent(st, b->sideDone, 0));
            }
            else
                hipLaunchKernelGGL(score, grid, dim3(256), 0, st, g, eb);
            if (Z > 1) {
                if ((int64_t)n_pairs * ((rows + 63) / 64) >= ctx->n_cus)
                    hipLaunchKernelGGL(ransac_sum_kernel<64>, dim3((rows + 63) / 64, n_pairs), dim3(SUM_THREADS), 0, st, g, eb);
                else
                    hipLaunchKernelGGL(ransac_sum_kernel<16>, dim3((rows + 15) / 16, n_pairs), dim3(SUM_THREADS), 0, st, g, eb);
            }
            if (!no_alloc) SFM_HIP(ctx, sfm_span_end(ctx, 1));
        }
        hipLaunchKernelGGL(ransac_select_kernel, dim3(n_pairs), dim3(256), 0, st, g);
    }
    if (!fit_only) {
        const int nBlk = (int)std::max<int64_t>(1, (maxN + 255) / 256);
        SFM_HIP(ctx, b->inlFlag.reserve((size_t)(total ? total : 1)));
        SFM_HIP(ctx, b->inlBlk.reserve((size_t)n_pairs * nBlk * sizeof(int)));
        unsigned char *flag = b->inlFlag.as<unsigned char>();
        int *blk = b->inlBlk.as<int>();
        hipLaunchKernelGGL(ransac_inlier_flag_kernel, dim3(nBlk, n_pairs), dim3(256), 0, st, g, flag, blk);
        hipLaunchKernelGGL(ransac_inlier_count_kernel, dim3(n_pairs), dim3(256), 0, st, g, blk, nBlk);
        hipLaunchKernelGGL(ransac_scan_kernel, dim3(1), dim3(1024), 0, st, g.inlCount, n_pairs, b->inlOff.as<int64_t>());
        hipLaunchKernelGGL(ransac_inlier_write_kernel, dim3(nBlk, n_pairs), dim3(256), 0, st, g, (const unsigned char *)flag,
                           (const int *)blk);
        hipLaunchKernelGGL(ransac_pack_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, g, b->packed.as<PackedBest>());
    }
    SFM_HIP(ctx, hipGetLastError());
    b->args = g;
    b->scored = !fit_only;
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_refit(sfm_ransac_batch *b)
{
    if (!b) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (!b->scored) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "refit: the batch holds no RANSAC result");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P1 = (size_t)(b->n_pairs ? b->n_pairs : 1);
    SFM_HIP(ctx, b->refitPose.reserve(P1 * 16 * sizeof(double)));
    SFM_HIP(ctx, b->refitUsed.reserve(P1 * sizeof(int)));
    SFM_HIP(ctx, b->refitPoints.reserve((size_t)(b->total ? b->total : 1) * 3 * sizeof(double)));
    if (b->n_pairs == 0) return SFM_OK;
    RefitArgs rf;
    rf.pose16 = b->refitPose.as<double>();
    rf.used = b->refitUsed.as<int>();
    rf.points = b->refitPoints.as<double>();
    hipLaunchKernelGGL(ransac_refit_kernel, dim3(b->n_pairs), dim3(256), 0, ctx->stream, b->args, rf);
    SFM_HIP(ctx, hipGetLastError());
    b->refitted = true;
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_handed_over(sfm_ransac_batch *b, int64_t *candidates_out, int64_t *passed_over_out,
                                            int64_t *handed_over_out)
{
    if (!b) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (!b->scored) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "handed_over: the batch holds no RANSAC result");
    unsigned int t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (b->args.heavyMax > 0) {
        SFM_HIP(ctx, hipSetDevice(ctx->device));
        SFM_HIP(ctx, hipMemcpyAsync(t, b->args.heavyCount, sizeof t, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (candidates_out) *candidates_out = t[2];
    if (passed_over_out) *passed_over_out = t[3];
    if (handed_over_out) *handed_over_out = t[4];
    return SFM_OK;
}

int sfm_ransac_batch_gather_view(sfm_ransac_batch *b, RansacGatherView *v)
{
    static_assert(sizeof(PackedBest) == SFM_RANSAC_RECORD_BYTES, "the gather moves the winners' records as they lie");
    if (!b || !v) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (!b->scored) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: the batch holds no RANSAC result");
    v->ctx = ctx;
    v->n_pairs = b->n_pairs;
    v->inlOff = b->inlOff.as<int64_t>();
    v->inliers = b->inliers.as<int32_t>();
    v->packed = b->packed.p;
    v->fence = &b->readFence;
    return SFM_OK;
}

int sfm_ransac_batch_refit_view(sfm_ransac_batch *b, RefitView *v, sfm_ctx **ctx_out)
{
    if (!b || !v || !ctx_out) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    *ctx_out = ctx;
    if (!b->scored || (!b->refitted && b->n_pairs > 0))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "refit: sfm_ransac_batch_refit has not run on this batch's result");
    v->n_pairs = b->n_pairs;
    v->total = b->total;
    v->corrOffHost = b->corrOffHost.data();
    v->corrOff = b->corrOff.as<int64_t>();
    v->inlCount = b->inlCount.as<int32_t>();
    v->inlOff = b->inlOff.as<int64_t>();
    v->inliers = b->inliers.as<int32_t>();
    v->pose16 = b->refitPose.as<double>();
    v->used = b->refitUsed.as<int32_t>();
    v->points = b->refitPoints.as<double>();
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_refit_fetch(sfm_ransac_batch *b, double *pose16_out, int32_t *used_out,
                                            double *points_out)
{
    if (!b) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (!b->refitPose.p) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "refit: sfm_ransac_batch_refit has not run");
    if (b->n_pairs == 0) return SFM_OK;
    const size_t P = (size_t)b->n_pairs;
    if (pose16_out)
        SFM_HIP(ctx, hipMemcpyAsync(pose16_out, b->refitPose.p, P * 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (used_out)
        SFM_HIP(ctx, hipMemcpyAsync(used_out, b->refitUsed.p, P * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (points_out && b->total > 0)
        SFM_HIP(ctx, hipMemcpyAsync(points_out, b->refitPoints.p, (size_t)b->total * 3 * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_run(sfm_ctx *ctx, int n_pairs, const int64_t *corr_offsets, const double *x,
                                    const double *y, const double *K, const int32_t *samples, int n_hyp,
                                    double max_squared_error, double success_prob, int adaptive, int loss,
                                    sfm_ransac_batch **batch_inout)
{
    return ransac_run(ctx, n_pairs, corr_offsets, x, y, K, samples, n_hyp, max_squared_error, success_prob, adaptive,
                      loss, false, false, batch_inout);
}

extern "C" int sfm_ransac_batch_run_device(sfm_ctx *ctx, int n_pairs, const int64_t *corr_offsets,
                                           const double *x_device, const double *y_device, const double *K,
                                           const int32_t *samples_device, int n_hyp, double max_squared_error,
                                           double success_prob, int adaptive, int loss,
                                           sfm_ransac_batch **batch_inout)
{
    return ransac_run(ctx, n_pairs, corr_offsets, x_device, y_device, K, samples_device, n_hyp, max_squared_error,
                      success_prob, adaptive, loss, true, false, batch_inout);
}

#define D2H(dst, buf, bytes)                                                                             \
    do {                                                                                                 \
        if ((dst) && (bytes) > 0)                                                                        \
            SFM_HIP(ctx, hipMemcpyAsync((dst), (buf).p, (bytes), hipMemcpyDeviceToHost, ctx->stream));   \
    } while (0)

extern "C" int sfm_ransac_batch_best(sfm_ransac_batch *b, int32_t *best_hyp_out, int32_t *best_count_out,
                                     double *best_toterr_out, double *pose12_out, int32_t *iterations_out)
{
    if (!b) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (!b->scored) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: the batch holds no RANSAC result");
    const size_t P = (size_t)b->n_pairs;
    if (P == 0) return SFM_OK;
    // one copy of the dense per-pair records (winner, count, error sum, iterations, winner's pose)
    std::vector<PackedBest> rec(P);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemcpyAsync(rec.data(), b->packed.p, P * sizeof(PackedBest), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t p = 0; p < P; p++) {
        if (best_hyp_out) best_hyp_out[p] = rec[p].bestH;
        if (best_count_out) best_count_out[p] = rec[p].bestCount;
        if (best_toterr_out) best_toterr_out[p] = rec[p].bestErr;
        if (iterations_out) iterations_out[p] = rec[p].iterations;
        if (pose12_out) memcpy(pose12_out + 12 * p, rec[p].pose12, sizeof rec[p].pose12);
    }
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_inlier_offsets(sfm_ransac_batch *b, int64_t *offsets_out)
{
    if (!b || !offsets_out) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (b->n_pairs == 0) {
        offsets_out[0] = 0;
        return SFM_OK;
    }
    D2H(offsets_out, b->inlOff, (size_t)(b->n_pairs + 1) * sizeof(int64_t));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_inliers(sfm_ransac_batch *b, int32_t *inliers_out)
{
    if (!b) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = b->ctx;
    if (b->n_pairs == 0) return SFM_OK;
    int64_t tot = 0;
    SFM_HIP(ctx, hipMemcpyAsync(&tot, b->inlOff.as<int64_t>() + b->n_pairs, sizeof tot, hipMemcpyDeviceToHost,
                                ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tot > 0) {
        if (!inliers_out) return SFM_ERR_INVALID_ARG;
        D2H(inliers_out, b->inliers, (size_t)tot * sizeof(int32_t));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFM_OK;
}

// The single-pair entry points keep their batch (device buffers) in the context between calls.
static sfm_ransac_batch **pair_batch(sfm_ctx *ctx) { return &ctx->pair_batch; }

void sfm_pair_batch_free(sfm_ctx *ctx)
{
    if (ctx->pair_batch) sfm_ransac_batch_destroy(ctx->pair_batch);
    ctx->pair_batch = nullptr;
}

extern "C" int sfm_ransac_pair(sfm_ctx *ctx, const double *x, const double *y, int n, const double *K,
                               const int32_t *samples, int n_hyp, int min_n, double max_squared_error,
                               double success_prob, int adaptive, int loss, int32_t *inliers_out, int capacity,
                               int *count_out, int *best_hyp_out, double *pose_out, int32_t *counts_out,
                               double *toterr_out, int *iterations_out)
{
    if (!ctx || !count_out) return SFM_ERR_INVALID_ARG;
    *count_out = 0;
    if (best_hyp_out) *best_hyp_out = -1;
    if (iterations_out) *iterations_out = 0;
    if (n < 0 || n_hyp < 0 || capacity < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: negative size");
    if (min_n != 8)
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "ransac: minN = %d has no kernel (the 8-point model needs exactly 8)", min_n);
    if (n < min_n) return SFM_OK; // RANSAC.cpp:148-149
    const int64_t off[2] = {0, n};
    // every row is scored when the caller asks for the per-row outputs
    int rc = ransac_run(ctx, 1, off, x, y, K, samples, n_hyp, max_squared_error, success_prob, adaptive, loss, false,
                        false, pair_batch(ctx), !counts_out && !toterr_out);
    if (rc != SFM_OK) return rc;
    sfm_ransac_batch *b = *pair_batch(ctx);
    // The winner's record and min(n, capacity) entries of its inlier list (the count is not known yet) go to
    // a pinned buffer in one go: one wait for the whole call.
    const size_t nInl = (size_t)std::min(n, capacity);
    const size_t rowBytes = (counts_out ? (size_t)n_hyp * sizeof(int32_t) : 0) + (toterr_out ? (size_t)n_hyp * sizeof(double) : 0);
    SFM_HIP(ctx, b->stage.reserve(sizeof(PackedBest) + nInl * sizeof(int32_t) + rowBytes + 16));
    unsigned char *stage = b->stage.as<unsigned char>();
    hipStream_t st = ctx->stream;
    SFM_HIP(ctx, hipMemcpyAsync(stage, b->packed.p, sizeof(PackedBest), hipMemcpyDeviceToHost, st));
    size_t at = sizeof(PackedBest);
    if (nInl > 0) SFM_HIP(ctx, hipMemcpyAsync(stage + at, b->inliers.p, nInl * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    at += (nInl * sizeof(int32_t) + 7) / 8 * 8;
    const size_t atTot = at;
    if (toterr_out && n_hyp > 0) {
        SFM_HIP(ctx, hipMemcpyAsync(stage + at, b->toterr.p, (size_t)n_hyp * sizeof(double), hipMemcpyDeviceToHost, st));
        at += (size_t)n_hyp * sizeof(double);
    }
    const size_t atCnt = at;
    if (counts_out && n_hyp > 0)
        SFM_HIP(ctx, hipMemcpyAsync(stage + at, b->counts.p, (size_t)n_hyp * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    PackedBest rec;
    memcpy(&rec, stage, sizeof rec);
    if (rec.bestCount > capacity) SFM_FAIL(ctx, SFM_ERR_CAPACITY, "ransac: inlier capacity too small");
    if (rec.bestCount > 0) {
        if (!inliers_out) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: null inlier buffer");
        memcpy(inliers_out, stage + sizeof(PackedBest), (size_t)rec.bestCount * sizeof(int32_t));
    }
    if (toterr_out && n_hyp > 0) memcpy(toterr_out, stage + atTot, (size_t)n_hyp * sizeof(double));
    if (counts_out && n_hyp > 0) memcpy(counts_out, stage + atCnt, (size_t)n_hyp * sizeof(int32_t));
    *count_out = rec.bestCount;
    if (best_hyp_out) *best_hyp_out = rec.bestH;
    if (iterations_out) *iterations_out = rec.iterations;
    if (pose_out) { // 4x4 column-major [R|T; 0 0 0 1]
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) pose_out[r + 4 * c] = (r == c) ? 1.0 : 0.0;
        if (rec.bestH >= 0)
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) pose_out[r + 4 * c] = rec.pose12[3 * r + c];
                pose_out[r + 12] = rec.pose12[9 + r];
            }
    }
    return SFM_OK;
}

extern "C" int sfm_eight_point_fit(sfm_ctx *ctx, const double *x, const double *y, int n, const int32_t *samples,
                                   int n_hyp, double *pose12_out)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    if (n < 8) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG,
                        "EightPointFromSubset: There have to be at least 8 pair-to-pair-correspondences (was: %d)", n);
    if (n_hyp < 0 || (n_hyp > 0 && !pose12_out)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "fit: bad output");
    const int64_t off[2] = {0, n};
    const double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc = ransac_run(ctx, 1, off, x, y, K, samples, n_hyp, 0.0, 0.0, 0, SFM_LOSS_REPROJECTION, false, true,
                        pair_batch(ctx));
    sfm_ransac_batch *b = *pair_batch(ctx);
    if (rc == SFM_OK && n_hyp > 0) {
        hipError_t e = hipMemcpyAsync(pose12_out, b->pose12.p, (size_t)n_hyp * 12 * sizeof(double),
                                      hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->last_error = std::string("fit: copy failed: ") + hipGetErrorString(e);
            rc = SFM_ERR_HIP;
        }
    }
    return rc;
}

extern "C" int sfm_eight_point_from_subset(sfm_ctx *ctx, const double *calc1, const double *calc2, int n_calc, const double *all1,
                                           const double *all2, int n_all, double *pose16_out, double *points_out)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    if (n_calc < 8) // eightPoint.cpp:228-231
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG,
                 "EightPointFromSubset: There have to be at least 8 pair-to-pair-correspondences (was: %d) to approximate a pose.", n_calc);
    if (!calc1 || !calc2 || n_all < 0 || (n_all > 0 && (!all1 || !all2))) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "EightPointFromSubset: null input");
    const double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double pose12[12];
    hipStream_t st = ctx->stream;
    sfm_ransac_batch *b = nullptr;
    if (n_calc == 8) { // the fit kernel: one row of a sample table
        const int64_t off[2] = {0, 8};
        const int32_t rows[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        int rc = ransac_run(ctx, 1, off, calc1, calc2, K, rows, 1, 0.0, 0.0, 0, SFM_LOSS_REPROJECTION, false, true, pair_batch(ctx));
        if (rc != SFM_OK) return rc;
        b = *pair_batch(ctx);
        SFM_HIP(ctx, hipMemcpyAsync(pose12, b->pose12.p, sizeof pose12, hipMemcpyDeviceToHost, st));
        if (!(n_all > 0 && points_out)) SFM_HIP(ctx, hipStreamSynchronize(st));
    } else { // more than eight sample points: the n x 9 system, as eightPointAlgorithm solves it (eightPoint.cpp:233-344 is :64-185)
        double pose16[16];
        int rc = sfm_eight_point_algorithm(ctx, calc1, calc2, n_calc, pose16, nullptr);
        if (rc != SFM_OK) return rc;
        b = *pair_batch(ctx);
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) pose12[3 * r + c] = pose16[r + 4 * c];
            pose12[9 + r] = pose16[r + 12];
        }
        if (n_all > 0 && points_out) {
            SFM_HIP(ctx, b->pose12.reserve(sizeof pose12));
            SFM_HIP(ctx, hipMemcpyAsync(b->pose12.p, pose12, sizeof pose12, hipMemcpyHostToDevice, st));
        }
    }
    if (n_all > 0 && points_out) { // the two point sets are staged behind each other in the (unused) error buffer
        SFM_HIP(ctx, b->errBuf.reserve((size_t)n_all * 2 * sizeof(double2)));
        SFM_HIP(ctx, b->refitPoints.reserve((size_t)n_all * 3 * sizeof(double)));
        double2 *d1 = b->errBuf.as<double2>(), *d2 = d1 + n_all;
        SFM_HIP(ctx, hipMemcpyAsync(d1, all1, (size_t)n_all * sizeof(double2), hipMemcpyHostToDevice, st));
        SFM_HIP(ctx, hipMemcpyAsync(d2, all2, (size_t)n_all * sizeof(double2), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ransac_triangulate_kernel, dim3((n_all + 255) / 256), dim3(256), 0, st, b->pose12.as<double>(), d1, d2, n_all,
                           b->refitPoints.as<double>());
        SFM_HIP(ctx, hipGetLastError());
        SFM_HIP(ctx, hipMemcpyAsync(points_out, b->refitPoints.p, (size_t)n_all * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        SFM_HIP(ctx, hipStreamSynchronize(st));
    }
    if (pose16_out) {
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) pose16_out[r + 4 * c] = (r == c) ? 1.0 : 0.0;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) pose16_out[r + 4 * c] = pose12[3 * r + c];
            pose16_out[r + 12] = pose12[9 + r];
        }
    }
    return SFM_OK;
}

extern "C" int sfm_eight_point_algorithm(sfm_ctx *ctx, const double *x, const double *y, int n, double *pose16_out,
                                         double *points_out)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    if (n < 8) // eightPoint.cpp:59-62
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG,
                 "EightPointAlgorithm: There have to be at least 8 pair-to-pair-correspondences (was: %d) to approximate a pose.", n);
    if (!x || !y) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "EightPointAlgorithm: null input");
    const int64_t off[2] = {0, n};
    const double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    // no hypotheses: no inliers, so the refit takes every point
    int rc = ransac_run(ctx, 1, off, x, y, K, nullptr, 0, 0.0, 0.0, 0, SFM_LOSS_REPROJECTION, false, false,
                        pair_batch(ctx));
    sfm_ransac_batch *b = *pair_batch(ctx);
    if (rc == SFM_OK) rc = sfm_ransac_batch_refit(b);
    if (rc == SFM_OK) rc = sfm_ransac_batch_refit_fetch(b, pose16_out, nullptr, points_out);
    return rc;
}

// ---------------------------------------------------------------------------------------
// sfm_ransac_pair for callers that cannot allocate
// ---------------------------------------------------------------------------------------
// Same work, same kernels, same results as sfm_ransac_pair, but every device buffer the kernels need
// beyond the caller's inputs and outputs (device copies of x, y and the sample table, the prepared
// records, per-row poses, counts and error sums, selection state, inlier flags) is a fixed slice of one
// file-scope device array, so the call performs no hipMalloc / hipFree and creates no events.  Each GPU
// holds its own copy of the array (about 3.5 MB) until the process ends; its address on a device is
// looked up once, when the first context on that device is created.  All calls on one GPU share the
// array: calls that overlap on one GPU (two contexts, two host threads) race.
#define SFM_STATIC_MAX_CORR 16384
#define SFM_STATIC_MAX_HYP 8192
#define SFM_STATIC_MAX_DEVICES 64

namespace {
constexpr size_t slice256(size_t bytes) { return (bytes + 255) / 256 * 256; }
constexpr size_t ST_XY = slice256((size_t)SFM_STATIC_MAX_CORR * sizeof(double2));
constexpr size_t ST_SAMPLES = slice256((size_t)SFM_STATIC_MAX_HYP * 8 * sizeof(int32_t));
constexpr size_t ST_REC = slice256((size_t)SFM_STATIC_MAX_CORR * REC_D * sizeof(double));
constexpr size_t ST_REC32 = slice256(((size_t)20 * ((SFM_STATIC_MAX_CORR + 6) / 4) + 40) * sizeof(float));
constexpr size_t ST_POSE = slice256((size_t)SFM_STATIC_MAX_HYP * 12 * sizeof(double));
constexpr size_t ST_COUNTS = slice256((size_t)SFM_STATIC_MAX_HYP * sizeof(int));
constexpr size_t ST_TOTERR = slice256((size_t)SFM_STATIC_MAX_HYP * sizeof(double));
constexpr size_t ST_INLIERS = slice256((size_t)SFM_STATIC_MAX_CORR * sizeof(int32_t));
constexpr size_t ST_FLAGS = slice256((size_t)SFM_STATIC_MAX_CORR);
constexpr size_t ST_BLK = slice256((size_t)((SFM_STATIC_MAX_CORR + 255) / 256) * sizeof(int));
constexpr size_t ST_SMALL = 256; // per-pair scalars of a one-pair batch, two offsets
constexpr size_t STATIC_SCRATCH_BYTES = 2 * ST_XY + ST_SAMPLES + ST_REC + ST_REC32 + ST_POSE + ST_COUNTS + ST_TOTERR +
                                        ST_INLIERS + ST_FLAGS + ST_BLK + 11 * ST_SMALL;
} // namespace

static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

// Host side: where `scratch` lives on each device (the array's name is no device address in host code),
// and a batch per device whose buffers are slices of it.
static unsigned char *g_scratchPtr[SFM_STATIC_MAX_DEVICES];
static sfm_ransac_batch *g_staticBatch[SFM_STATIC_MAX_DEVICES];

static std::mutex g_staticMutex; // contexts may be created from several host threads at once

int sfm_ransac_static_setup(int device)
{
    if (device < 0 || device >= SFM_STATIC_MAX_DEVICES) return SFM_OK; // such a device has no static entry point
    std::lock_guard<std::mutex> lock(g_staticMutex);
    if (g_scratchPtr[device]) return SFM_OK;
    void *ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return SFM_ERR_HIP;
    sfm_ransac_batch *b = new (std::nothrow) sfm_ransac_batch();
    if (!b) return SFM_ERR_OOM;
    unsigned char *at = static_cast<unsigned char *>(ptr);
    auto slice = [&](DevBuf &d, size_t bytes) {
        d.p = at;
        d.cap = bytes;
        d.fixed = true;
        at += bytes;
    };
    slice(b->x, ST_XY);
    slice(b->y, ST_XY);
    slice(b->samples, ST_SAMPLES);
    slice(b->rec, ST_REC);
    slice(b->rec32, ST_REC32);
    slice(b->pose12, ST_POSE);
    slice(b->counts, ST_COUNTS);
    slice(b->toterr, ST_TOTERR);
    slice(b->inliers, ST_INLIERS);
    slice(b->inlFlag, ST_FLAGS);
    slice(b->inlBlk, ST_BLK);
    slice(b->corrOff, ST_SMALL);
    slice(b->bestH, ST_SMALL);
    slice(b->bestCount, ST_SMALL);
    slice(b->bestErr, ST_SMALL);
    slice(b->iterations, ST_SMALL);
    slice(b->inlCount, ST_SMALL);
    slice(b->inlOff, ST_SMALL);
    slice(b->active, ST_SMALL);
    slice(b->maxIter, ST_SMALL);
    slice(b->clock, ST_SMALL);
    slice(b->packed, ST_SMALL);
    if ((size_t)(at - static_cast<unsigned char *>(ptr)) != STATIC_SCRATCH_BYTES) { // the table above and the size disagree
        delete b;
        return SFM_ERR_HIP;
    }
    b->corrOffHost.reserve(2);
    g_staticBatch[device] = b;
    g_scratchPtr[device] = static_cast<unsigned char *>(ptr);
    return SFM_OK;
}

extern "C" int sfm_ransac_pair_static(sfm_ctx *ctx, const double *x, const double *y, int n, const double *K,
                                      const int32_t *samples, int n_hyp, int min_n, double max_squared_error,
                                      double success_prob, int adaptive, int loss, int32_t *inliers_out, int capacity,
                                      int *count_out, int *best_hyp_out, double *pose_out, int32_t *counts_out,
                                      double *toterr_out, int *iterations_out)
{
    if (!ctx || !count_out) return SFM_ERR_INVALID_ARG;
    *count_out = 0;
    if (best_hyp_out) *best_hyp_out = -1;
    if (iterations_out) *iterations_out = 0;
    if (n < 0 || n_hyp < 0 || capacity < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: negative size");
    if (min_n != 8)
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "ransac: minN = %d has no kernel (the 8-point model needs exactly 8)", min_n);
    if (n > SFM_STATIC_MAX_CORR || n_hyp > SFM_STATIC_MAX_HYP)
        SFM_FAIL(ctx, SFM_ERR_CAPACITY,
                 "ransac (static workspace): %d correspondences x %d sample rows exceed its fixed size of %d x %d; "
                 "use sfm_ransac_pair",
                 n, n_hyp, SFM_STATIC_MAX_CORR, SFM_STATIC_MAX_HYP);
    if (ctx->device >= SFM_STATIC_MAX_DEVICES || !g_staticBatch[ctx->device])
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "ransac (static workspace): not set up on device %d", ctx->device);
    if (n < min_n) return SFM_OK; // RANSAC.cpp:148-149
    sfm_ransac_batch *b = g_staticBatch[ctx->device];
    b->ctx = ctx; // the workspace belongs to the device, whichever context calls
    const int64_t off[2] = {0, n};
    int rc = ransac_run(ctx, 1, off, x, y, K, samples, n_hyp, max_squared_error, success_prob, adaptive, loss, false, false, &b,
                        !counts_out && !toterr_out, true);
    if (rc != SFM_OK) return rc;
    hipStream_t st = ctx->stream;
    int32_t bh = -1, bc = 0, it = 0;
    SFM_HIP(ctx, hipMemcpyAsync(&bh, b->bestH.p, sizeof bh, hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipMemcpyAsync(&bc, b->bestCount.p, sizeof bc, hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipMemcpyAsync(&it, b->iterations.p, sizeof it, hipMemcpyDeviceToHost, st));
    if (counts_out && n_hyp > 0)
        SFM_HIP(ctx, hipMemcpyAsync(counts_out, b->counts.p, (size_t)n_hyp * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (toterr_out && n_hyp > 0)
        SFM_HIP(ctx, hipMemcpyAsync(toterr_out, b->toterr.p, (size_t)n_hyp * sizeof(double), hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    if (bc > capacity) SFM_FAIL(ctx, SFM_ERR_CAPACITY, "ransac: inlier capacity too small");
    double pose12[12];
    if (bh >= 0 && pose_out)
        SFM_HIP(ctx, hipMemcpyAsync(pose12, b->pose12.as<double>() + (size_t)bh * 12, sizeof pose12, hipMemcpyDeviceToHost, st));
    if (bc > 0) {
        if (!inliers_out) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: null inlier buffer");
        SFM_HIP(ctx, hipMemcpyAsync(inliers_out, b->inliers.p, (size_t)bc * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    SFM_HIP(ctx, hipStreamSynchronize(st));
    *count_out = bc;
    if (best_hyp_out) *best_hyp_out = bh;
    if (iterations_out) *iterations_out = it;
    if (pose_out) { // 4x4 column-major [R|T; 0 0 0 1]
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) pose_out[r + 4 * c] = (r == c) ? 1.0 : 0.0;
        if (bh >= 0)
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) pose_out[r + 4 * c] = pose12[3 * r + c];
                pose_out[r + 12] = pose12[9 + r];
            }
    }
    return SFM_OK;
}
