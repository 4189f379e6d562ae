/*
 * sfm_oracle.c -- CPU restatement of the reference's pairwise matcher and
 * RANSAC two-view path.  TEST INFRASTRUCTURE ONLY: nothing in the shipped
 * library links, loads or calls this file; only tests/, the smoke check and
 * the cpu_baseline leg of bench.py do, and only as the checker / the timed
 * CPU baseline.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures
 * for this path, and it cannot be compiled in this environment (it needs
 * Eigen3, OpenCV, TBB and Ceres headers, none of which are installed), so no
 * output of the reference itself pins this file.  What pins it instead:
 *   - the matcher uses no third-party arithmetic: every fp32 operation below
 *     is the one the reference source spells out, in the same order
 *     (build with -ffp-contract=off, no -ffast-math);
 *   - the linear algebra that the reference delegates to Eigen >= 3.3 (a
 *     dependency absent from the reference tree and from this environment;
 *     CMakeLists.txt:27 pins no version) is restated from Eigen's published
 *     algorithms, operation by operation: JacobiSVD of the 8x9 / n x 9 system
 *     and of the 3x3 essential matrix (column-pivoted Householder QR as the
 *     preconditioner, two-sided Jacobi sweeps, sign fix, descending sort), so
 *     that signs and order -- which decide a tied cheirality vote -- are
 *     Eigen's; and colPivHouseholderQr().solve of the 3x2 triangulation.
 *     Tests cross-check all of them against LAPACK (numpy.linalg.svd,
 *     SVD-based least squares) and count, over many seeds, the (hypothesis,
 *     correspondence) inlier decisions on which the restatement and LAPACK
 *     disagree.
 *
 * Reference files followed (paths relative to the reference tree):
 *   submodules/match/match.hpp:18-46     dotProduct, cosineSimilarity
 *   submodules/match/match.cpp:24-94     matchTwoSided
 *   submodules/solve/eightPoint.cpp:225-372  eightPointAlgorithmFromSubset
 *   submodules/solve/eightPoint.cpp:57-223   eightPointAlgorithm (refit)
 *   submodules/solve/solve.hpp:85-95,119-124 reprojectionError, default loss
 *   submodules/solve/RANSAC.cpp:14-32,146-270 safeCastToInt, RANSAC
 *   submodules/scene.cpp:495-501         normalizePoints
 *   submodules/SfM.hpp:8,23              REAL = double, EPSILON = 1e-6
 *
 * All matrices cross this API in Eigen's default layout: column-major.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SFM_EPSILON 1e-6 /* SfM.hpp:23 */

/* ------------------------------------------------------------------------ */
/* Matching                                                                  */
/* ------------------------------------------------------------------------ */

typedef float (*similarity_fn)(const float *a, const float *b, size_t n);

/* match.hpp:18-27 */
static float dot_product(const float *a, const float *b, size_t n)
{
    float result = 0.0f;
    for (size_t i = 0; i < n; ++i)
        result += a[i] * b[i];
    return result;
}

/* match.hpp:32-46: one loop, three fp32 running sums, then
 * dot / (sqrt(normA) * sqrt(normB)) with float sqrt. */
static float cosine_similarity(const float *a, const float *b, size_t n)
{
    float dot = 0.0f;
    float normA = 0.0f;
    float normB = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        dot += a[i] * b[i];
        normA += a[i] * a[i];
        normB += b[i] * b[i];
    }
    return dot / (sqrtf(normA) * sqrtf(normB));
}

/* The reference reaches its metric through a std::function; keep the call
 * indirect so the timed baseline pays the same kind of dispatch. */
static similarity_fn volatile g_metric_table[2] = {cosine_similarity, dot_product};

int sfm_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void sfm_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Single similarity value, for tests. metric: 0 cosine, 1 dot. */
float sfm_oracle_similarity(const float *a, const float *b, int dim, int metric)
{
    return g_metric_table[metric ? 1 : 0](a, b, (size_t)dim);
}

/*
 * match.cpp:24-94 (matchTwoSided).  descA: nA x dim row-major floats, ptsA:
 * nA x 2 doubles (pixels); same for B.  Writes matches1[nA], matches2[nB]
 * (either may be NULL) and up to min(nA,nB) (i, j) pairs into pairs_out.
 * Returns the number of pairs, or -1 when the similarity matrix cannot be
 * allocated.  The nA x nB float matrix is materialised column-major exactly
 * as the reference does.
 */
int sfm_oracle_match(const float *descA, const double *ptsA, int nA,
                     const float *descB, const double *ptsB, int nB, int dim,
                     int metric, float threshold, float maxDistancePxSquared,
                     int *matches1_out, int *matches2_out, int *pairs_out)
{
    const int lenA = nA, lenB = nB;
    similarity_fn similarityMetric = g_metric_table[metric ? 1 : 0];

    size_t cells = (size_t)(lenA > 0 ? lenA : 0) * (size_t)(lenB > 0 ? lenB : 0);
    float *mat = (float *)malloc((cells ? cells : 1) * sizeof(float));
    int *matches1 = (int *)malloc((size_t)(lenA > 0 ? lenA : 1) * sizeof(int));
    int *matches2 = (int *)malloc((size_t)(lenB > 0 ? lenB : 1) * sizeof(int));
    if (!mat || !matches1 || !matches2) {
        free(mat);
        free(matches1);
        free(matches2);
        return -1;
    }

    /* match.cpp:34-61: similarity matrix + best row per column */
#pragma omp parallel for schedule(static)
    for (int idx2 = 0; idx2 < lenB; idx2++) {
        int maxIdx = -1;
        float maxSimilarity = 0;
        for (int idx1 = 0; idx1 < lenA; idx1++) {
            const float *d1 = descA + (size_t)idx1 * dim;
            const float *d2 = descB + (size_t)idx2 * dim;
            float similarity = similarityMetric(d1, d2, (size_t)dim);
            double dx = ptsA[2 * idx1] - ptsB[2 * idx2];
            double dy = ptsA[2 * idx1 + 1] - ptsB[2 * idx2 + 1];
            double squaredNorm = dx * dx + dy * dy;
            if (similarity > threshold && squaredNorm < maxDistancePxSquared) {
                mat[(size_t)idx2 * lenA + idx1] = similarity;
                if (similarity > maxSimilarity) {
                    maxSimilarity = similarity;
                    maxIdx = idx1;
                }
            } else {
                mat[(size_t)idx2 * lenA + idx1] = 0;
            }
        }
        matches2[idx2] = maxIdx;
    }

    /* match.cpp:64-79: best column per row, read back from the matrix */
#pragma omp parallel for schedule(static)
    for (int idx1 = 0; idx1 < lenA; idx1++) {
        int maxIdx = -1;
        float maxSimilarity = 0;
        for (int idx2 = 0; idx2 < lenB; idx2++) {
            float v = mat[(size_t)idx2 * lenA + idx1];
            if (v > maxSimilarity) {
                maxSimilarity = v;
                maxIdx = idx2;
            }
        }
        matches1[idx1] = maxIdx;
    }

    /* match.cpp:84-91: mutual check, ascending i */
    int count = 0;
    for (int i = 0; i < lenA; i++) {
        int matchIn2 = matches1[i];
        if (matchIn2 != -1 && matches2[matchIn2] == i) {
            if (pairs_out) {
                pairs_out[2 * count] = i;
                pairs_out[2 * count + 1] = matchIn2;
            }
            count++;
        }
    }

    if (matches1_out && lenA > 0)
        memcpy(matches1_out, matches1, (size_t)lenA * sizeof(int));
    if (matches2_out && lenB > 0)
        memcpy(matches2_out, matches2, (size_t)lenB * sizeof(int));
    free(mat);
    free(matches1);
    free(matches2);
    return count;
}

/* ------------------------------------------------------------------------ */
/* Eigen's JacobiSVD, restated                                               */
/* ------------------------------------------------------------------------ */
/*
 * The reference takes its null vector and the factors of the essential matrix from
 * Eigen::JacobiSVD (eightPoint.cpp:85-86,93-96 and :254-255,262-265).  Which of the four (R, T)
 * candidates comes first -- and so which one a tied cheirality vote keeps (:296-337) -- depends on
 * the signs and the order that algorithm happens to return, so the algorithm itself is restated
 * here from Eigen's published sources (Eigen 3.3 / 3.4, the versions CMakeLists.txt:27 admits;
 * the two agree for real matrices): src/SVD/JacobiSVD.h (compute, qr_preconditioner_impl for
 * ColPivHouseholderQRPreconditioner, the default), src/Jacobi/Jacobi.h (makeJacobi,
 * apply_rotation_in_the_plane), src/misc/RealSvd2x2.h (real_2x2_jacobi_svd),
 * src/QR/ColPivHouseholderQR.h (computeInPlace), src/Householder/Householder.h (makeHouseholder,
 * applyHouseholderOnTheLeft) and HouseholderSequence.h (evalTo).
 *
 *   compute(A):  scale = max |A_ij| (1 when that is 0); everything below works on A / scale.
 *     rows < cols (the 8x9 system of a sample): column-pivoted Householder QR of A^T (9x8);
 *        V = its 9x9 Q; the Jacobi sweeps then rotate V's first 8 columns only, and the final
 *        sort moves only those, so matrixV().col(8) is Q's last column, H_0 H_1 ... H_7 e_8,
 *        formed by evalTo from the last reflector to the first.
 *     rows > cols (a refit on more than nine points): column-pivoted Householder QR of A; the
 *        work matrix is the upper triangle R (9x9), V starts as the column permutation.
 *     rows == cols: the work matrix is A / scale, V starts as the identity.
 *     sweeps: for p = 1..m-1, q = 0..p-1: when |W_pq| or |W_qp| exceeds
 *        max(DBL_MIN, 2 eps max|diagonal entry seen so far|): real_2x2_jacobi_svd gives a left
 *        and a right rotation; W <- J_left W, U <- U J_left^T, W <- W J_right, V <- V J_right;
 *        repeated until a whole sweep finds nothing to do.
 *     then S_i = |W_ii|, U's column i negated where W_ii < 0, S *= scale, and a selection sort
 *        into descending order (the first of equal maxima), swapping the columns of U and V.
 *
 * PARITY UNPINNED at the level of single roundings: Eigen is absent here, so no output of Eigen
 * pins this file.  What is restated is the sequence of operations; where Eigen leaves the order
 * of a sum to its vectorised kernels (the squared norms and the reflector-times-block products
 * of the QR) the sums below run top to bottom, and an Eigen built with fused multiply-adds
 * would round differently again.  Such differences are of the size of a rounding error; they do
 * not change a sign, a pivot or the order of singular values except where two of them agree to
 * rounding.
 */
typedef struct {
    double c, s;
} jrot_t;

/* Jacobi.h, JacobiRotation::makeJacobi(x, y, z): the rotation that diagonalises the symmetric
 * 2x2 matrix [x y; y z]. */
static void make_jacobi(double x, double y, double z, jrot_t *j)
{
    const double deno = 2.0 * fabs(y);
    if (deno < DBL_MIN) {
        j->c = 1.0;
        j->s = 0.0;
        return;
    }
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    double t;
    if (tau > 0.0)
        t = 1.0 / (tau + w);
    else
        t = 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    j->s = -sign_t * (y / fabs(y)) * fabs(t) * n;
    j->c = n;
}

/* RealSvd2x2.h, real_2x2_jacobi_svd on m = [m00 m01; m10 m11]: J_left m J_right is diagonal. */
static void real_2x2_jacobi_svd(double m00, double m01, double m10, double m11, jrot_t *j_left, jrot_t *j_right)
{
    jrot_t rot1;
    const double t = m00 + m11;
    const double d = m10 - m01;
    if (fabs(d) < DBL_MIN) {
        rot1.s = 0.0;
        rot1.c = 1.0;
    } else {
        const double u = t / d;
        const double tmp = sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    /* m.applyOnTheLeft(0, 1, rot1); only m00, m01, m11 of the result are used */
    double r00 = m00, r01 = m01, r11 = m11;
    if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
        r00 = rot1.c * m00 + rot1.s * m10;
        r01 = rot1.c * m01 + rot1.s * m11;
        r11 = -rot1.s * m01 + rot1.c * m11;
    }
    make_jacobi(r00, r01, r11, j_right);
    /* *j_left = rot1 * j_right->transpose() */
    const double oc = j_right->c, os = -j_right->s;
    j_left->c = rot1.c * oc - rot1.s * os;
    j_left->s = rot1.c * os + rot1.s * oc;
}

/* Jacobi.h, apply_rotation_in_the_plane: (x_i, y_i) <- (c x_i + s y_i, -s x_i + c y_i). */
static void rotate_plane(double *x, double *y, int n, int stride, double c, double s)
{
    if (c == 1.0 && s == 0.0)
        return;
    for (int i = 0; i < n; i++) {
        const double xi = x[i * stride], yi = y[i * stride];
        x[i * stride] = c * xi + s * yi;
        y[i * stride] = -s * xi + c * yi;
    }
}

/*
 * JacobiSVD::compute from "step 2" on, for a square m x m work matrix W (row-major, already
 * divided by scale): the sweeps, the sign fix and the sort.  U (urows x m, row-major) and V
 * (vrows x m, row-major) accumulate the rotations from whatever they hold on entry; either may be
 * NULL.  S receives the singular values, descending.
 */
static void jacobi_svd_square(int m, double *W, double *U, int urows, double *V, int vrows, double scale, double *S)
{
    const double precision = 2.0 * DBL_EPSILON;
    const double considerAsZero = DBL_MIN;
    double maxDiagEntry = 0.0;
    for (int i = 0; i < m; i++)
        if (fabs(W[i * m + i]) > maxDiagEntry)
            maxDiagEntry = fabs(W[i * m + i]);
    int finished = 0;
    /* (Eigen has no sweep limit: a NaN fails every comparison and ends the loop at once; the cap
     * only guards this restatement against an input Eigen would spin on.) */
    for (int sweep = 0; !finished && sweep < 1000; sweep++) {
        finished = 1;
        for (int p = 1; p < m; p++) {
            for (int q = 0; q < p; q++) {
                const double pm = precision * maxDiagEntry;
                const double threshold = (considerAsZero < pm) ? pm : considerAsZero;
                if (fabs(W[p * m + q]) > threshold || fabs(W[q * m + p]) > threshold) {
                    finished = 0;
                    jrot_t jl, jr;
                    real_2x2_jacobi_svd(W[p * m + p], W[p * m + q], W[q * m + p], W[q * m + q], &jl, &jr);
                    rotate_plane(W + p * m, W + q * m, m, 1, jl.c, jl.s);   /* rows p, q */
                    if (U)
                        rotate_plane(U + p, U + q, urows, m, jl.c, jl.s);       /* columns p, q, by (J_left^T)^T */
                    rotate_plane(W + p, W + q, m, m, jr.c, -jr.s);          /* columns p, q, by J_right^T */
                    if (V)
                        rotate_plane(V + p, V + q, vrows, m, jr.c, -jr.s);
                    const double app = fabs(W[p * m + p]), aqq = fabs(W[q * m + q]);
                    const double mx = (app < aqq) ? aqq : app;
                    if (maxDiagEntry < mx)
                        maxDiagEntry = mx;
                }
            }
        }
    }
    for (int i = 0; i < m; i++) {
        const double a = W[i * m + i];
        S[i] = fabs(a);
        if (U && a < 0.0)
            for (int r = 0; r < urows; r++)
                U[r * m + i] = -U[r * m + i];
    }
    for (int i = 0; i < m; i++)
        S[i] = S[i] * scale;
    for (int i = 0; i < m; i++) {
        int pos = 0;
        double best = S[i];
        for (int k = i + 1; k < m; k++)
            if (S[k] > best) {
                best = S[k];
                pos = k - i;
            }
        if (best == 0.0)
            break;
        if (pos) {
            pos += i;
            double t = S[i];
            S[i] = S[pos];
            S[pos] = t;
            if (U)
                for (int r = 0; r < urows; r++) {
                    t = U[r * m + i];
                    U[r * m + i] = U[r * m + pos];
                    U[r * m + pos] = t;
                }
            if (V)
                for (int r = 0; r < vrows; r++) {
                    t = V[r * m + i];
                    V[r * m + i] = V[r * m + pos];
                    V[r * m + pos] = t;
                }
        }
    }
}

/* Householder.h, makeHouseholder on the vector v[0..len-1]: the tail becomes the essential part
 * (in place), *beta the new leading coefficient, *tau the reflector's factor. */
static void make_householder_in_place(double *v, int len, double *tau, double *beta)
{
    double tailSqNorm = 0.0;
    for (int i = 1; i < len; i++)
        tailSqNorm = tailSqNorm + v[i] * v[i];
    const double c0 = v[0];
    if (tailSqNorm <= DBL_MIN) {
        *tau = 0.0;
        *beta = c0;
        for (int i = 1; i < len; i++)
            v[i] = 0.0;
    } else {
        double b = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0)
            b = -b;
        for (int i = 1; i < len; i++)
            v[i] = v[i] / (c0 - b);
        *tau = (b - c0) / b;
        *beta = b;
    }
}

/* Householder.h, applyHouseholderOnTheLeft on one column w[0..len-1] (essential has len-1 entries):
 * tmp = essential . w[1:] + w[0]; w[0] -= tau tmp; w[1:] -= tmp (tau essential). */
static void apply_householder_left(double *w, int len, const double *essential, double tau)
{
    if (len == 1) {
        w[0] = w[0] * (1.0 - tau);
        return;
    }
    if (tau == 0.0)
        return;
    double tmp = 0.0;
    for (int i = 1; i < len; i++)
        tmp = tmp + essential[i - 1] * w[i];
    tmp = tmp + w[0];
    w[0] = w[0] - tau * tmp;
    for (int i = 1; i < len; i++)
        w[i] = w[i] - tmp * (tau * essential[i - 1]);
}

static double column_norm(const double *v, int len)
{
    double s = 0.0;
    for (int i = 0; i < len; i++)
        s = s + v[i] * v[i];
    return sqrt(s);
}

/*
 * ColPivHouseholderQR.h, computeInPlace, on a rows x cols matrix stored by columns
 * (M[c * rows + r]).  On return the upper triangle holds R, the columns below the diagonal the
 * essential parts, hcoeffs the factors and transp[k] the column that step k swapped into place.
 * The count of pivots above Eigen's threshold is returned (it decides nothing on this path).
 */
static int colpiv_householder_qr(double *M, int rows, int cols, double *hcoeffs, int *transp)
{
    const int size = rows < cols ? rows : cols;
    double *normsUpdated = (double *)malloc(sizeof(double) * (size_t)cols * 2);
    double *normsDirect = normsUpdated + cols;
    double maxNorm = 0.0;
    for (int k = 0; k < cols; k++) {
        normsDirect[k] = column_norm(M + (size_t)k * rows, rows);
        normsUpdated[k] = normsDirect[k];
        if (k == 0 || normsUpdated[k] > maxNorm)
            maxNorm = normsUpdated[k];
    }
    const double threshold_helper = ((maxNorm * DBL_EPSILON) * (maxNorm * DBL_EPSILON)) / (double)rows;
    const double norm_downdate_threshold = sqrt(DBL_EPSILON);
    int nonzero_pivots = size;
    for (int k = 0; k < size; k++) {
        int biggest = k;
        double bn = normsUpdated[k];
        for (int j = k + 1; j < cols; j++)
            if (normsUpdated[j] > bn) {
                bn = normsUpdated[j];
                biggest = j;
            }
        const double biggest_col_sq_norm = bn * bn;
        if (nonzero_pivots == size && biggest_col_sq_norm < threshold_helper * (double)(rows - k))
            nonzero_pivots = k;
        transp[k] = biggest;
        if (k != biggest) {
            double *a = M + (size_t)k * rows, *b = M + (size_t)biggest * rows;
            for (int r = 0; r < rows; r++) {
                const double t = a[r];
                a[r] = b[r];
                b[r] = t;
            }
            double t = normsUpdated[k];
            normsUpdated[k] = normsUpdated[biggest];
            normsUpdated[biggest] = t;
            t = normsDirect[k];
            normsDirect[k] = normsDirect[biggest];
            normsDirect[biggest] = t;
        }
        double beta;
        double *colk = M + (size_t)k * rows + k;
        make_householder_in_place(colk, rows - k, &hcoeffs[k], &beta);
        colk[0] = beta;
        for (int j = k + 1; j < cols; j++)
            apply_householder_left(M + (size_t)j * rows + k, rows - k, colk + 1, hcoeffs[k]);
        for (int j = k + 1; j < cols; j++) {
            if (normsUpdated[j] != 0.0) {
                double temp = fabs(M[(size_t)j * rows + k]) / normsUpdated[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                const double ratio = normsUpdated[j] / normsDirect[j];
                const double temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    normsDirect[j] = column_norm(M + (size_t)j * rows + k + 1, rows - k - 1);
                    normsUpdated[j] = normsDirect[j];
                } else {
                    normsUpdated[j] = normsUpdated[j] * sqrt(temp);
                }
            }
        }
    }
    free(normsUpdated);
    return nonzero_pivots;
}

/*
 * JacobiSVD(A, ComputeFullV).matrixV().col(8) for the 8x9 system of a sample (a[8][9], row-major;
 * eightPoint.cpp:254-255): the last column of the Q of the pivoted QR of (A / scale)^T.
 */
static void null_vector_8x9(const double a[8][9], double q[9])
{
    double scale = 0.0;
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 9; c++)
            if (fabs(a[r][c]) > scale)
                scale = fabs(a[r][c]);
    if (scale == 0.0)
        scale = 1.0;
    double M[8 * 9]; /* column c of A^T is row c of A */
    for (int c = 0; c < 8; c++)
        for (int r = 0; r < 9; r++)
            M[c * 9 + r] = a[c][r] / scale;
    double h[8];
    int transp[8];
    colpiv_householder_qr(M, 9, 8, h, transp);
    /* HouseholderSequence::evalTo, the column it makes of e_8 */
    for (int r = 0; r < 9; r++)
        q[r] = 0.0;
    q[8] = 1.0;
    for (int k = 7; k >= 0; k--)
        apply_householder_left(q + k, 9 - k, M + k * 9 + k + 1, h[k]);
}

/*
 * JacobiSVD(A, ComputeFullV).matrixV().col(8) for an n x 9 system, n >= 9 (row-major a; the
 * refit, eightPoint.cpp:85-86).
 */
static void smallest_right_singular_vector_nx9(const double *a, int n, double q[9])
{
    double scale = 0.0;
    for (size_t i = 0; i < (size_t)n * 9; i++)
        if (fabs(a[i]) > scale)
            scale = fabs(a[i]);
    if (scale == 0.0)
        scale = 1.0;
    double W[81], V[81], S[9];
    if (n == 9) {
        for (int i = 0; i < 81; i++) {
            W[i] = a[i] / scale;
            V[i] = (i / 9 == i % 9) ? 1.0 : 0.0;
        }
    } else {
        double *M = (double *)malloc(sizeof(double) * (size_t)n * 9);
        for (int c = 0; c < 9; c++)
            for (int r = 0; r < n; r++)
                M[(size_t)c * n + r] = a[(size_t)r * 9 + c] / scale;
        double h[9];
        int transp[9], perm[9];
        colpiv_householder_qr(M, n, 9, h, transp);
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++)
                W[i * 9 + j] = (i <= j) ? M[(size_t)j * n + i] : 0.0;
        /* colsPermutation(): the transpositions applied on the right of the identity */
        for (int i = 0; i < 9; i++)
            perm[i] = i;
        for (int k = 0; k < 9; k++) {
            const int t = perm[k];
            perm[k] = perm[transp[k]];
            perm[transp[k]] = t;
        }
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++)
                V[i * 9 + j] = (i == perm[j]) ? 1.0 : 0.0;
        free(M);
    }
    jacobi_svd_square(9, W, NULL, 0, V, 9, scale, S);
    for (int i = 0; i < 9; i++)
        q[i] = V[i * 9 + 8];
}

/*
 * JacobiSVD<Mat3>(E, ComputeFullU | ComputeFullV) (eightPoint.cpp:93-96, :262-265): E (row-major)
 * = U diag(S) V^T, S descending.
 */
static void svd3(const double e[3][3], double U[3][3], double S[3], double V[3][3])
{
    double scale = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (fabs(e[i][j]) > scale)
                scale = fabs(e[i][j]);
    if (scale == 0.0)
        scale = 1.0;
    double W[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            W[i * 3 + j] = e[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    jacobi_svd_square(3, W, &U[0][0], 3, &V[0][0], 3, scale, S);
}

/*
 * Least-squares solution of [-a, b] * lambda = T (3x2 system) the way the reference gets it:
 * A_tri.colPivHouseholderQr().solve(T) (eightPoint.cpp:321,363; twins :162,:214).  Eigen is a
 * third-party dependency that is absent here (CMakeLists.txt:27 asks for any version >= 3.3), so
 * this restates its published algorithm -- ColPivHouseholderQR::computeInPlace and _solve_impl,
 * MatrixBase::makeHouseholder and applyHouseholderOnTheLeft of Eigen 3.3/3.4 -- operation by
 * operation for a 3x2 matrix:
 *   - column norms sqrt((c0^2 + c1^2) + c2^2); the larger column becomes the first pivot, the
 *     first of equals (maxCoeff);
 *   - reflector of a column (c0, tail): beta = -sign(c0) sqrt(c0^2 + |tail|^2) (sign(0) = +),
 *     essential = tail / (c0 - beta), tau = (beta - c0) / beta; a tail with |tail|^2 <= DBL_MIN
 *     gives tau = 0, beta = c0;
 *   - applying it to a vector (w0, w): tmp = essential.w + w0; w0 -= tau tmp;
 *     w -= tmp (tau essential); skipped when tau = 0;
 *   - the second column's norm is downdated as in LAPACK's xGEQPF (it only feeds the rank
 *     decision: a second pivot whose squared norm is below (eps max norm)^2 * 2/3 is dropped and
 *     its unknown set to 0);
 *   - c = Q^T T by the two reflectors in turn, back substitution (c1 /= R11; c0 -= c1 R01;
 *     c0 /= R00, each division skipped for a zero numerator), then the column permutation.
 * PARITY UNPINNED at the level of single roundings: no output of Eigen itself pins this order of
 * operations; tests compare it with LAPACK (SVD-based least squares) to a few ulps times the
 * condition number.  bb = b.b is passed in because callers hoist it per correspondence.
 */
static inline void householder2(double c0, double t1, double t2, int tail_len, double *beta, double *tau,
                                double *e1, double *e2)
{
    /* MatrixBase::makeHouseholder on (c0, t1[, t2]) */
    const double tailSqNorm = (tail_len == 2) ? t1 * t1 + t2 * t2 : t1 * t1;
    if (tailSqNorm <= 2.2250738585072014e-308) { /* (std::numeric_limits<double>::min)() */
        *tau = 0.0;
        *beta = c0;
        *e1 = 0.0;
        *e2 = 0.0;
    } else {
        double b = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0)
            b = -b;
        *e1 = t1 / (c0 - b);
        *e2 = (tail_len == 2) ? t2 / (c0 - b) : 0.0;
        *tau = (b - c0) / b;
        *beta = b;
    }
}

static inline void triangulate_lambdas(const double a[3], const double b[3], double bb,
                                       const double T[3], double *lambda0, double *lambda1)
{
    const double eps = 2.220446049250313e-16;
    double p[3] = {-a[0], -a[1], -a[2]}; /* A_tri.col(0) = -Rx1 */
    double q[3] = {b[0], b[1], b[2]};    /* A_tri.col(1) = x2 */
    double n0 = sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
    double n1 = sqrt(bb);
    const double maxn = (n1 > n0) ? n1 : n0;
    const double threshold_helper = ((maxn * eps) * (maxn * eps)) / 3.0;
    int nonzero_pivots = 2;
    const int swapped = (n1 > n0); /* first pivot: the column of the largest norm, first of equals */
    /* (the first pivot's squared norm is never below (eps max norm)^2: both pivots count so far) */
    if (swapped) {
        for (int i = 0; i < 3; i++) {
            double t = p[i];
            p[i] = q[i];
            q[i] = t;
        }
        n1 = n0;
    }
    /* k = 0 */
    double beta0, tau0, e01, e02;
    householder2(p[0], p[1], p[2], 2, &beta0, &tau0, &e01, &e02);
    if (tau0 != 0.0) {
        double tmp = e01 * q[1] + e02 * q[2];
        tmp = tmp + q[0];
        q[0] = q[0] - tau0 * tmp;
        q[1] = q[1] - tmp * (tau0 * e01);
        q[2] = q[2] - tmp * (tau0 * e02);
    }
    /* norm of the remaining column, downdated (it was computed directly a moment ago, so the
     * ratio updated / direct is 1) */
    double upd = n1;
    if (upd != 0.0) {
        double temp = fabs(q[0]) / upd;
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double temp2 = temp * ((upd / n1) * (upd / n1));
        if (temp2 <= 1.4901161193847656e-08) /* sqrt(eps) */
            upd = sqrt(q[1] * q[1] + q[2] * q[2]);
        else
            upd = upd * sqrt(temp);
    }
    /* k = 1 */
    if (nonzero_pivots == 2 && upd * upd < threshold_helper * 2.0)
        nonzero_pivots = 1;
    double beta1, tau1, e11, unused;
    householder2(q[1], q[2], 0.0, 1, &beta1, &tau1, &e11, &unused);

    /* _solve_impl */
    double c[3] = {T[0], T[1], T[2]};
    double s0 = 0.0, s1 = 0.0;
    if (nonzero_pivots > 0) {
        if (tau0 != 0.0) {
            double tmp = e01 * c[1] + e02 * c[2];
            tmp = tmp + c[0];
            c[0] = c[0] - tau0 * tmp;
            c[1] = c[1] - tmp * (tau0 * e01);
            c[2] = c[2] - tmp * (tau0 * e02);
        }
        if (nonzero_pivots == 2) {
            if (tau1 != 0.0) {
                double tmp = e11 * c[2];
                tmp = tmp + c[1];
                c[1] = c[1] - tau1 * tmp;
            }
            if (c[1] != 0.0) {
                c[1] = c[1] / beta1;
                c[0] = c[0] - c[1] * q[0];
            }
            s1 = c[1];
        }
        if (c[0] != 0.0)
            c[0] = c[0] / beta0;
        s0 = c[0];
    }
    *lambda0 = swapped ? s1 : s0;
    *lambda1 = swapped ? s0 : s1;
}

/* The same least squares through the normal equations and Cramer's rule.  Not what the reference
 * does (it squares the condition number); kept only so that the tests can measure how often it
 * would have decided differently from the QR solve above. */
static inline void triangulate_lambdas_normal_equations(const double a[3], const double b[3], double bb,
                                                        const double T[3], double *lambda0, double *lambda1)
{
    double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    double ab = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    double aT = (a[0] * T[0] + a[1] * T[1]) + a[2] * T[2];
    double bT = (b[0] * T[0] + b[1] * T[1]) + b[2] * T[2];
    double det = aa * bb - ab * ab;
    *lambda0 = (ab * bT - bb * aT) / det;
    *lambda1 = (aa * bT - ab * aT) / det;
}

/* R (row-major 3x3) times (x, y, 1): Eigen's coefficient-wise 3x3 * 3x1
 * product, k ascending, with x1[2] == 1.0 so the last product is exact. */
static inline void rot_apply_h(const double R[3][3], double x, double y, double out[3])
{
    for (int i = 0; i < 3; i++)
        out[i] = (R[i][0] * x + R[i][1] * y) + R[i][2];
}

static inline double det3(const double M[3][3])
{
    return (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
            M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0])) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

/*
 * The fit half of eightPointAlgorithmFromSubset / eightPointAlgorithm
 * (eightPoint.cpp:233-344 and :64-185): sample points -> (R, T).
 * p1, p2: n x 2 doubles (K^-1-normalised).  Returns the chosen candidate
 * index 0..3 (order (R1,T),(R1,-T),(R2,T),(R2,-T)), and optionally the four
 * cheirality counts.
 */
static int fit_pose(const double *p1, const double *p2, int n, double R[3][3], double T[3],
                    int cheirality_counts[4])
{
    double q[9];
    /* eightPoint.cpp:243-251: row = [x1*x2, y1*x2, x2, x1*y2, y1*y2, y2, x1, y1, 1] */
    if (n == 8) {
        double A[8][9];
        for (int i = 0; i < 8; i++) {
            double x1 = p1[2 * i], y1 = p1[2 * i + 1];
            double x2 = p2[2 * i], y2 = p2[2 * i + 1];
            A[i][0] = x1 * x2; A[i][1] = y1 * x2; A[i][2] = x2;
            A[i][3] = x1 * y2; A[i][4] = y1 * y2; A[i][5] = y2;
            A[i][6] = x1;      A[i][7] = y1;      A[i][8] = 1.0;
        }
        null_vector_8x9(A, q);
    } else {
        double *A = (double *)malloc((size_t)n * 9 * sizeof(double));
        for (int i = 0; i < n; i++) {
            double x1 = p1[2 * i], y1 = p1[2 * i + 1];
            double x2 = p2[2 * i], y2 = p2[2 * i + 1];
            double *r = A + (size_t)i * 9;
            r[0] = x1 * x2; r[1] = y1 * x2; r[2] = x2;
            r[3] = x1 * y2; r[4] = y1 * y2; r[5] = y2;
            r[6] = x1;      r[7] = y1;      r[8] = 1.0;
        }
        smallest_right_singular_vector_nx9(A, n, q);
        free(A);
    }

    /* eightPoint.cpp:257-260: row-major reshape */
    double E[3][3] = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}};
    double U[3][3], S[3], V[3][3];
    svd3(E, U, S, V);

    /* eightPoint.cpp:274-289: R1 = U Rz(+90) V^T, R2 = U Rz(-90) V^T, T = u2.
     * U Rz(+90) has columns (u1, -u0, u2); U Rz(-90) has (-u1, u0, u2). */
    double R1[3][3], R2[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double w = U[i][2] * V[j][2];
            R1[i][j] = (U[i][1] * V[j][0] - U[i][0] * V[j][1]) + w;
            R2[i][j] = (U[i][0] * V[j][1] - U[i][1] * V[j][0]) + w;
        }
    if (det3(R1) < 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R1[i][j] = -R1[i][j];
    if (det3(R2) < 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R2[i][j] = -R2[i][j];
    const double T12[3] = {U[0][2], U[1][2], U[2][2]};

    /* eightPoint.cpp:296-337: cheirality vote over the sample points */
    int maxIndex = 0, maxPoints = 0;
    for (int c = 0; c < 4; c++) {
        const double(*Rc)[3] = (c < 2) ? R1 : R2;
        const double sgn = (c & 1) ? -1.0 : 1.0;
        const double Tc[3] = {sgn * T12[0], sgn * T12[1], sgn * T12[2]};
        int numPoints = 0;
        for (int i = 0; i < n; i++) {
            double a[3];
            rot_apply_h(Rc, p1[2 * i], p1[2 * i + 1], a);
            const double b[3] = {p2[2 * i], p2[2 * i + 1], 1.0};
            const double bb = (b[0] * b[0] + b[1] * b[1]) + 1.0;
            double l0, l1;
            triangulate_lambdas(a, b, bb, Tc, &l0, &l1);
            if (l0 > 0 && l1 > 0)
                numPoints++;
        }
        if (cheirality_counts)
            cheirality_counts[c] = numPoints;
        if (numPoints > maxPoints) {
            maxIndex = c;
            maxPoints = numPoints;
        }
    }
    const double(*Rb)[3] = (maxIndex < 2) ? R1 : R2;
    const double sgn = (maxIndex & 1) ? -1.0 : 1.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            R[i][j] = Rb[i][j];
        T[i] = sgn * T12[i];
    }
    return maxIndex;
}

/* eightPoint.cpp:349-369: depth of every correspondence in camera 1 with the
 * fp64 -> fp32 -> fp64 round trip of lambda1 (":365 float lambda1 = lambdas[0]"),
 * point = lambda1 * (x1, y1, 1). */
static inline void triangulate_point(const double R[3][3], const double T[3], double x1, double y1,
                                     double x2, double y2, double pt[3])
{
    double a[3];
    rot_apply_h(R, x1, y1, a);
    const double b[3] = {x2, y2, 1.0};
    const double bb = (x2 * x2 + y2 * y2) + 1.0;
    double l0, l1;
    triangulate_lambdas(a, b, bb, T, &l0, &l1);
    (void)l1;
    float lambda1 = (float)l0;
    double l = (double)lambda1;
    pt[0] = l * x1;
    pt[1] = l * y1;
    pt[2] = l;
}

/* solve.hpp:85-95 with K column-major (K[r + 3*c]) and view = [R|T]. */
static inline double reprojection_error_RT(const double *K, double ox, double oy, const double pt[3],
                                           const double R[3][3], const double T[3])
{
    double q[3];
    for (int i = 0; i < 3; i++)
        q[i] = ((R[i][0] * pt[0] + R[i][1] * pt[1]) + R[i][2] * pt[2]) + T[i];
    double p[3];
    for (int i = 0; i < 3; i++)
        p[i] = (K[i] * q[0] + K[i + 3] * q[1]) + K[i + 6] * q[2];
    p[0] /= p[2];
    p[1] /= p[2];
    double obs0 = (K[0] * ox + K[3] * oy) + K[6];
    double obs1 = (K[1] * ox + K[4] * oy) + K[7];
    double dx = p[0] - obs0;
    double dy = p[1] - obs1;
    return dx * dx + dy * dy;
}

/* View matrix = identity: (I * [pt;1]).head<3>() == pt for finite pt. */
static inline double reprojection_error_I(const double *K, double ox, double oy, const double pt[3])
{
    double p[3];
    for (int i = 0; i < 3; i++)
        p[i] = (K[i] * pt[0] + K[i + 3] * pt[1]) + K[i + 6] * pt[2];
    p[0] /= p[2];
    p[1] /= p[2];
    double obs0 = (K[0] * ox + K[3] * oy) + K[6];
    double obs1 = (K[1] * ox + K[4] * oy) + K[7];
    double dx = p[0] - obs0;
    double dy = p[1] - obs1;
    return dx * dx + dy * dy;
}

/* solve.hpp:119-124: std::max(loss1, loss2) == (loss1 < loss2) ? loss2 : loss1 */
static inline double default_loss(const double *K, double x1, double y1, double x2, double y2,
                                  const double pt[3], const double R[3][3], const double T[3])
{
    double loss1 = reprojection_error_I(K, x1, y1, pt);
    double loss2 = reprojection_error_RT(K, x2, y2, pt, R, T);
    return (loss1 < loss2) ? loss2 : loss1;
}

/* Sampson distance in pixel^2 for E = [T]x R with K-normalised points scaled
 * by the focal lengths of K; an extension loss with no counterpart in the
 * reference (its algebraic residual eightPointError, solve.hpp:142-157, is
 * unused there). */
static inline double sampson_loss(const double *K, double x1, double y1, double x2, double y2,
                                  const double R[3][3], const double T[3])
{
    double Tx[3][3] = {{0, -T[2], T[1]}, {T[2], 0, -T[0]}, {-T[1], T[0], 0}};
    double E[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            E[i][j] = (Tx[i][0] * R[0][j] + Tx[i][1] * R[1][j]) + Tx[i][2] * R[2][j];
    double Ex1[3], Etx2[2];
    for (int i = 0; i < 3; i++)
        Ex1[i] = (E[i][0] * x1 + E[i][1] * y1) + E[i][2];
    for (int j = 0; j < 2; j++)
        Etx2[j] = (E[0][j] * x2 + E[1][j] * y2) + E[2][j];
    double num = (x2 * Ex1[0] + y2 * Ex1[1]) + Ex1[2];
    double den = (Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1]) + (Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1]);
    double f = 0.5 * (K[0] + K[4]);
    return (num * num / den) * (f * f);
}

static void pose_to_colmajor16(const double R[3][3], const double T[3], double *pose16)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            pose16[r + 4 * c] = (r == c) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            pose16[r + 4 * c] = R[r][c];
        pose16[r + 12] = T[r];
    }
}

/*
 * eightPointAlgorithmFromSubset (eightPoint.cpp:225-372).  calc1/calc2:
 * n_calc x 2; all1/all2: n_all x 2.  pose16_out: 4x4 column-major;
 * points_out: n_all x 3.  Returns the chosen candidate index, or -1 when
 * n_calc < 8 (the reference throws std::invalid_argument, :228-231).
 */
int sfm_oracle_eight_point_from_subset(const double *calc1, const double *calc2, int n_calc,
                                       const double *all1, const double *all2, int n_all,
                                       double *pose16_out, double *points_out,
                                       int *cheirality_counts_out)
{
    if (n_calc < 8)
        return -1;
    double R[3][3], T[3];
    int idx = fit_pose(calc1, calc2, n_calc, R, T, cheirality_counts_out);
    if (pose16_out)
        pose_to_colmajor16(R, T, pose16_out);
    if (points_out)
        for (int i = 0; i < n_all; i++)
            triangulate_point(R, T, all1[2 * i], all1[2 * i + 1], all2[2 * i], all2[2 * i + 1],
                              points_out + 3 * (size_t)i);
    return idx;
}

/* eightPointAlgorithm(points1, points2) (eightPoint.cpp:57-223): the same on
 * one set; used for the post-RANSAC refit (scene.cpp:288). */
int sfm_oracle_eight_point(const double *points1, const double *points2, int n,
                           double *pose16_out, double *points_out)
{
    return sfm_oracle_eight_point_from_subset(points1, points2, n, points1, points2, n, pose16_out,
                                              points_out, NULL);
}

/* The default loss for one correspondence, for tests.  pose16 column-major. */
double sfm_oracle_loss(const double *K, const double *obs1, const double *obs2, const double *pt,
                       const double *pose16, int loss_kind)
{
    double R[3][3], T[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            R[r][c] = pose16[r + 4 * c];
        T[r] = pose16[r + 12];
    }
    if (loss_kind == 1)
        return sampson_loss(K, obs1[0], obs1[1], obs2[0], obs2[1], R, T);
    return default_loss(K, obs1[0], obs1[1], obs2[0], obs2[1], pt, R, T);
}

/* 3x3 SVD and 8x9 null vector exposed for the LAPACK cross-checks. */
void sfm_oracle_svd3(const double *e_rowmajor, double *U_rowmajor, double *S, double *V_rowmajor)
{
    double e[3][3], U[3][3], V[3][3];
    memcpy(e, e_rowmajor, sizeof e);
    svd3(e, U, S, V);
    memcpy(U_rowmajor, U, sizeof U);
    memcpy(V_rowmajor, V, sizeof V);
}

void sfm_oracle_null_vector(const double *a_rowmajor, int n, double *q9)
{
    if (n == 8) {
        double a[8][9];
        memcpy(a, a_rowmajor, sizeof a);
        null_vector_8x9(a, q9);
    } else {
        smallest_right_singular_vector_nx9(a_rowmajor, n, q9);
    }
}

/*
 * JacobiSVD<MatrixXd>(A, ComputeThinU | ComputeThinV) for rows > cols (a row-major): thin U
 * (rows x cols), S, V (cols x cols), row-major.  Not on the reference's path with these options;
 * it runs the same restated pieces (pivoted QR, householderQ applied to the identity, the sweeps,
 * the sign fix, the sort) and exists so that a test can compare them with the one output of
 * Eigen's JacobiSVD that is widely published: the example in its class documentation.
 */
void sfm_oracle_jacobi_svd_tall(const double *a, int rows, int cols, double *U, double *S, double *V)
{
    double scale = 0.0;
    for (size_t i = 0; i < (size_t)rows * cols; i++)
        if (fabs(a[i]) > scale)
            scale = fabs(a[i]);
    if (scale == 0.0)
        scale = 1.0;
    double *M = (double *)malloc(sizeof(double) * ((size_t)rows * cols + (size_t)cols * cols + cols + rows));
    double *W = M + (size_t)rows * cols, *h = W + (size_t)cols * cols, *col = h + cols;
    int *transp = (int *)malloc(sizeof(int) * 2 * (size_t)cols), *perm = transp + cols;
    for (int c = 0; c < cols; c++)
        for (int r = 0; r < rows; r++)
            M[(size_t)c * rows + r] = a[(size_t)r * cols + c] / scale;
    colpiv_householder_qr(M, rows, cols, h, transp);
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < cols; j++)
            W[i * cols + j] = (i <= j) ? M[(size_t)j * rows + i] : 0.0;
    /* thin U: householderQ() applied to the first cols columns of the identity, last reflector first */
    for (int j = 0; j < cols; j++) {
        for (int r = 0; r < rows; r++)
            col[r] = (r == j) ? 1.0 : 0.0;
        for (int k = cols - 1; k >= 0; k--)
            apply_householder_left(col + k, rows - k, M + (size_t)k * rows + k + 1, h[k]);
        for (int r = 0; r < rows; r++)
            U[(size_t)r * cols + j] = col[r];
    }
    for (int i = 0; i < cols; i++)
        perm[i] = i;
    for (int k = 0; k < cols; k++) {
        const int t = perm[k];
        perm[k] = perm[transp[k]];
        perm[transp[k]] = t;
    }
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < cols; j++)
            V[i * cols + j] = (i == perm[j]) ? 1.0 : 0.0;
    jacobi_svd_square(cols, W, U, rows, V, cols, scale, S);
    free(M);
    free(transp);
}

void sfm_oracle_triangulate(const double *a, const double *b, const double *T, double *lambdas2)
{
    double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    triangulate_lambdas(a, b, bb, T, &lambdas2[0], &lambdas2[1]);
}

void sfm_oracle_triangulate_normal_equations(const double *a, const double *b, const double *T, double *lambdas2)
{
    double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    triangulate_lambdas_normal_equations(a, b, bb, T, &lambdas2[0], &lambdas2[1]);
}

/* Every (hypothesis, correspondence) cell of RANSAC.cpp:213-221 on its own, for the tests that count
 * decisions: lambda_out[h][k] = the camera-1 depth before the float cast, err_out[h][k] = the default
 * loss.  pose12[h] = R row-major then T.  method 0 = the QR solve, 1 = the normal equations. */
void sfm_oracle_score_cells(const double *x, const double *y, int n, const double *K, const double *pose12, int H,
                            int method, double *lambda_out, double *err_out)
{
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++) {
        double R[3][3], T[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                R[i][j] = pose12[(size_t)h * 12 + 3 * i + j];
            T[i] = pose12[(size_t)h * 12 + 9 + i];
        }
        for (int k = 0; k < n; k++) {
            const double x1 = x[2 * k], y1 = x[2 * k + 1], x2 = y[2 * k], y2 = y[2 * k + 1];
            double a[3];
            rot_apply_h(R, x1, y1, a);
            const double b[3] = {x2, y2, 1.0};
            const double bb = (x2 * x2 + y2 * y2) + 1.0;
            double l0, l1;
            if (method == 1)
                triangulate_lambdas_normal_equations(a, b, bb, T, &l0, &l1);
            else
                triangulate_lambdas(a, b, bb, T, &l0, &l1);
            const double l = (double)(float)l0;
            const double pt[3] = {l * x1, l * y1, l};
            lambda_out[(size_t)h * n + k] = l0;
            err_out[(size_t)h * n + k] = default_loss(K, x1, y1, x2, y2, pt, R, T);
        }
    }
}

/* Timing aid of the CPU baseline: 1 = sfm_oracle_ransac also performs the per-iteration shuffle of all
 * correspondence indices that the reference's loop does (RANSAC.cpp:201); results do not change. */
static int g_emulate_shuffle = 0;
static volatile int g_shuffle_sink = 0;
void sfm_oracle_emulate_shuffle(int on) { g_emulate_shuffle = on; }

/* RANSAC.cpp:14-32 */
static int safe_cast_to_int(double value)
{
    if (isnan(value))
        return 0;
    if (value >= (double)INT_MAX)
        return INT_MAX;
    if (value <= (double)INT_MIN)
        return INT_MIN;
    return (int)value;
}

int sfm_oracle_adaptive_max_iter(int n_inliers, int n_total, int minN, double successProb)
{
    /* RANSAC.cpp:245-250; returns -1 when the cap does not apply (:248) */
    double inlierRatio = (double)n_inliers / (double)n_total;
    double succesProb = pow(inlierRatio, minN);
    if (succesProb >= SFM_EPSILON)
        return safe_cast_to_int(ceil(log(1. - successProb) / log(1 - succesProb)));
    return -1;
}

/* Steps 2-3 of one RANSAC iteration (RANSAC.cpp:201-221) for sample row s[minN]. */
static void score_hypothesis(const double *x, const double *y, int n, const double *K,
                             const int *s, int minN, double maxSquaredError, int loss_kind,
                             int *count_out, double *toterr_out, double R[3][3], double T[3])
{
    double cx[2 * 64], cy[2 * 64];
    for (int k = 0; k < minN; k++) {
        cx[2 * k] = x[2 * s[k]];
        cx[2 * k + 1] = x[2 * s[k] + 1];
        cy[2 * k] = y[2 * s[k]];
        cy[2 * k + 1] = y[2 * s[k] + 1];
    }
    fit_pose(cx, cy, minN, R, T, NULL);
    int count = 0;
    double currentTotError = 0;
    for (int k = 0; k < n; k++) {
        double error;
        if (loss_kind == 1) {
            error = sampson_loss(K, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], R, T);
        } else {
            double pt[3];
            triangulate_point(R, T, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], pt);
            error = default_loss(K, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], pt, R, T);
        }
        if (error < maxSquaredError) {
            count++;
            currentTotError += error;
        }
    }
    *count_out = count;
    *toterr_out = currentTotError;
}

/*
 * solve::RANSAC (RANSAC.cpp:146-270) with an explicit sample table
 * samples[H][minN] instead of the reference's time-seeded shuffle, evaluated
 * in the order one TBB worker would take them (h ascending), which is one of
 * the schedules the reference admits.  Selection: more inliers wins, equal
 * count -> smaller total error wins (:237-238), first found keeps exact ties.
 * adaptive != 0 applies the maxIter update of :245-258.  Every hypothesis is
 * scored (in parallel when OpenMP is on); the serial scan afterwards decides
 * which of them the h-ascending schedule would have reached.
 *
 * Outputs (any may be NULL): inliers_out[n] ascending indices, best_h,
 * pose16 (column-major 4x4 of the best hypothesis), counts[H], toterr[H],
 * iterations_done.  Returns the inlier count, 0 for the guard of :148-149
 * (n < minN) and -1 for minN outside 8..64.
 */
int sfm_oracle_ransac(const double *x, const double *y, int n, const double *K, const int *samples,
                      int H, int minN, double maxSquaredError, double successProb, int adaptive,
                      int loss_kind, int *inliers_out, int *best_h_out, double *pose16_out,
                      int *counts_out, double *toterr_out, int *iterations_done_out)
{
    if (best_h_out)
        *best_h_out = -1;
    if (iterations_done_out)
        *iterations_done_out = 0;
    if (minN < 8 || minN > 64)
        return -1;
    if (n < minN)
        return 0;

    int *counts = (int *)malloc((size_t)(H > 0 ? H : 1) * sizeof(int));
    double *toterr = (double *)malloc((size_t)(H > 0 ? H : 1) * sizeof(double));
#pragma omp parallel
    {
        /* RANSAC.cpp:170-172: every worker owns a random engine and an index vector; :201 shuffles the whole
         * vector once per iteration and keeps its first minN entries.  The rows come from the caller's table
         * here, so the shuffle changes no result; the timed CPU baseline switches it on to pay its cost. */
        int *indices = NULL;
        uint64_t state = 0x9E3779B97F4A7C15ull;
        if (g_emulate_shuffle) {
#ifdef _OPENMP
            state += (uint64_t)omp_get_thread_num();
#endif
            indices = (int *)malloc((size_t)n * sizeof(int));
            if (indices)
                for (int i = 0; i < n; i++)
                    indices[i] = i;
        }
#pragma omp for schedule(dynamic, 4)
        for (int h = 0; h < H; h++) {
            if (indices) {
                for (int i = n - 1; i > 0; i--) { /* Fisher-Yates, as std::shuffle */
                    state = state * 6364136223846793005ull + 1442695040888963407ull;
                    const int j = (int)(((state >> 33) * (uint64_t)(i + 1)) >> 31);
                    const int t = indices[i];
                    indices[i] = indices[j];
                    indices[j] = t;
                }
                g_shuffle_sink += indices[0];
            }
            double R[3][3], T[3];
            score_hypothesis(x, y, n, K, samples + (size_t)h * minN, minN, maxSquaredError, loss_kind,
                             &counts[h], &toterr[h], R, T);
        }
        free(indices);
    }

    int maxIter = H;
    int iterationsDone = 0;
    int bestCount = 0;
    double bestError = INFINITY;
    int best = -1;
    for (int h = 0; h < H; h++) {
        if (iterationsDone >= maxIter)
            break;
        iterationsDone++;
        if (counts[h] > 0 &&
            (counts[h] > bestCount || (counts[h] == bestCount && toterr[h] < bestError))) {
            bestError = toterr[h];
            bestCount = counts[h];
            best = h;
            if (adaptive) {
                int newMax = sfm_oracle_adaptive_max_iter(bestCount, n, minN, successProb);
                if (newMax >= 0 && newMax < maxIter)
                    maxIter = newMax;
            }
        }
    }

    if (counts_out)
        memcpy(counts_out, counts, (size_t)H * sizeof(int));
    if (toterr_out)
        memcpy(toterr_out, toterr, (size_t)H * sizeof(double));
    if (iterations_done_out)
        *iterations_done_out = iterationsDone;
    if (best_h_out)
        *best_h_out = best;
    free(counts);
    free(toterr);
    if (best < 0)
        return 0;

    /* Re-run the winner to list its inliers (the reference keeps the list it
     * built while scoring, RANSAC.cpp:218,241). */
    double R[3][3], T[3];
    const int *s = samples + (size_t)best * minN;
    double cx[2 * 64], cy[2 * 64];
    for (int k = 0; k < minN; k++) {
        cx[2 * k] = x[2 * s[k]];
        cx[2 * k + 1] = x[2 * s[k] + 1];
        cy[2 * k] = y[2 * s[k]];
        cy[2 * k + 1] = y[2 * s[k] + 1];
    }
    fit_pose(cx, cy, minN, R, T, NULL);
    if (pose16_out)
        pose_to_colmajor16(R, T, pose16_out);
    int count = 0;
    for (int k = 0; k < n; k++) {
        double error;
        if (loss_kind == 1) {
            error = sampson_loss(K, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], R, T);
        } else {
            double pt[3];
            triangulate_point(R, T, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], pt);
            error = default_loss(K, x[2 * k], x[2 * k + 1], y[2 * k], y[2 * k + 1], pt, R, T);
        }
        if (error < maxSquaredError) {
            if (inliers_out)
                inliers_out[count] = k;
            count++;
        }
    }
    return count;
}

/* Per-hypothesis poses for a sample table: pose12[h] = R row-major (9) then T (3). */
/* votes_out (may be NULL): the four cheirality counts of every row, in candidate order. */
void sfm_oracle_fit_hypotheses_votes(const double *x, const double *y, const int *samples, int H,
                                     double *pose12_out, int *candidate_out, int *votes_out)
{
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++) {
        double cx[16], cy[16], R[3][3], T[3];
        const int *s = samples + (size_t)h * 8;
        for (int k = 0; k < 8; k++) {
            cx[2 * k] = x[2 * s[k]];
            cx[2 * k + 1] = x[2 * s[k] + 1];
            cy[2 * k] = y[2 * s[k]];
            cy[2 * k + 1] = y[2 * s[k] + 1];
        }
        int c = fit_pose(cx, cy, 8, R, T, votes_out ? votes_out + (size_t)h * 4 : NULL);
        double *o = pose12_out + (size_t)h * 12;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                o[3 * i + j] = R[i][j];
        for (int i = 0; i < 3; i++)
            o[9 + i] = T[i];
        if (candidate_out)
            candidate_out[h] = c;
    }
}

void sfm_oracle_fit_hypotheses(const double *x, const double *y, const int *samples, int H,
                               double *pose12_out, int *candidate_out)
{
    sfm_oracle_fit_hypotheses_votes(x, y, samples, H, pose12_out, candidate_out, NULL);
}

/* scene.cpp:495-501: (K^-1 [u, v, 1])_{0,1}; Kinv column-major. */
void sfm_oracle_normalize_points(const double *Kinv, const double *pixels, int n, double *out)
{
    for (int i = 0; i < n; i++) {
        double u = pixels[2 * i], v = pixels[2 * i + 1];
        out[2 * i] = (Kinv[0] * u + Kinv[3] * v) + Kinv[6];
        out[2 * i + 1] = (Kinv[1] * u + Kinv[4] * v) + Kinv[7];
    }
}

/* The sample-table generator of the device library (no counterpart in the reference, whose
 * shuffle is time-seeded, RANSAC.cpp:170,201): row h of pair p, n correspondences. */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void sfm_oracle_sample_table(int64_t n, int n_hyp, uint64_t seed, int pair_index, int32_t *out)
{
    for (int h = 0; h < n_hyp; h++) {
        int32_t *row = out + (size_t)h * 8;
        if (n < 8) {
            for (int k = 0; k < 8; k++) row[k] = 0;
            continue;
        }
        uint64_t s = mix64(seed ^ mix64(((uint64_t)pair_index << 32) | (uint64_t)(uint32_t)h));
        for (int k = 0; k < 8; k++) {
            uint64_t r = mix64(s + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
            int64_t v = (int64_t)(((unsigned __int128)r * (uint64_t)(n - k)) >> 64);
            /* v-th index not taken yet: walk the taken ones in ascending order */
            int32_t taken[8];
            for (int t = 0; t < k; t++) taken[t] = row[t];
            for (int a = 1; a < k; a++) { /* insertion sort */
                int32_t key = taken[a];
                int b = a - 1;
                while (b >= 0 && taken[b] > key) { taken[b + 1] = taken[b]; b--; }
                taken[b + 1] = key;
            }
            for (int t = 0; t < k; t++)
                if (taken[t] <= v) v++;
            row[k] = (int32_t)v;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Scale chaining and running point average of Scene::solveForLastAddedFrame, scene.cpp:301-421
 * (geometry only: the colour average of :378,:394-407 needs the images).
 *
 * Step s joins frames s and s + 1.  Its inputs are what the refit on the inliers produced for that
 * pair: the ascending track ids of its points (m_trackIndices23), the points in the older camera's
 * frame (m_frame23.points) and the pose (m_frame23.pose, column-major 4x4).  CSR over steps.
 * State: accumulated scale (1), accumulated pose (start_pose16), per-track running mean and count.
 * extrinsics_out[0] = start pose (scene.cpp:181), extrinsics_out[s + 1] = pose after step s (:421).
 *
 * Eigen's fixed-size products and Matrix4::inverse() are restated with a stated operation order
 * (rows as sums left to right; adjugate / determinant by 2x2 sub-determinants), so parity with the
 * reference is to rounding, not bitwise.
 * --------------------------------------------------------------------------------------------- */
static void mat4_mul(const double *A, const double *B, double *C) /* column-major */
{
    double T[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            T[r + 4 * c] = ((A[r] * B[4 * c] + A[r + 4] * B[1 + 4 * c]) + A[r + 8] * B[2 + 4 * c]) + A[r + 12] * B[3 + 4 * c];
    memcpy(C, T, sizeof T);
}

static void mat4_inverse(const double *m, double *inv) /* column-major; adjugate / determinant */
{
#define M(r, c) m[(r) + 4 * (c)]
    const double s0 = M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1);
    const double s1 = M(0, 0) * M(1, 2) - M(1, 0) * M(0, 2);
    const double s2 = M(0, 0) * M(1, 3) - M(1, 0) * M(0, 3);
    const double s3 = M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2);
    const double s4 = M(0, 1) * M(1, 3) - M(1, 1) * M(0, 3);
    const double s5 = M(0, 2) * M(1, 3) - M(1, 2) * M(0, 3);
    const double c5 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    const double c4 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const double c3 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    const double c2 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const double c1 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    const double c0 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    const double id = 1.0 / det;
    double b[16];
#define B_(r, c) b[(r) + 4 * (c)]
    B_(0, 0) = ((M(1, 1) * c5 - M(1, 2) * c4) + M(1, 3) * c3) * id;
    B_(0, 1) = ((-M(0, 1) * c5 + M(0, 2) * c4) - M(0, 3) * c3) * id;
    B_(0, 2) = ((M(3, 1) * s5 - M(3, 2) * s4) + M(3, 3) * s3) * id;
    B_(0, 3) = ((-M(2, 1) * s5 + M(2, 2) * s4) - M(2, 3) * s3) * id;
    B_(1, 0) = ((-M(1, 0) * c5 + M(1, 2) * c2) - M(1, 3) * c1) * id;
    B_(1, 1) = ((M(0, 0) * c5 - M(0, 2) * c2) + M(0, 3) * c1) * id;
    B_(1, 2) = ((-M(3, 0) * s5 + M(3, 2) * s2) - M(3, 3) * s1) * id;
    B_(1, 3) = ((M(2, 0) * s5 - M(2, 2) * s2) + M(2, 3) * s1) * id;
    B_(2, 0) = ((M(1, 0) * c4 - M(1, 1) * c2) + M(1, 3) * c0) * id;
    B_(2, 1) = ((-M(0, 0) * c4 + M(0, 1) * c2) - M(0, 3) * c0) * id;
    B_(2, 2) = ((M(3, 0) * s4 - M(3, 1) * s2) + M(3, 3) * s0) * id;
    B_(2, 3) = ((-M(2, 0) * s4 + M(2, 1) * s2) - M(2, 3) * s0) * id;
    B_(3, 0) = ((-M(1, 0) * c3 + M(1, 1) * c1) - M(1, 2) * c0) * id;
    B_(3, 1) = ((M(0, 0) * c3 - M(0, 1) * c1) + M(0, 2) * c0) * id;
    B_(3, 2) = ((-M(3, 0) * s3 + M(3, 1) * s1) - M(3, 2) * s0) * id;
    B_(3, 3) = ((M(2, 0) * s3 - M(2, 1) * s1) + M(2, 2) * s0) * id;
#undef B_
#undef M
    memcpy(inv, b, sizeof b);
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* (pose * p.homogeneous()).head<3>() */
static void pose_apply(const double *P, const double *p, double *out)
{
    for (int r = 0; r < 3; r++)
        out[r] = ((P[r] * p[0] + P[r + 4] * p[1]) + P[r + 8] * p[2]) + P[r + 12];
}

/* colors (optional): the colour of the observation behind every point (scene.cpp:377), three bytes per row;
 * colors_out[n_tracks][3] then receives the running average of scene.cpp:394-407. */
static int scale_chain_impl(int n_steps, const int64_t *offsets, const int32_t *track_ids, const double *points,
                            const unsigned char *colors, const double *poses16, const double *start_pose16,
                            double max_translation, int n_tracks, double *extrinsics_out, double *points3d_out,
                            unsigned char *colors_out, int32_t *counts_out, double *acc_scale_out)
{
    double accScale = 1.0, accPose[16];
    memcpy(accPose, start_pose16, sizeof accPose);
    memcpy(extrinsics_out, accPose, sizeof accPose);
    for (int t = 0; t < n_tracks; t++) {
        points3d_out[3 * t] = points3d_out[3 * t + 1] = points3d_out[3 * t + 2] = 0.0;
        counts_out[t] = 0;
        if (colors_out) colors_out[3 * t] = colors_out[3 * t + 1] = colors_out[3 * t + 2] = 0;
    }
    double *ratios = (double *)malloc(sizeof(double) * (size_t)(offsets[n_steps] > 0 ? offsets[n_steps] : 1));
    if (!ratios) return -1;
    for (int s = 0; s < n_steps; s++) {
        const int64_t b23 = offsets[s], n23 = offsets[s + 1] - b23;
        if (s > 0) { /* scene.cpp:301-354 */
            const int64_t b12 = offsets[s - 1], n12 = offsets[s] - b12;
            const double *pose12 = poses16 + 16 * (s - 1);
            int64_t idx12 = 0, nr = 0;
            for (int64_t idx23 = 0; idx23 < n23; idx23++) {
                const int cur = track_ids[b23 + idx23];
                while (idx12 < n12 && track_ids[b12 + idx12] < cur) idx12++;
                if (idx12 >= n12) break;
                if (track_ids[b12 + idx12] == cur) {
                    double q[3];
                    pose_apply(pose12, points + 3 * (b12 + idx12), q);
                    const double *p23 = points + 3 * (b23 + idx23);
                    const double dist12 = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
                    const double dist23 = sqrt((p23[0] * p23[0] + p23[1] * p23[1]) + p23[2] * p23[2]);
                    if (dist23 > 0.1 && dist12 > 0.1 && dist23 < 100.0 && dist12 < 100.0) ratios[nr++] = dist12 / dist23;
                }
            }
            if (nr >= 5) {
                qsort(ratios, (size_t)nr, sizeof(double), cmp_double); /* getMedian, scene.cpp:486-493: element n/2 */
                const double rel = ratios[nr / 2];
                if (rel >= 0.1 && rel < 10) accScale *= rel;
            }
        }
        if (acc_scale_out) acc_scale_out[s] = accScale;
        double view[16];
        memcpy(view, poses16 + 16 * s, sizeof view);
        for (int r = 0; r < 3; r++) view[r + 12] *= accScale; /* :356-357 */
        const double len = sqrt((view[12] * view[12] + view[13] * view[13]) + view[14] * view[14]);
        if (len > max_translation) { /* :359-365 */
            const double f = max_translation / len;
            for (int r = 0; r < 3; r++) view[r + 12] *= f;
        }
        for (int64_t j = 0; j < n23; j++) { /* :367-411 */
            const int t = track_ids[b23 + j];
            const double *p = points + 3 * (b23 + j);
            if (p[2] < 0 || t < 0 || t >= n_tracks) continue;
            const double sp[3] = {accScale * p[0], accScale * p[1], accScale * p[2]};
            double g[3];
            pose_apply(accPose, sp, g);
            double *cur = points3d_out + 3 * t;
            const unsigned char *nc = (colors && colors_out) ? colors + 3 * (b23 + j) : NULL;
            if (counts_out[t] == 0) {
                cur[0] = g[0]; cur[1] = g[1]; cur[2] = g[2];
                if (nc) for (int c = 0; c < 3; c++) colors_out[3 * t + c] = nc[c];
                counts_out[t] = 1;
            } else {
                const double d0 = cur[0] - g[0], d1 = cur[1] - g[1], d2 = cur[2] - g[2];
                if (sqrt((d0 * d0 + d1 * d1) + d2 * d2) < 50.0) {
                    const int N = counts_out[t];
                    const double N1 = (double)(N + 1);
                    for (int r = 0; r < 3; r++) cur[r] = cur[r] + (g[r] - cur[r]) / N1;
                    if (nc) { /* accumulateUchar, scene.cpp:394-402: float arithmetic, clamp, cut off */
                        for (int c = 0; c < 3; c++) {
                            const float cNew = (float)nc[c], cOld = (float)colors_out[3 * t + c];
                            float res = cOld + (cNew - cOld) / (float)(N + 1);
                            res = res < 0.f ? 0.f : (res > 255.f ? 255.f : res);
                            colors_out[3 * t + c] = (unsigned char)res;
                        }
                    }
                    counts_out[t]++;
                }
            }
        }
        double inv[16];
        mat4_inverse(view, inv); /* :421 */
        mat4_mul(accPose, inv, accPose);
        memcpy(extrinsics_out + 16 * (s + 1), accPose, sizeof accPose);
    }
    free(ratios);
    return 0;
}

int sfm_oracle_scale_chain(int n_steps, const int64_t *offsets, const int32_t *track_ids, const double *points,
                           const double *poses16, const double *start_pose16, double max_translation, int n_tracks,
                           double *extrinsics_out, double *points3d_out, int32_t *counts_out, double *acc_scale_out)
{
    return scale_chain_impl(n_steps, offsets, track_ids, points, NULL, poses16, start_pose16, max_translation, n_tracks,
                            extrinsics_out, points3d_out, NULL, counts_out, acc_scale_out);
}

int sfm_oracle_scale_chain_colors(int n_steps, const int64_t *offsets, const int32_t *track_ids, const double *points,
                                  const unsigned char *colors, const double *poses16, const double *start_pose16,
                                  double max_translation, int n_tracks, double *extrinsics_out, double *points3d_out,
                                  unsigned char *colors_out, int32_t *counts_out, double *acc_scale_out)
{
    return scale_chain_impl(n_steps, offsets, track_ids, points, colors, poses16, start_pose16, max_translation, n_tracks,
                            extrinsics_out, points3d_out, colors_out, counts_out, acc_scale_out);
}

/* util::getPixelBilinearUchar (submodules/util/util.cpp:40-78) for n positions (u, v) of one image of BGR bytes:
 * RGB bytes out.  cv::saturate_cast<uchar>(double) is cvRound (nearest, ties to even) and a clamp. */
void sfm_oracle_pixel_bilinear_u8(const unsigned char *bgr, int rows, int cols, const double *pixels, int n,
                                  unsigned char *rgb_out)
{
    for (int i = 0; i < n; i++) {
        const double u = pixels[2 * i], v = pixels[2 * i + 1];
        int x0 = (int)floor(u), y0 = (int)floor(v);
        x0 = x0 < 0 ? 0 : (x0 > cols - 2 ? cols - 2 : x0);
        y0 = y0 < 0 ? 0 : (y0 > rows - 2 ? rows - 2 : y0);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const double dx = u - x0, dy = v - y0;
        const double w00 = (1.0 - dx) * (1.0 - dy), w10 = dx * (1.0 - dy), w01 = (1.0 - dx) * dy, w11 = dx * dy;
        const unsigned char *p00 = bgr + 3 * ((size_t)y0 * cols + x0), *p10 = bgr + 3 * ((size_t)y0 * cols + x1);
        const unsigned char *p01 = bgr + 3 * ((size_t)y1 * cols + x0), *p11 = bgr + 3 * ((size_t)y1 * cols + x1);
        unsigned char res[3];
        for (int c = 0; c < 3; c++) {
            const double val = ((w00 * p00[c] + w10 * p10[c]) + w01 * p01[c]) + w11 * p11[c];
            const double r = nearbyint(val);
            res[c] = (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
        }
        rgb_out[3 * i] = res[2]; /* BGR -> RGB */
        rgb_out[3 * i + 1] = res[1];
        rgb_out[3 * i + 2] = res[0];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Extension with no counterpart in the reference (PARITY UNPINNED by it): matchTwoSided
 * (match.cpp:24-94) with Lowe's ratio test on the unit sphere between the two best-column scans.
 * Row idx1 with best column j1 (similarity s1, as match.cpp:64-79 finds it) keeps it only if every
 * other column j inside the pixel gate has (1 - s1) < ratio^2 (1 - sim(idx1, j)), in double on the
 * float similarities; NaN similarities are ignored.  Rows that fail get matches1 = -1; the mutual
 * check (match.cpp:84-91) follows.  Cosine metric.
 * --------------------------------------------------------------------------------------------- */
int sfm_oracle_match_ratio(const float *descA, const double *ptsA, int nA, const float *descB, const double *ptsB,
                           int nB, int dim, float threshold, float maxDistancePxSquared, float ratio,
                           int *matches1_out, int *matches2_out, int *pairs_out)
{
    int *m1 = (int *)malloc((size_t)(nA > 0 ? nA : 1) * sizeof(int));
    int *m2 = (int *)malloc((size_t)(nB > 0 ? nB : 1) * sizeof(int));
    if (!m1 || !m2) {
        free(m1);
        free(m2);
        return -1;
    }
    int n = sfm_oracle_match(descA, ptsA, nA, descB, ptsB, nB, dim, 0, threshold, maxDistancePxSquared, m1, m2, NULL);
    if (n < 0) {
        free(m1);
        free(m2);
        return -1;
    }
    const double r2 = (double)ratio * (double)ratio;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; i++) {
        const int j1 = m1[i];
        if (j1 < 0) continue;
        const float *a = descA + (size_t)i * dim;
        const double s1 = (double)cosine_similarity(a, descB + (size_t)j1 * dim, (size_t)dim);
        for (int j = 0; j < nB; j++) {
            if (j == j1) continue;
            const double dx = ptsA[2 * i] - ptsB[2 * j], dy = ptsA[2 * i + 1] - ptsB[2 * j + 1];
            if (!(dx * dx + dy * dy < maxDistancePxSquared)) continue;
            const float s = cosine_similarity(a, descB + (size_t)j * dim, (size_t)dim);
            if (!(s == s)) continue;
            if (!((1.0 - s1) < r2 * (1.0 - (double)s))) {
                m1[i] = -1;
                break;
            }
        }
    }
    int count = 0;
    for (int i = 0; i < nA; i++) {
        const int j = m1[i];
        if (j != -1 && m2[j] == i) {
            if (pairs_out) {
                pairs_out[2 * count] = i;
                pairs_out[2 * count + 1] = j;
            }
            count++;
        }
    }
    if (matches1_out) memcpy(matches1_out, m1, (size_t)nA * sizeof(int));
    if (matches2_out) memcpy(matches2_out, m2, (size_t)nB * sizeof(int));
    free(m1);
    free(m2);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * Extension with no counterpart in the reference (PARITY UNPINNED by it; the reference has float
 * descriptors only, SfM.hpp:132): the structure of matchTwoSided (match.cpp:24-94) on binary
 * descriptors under the Hamming distance.  A cell is admissible when popcount(a xor b) <=
 * max_distance and the keypoints are inside the pixel gate; every column takes its nearest admissible
 * row and every row its nearest admissible column, the first of equals (the strict scans of
 * match.cpp:49,72); mutual pairs in ascending row order (match.cpp:84-91).
 * --------------------------------------------------------------------------------------------- */
int sfm_oracle_match_hamming(const uint8_t *descA, const double *ptsA, int nA, const uint8_t *descB, const double *ptsB,
                             int nB, int bytes, int max_distance, float maxDistancePxSquared, int *matches1_out,
                             int *matches2_out, int *pairs_out)
{
    int *m1 = (int *)malloc((size_t)(nA > 0 ? nA : 1) * sizeof(int));
    int *m2 = (int *)malloc((size_t)(nB > 0 ? nB : 1) * sizeof(int));
    int *d2 = (int *)malloc((size_t)(nB > 0 ? nB : 1) * sizeof(int));
    if (!m1 || !m2 || !d2) {
        free(m1);
        free(m2);
        free(d2);
        return -1;
    }
    for (int j = 0; j < nB; j++) {
        m2[j] = -1;
        d2[j] = 0x7FFFFFFF;
    }
    for (int i = 0; i < nA; i++) {
        int best = 0x7FFFFFFF, bestJ = -1;
        for (int j = 0; j < nB; j++) {
            int d = 0;
            for (int k = 0; k < bytes; k++) d += __builtin_popcount((unsigned)(descA[(size_t)i * bytes + k] ^ descB[(size_t)j * bytes + k]));
            const double dx = ptsA[2 * i] - ptsB[2 * j], dy = ptsA[2 * i + 1] - ptsB[2 * j + 1];
            if (d > max_distance || !(dx * dx + dy * dy < maxDistancePxSquared)) continue;
            if (d < best) {
                best = d;
                bestJ = j;
            }
            if (d < d2[j]) { /* rows come in ascending order: the first of equals stays */
                d2[j] = d;
                m2[j] = i;
            }
        }
        m1[i] = bestJ;
    }
    int count = 0;
    for (int i = 0; i < nA; i++) {
        const int j = m1[i];
        if (j != -1 && m2[j] == i) {
            if (pairs_out) {
                pairs_out[2 * count] = i;
                pairs_out[2 * count + 1] = j;
            }
            count++;
        }
    }
    if (matches1_out) memcpy(matches1_out, m1, (size_t)nA * sizeof(int));
    if (matches2_out) memcpy(matches2_out, m2, (size_t)nB * sizeof(int));
    free(m1);
    free(m2);
    free(d2);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * Extension with no counterpart in the reference (PARITY UNPINNED by it; its metrics are cosine and
 * dot, match.hpp:18-46): the structure of matchTwoSided (match.cpp:24-94) under the squared Euclidean
 * distance in its expanded form.  d^2 = (|a|^2 + |b|^2) - 2 a.b with the three sums accumulated in one
 * loop in index order (as cosineSimilarity accumulates them, match.hpp:38-43), negative values clamped
 * to 0.  A cell is admissible when d^2 < maxDistanceSquared and the keypoints are inside the pixel gate;
 * columns and rows take their nearest admissible cell, the first of equals; ratio > 0: a row keeps its
 * best at d1 only if every other column inside the gate has d1^2 < ratio^2 d2^2 (double); mutual pairs
 * in ascending row order (match.cpp:84-91).
 * --------------------------------------------------------------------------------------------- */
static float squared_distance(const float *a, const float *b, size_t n)
{
    float dot = 0.0f, normA = 0.0f, normB = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        dot += a[i] * b[i];
        normA += a[i] * a[i];
        normB += b[i] * b[i];
    }
    float d2 = (normA + normB) - 2.0f * dot;
    return d2 < 0.0f ? 0.0f : d2;
}

int sfm_oracle_match_l2(const float *descA, const double *ptsA, int nA, const float *descB, const double *ptsB, int nB,
                        int dim, float maxDistanceSquared, float maxDistancePxSquared, float ratio, int *matches1_out,
                        int *matches2_out, int *pairs_out)
{
    int *m1 = (int *)malloc((size_t)(nA > 0 ? nA : 1) * sizeof(int));
    int *m2 = (int *)malloc((size_t)(nB > 0 ? nB : 1) * sizeof(int));
    float *d1 = (float *)malloc((size_t)(nA > 0 ? nA : 1) * sizeof(float));
    if (!m1 || !m2 || !d1) {
        free(m1);
        free(m2);
        free(d1);
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < nB; j++) {
        int best = -1;
        float bestD = INFINITY;
        for (int i = 0; i < nA; i++) {
            const float d = squared_distance(descA + (size_t)i * dim, descB + (size_t)j * dim, (size_t)dim);
            const double dx = ptsA[2 * i] - ptsB[2 * j], dy = ptsA[2 * i + 1] - ptsB[2 * j + 1];
            if (d < maxDistanceSquared && dx * dx + dy * dy < maxDistancePxSquared && d < bestD) {
                bestD = d;
                best = i;
            }
        }
        m2[j] = best;
    }
    const double r2 = (double)ratio * (double)ratio;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; i++) {
        int best = -1;
        float bestD = INFINITY;
        for (int j = 0; j < nB; j++) {
            const float d = squared_distance(descA + (size_t)i * dim, descB + (size_t)j * dim, (size_t)dim);
            const double dx = ptsA[2 * i] - ptsB[2 * j], dy = ptsA[2 * i + 1] - ptsB[2 * j + 1];
            if (d < maxDistanceSquared && dx * dx + dy * dy < maxDistancePxSquared && d < bestD) {
                bestD = d;
                best = j;
            }
        }
        m1[i] = best;
        d1[i] = bestD;
        if (best >= 0 && ratio > 0.0f) {
            for (int j = 0; j < nB; j++) {
                if (j == best) continue;
                const double dx = ptsA[2 * i] - ptsB[2 * j], dy = ptsA[2 * i + 1] - ptsB[2 * j + 1];
                if (!(dx * dx + dy * dy < maxDistancePxSquared)) continue;
                const float d = squared_distance(descA + (size_t)i * dim, descB + (size_t)j * dim, (size_t)dim);
                if (!(d == d)) continue;
                if (!((double)bestD < r2 * (double)d)) {
                    m1[i] = -1;
                    break;
                }
            }
        }
    }
    int count = 0;
    for (int i = 0; i < nA; i++) {
        const int j = m1[i];
        if (j != -1 && m2[j] == i) {
            if (pairs_out) {
                pairs_out[2 * count] = i;
                pairs_out[2 * count + 1] = j;
            }
            count++;
        }
    }
    if (matches1_out) memcpy(matches1_out, m1, (size_t)nA * sizeof(int));
    if (matches2_out) memcpy(matches2_out, m2, (size_t)nB * sizeof(int));
    free(m1);
    free(m2);
    free(d1);
    return count;
}
