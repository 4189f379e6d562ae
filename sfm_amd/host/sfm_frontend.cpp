// Implementation of the host-side mirror over the C-ABI.  Host work here is packing
// (Keypoint AoS -> contiguous tables), drawing the sample table, and the in-order scan that
// applies RANSAC's adaptive iteration cap and wall-clock cap between device chunks.
#include "sfm_frontend.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <random>
#include <string>

#include "../../include/sfm_amd.h"

namespace sfm_amd
{
    namespace
    {
        int g_device = 0;
        struct CtxHolder
        {
            sfm_ctx *ctx = nullptr;
            ~CtxHolder()
            {
                if (ctx) sfm_ctx_destroy(ctx);
            }
        };
        thread_local CtxHolder t_ctx;

        [[noreturn]] void fail(sfm_ctx *ctx, int rc, const char *what)
        {
            std::string msg = std::string(what) + ": sfm_amd status " + std::to_string(rc);
            if (ctx) msg += std::string(" (") + sfm_last_error(ctx) + ")";
            if (rc == SFM_ERR_INVALID_ARG || rc == SFM_ERR_UNSUPPORTED) throw std::invalid_argument(msg);
            throw std::runtime_error(msg);
        }
    } // namespace

    void set_device(int device) { g_device = device; }

    sfm_ctx *default_context()
    {
        if (!t_ctx.ctx) {
            int rc = sfm_ctx_create(g_device, &t_ctx.ctx);
            if (rc != SFM_OK) fail(nullptr, rc, "sfm_ctx_create");
        }
        return t_ctx.ctx;
    }

    namespace match
    {
        float dotProduct(const std::vector<float> &a, const std::vector<float> &b)
        {
            float result = 0.0f;
            for (size_t i = 0; i < a.size(); ++i) result += a[i] * b[i];
            return result;
        }

        float cosineSimilarity(const std::vector<float> &a, const std::vector<float> &b)
        {
            float dot = 0.0f, normA = 0.0f, normB = 0.0f;
            for (size_t i = 0; i < a.size(); ++i) {
                dot += a[i] * b[i];
                normA += a[i] * a[i];
                normB += b[i] * b[i];
            }
            return dot / (std::sqrt(normA) * std::sqrt(normB));
        }

        static void pack(const std::vector<Keypoint> &kps, size_t dim, std::vector<float> &desc, std::vector<double> &pts)
        {
            desc.resize(kps.size() * dim);
            pts.resize(kps.size() * 2);
            for (size_t i = 0; i < kps.size(); i++) {
                if (kps[i].descriptor.size() != dim)
                    throw std::invalid_argument("Match: descriptors of different lengths (" + std::to_string(dim) + " vs " +
                                                std::to_string(kps[i].descriptor.size()) + ")");
                std::memcpy(desc.data() + i * dim, kps[i].descriptor.data(), dim * sizeof(float));
                pts[2 * i] = kps[i].point[0];
                pts[2 * i + 1] = kps[i].point[1];
            }
        }

        std::vector<std::tuple<int, int>> match(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                                const MATCHING_OPTIONS &options)
        {
            if (options.algorithm != matchingAlgorithm::TWO_SIDED_MATHICHNG)
                throw std::invalid_argument("Match: Unknown matching algorithm. Was " + std::to_string((int)options.algorithm));
            using fn_t = float (*)(const std::vector<float> &, const std::vector<float> &);
            const fn_t *target = options.similarityMetric.target<fn_t>();
            int metric;
            if (target && *target == &cosineSimilarity)
                metric = SFM_METRIC_COSINE;
            else if (target && *target == &dotProduct)
                metric = SFM_METRIC_DOT;
            else
                throw std::invalid_argument("Match: similarityMetric must be cosineSimilarity or dotProduct; a user functor has no device kernel");

            std::vector<std::tuple<int, int>> result;
            if (keypoints1.empty() || keypoints2.empty()) return result; // the reference's loops do not run
            const size_t dim = keypoints1[0].descriptor.size();
            std::vector<float> dA, dB;
            std::vector<double> pA, pB;
            pack(keypoints1, dim, dA, pA);
            pack(keypoints2, dim, dB, pB);
            const int cap = (int)std::min(keypoints1.size(), keypoints2.size());
            std::vector<int32_t> pairs(2 * (size_t)cap);
            int count = 0;
            sfm_ctx *ctx = default_context();
            int rc = sfm_match_pair(ctx, SFM_MATCH_TWO_SIDED, dA.data(), pA.data(), (int)keypoints1.size(), dB.data(), pB.data(),
                                    (int)keypoints2.size(), (int)dim, metric, options.threshold, options.maxDistancePxSquared,
                                    pairs.data(), cap, &count, nullptr, nullptr);
            if (rc != SFM_OK) fail(ctx, rc, "sfm_match_pair");
            result.reserve(count);
            for (int k = 0; k < count; k++) result.push_back({pairs[2 * k], pairs[2 * k + 1]});
            return result;
        }
    } // namespace match

    namespace solve
    {
        static int safeCastToInt(REAL value) // RANSAC.cpp:14-32
        {
            if (std::isnan(value)) return 0;
            if (value >= static_cast<REAL>(std::numeric_limits<int>::max())) return std::numeric_limits<int>::max();
            if (value <= static_cast<double>(std::numeric_limits<int>::min())) return std::numeric_limits<int>::min();
            return static_cast<int>(value);
        }

        EightPointResult eightPointAlgorithm(const std::vector<Vec2> &points1, const std::vector<Vec2> &points2)
        {
            if (points1.size() < 8) // eightPoint.cpp:59-62
                throw std::invalid_argument("EightPointAlgorithm: There have to be at least 8 pair-to-pair-correspondences (was: " +
                                            std::to_string(points1.size()) + ") to approximate a pose.");
            if (points2.size() < points1.size())
                throw std::invalid_argument("EightPointAlgorithm: points2 is shorter than points1");
            sfm_ctx *ctx = default_context();
            EightPointResult r;
            r.points.resize(points1.size());
            double pose16[16];
            int rc = sfm_eight_point_algorithm(ctx, &points1[0][0], &points2[0][0], (int)points1.size(), pose16, &r.points[0][0]);
            if (rc != SFM_OK) fail(ctx, rc, "sfm_eight_point_algorithm");
            for (int c = 0; c < 4; c++)
                for (int q = 0; q < 4; q++) r.pose(q, c) = pose16[q + 4 * c];
            return r;
        }

        EightPointResult eightPointAlgorithmFromSubset(const std::vector<Vec2> &calcPoints1, const std::vector<Vec2> &calcPoints2,
                                                       const std::vector<Vec2> &allPoints1, const std::vector<Vec2> &allPoints2)
        {
            if (calcPoints1.size() < 8)
                throw std::invalid_argument("EightPointFromSubset: There have to be at least 8 pair-to-pair-correspondences (was: " +
                                            std::to_string(calcPoints1.size()) + ") to approximate a pose.");
            if (calcPoints1.size() != 8 || calcPoints2.size() != 8)
                throw std::invalid_argument("EightPointFromSubset: the device model fits exactly 8 correspondences");
            (void)allPoints1;
            (void)allPoints2;
            sfm_ctx *ctx = default_context();
            const int32_t rows[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            double pose12[12];
            int rc = sfm_eight_point_fit(ctx, &calcPoints1[0][0], &calcPoints2[0][0], 8, rows, 1, pose12);
            if (rc != SFM_OK) fail(ctx, rc, "sfm_eight_point_fit");
            EightPointResult r;
            for (int c = 0; c < 4; c++)
                for (int q = 0; q < 4; q++) r.pose(q, c) = (q == c) ? 1.0 : 0.0;
            for (int q = 0; q < 3; q++) {
                for (int c = 0; c < 3; c++) r.pose(q, c) = pose12[3 * q + c];
                r.pose(q, 3) = pose12[9 + q];
            }
            return r; // points are produced and consumed on the device inside RANSAC()
        }

        REAL defaultLoss(const Mat3 &, const Vec2, const Vec2, const Vec3, const Mat4 &)
        {
            throw std::logic_error("defaultLoss is evaluated on the device inside RANSAC(); it marks the default loss");
        }

        std::vector<int> RANSAC(const std::vector<Vec2> &x, const std::vector<Vec2> &y, const Mat3 &K, const RANSAC_OPTIONS &options,
                                const bool verbose)
        {
            if (x.size() != y.size() || x.size() < (size_t)options.minN) return {};

            using fit_t = EightPointResult (*)(const std::vector<Vec2> &, const std::vector<Vec2> &, const std::vector<Vec2> &,
                                               const std::vector<Vec2> &);
            using loss_t = REAL (*)(const Mat3 &, const Vec2, const Vec2, const Vec3, const Mat4 &);
            const fit_t *fit = options.model.target<fit_t>();
            const loss_t *loss = options.loss.target<loss_t>();
            if (!fit || *fit != &eightPointAlgorithmFromSubset)
                throw std::invalid_argument("RANSAC: model must be eightPointAlgorithmFromSubset; a user functor has no device kernel");
            if (!loss || *loss != &defaultLoss)
                throw std::invalid_argument("RANSAC: loss must be the default reprojection loss; a user functor has no device kernel");
            if (options.minN != 8) throw std::invalid_argument("RANSAC: the device model needs minN == 8");

            auto start = std::chrono::steady_clock::now();
            auto limit = std::chrono::milliseconds(options.maxTimeMs);
            sfm_ctx *ctx = default_context();
            const int n = (int)x.size();
            const int H = std::max(options.maxIter, 0);

            // Sample table: H rows of 8 distinct indices.  The reference shuffles the whole index
            // vector per iteration with a clock-seeded engine (RANSAC.cpp:170,201) and keeps the
            // first 8: a uniform 8-subset in random order, which is what is drawn here.
            unsigned long long seed = options.seed;
            if (seed == 0) seed = (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
            std::mt19937_64 rng(seed);
            std::vector<int32_t> samples((size_t)H * 8);
            for (int h = 0; h < H; h++) {
                int32_t *row = samples.data() + (size_t)h * 8;
                for (int k = 0; k < 8; k++) {
                    for (;;) {
                        int32_t v = (int32_t)(rng() % (unsigned long long)n);
                        bool dup = false;
                        for (int q = 0; q < k; q++) dup |= row[q] == v;
                        if (!dup) {
                            row[k] = v;
                            break;
                        }
                    }
                }
            }

            // Device chunks; between chunks the in-order rules of RANSAC.cpp:182-258.
            const int chunk = 1024;
            std::vector<int32_t> counts(chunk);
            std::vector<double> toterr(chunk);
            std::vector<int32_t> scratch((size_t)n);
            int maxIter = H, iterationsDone = 0, best = -1;
            size_t bestCount = 0;
            REAL bestError = std::numeric_limits<REAL>::infinity();
            for (int h0 = 0; h0 < maxIter; h0 += chunk) {
                if (std::chrono::steady_clock::now() - start > limit) break; // RANSAC.cpp:189-196
                const int m = std::min(chunk, H - h0);
                int cnt = 0;
                int rc = sfm_ransac_pair(ctx, &x[0][0], &y[0][0], n, K.data(), samples.data() + (size_t)h0 * 8, m, 8,
                                         options.maxSquaredError, options.successProb, 0, SFM_LOSS_REPROJECTION, scratch.data(), n,
                                         &cnt, nullptr, nullptr, counts.data(), toterr.data(), nullptr);
                if (rc != SFM_OK) fail(ctx, rc, "sfm_ransac_pair");
                for (int q = 0; q < m; q++) {
                    if (iterationsDone >= maxIter) break; // RANSAC.cpp:182-186
                    iterationsDone++;
                    const size_t c = (size_t)counts[q];
                    if (c > 0 && (c > bestCount || (c == bestCount && toterr[q] < bestError))) { // RANSAC.cpp:229,237-238
                        bestError = toterr[q];
                        bestCount = c;
                        best = h0 + q;
                        REAL inlierRatio = static_cast<REAL>(bestCount) / static_cast<REAL>(x.size());
                        REAL succesProb = std::pow(inlierRatio, options.minN);
                        if (succesProb >= EPSILON) {
                            int newMax = safeCastToInt(std::ceil(std::log(1. - options.successProb) / std::log(1 - succesProb)));
                            if (verbose)
                                std::cout << "RANSAC: found better model with " << bestCount << " inliers that have a total error of "
                                          << toterr[q] << ", maxIter adjusted to " << std::min(newMax, maxIter) << ".\n";
                            if (newMax < maxIter) maxIter = newMax;
                        }
                    }
                }
            }

            std::vector<int> bestInliers;
            if (best >= 0) {
                int cnt = 0;
                int rc = sfm_ransac_pair(ctx, &x[0][0], &y[0][0], n, K.data(), samples.data() + (size_t)best * 8, 1, 8,
                                         options.maxSquaredError, options.successProb, 0, SFM_LOSS_REPROJECTION, scratch.data(), n,
                                         &cnt, nullptr, nullptr, nullptr, nullptr, nullptr);
                if (rc != SFM_OK) fail(ctx, rc, "sfm_ransac_pair");
                bestInliers.assign(scratch.begin(), scratch.begin() + cnt);
            }
            if (verbose)
                std::cout << "RANSAC: finished after " << iterationsDone << " iterations ("
                          << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count()
                          << "ms)." << std::endl;
            return bestInliers;
        }
    } // namespace solve
} // namespace sfm_amd
