// Adaptor: replaces the reference's submodules/solve/RANSAC.cpp.
//
// Defines SfM::solve::RANSAC (solve.hpp:137).  With the default model (eightPointAlgorithmFromSubset), the
// default loss (the lambda of solve.hpp:119-124) and minN = 8 the whole loop runs in libsfm_amd.so: one call
// that draws the sample table on the device, fits and scores the rows in growing slices, applies the adaptive
// iteration cap (RANSAC.cpp:245-258) and the wall-clock cap (RANSAC.cpp:189-196) between slices and returns the
// winner's inlier list.  Any other model or loss is a plug-in the device has no kernel for: the CPU loop below
// runs it with the reference's rules.
// The sample table's seed comes from the environment variable SFM_AMD_RANSAC_SEED when it is set (repeatable
// runs); otherwise from the clock, as the reference seeds its engines (RANSAC.cpp:170).
#include "solve/solve.hpp"

#include "sfm_amd_context.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <iostream>
#include <limits>
#include <numeric>
#include <random>

namespace SfM::solve
{
    namespace
    {
        // RANSAC.cpp:14-32
        int safeCastToInt(REAL value)
        {
            if (std::isnan(value)) return 0;
            if (value >= static_cast<REAL>(std::numeric_limits<int>::max())) return std::numeric_limits<int>::max();
            if (value <= static_cast<REAL>(std::numeric_limits<int>::min())) return std::numeric_limits<int>::min();
            return static_cast<int>(value);
        }

        bool is_default(const RANSAC_OPTIONS &options)
        {
            using fit_t = EightPointResult (*)(const std::vector<Vec2> &, const std::vector<Vec2> &, const std::vector<Vec2> &,
                                               const std::vector<Vec2> &);
            const fit_t *fit = options.model.target<fit_t>();
            // the default loss is a captureless lambda in the struct's member initialiser: one closure type
            return fit && *fit == &eightPointAlgorithmFromSubset && options.minN == 8 &&
                   options.loss.target_type() == RANSAC_OPTIONS{}.loss.target_type();
        }

        // The reference's loop for plug-in models and losses, one iteration after the other (RANSAC.cpp:166-263
        // as a single worker takes them): shuffle, fit the first minN, score every correspondence, keep the
        // hypothesis with more inliers or, among equals, the smaller error sum, lower the iteration cap.
        std::vector<int> ransac_on_host(const std::vector<Vec2> &x, const std::vector<Vec2> &y, const Mat3 &K, const RANSAC_OPTIONS &options,
                                        const bool verbose)
        {
            const auto start = std::chrono::steady_clock::now();
            const auto limit = std::chrono::milliseconds(options.maxTimeMs);
            const char *env = std::getenv("SFM_AMD_RANSAC_SEED");
            std::default_random_engine rng(env ? std::strtoull(env, nullptr, 10)
                                               : static_cast<unsigned long long>(std::chrono::system_clock::now().time_since_epoch().count()));
            std::vector<int> indices(x.size());
            std::iota(indices.begin(), indices.end(), 0);
            std::vector<Vec2> sampleX(options.minN), sampleY(options.minN);
            std::vector<int> inliers, bestInliers;
            inliers.reserve(x.size());
            REAL bestError = std::numeric_limits<REAL>::infinity();
            int maxIter = options.maxIter, iterationsDone = 0;
            for (int i = 0; i < options.maxIter && iterationsDone < maxIter; ++i) {
                if (i % 50 == 0 && std::chrono::steady_clock::now() - start > limit) break;
                iterationsDone++;
                std::shuffle(indices.begin(), indices.end(), rng);
                for (int k = 0; k < options.minN; k++) {
                    sampleX[k] = x[indices[k]];
                    sampleY[k] = y[indices[k]];
                }
                const EightPointResult fit = options.model(sampleX, sampleY, x, y);
                inliers.clear();
                REAL totalError = 0;
                for (size_t k = 0; k < x.size(); k++) {
                    const REAL error = options.loss(K, x[k], y[k], fit.points[k], fit.pose);
                    if (error < options.maxSquaredError) {
                        inliers.push_back(static_cast<int>(k));
                        totalError += error;
                    }
                }
                if (inliers.empty()) continue;
                if (inliers.size() > bestInliers.size() || (inliers.size() == bestInliers.size() && totalError < bestError)) {
                    bestInliers = inliers;
                    bestError = totalError;
                    const REAL ratio = static_cast<REAL>(bestInliers.size()) / static_cast<REAL>(x.size());
                    const REAL allInliers = std::pow(ratio, options.minN);
                    if (allInliers >= EPSILON) {
                        const int cap = safeCastToInt(std::ceil(std::log(1. - options.successProb) / std::log(1 - allInliers)));
                        if (verbose)
                            std::cout << "RANSAC: found better model with " << bestInliers.size() << " inliers that have a total error of "
                                      << totalError << ", maxIter adjusted to " << std::min(cap, maxIter) << ".\n";
                        if (cap < maxIter) maxIter = cap;
                    }
                }
            }
            if (verbose)
                std::cout << "RANSAC: finished after " << iterationsDone << " iterations ("
                          << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() << "ms)."
                          << std::endl;
            return bestInliers;
        }
    } // namespace

    std::vector<int> RANSAC(const std::vector<Vec2> &x, const std::vector<Vec2> &y, const Mat3 &K, const RANSAC_OPTIONS &options,
                            const bool verbose)
    {
        if (x.size() != y.size() || x.size() < static_cast<size_t>(options.minN)) return {}; // RANSAC.cpp:148-149
        if (!is_default(options)) return ransac_on_host(x, y, K, options, verbose);

        const auto start = std::chrono::steady_clock::now();
        if (options.maxTimeMs <= 0) {
            // RANSAC.cpp:189-196: the clock is read before the first iteration (i = 0), any time at all has passed
            // by then, so a limit of zero or less stops the loop before a single row is fitted.
            if (verbose) std::cout << "RANSAC: finished after 0 iterations (0ms)." << std::endl;
            return {};
        }
        sfm_ctx *ctx = sfm_amd_adaptor::context();
        const char *env = std::getenv("SFM_AMD_RANSAC_SEED");
        int rc = sfm_ctx_set_option(ctx, "ransac_seed", env ? static_cast<int64_t>(std::strtoull(env, nullptr, 10)) : 0);
        if (rc == SFM_OK) rc = sfm_ctx_set_option(ctx, "ransac_max_time_ms", std::max(options.maxTimeMs, 0));
        if (rc != SFM_OK) sfm_amd_adaptor::fail(ctx, rc, "sfm_ctx_set_option");
        // Vec2 is two contiguous doubles (Eigen::Matrix<double, 2, 1>), so the vectors are double[2n]; Mat3 is
        // column-major, which is what the library takes.
        const int n = static_cast<int>(x.size());
        const int rows = std::max(options.maxIter, 0);
        std::vector<int> inliers(x.size());
        int count = 0, iterations = 0;
        static_assert(sizeof(Vec2) == 2 * sizeof(double), "Vec2 must be two packed doubles");
        // The reference reports every improvement of the kept model as it happens (RANSAC.cpp:251-254).  The device
        // scans the rows itself; for a verbose call it also returns every row's inlier count and error sum (all rows
        // are then scored in one go, so the time limit is not looked at during this call), and the lines are made
        // from them below by the reference's own rule.
        std::vector<int32_t> rowCount(verbose ? rows : 0);
        std::vector<double> rowError(verbose ? rows : 0);
        rc = sfm_ransac_pair(ctx, x[0].data(), y[0].data(), n, K.data(), /*samples: drawn on the device*/ nullptr, rows, options.minN,
                             options.maxSquaredError, options.successProb, /*adaptive*/ 1, SFM_LOSS_REPROJECTION, inliers.data(), n,
                             &count, nullptr, nullptr, verbose && rows ? rowCount.data() : nullptr,
                             verbose && rows ? rowError.data() : nullptr, &iterations);
        if (rc != SFM_OK) sfm_amd_adaptor::fail(ctx, rc, "sfm_ransac_pair");
        inliers.resize(count);
        if (verbose) {
            int bestCount = 0, maxIter = options.maxIter;
            REAL bestError = std::numeric_limits<REAL>::infinity();
            for (int h = 0; h < iterations; h++) { // the rows an in-order loop visits before it stops
                if (rowCount[h] <= 0) continue;
                if (rowCount[h] > bestCount || (rowCount[h] == bestCount && rowError[h] < bestError)) {
                    bestCount = rowCount[h];
                    bestError = rowError[h];
                    const REAL allInliers = std::pow(static_cast<REAL>(bestCount) / static_cast<REAL>(n), options.minN);
                    if (allInliers >= EPSILON) {
                        const int cap = safeCastToInt(std::ceil(std::log(1. - options.successProb) / std::log(1 - allInliers)));
                        std::cout << "RANSAC: found better model with " << bestCount << " inliers that have a total error of " << bestError
                                  << ", maxIter adjusted to " << std::min(cap, maxIter) << ".\n";
                        if (cap < maxIter) maxIter = cap;
                    }
                }
            }
        }
        if (verbose)
            std::cout << "RANSAC: finished after " << iterations << " iterations ("
                      << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() << "ms)."
                      << std::endl;
        return inliers;
    }
} // namespace SfM::solve
