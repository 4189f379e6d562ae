// Adaptor: replaces the reference's submodules/match/match.cpp.
//
// Defines SfM::match::match (match.hpp:70) on top of libsfm_amd.so, and keeps a CPU path with the
// reference's semantics for options.similarityMetric functors the device has no kernel for
// (matchTwoSided / matchTwoSided2, declared in match.hpp:81-83).  Build: add this file to the reference's
// CMake target in place of submodules/match/match.cpp, with <reference>/submodules and
// <sfm_amd>/include on the include path, and link libsfm_amd.so.
#include "match/match.hpp"

#include "sfm_amd_context.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace SfM::match
{
    namespace
    {
        // Keypoint is an array of structures with one heap block per descriptor (SfM.hpp:125-140); the
        // library takes one contiguous descriptor table and one point table per image.
        void pack(const std::vector<Keypoint> &kps, size_t dim, float *desc, double *pts)
        {
            for (size_t i = 0; i < kps.size(); i++) {
                if (kps[i].descriptor.size() != dim)
                    throw std::invalid_argument("Match: descriptors of different lengths (" + std::to_string(dim) + " and " +
                                                std::to_string(kps[i].descriptor.size()) + ")");
                std::memcpy(desc + i * dim, kps[i].descriptor.data(), dim * sizeof(float));
                pts[2 * i] = kps[i].point[0];
                pts[2 * i + 1] = kps[i].point[1];
            }
        }

        // Which of the two metrics of match.hpp:18-46 the std::function holds (they are inline functions
        // with linkage, so their addresses identify them); -1 for anything else.
        int device_metric(const similarityFunction &f)
        {
            using fn_t = float (*)(const std::vector<float> &, const std::vector<float> &);
            const fn_t *target = f.target<fn_t>();
            if (target && *target == &cosineSimilarity) return SFM_METRIC_COSINE;
            if (target && *target == &dotProduct) return SFM_METRIC_DOT;
            return -1;
        }
    } // namespace

    std::vector<std::tuple<int, int>> match(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                            const MATCHING_OPTIONS &options)
    {
        if (options.algorithm != matchingAlgorithm::TWO_SIDED_MATHICHNG) // match.cpp:12-19
            throw std::invalid_argument("Match: Unknown matching algorithm. Was " + std::to_string(static_cast<int>(options.algorithm)));
        const int metric = device_metric(options.similarityMetric);
        if (metric < 0) return matchTwoSided(keypoints1, keypoints2, options); // a user functor: the CPU path below
        std::vector<std::tuple<int, int>> result;
        if (keypoints1.empty() || keypoints2.empty()) return result; // the reference's loops do not run

        const size_t dim = keypoints1[0].descriptor.size();
        sfm_ctx *ctx = sfm_amd_adaptor::context();
        static thread_local sfm_amd_adaptor::PinnedTable tables[4]; // packed tables stay page-locked from call to call
        // A video is matched frame i against i + 1, then i + 1 against i + 2 (scene.cpp:60): image A of a call is
        // image B of the call before.  With SFM_AMD_REUSE_FRAMES=1 in the environment the caller promises that a
        // keypoint vector seen again at the same address and size has not changed; the table packed for it last
        // time becomes this call's table A, and the library (option "match_pair_reuse") takes it from the device.
        static thread_local const Keypoint *lastB = nullptr;
        static thread_local size_t lastCount = 0, lastDim = 0;
        static const bool reuse = [] {
            const char *v = std::getenv("SFM_AMD_REUSE_FRAMES");
            return v && std::atoi(v) != 0;
        }();
        static thread_local bool optionSet = false;
        if (reuse && !optionSet) {
            if (sfm_ctx_set_option(ctx, "match_pair_reuse", 1) != SFM_OK) sfm_amd_adaptor::fail(ctx, SFM_ERR_INVALID_ARG, "sfm_ctx_set_option");
            optionSet = true;
        }
        const bool sameA = reuse && keypoints1.data() == lastB && keypoints1.size() == lastCount && dim == lastDim;
        lastB = nullptr;
        if (sameA) {
            std::swap(tables[0].p, tables[1].p);
            std::swap(tables[0].cap, tables[1].cap);
            std::swap(tables[2].p, tables[3].p);
            std::swap(tables[2].cap, tables[3].cap);
        }
        float *descA = tables[0].reserve<float>(ctx, keypoints1.size() * dim), *descB = tables[1].reserve<float>(ctx, keypoints2.size() * dim);
        double *ptsA = tables[2].reserve<double>(ctx, keypoints1.size() * 2), *ptsB = tables[3].reserve<double>(ctx, keypoints2.size() * 2);
        if (!sameA) pack(keypoints1, dim, descA, ptsA);
        pack(keypoints2, dim, descB, ptsB);
        const int capacity = static_cast<int>(std::min(keypoints1.size(), keypoints2.size()));
        std::vector<int32_t> pairs(2 * static_cast<size_t>(capacity));
        int count = 0;
        const int rc = sfm_match_pair(ctx, SFM_MATCH_TWO_SIDED, descA, ptsA, static_cast<int>(keypoints1.size()),
                                      descB, ptsB, static_cast<int>(keypoints2.size()), static_cast<int>(dim), metric,
                                      options.threshold, options.maxDistancePxSquared, pairs.data(), capacity, &count, nullptr, nullptr);
        if (rc != SFM_OK) sfm_amd_adaptor::fail(ctx, rc, "sfm_match_pair");
        lastB = keypoints2.data();
        lastCount = keypoints2.size();
        lastDim = dim;
        result.reserve(count);
        for (int k = 0; k < count; k++) result.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
        return result;
    }

    // CPU path for similarityMetric plug-ins: the semantics of matchTwoSided (match.cpp:24-94).  Every cell is
    // scored through the functor; a cell counts when its similarity exceeds the threshold and the keypoints
    // are closer than the pixel gate; every column and every row keeps its best counting cell (the first of
    // equals, similarity above zero); mutual pairs are returned in ascending row order.
    std::vector<std::tuple<int, int>> matchTwoSided(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                                    const MATCHING_OPTIONS &options)
    {
        const int rows = static_cast<int>(keypoints1.size()), cols = static_cast<int>(keypoints2.size());
        std::vector<float> score(static_cast<size_t>(rows) * cols, 0.0f); // column after column
        std::vector<int> bestRowOfCol(cols, -1), bestColOfRow(rows, -1);
#pragma omp parallel for schedule(static)
        for (int c = 0; c < cols; c++) {
            float top = 0;
            for (int r = 0; r < rows; r++) {
                const float s = options.similarityMetric(keypoints1[r].descriptor, keypoints2[c].descriptor);
                const REAL dx = keypoints1[r].point[0] - keypoints2[c].point[0], dy = keypoints1[r].point[1] - keypoints2[c].point[1];
                if (s > options.threshold && dx * dx + dy * dy < options.maxDistancePxSquared) {
                    score[static_cast<size_t>(c) * rows + r] = s;
                    if (s > top) {
                        top = s;
                        bestRowOfCol[c] = r;
                    }
                }
            }
        }
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; r++) {
            float top = 0;
            for (int c = 0; c < cols; c++) {
                const float s = score[static_cast<size_t>(c) * rows + r];
                if (s > top) {
                    top = s;
                    bestColOfRow[r] = c;
                }
            }
        }
        std::vector<std::tuple<int, int>> result;
        for (int r = 0; r < rows; r++)
            if (bestColOfRow[r] != -1 && bestRowOfCol[bestColOfRow[r]] == r) result.emplace_back(r, bestColOfRow[r]);
        return result;
    }

    // match.cpp:96-191 computes the same lists without the matrix (match.cpp:154-167); one implementation serves both names.
    std::vector<std::tuple<int, int>> matchTwoSided2(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                                     const MATCHING_OPTIONS &options)
    {
        return matchTwoSided(keypoints1, keypoints2, options);
    }
} // namespace SfM::match
