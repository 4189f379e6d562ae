// STAND-IN for the reference's submodules/match/match.hpp (compile test of adaptors/match_amd.cpp without
// OpenCV / Eigen): the declarations of match.hpp:8-13 (types), :18-46 (the two metrics), :55-61
// (MATCHING_OPTIONS) and :70-83 (the three functions).  A build against the real tree uses the real header.
#pragma once
#include "../SfM.hpp"
#include <cmath>
#include <functional>
#include <tuple>

namespace SfM::match
{
    using similarityFunction = std::function<float(const std::vector<float> &, const std::vector<float> &)>;

    enum matchingAlgorithm
    {
        TWO_SIDED_MATHICHNG,
    };

    inline float dotProduct(const std::vector<float> &a, const std::vector<float> &b)
    {
        float acc = 0.0f;
        for (size_t k = 0; k < a.size(); ++k) acc += a[k] * b[k];
        return acc;
    }

    inline float cosineSimilarity(const std::vector<float> &a, const std::vector<float> &b)
    {
        float ab = 0.0f, aa = 0.0f, bb = 0.0f; // three running sums in one pass, index order
        for (size_t k = 0; k < a.size(); ++k) {
            ab += a[k] * b[k];
            aa += a[k] * a[k];
            bb += b[k] * b[k];
        }
        return ab / (std::sqrt(aa) * std::sqrt(bb));
    }

    struct MATCHING_OPTIONS
    {
        matchingAlgorithm algorithm = matchingAlgorithm::TWO_SIDED_MATHICHNG;
        similarityFunction similarityMetric = cosineSimilarity;
        float threshold = 0.5f;
        float maxDistancePxSquared = 10000 * 10000;
    };

    std::vector<std::tuple<int, int>> match(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                            const MATCHING_OPTIONS &options);
    std::vector<std::tuple<int, int>> matchTwoSided(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                                    const MATCHING_OPTIONS &options);
    std::vector<std::tuple<int, int>> matchTwoSided2(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2,
                                                     const MATCHING_OPTIONS &options);
} // namespace SfM::match
