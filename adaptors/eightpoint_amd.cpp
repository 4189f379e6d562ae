// Adaptor (optional): the two point-set functions of the reference's submodules/solve/eightPoint.cpp on the
// device.  The reference keeps eightPoint.cpp in its build (it needs eightPointAlgorithm for the refit after
// RANSAC, scene.cpp:288); a maintainer who wants that refit on the GPU as well takes these two definitions and
// keeps the third overload of eightPoint.cpp (the synthetic-data harness, eightPoint.cpp:10-55) in a file of
// its own.  In this repository the file also completes the link of the compile test.
#include "solve/solve.hpp"

#include "sfm_amd_context.hpp"

namespace SfM::solve
{
    static_assert(sizeof(Vec2) == 2 * sizeof(double) && sizeof(Vec3) == 3 * sizeof(double), "packed doubles");

    // eightPoint.cpp:57-223
    EightPointResult eightPointAlgorithm(const std::vector<Vec2> &points1, const std::vector<Vec2> &points2)
    {
        if (points1.size() < 8)
            throw std::invalid_argument("EightPointAlgorithm: There have to be at least 8 pair-to-pair-correspondences (was: " +
                                        std::to_string(points1.size()) + ") to approximate a pose.");
        if (points2.size() < points1.size()) throw std::invalid_argument("EightPointAlgorithm: points2 is shorter than points1");
        sfm_ctx *ctx = sfm_amd_adaptor::context();
        EightPointResult result;
        result.points.resize(points1.size());
        const int rc = sfm_eight_point_algorithm(ctx, points1[0].data(), points2[0].data(), static_cast<int>(points1.size()),
                                                 result.pose.data(), result.points[0].data());
        if (rc != SFM_OK) sfm_amd_adaptor::fail(ctx, rc, "sfm_eight_point_algorithm");
        return result;
    }

    // eightPoint.cpp:225-372
    EightPointResult eightPointAlgorithmFromSubset(const std::vector<Vec2> &calcPoints1, const std::vector<Vec2> &calcPoints2,
                                                   const std::vector<Vec2> &allPoints1, const std::vector<Vec2> &allPoints2)
    {
        if (calcPoints1.size() < 8)
            throw std::invalid_argument("EightPointFromSubset: There have to be at least 8 pair-to-pair-correspondences (was: " +
                                        std::to_string(calcPoints1.size()) + ") to approximate a pose.");
        if (calcPoints2.size() < calcPoints1.size() || allPoints2.size() < allPoints1.size())
            throw std::invalid_argument("EightPointFromSubset: the second point set is shorter than the first");
        sfm_ctx *ctx = sfm_amd_adaptor::context();
        EightPointResult result;
        result.points.resize(allPoints1.size());
        const int rc = sfm_eight_point_from_subset(ctx, calcPoints1[0].data(), calcPoints2[0].data(), static_cast<int>(calcPoints1.size()),
                                                   allPoints1.empty() ? nullptr : allPoints1[0].data(),
                                                   allPoints2.empty() ? nullptr : allPoints2[0].data(), static_cast<int>(allPoints1.size()),
                                                   result.pose.data(), allPoints1.empty() ? nullptr : result.points[0].data());
        if (rc != SFM_OK) sfm_amd_adaptor::fail(ctx, rc, "sfm_eight_point_from_subset");
        return result;
    }
} // namespace SfM::solve
