// STAND-IN for the reference's submodules/solve/solve.hpp (compile test of adaptors/ransac_amd.cpp and
// adaptors/eightpoint_amd.cpp without Eigen / Ceres): the declarations of solve.hpp:73-79 (eight-point
// functions), :85-95 (reprojectionError), :108-125 (RANSAC_OPTIONS with its default model and loss) and :137
// (RANSAC).  Bundle adjustment (solve.hpp:41-68) is not part of the path and is left out.  A build against
// the real tree uses the real header.
#pragma once
#include "../SfM.hpp"
#include <algorithm>
#include <functional>

namespace SfM::solve
{
    EightPointResult eightPointAlgorithm(const std::vector<Vec2> &points1, const std::vector<Vec2> &points2);

    EightPointResult eightPointAlgorithmFromSubset(const std::vector<Vec2> &calcPoints1, const std::vector<Vec2> &calcPoints2,
                                                   const std::vector<Vec2> &allPoints1, const std::vector<Vec2> &allPoints2);

    // Squared pixel distance between the projection K * (viewMatrix * [point3d; 1])_{0..2} (divided by its
    // depth) and the normalised observation mapped to pixels by K.
    inline REAL reprojectionError(const Mat3 &K, const Vec2 observation, const Vec3 point3d, const Mat4 &viewMatrix)
    {
        REAL cam[3], p[3], obs[3];
        for (int r = 0; r < 3; r++)
            cam[r] = viewMatrix(r, 0) * point3d[0] + viewMatrix(r, 1) * point3d[1] + viewMatrix(r, 2) * point3d[2] + viewMatrix(r, 3);
        for (int r = 0; r < 3; r++) p[r] = K(r, 0) * cam[0] + K(r, 1) * cam[1] + K(r, 2) * cam[2];
        p[0] /= p[2];
        p[1] /= p[2];
        for (int r = 0; r < 3; r++) obs[r] = K(r, 0) * observation[0] + K(r, 1) * observation[1] + K(r, 2);
        const REAL dx = p[0] - obs[0], dy = p[1] - obs[1];
        return dx * dx + dy * dy;
    }

    struct RANSAC_OPTIONS
    {
        using fittingFunction = std::function<EightPointResult(const std::vector<Vec2> &, const std::vector<Vec2> &,
                                                               const std::vector<Vec2> &, const std::vector<Vec2> &)>;
        using lossFunction = std::function<REAL(const Mat3 &, const Vec2, const Vec2, const Vec3, const Mat4 &)>;

        int minN = 8;
        int maxIter = 512;
        int maxTimeMs = 1000;
        REAL maxSquaredError = 10;
        REAL successProb = static_cast<REAL>(0.99);
        fittingFunction model = eightPointAlgorithmFromSubset;
        lossFunction loss = [](const Mat3 &K, const Vec2 obs1, const Vec2 obs2, const Vec3 point3d, const Mat4 &viewMat)
        {
            REAL loss1 = reprojectionError(K, obs1, point3d, Mat4::Identity());
            REAL loss2 = reprojectionError(K, obs2, point3d, viewMat);
            return std::max(loss1, loss2);
        };
    };

    std::vector<int> RANSAC(const std::vector<Vec2> &x, const std::vector<Vec2> &y, const Mat3 &K, const RANSAC_OPTIONS &options,
                            const bool verbose);
} // namespace SfM::solve
