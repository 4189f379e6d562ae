// Host-side C++ mirror of the two reference entry points this library accelerates, written
// over the C-ABI of include/sfm_amd.h.  Same names, argument meaning and error behaviour as
//   SfM::match::match            (reference submodules/match/match.hpp:70, match.cpp:9-22)
//   SfM::match::MATCHING_OPTIONS (match.hpp:55-61)
//   SfM::solve::RANSAC           (solve/solve.hpp:137, RANSAC.cpp:146-270)
//   SfM::solve::RANSAC_OPTIONS   (solve.hpp:108-125)
// The value types below are layout-compatible stand-ins for the Eigen typedefs of
// submodules/SfM.hpp:8-21 (column-major doubles), so an adaptor compiled against the reference's
// headers forwards Eigen's .data() pointers unchanged.  This header needs neither Eigen nor HIP.
#pragma once
#include <functional>
#include <stdexcept>
#include <tuple>
#include <vector>

struct sfm_ctx;

namespace sfm_amd
{
    using REAL = double;                  // SfM.hpp:8
    constexpr REAL EPSILON = 1e-6;        // SfM.hpp:23

    struct Vec2                           // Eigen::Matrix<REAL,2,1>: two contiguous doubles
    {
        REAL v[2];
        REAL &operator[](int i) { return v[i]; }
        const REAL &operator[](int i) const { return v[i]; }
    };
    struct Vec3 { REAL v[3]; REAL &operator[](int i) { return v[i]; } const REAL &operator[](int i) const { return v[i]; } };
    struct Mat3                           // column-major, like Eigen's default
    {
        REAL m[9];
        REAL &operator()(int r, int c) { return m[r + 3 * c]; }
        const REAL &operator()(int r, int c) const { return m[r + 3 * c]; }
        const REAL *data() const { return m; }
    };
    struct Mat4
    {
        REAL m[16];
        REAL &operator()(int r, int c) { return m[r + 4 * c]; }
        const REAL &operator()(int r, int c) const { return m[r + 4 * c]; }
        const REAL *data() const { return m; }
    };

    struct Keypoint                       // SfM.hpp:125-140
    {
        std::vector<float> descriptor;
        Vec2 point;
        REAL size = 0, angle = 0, response = 0;
        int octave = 0;
        int trackId = -1;
        void *observation = nullptr;
    };

    struct EightPointResult               // SfM.hpp:169-173
    {
        Mat4 pose;
        std::vector<Vec3> points;
    };

    // One device context per host thread, created on first use on device 0 (or the device set
    // with set_device before the first call).  Throws std::runtime_error without a gfx950 GPU.
    sfm_ctx *default_context();
    void set_device(int device);

    namespace match
    {
        using similarityFunction = std::function<float(const std::vector<float> &, const std::vector<float> &)>;

        enum matchingAlgorithm
        {
            TWO_SIDED_MATHICHNG, // (sic) match.hpp:12
        };

        // match.hpp:18-46.  Their addresses select the device metric; they are also callable.
        float dotProduct(const std::vector<float> &a, const std::vector<float> &b);
        float cosineSimilarity(const std::vector<float> &a, const std::vector<float> &b);

        struct MATCHING_OPTIONS           // match.hpp:55-61
        {
            matchingAlgorithm algorithm = matchingAlgorithm::TWO_SIDED_MATHICHNG;
            similarityFunction similarityMetric = cosineSimilarity;
            float threshold = 0.5f;
            float maxDistancePxSquared = 10000 * 10000;
        };

        // Throws std::invalid_argument for an unknown algorithm (match.cpp:18) and for a
        // similarityMetric other than the two above (no device kernel, no CPU fallback).
        std::vector<std::tuple<int, int>> match(const std::vector<Keypoint> &keypoints1,
                                                const std::vector<Keypoint> &keypoints2,
                                                const MATCHING_OPTIONS &options);
    } // namespace match

    namespace solve
    {
        // solve.hpp:69 / eightPoint.cpp:57-223: pose and triangulated points from all correspondences
        // (the refit on RANSAC's inliers, scene.cpp:288).  Throws std::invalid_argument below 8 points.
        EightPointResult eightPointAlgorithm(const std::vector<Vec2> &points1, const std::vector<Vec2> &points2);

        // solve.hpp:78.  Runs on the device inside RANSAC(); its address marks the default
        // model.  Calling it on the host fits the pose on the device and triangulates there too.
        EightPointResult eightPointAlgorithmFromSubset(const std::vector<Vec2> &calcPoints1,
                                                       const std::vector<Vec2> &calcPoints2,
                                                       const std::vector<Vec2> &allPoints1,
                                                       const std::vector<Vec2> &allPoints2);

        // The default loss of solve.hpp:119-124 (max of the two reprojection errors, pixel^2);
        // its address marks the default loss.
        REAL defaultLoss(const Mat3 &K, const Vec2 obs1, const Vec2 obs2, const Vec3 point3d, const Mat4 &viewMat);

        struct RANSAC_OPTIONS             // solve.hpp:108-125
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
            lossFunction loss = defaultLoss;
            // Not in the reference: seed of the sample table (the reference seeds from the wall
            // clock, RANSAC.cpp:170; 0 keeps that behaviour).
            unsigned long long seed = 0;
        };

        // Returns {} when sizes differ or are below minN (RANSAC.cpp:148-149).  Throws
        // std::invalid_argument for a model/loss other than the defaults or minN != 8.
        std::vector<int> RANSAC(const std::vector<Vec2> &x, const std::vector<Vec2> &y, const Mat3 &K,
                                const RANSAC_OPTIONS &options, const bool verbose);
    } // namespace solve
} // namespace sfm_amd
