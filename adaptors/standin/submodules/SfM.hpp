// STAND-IN for the reference's submodules/SfM.hpp, for compile-testing the adaptor translation units
// where Eigen is not installed.  It reproduces the names and memory layouts the adaptors touch
// (submodules/SfM.hpp:8-23 REAL / Vec2 / Vec3 / Mat3 / Mat4 / EPSILON, :125-140 Keypoint,
// :169-173 EightPointResult) and nothing else; a build against the real tree uses the real header.
#pragma once
#include <cstddef>
#include <vector>

namespace SfM
{
    using REAL = double;
    constexpr REAL EPSILON = 1e-6;

    // Fixed-size column vectors and column-major matrices with Eigen's accessors: operator[], operator(),
    // data(), Identity().
    template <int N> struct VecN
    {
        REAL v[N];
        REAL &operator[](int i) { return v[i]; }
        const REAL &operator[](int i) const { return v[i]; }
        REAL *data() { return v; }
        const REAL *data() const { return v; }
    };
    using Vec2 = VecN<2>;
    using Vec3 = VecN<3>;

    template <int N> struct MatN
    {
        REAL m[N * N];
        REAL &operator()(int r, int c) { return m[r + N * c]; }
        const REAL &operator()(int r, int c) const { return m[r + N * c]; }
        REAL *data() { return m; }
        const REAL *data() const { return m; }
        static MatN Identity()
        {
            MatN I{};
            for (int i = 0; i < N; i++) I.m[i + N * i] = 1;
            return I;
        }
    };
    using Mat3 = MatN<3>;
    using Mat4 = MatN<4>;

    struct Observation;

    struct Keypoint
    {
        enum Status
        {
            UNINITIALIZED = -1,
        };
        std::vector<float> descriptor;
        Vec2 point;
        REAL size;
        REAL angle;
        REAL response;
        int octave;
        int trackId = UNINITIALIZED;
        Observation *observation = nullptr;
    };

    struct EightPointResult
    {
        Mat4 pose;
        std::vector<Vec3> points;
    };
} // namespace SfM
