// Drives the C++ host mirror end to end: reads a binary case file written by the test suite,
// runs match() and RANSAC() with the reference's option structs, writes the results.
//   test_frontend <case.bin> <out.bin>
// case.bin: int32 nA, nB, dim; float thr, gate; float descA[nA*dim]; double ptsA[2nA]; descB; ptsB;
//           int32 n, maxIter; uint64 seed; double maxSqErr, successProb; double K[9]; double x[2n]; double y[2n]
// out.bin:  int32 nMatches; int32 pairs[2*nMatches]; int32 nInliers; int32 inliers[nInliers]
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "sfm_frontend.hpp"

using namespace sfm_amd;

template <typename T> static void rd(FILE *f, T *dst, size_t n)
{
    if (n && fread(dst, sizeof(T), n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
}

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t nA, nB, dim;
    float thr, gate;
    rd(f, &nA, 1); rd(f, &nB, 1); rd(f, &dim, 1); rd(f, &thr, 1); rd(f, &gate, 1);
    auto read_kps = [&](int n) {
        std::vector<float> d((size_t)n * dim);
        std::vector<double> p((size_t)n * 2);
        rd(f, d.data(), d.size());
        rd(f, p.data(), p.size());
        std::vector<Keypoint> k(n);
        for (int i = 0; i < n; i++) {
            k[i].descriptor.assign(d.begin() + (size_t)i * dim, d.begin() + (size_t)(i + 1) * dim);
            k[i].point = Vec2{{p[2 * i], p[2 * i + 1]}};
        }
        return k;
    };
    auto kA = read_kps(nA), kB = read_kps(nB);
    int32_t n, maxIter;
    unsigned long long seed;
    double maxSqErr, successProb;
    Mat3 K;
    rd(f, &n, 1); rd(f, &maxIter, 1); rd(f, &seed, 1); rd(f, &maxSqErr, 1); rd(f, &successProb, 1); rd(f, K.m, 9);
    std::vector<Vec2> x(n), y(n);
    rd(f, &x[0][0], (size_t)2 * n);
    rd(f, &y[0][0], (size_t)2 * n);
    fclose(f);

    match::MATCHING_OPTIONS mo;
    mo.threshold = thr;
    mo.maxDistancePxSquared = gate;
    auto m = match::match(kA, kB, mo);

    // error behaviour of the reference: unknown algorithm throws std::invalid_argument
    bool threw = false;
    try {
        match::MATCHING_OPTIONS bad;
        bad.algorithm = (match::matchingAlgorithm)5;
        match::match(kA, kB, bad);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) return 3;
    threw = false;
    try {
        match::MATCHING_OPTIONS bad;
        bad.similarityMetric = [](const std::vector<float> &, const std::vector<float> &) { return 0.f; };
        match::match(kA, kB, bad);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) return 4;

    solve::RANSAC_OPTIONS ro;
    ro.maxIter = maxIter;
    ro.maxTimeMs = 20000;
    ro.maxSquaredError = maxSqErr;
    ro.successProb = successProb;
    ro.seed = seed;
    auto inl = solve::RANSAC(x, y, K, ro, true);
    if (!solve::RANSAC(x, std::vector<Vec2>(y.begin(), y.begin() + 3), K, ro, false).empty()) return 5;

    // scene.cpp:264-298: refit on the inliers
    std::vector<Vec2> in1, in2;
    for (int v : inl) {
        in1.push_back(x[v]);
        in2.push_back(y[v]);
    }
    EightPointResult refit;
    if (in1.size() >= 8) refit = solve::eightPointAlgorithm(in1, in2);
    threw = false;
    try {
        solve::eightPointAlgorithm(std::vector<Vec2>(x.begin(), x.begin() + 7), std::vector<Vec2>(y.begin(), y.begin() + 7));
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) return 6;

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    int32_t nm = (int32_t)m.size(), ni = (int32_t)inl.size();
    fwrite(&nm, 4, 1, o);
    for (auto &t : m) {
        int32_t a = std::get<0>(t), b = std::get<1>(t);
        fwrite(&a, 4, 1, o);
        fwrite(&b, 4, 1, o);
    }
    fwrite(&ni, 4, 1, o);
    for (int v : inl) {
        int32_t q = v;
        fwrite(&q, 4, 1, o);
    }
    int32_t np3 = (int32_t)refit.points.size();
    fwrite(&np3, 4, 1, o);
    fwrite(refit.pose.data(), 8, 16, o);
    if (np3) fwrite(&refit.points[0][0], 8, (size_t)np3 * 3, o);
    fclose(o);
    return 0;
}
