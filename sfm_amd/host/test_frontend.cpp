// Drives the adaptor translation units (adaptors/*.cpp, compiled against the stand-in headers) end to end:
// reads a binary case file written by the test suite, runs SfM::match::match and SfM::solve::RANSAC with the
// reference's option structs, writes the results.
//   test_frontend <case.bin> <out.bin>
//   test_frontend --lifetime        (makes the thread's context and exits without releasing anything)
//   test_frontend --time <case.bin> (wall time of match() and RANSAC() per call, as a caller of the reference sees them:
//                                    packing of the keypoints, upload, kernels, download and the result vector included)
//   test_frontend --sequence <case.bin> (three frames - A, B and A's keypoints in reverse order - matched in the order
//                                    of a video, B of one call being A of the next, and once out of that order;
//                                    every list is compared with the CPU path's when the frames are small, and the
//                                    time per call is printed.  With SFM_AMD_REUSE_FRAMES=1 the adaptor and the
//                                    library skip the packing and the upload of the shared frame)
// case.bin: int32 nA, nB, dim; float thr, gate; float descA[nA*dim]; double ptsA[2nA]; descB; ptsB;
//           int32 n, maxIter; double maxSqErr, successProb; double K[9]; double x[2n]; double y[2n]
// out.bin:  int32 nMatches; int32 pairs[2*nMatches]; int32 nInliers; int32 inliers[nInliers];
//           int32 nPoints; double pose[16]; double points[3*nPoints];          (refit on the inliers)
//           int32 nMatchesPlugin; int32 pairs[...];                            (similarityMetric plug-in, CPU path)
//           int32 nInliersPlugin; int32 inliers[...]                           (loss plug-in, CPU path)
// The sample table's seed comes from the environment (SFM_AMD_RANSAC_SEED), as adaptors/ransac_amd.cpp reads it.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <iostream>

#include "match/match.hpp"
#include "solve/solve.hpp"

using namespace SfM;

template <typename T> static void rd(FILE *f, T *dst, size_t n)
{
    if (n && fread(dst, sizeof(T), n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
}

static void wr_list(FILE *o, const std::vector<int> &v)
{
    const int32_t n = (int32_t)v.size();
    fwrite(&n, 4, 1, o);
    for (int q : v) {
        const int32_t w = q;
        fwrite(&w, 4, 1, o);
    }
}

static void wr_pairs(FILE *o, const std::vector<std::tuple<int, int>> &m)
{
    const int32_t n = (int32_t)m.size();
    fwrite(&n, 4, 1, o);
    for (auto &t : m) {
        const int32_t a = std::get<0>(t), b = std::get<1>(t);
        fwrite(&a, 4, 1, o);
        fwrite(&b, 4, 1, o);
    }
}

static int lifetime()
{
    std::vector<Keypoint> k(40);
    for (size_t i = 0; i < k.size(); i++) {
        k[i].descriptor.assign(128, 1.0f);
        k[i].descriptor[i] = 200.0f;
        k[i].point = Vec2{{(double)i, 0.0}};
    }
    match::MATCHING_OPTIONS mo;
    mo.threshold = 0.9f;
    const auto m = match::match(k, k, mo);
    if (m.size() != k.size()) return 3;
    std::vector<Vec2> x(30), y(30);
    for (int i = 0; i < 30; i++) {
        x[i] = Vec2{{0.01 * i, 0.02 * (i % 7)}};
        y[i] = Vec2{{0.01 * i + 0.05, 0.02 * (i % 7) + 0.001 * i}};
    }
    Mat3 K = Mat3::Identity();
    K(0, 0) = K(1, 1) = 1000;
    solve::RANSAC(x, y, K, solve::RANSAC_OPTIONS{}, false);
    std::cout << "lifetime ok" << std::endl;
    return 0; // the thread-local context, its pair buffers and streams are released by static destruction
}

int main(int argc, char **argv)
{
    if (argc == 2 && !strcmp(argv[1], "--lifetime")) return lifetime();
    const bool timing = argc == 3 && !strcmp(argv[1], "--time");
    const bool sequence = argc == 3 && !strcmp(argv[1], "--sequence");
    if (argc != 3) return 2;
    FILE *f = fopen(argv[timing || sequence ? 2 : 1], "rb");
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
    double maxSqErr, successProb;
    Mat3 K;
    rd(f, &n, 1); rd(f, &maxIter, 1); rd(f, &maxSqErr, 1); rd(f, &successProb, 1); rd(f, K.m, 9);
    std::vector<Vec2> x(n), y(n);
    rd(f, &x[0][0], (size_t)2 * n);
    rd(f, &y[0][0], (size_t)2 * n);
    fclose(f);

    match::MATCHING_OPTIONS mo;
    mo.threshold = thr;
    mo.maxDistancePxSquared = gate;
    if (sequence) {
        std::vector<Keypoint> kC(kA.rbegin(), kA.rend());
        const std::vector<Keypoint> *frames[3] = {&kA, &kB, &kC};
        const int order[5][2] = {{0, 1}, {1, 2}, {0, 2}, {2, 1}, {1, 0}}; // calls 2, 4 and 5 continue the call before them
        const bool check = (double)nA * nB <= 2e6;
        double ms = 0;
        size_t total = 0;
        for (int round = 0; round < (check ? 2 : 12); round++)
            for (int c = 0; c < 5; c++) {
                const auto &k1 = *frames[order[c][0]], &k2 = *frames[order[c][1]];
                const auto t0 = std::chrono::steady_clock::now();
                const auto got = match::match(k1, k2, mo);
                if (round >= 2 || check) ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                total += got.size();
                if (check && got != match::matchTwoSided(k1, k2, mo)) {
                    fprintf(stderr, "sequence: call %d of round %d differs from the CPU path\n", c, round);
                    return 7;
                }
            }
        printf("sequence: %.3f ms per call (%d x %d keypoints, %zu matches in all)\n", ms / (check ? 10 : 50), nA, nB, total);
        return 0;
    }
    if (timing) {
        solve::RANSAC_OPTIONS ro;
        ro.maxIter = maxIter;
        ro.maxTimeMs = 20000;
        ro.maxSquaredError = maxSqErr;
        ro.successProb = successProb;
        auto per_call_ms = [](auto &&fn) {
            for (int i = 0; i < 10; i++) fn();
            double best = 1e30;
            for (int round = 0; round < 3; round++) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 20; i++) fn();
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 20);
            }
            return best;
        };
        size_t nm = 0, ni = 0;
        const double tm = per_call_ms([&] { nm = match::match(kA, kB, mo).size(); });
        const double tr = per_call_ms([&] { ni = solve::RANSAC(x, y, K, ro, false).size(); });
        printf("match(): %.3f ms per call (%d x %d keypoints, %zu matches); RANSAC(): %.3f ms per call (%d correspondences, maxIter %d, %zu inliers)\n",
               tm, nA, nB, nm, tr, n, maxIter, ni);
        return 0;
    }
    auto m = match::match(kA, kB, mo);

    // error behaviour of the reference: unknown algorithm throws std::invalid_argument (match.cpp:18)
    bool threw = false;
    try {
        match::MATCHING_OPTIONS bad;
        bad.algorithm = (match::matchingAlgorithm)5;
        match::match(kA, kB, bad);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) return 3;
    // a similarityMetric plug-in (match.hpp:58) runs on the CPU path with the same semantics
    match::MATCHING_OPTIONS plug = mo;
    plug.similarityMetric = [](const std::vector<float> &a, const std::vector<float> &b) { return match::cosineSimilarity(a, b); };
    auto mPlug = match::match(kA, kB, plug);

    solve::RANSAC_OPTIONS ro;
    ro.maxIter = maxIter;
    ro.maxTimeMs = 20000;
    ro.maxSquaredError = maxSqErr;
    ro.successProb = successProb;
    auto inl = solve::RANSAC(x, y, K, ro, true);
    if (!solve::RANSAC(x, std::vector<Vec2>(y.begin(), y.begin() + 3), K, ro, false).empty()) return 5; // RANSAC.cpp:148
    // a loss plug-in (solve.hpp:118-124) runs the CPU loop; the model stays the default
    solve::RANSAC_OPTIONS roPlug = ro;
    roPlug.maxIter = 400;
    const auto defaultLoss = ro.loss;
    roPlug.loss = [defaultLoss](const Mat3 &Km, const Vec2 o1, const Vec2 o2, const Vec3 p, const Mat4 &view) {
        return defaultLoss(Km, o1, o2, p, view);
    };
    auto inlPlug = solve::RANSAC(x, y, K, roPlug, false);
    // a time limit of zero ends the loop before its first iteration (RANSAC.cpp:189-196): no inliers
    solve::RANSAC_OPTIONS roTime = ro;
    roTime.maxTimeMs = 0;
    if (!solve::RANSAC(x, y, K, roTime, false).empty()) return 8;

    // scene.cpp:264-298: refit on the inliers
    std::vector<Vec2> in1, in2;
    for (int v : inl) {
        in1.push_back(x[v]);
        in2.push_back(y[v]);
    }
    EightPointResult refit;
    refit.pose = Mat4::Identity();
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
    wr_pairs(o, m);
    wr_list(o, inl);
    const int32_t np3 = (int32_t)refit.points.size();
    fwrite(&np3, 4, 1, o);
    fwrite(refit.pose.data(), 8, 16, o);
    if (np3) fwrite(&refit.points[0][0], 8, (size_t)np3 * 3, o);
    wr_pairs(o, mPlug);
    wr_list(o, inlPlug);
    fclose(o);
    return 0;
}
