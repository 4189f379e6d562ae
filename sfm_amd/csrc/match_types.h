// Types shared by the matcher and the stages that consume its results on the device.
#pragma once
#include "common.h"

#include <vector>

struct PairRec {
    int32_t nA, nB;
    int64_t rowA32, rowB32; // first row in desc32 / norms / pts
    int64_t rowA16, rowB16; // first row in desc16
    int64_t bestRow, bestCol; // first slot in the best[] array of the chunk
    float thr_m;            // candidate bound on the approximate cosine
    int32_t thr_i;          // the same bound on the int8 dot product (instances that use the int8 table)
    int32_t chunkLocal;     // index inside the chunk
    int32_t pad_;
};

struct ImageRec {
    int32_t n;
    int32_t pad;
    int64_t row32;
    int64_t row16;
};

struct sfm_imageset {
    sfm_ctx *ctx = nullptr;
    int n_images = 0;
    int dim = 0;
    std::vector<ImageRec> images;
    std::vector<int> flags;
    int64_t rows32 = 0, rows16 = 0;
    DevBuf desc32, desc16, desc8, desc8u, norms, pts, imageRecs, imageFlags, imageStats;
    std::vector<float> scale8; // per image: int8 value = round(unit-normalised element * scale8)
    std::vector<float> l1max;  // per image: largest L1 norm of a unit-normalised descriptor
    bool owns_desc32 = true, owns_pts = true;
    bool u8_exact = false;     // desc8u holds a byte copy of desc32 (every element an integer in [0, 255], dim 128)
};

struct sfm_match_result {
    sfm_ctx *ctx = nullptr;
    int n_pairs = 0;
    DevBuf offsets; // int64[n_pairs+1]
    DevBuf pairs;   // int2[sum min(nA,nB)]
    DevBuf pairRecs, work, cand, candCount, pairOvf, best, ovfList, counts, stats, m1, m2, kill;
    PinBuf hostRecs, hostWork;
    int64_t total = -1;
};

