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
    float thr6;             // the same bound on the e2m3 dot product (instances that use the 6-bit table)
    int32_t chunkLocal;     // index inside the chunk
    int32_t ia, ib;         // the two images in their set (float matcher: the bounds are made from their statistics on the device)
    int32_t gateOff;        // 1: the pixel gate cannot exclude any cell of this pair (made on the device with the bounds)
};

#define FLAG_VALUES 5 // values a pair's overflow flag takes, 0 included
#define STATS_WORDS 5 // counters of a batch call (MatchArgs::stats)

struct ImageRec {
    int32_t n;
    int32_t pad;
    int64_t row32;
    int64_t row16;
};

typedef unsigned long long u64;

struct MatchArgs {
    const _Float16 *desc16;
    const signed char *desc8; // unit-normalised descriptors quantised to int8, 128 bytes per row
    const unsigned char *desc6; // unit-normalised descriptors as e2m3 (6-bit floats), 128 bytes per row (see ingest_f6_kernel), or null
    const unsigned char *desc8u; // byte copy of desc32 (dim 128, rows as desc32) when every element is an integer in [0, 255], else null
    const float *desc32;
    const float *norms;
    const float *norms2;  // squared norms, fp32 sums in index order (L2 mode)
    const double2 *pts;
    const PairRec *pairs; // chunk base
    const int2 *work;     // (pair in chunk, row block), -1 = no work
    uint2 *cand;          // [pairs in chunk][candCap]
    int *candCount;       // [pairs in chunk][maxQueues]: entries in each wave's slice
    int *pairOvf;         // [pairs in chunk]: a slice overflowed in 4 = the e2m3 strict pass, 3 = the int8 strict pass, 1 = the fp16 strict pass, 2 = the general pass
    int *flagCount;       // [FLAG_VALUES]: how often a flag value was set in this chunk (only zero / non-zero matters)
    int *flagOut;         // single-pair calls: where the pair's final flag goes (beside the packed count), or null
    int onlyFlagged;      // != 0: this pass handles only the pairs whose flag has this value
    int nWork;            // entries of the work list (passes that walk it)
    int orient;           // filtered passes: 0 = image A's rows stay in registers, 1 = image B's
    int gateMode;         // 0 = every pair; 1 = only pairs whose pixel gate excludes nothing; 2 = only the others
    float *rowMax;        // [rows of the chunk, as best]: minus each row's bar, from its largest approximate cosine (two-phase filtered passes)
    int maxQueues;        // slices per pair (row blocks * waves * column parts), upper bound over the batch
    int dotStrict;        // dot metric: 1 = the pairs take the strict passes first, under a cosine bound made from the images' largest norms
    int colSplit;         // strict passes: column parts per row block (0 or 1 = a block walks all columns); work item y = row block | part << 24
    u64 *best;            // row keys and column keys of the chunk
    int *ovfList;         // pairs of the chunk whose candidate list overflowed, [0] = how many
    u64 *stats;           // [0] candidates scored from lists, [1] pairs scored cell by cell, [2] pairs that needed the filtered passes, [3] pairs settled by the fp16 strict pass after the int8 one flooded, [4] pairs the e2m3 pass flagged
    int candCap;
    int blockRows;        // rows of image A per block of the similarity kernel instance in use
    int wavesPerBlock;
    int dim;
    int metric;
    float threshold;
    float maxDistSq;
    double ratioSq;       // ratio mode (> 0): squared ratio of the second-best test
    int *kill;            // ratio mode: [rows of the chunk, as best] 1 = the row's best fails the test
};

// Strict similarity pass over a table of 256 signs per row (g.desc8, rows of 256 B); candidates are the
// cells whose int32 sum exceeds PairRec::thr_i.  Defined beside the similarity kernel.
void sfm_launch_sim_signs(const MatchArgs &g, int blocks, hipStream_t st);

struct sfm_imageset {
    sfm_ctx *ctx = nullptr;
    int n_images = 0;
    int dim = 0;
    std::vector<ImageRec> images;
    std::vector<int> flags;
    int64_t rows32 = 0, rows16 = 0;
    DevBuf desc32, desc16, desc8, desc6, desc8u, norms, norms2, pts, imageRecs, imageStats;
    // imageStats (device, ints holding float bit patterns): [image][2] = largest |element| and largest L1 norm of the
    // unit-normalised descriptors (the int8 value of an element is round(element * 127 / largest)), then
    // [image][2] = 0x7FFFFFFF - bits of the smallest norm, bits of the largest (norms inside the safe range),
    // then [image] = IMG_* flags, [image][4] = the extent of the points, and [image] = the largest distance between a
    // unit-normalised descriptor and its e2m3 copy (float bits, rounded up).
    bool owns_desc32 = true, owns_pts = true;
    bool flags_on_host = true; // false: built without waiting for the device (single-pair calls); flags stays zero
    bool u8_exact = false;     // desc8u holds a byte copy of desc32 (every element an integer in [0, 255], dim 128)
};

// What an observed candidate-list size was observed for.
struct MatchKey {
    const void *set;
    int metric;
    float threshold, maxDistSq, ratio;
    int capMax, simVariant;
};

struct sfm_match_result {
    sfm_ctx *ctx = nullptr;
    int n_pairs = 0;
    DevBuf offsets; // int64[n_pairs+1]
    DevBuf pairs;   // int2[sum min(nA,nB)]
    DevBuf pairRecs, work, cand, candCount, pairOvf, best, ovfList, counts, stats, m1, kill; // m1: packed outputs of a single-pair call
    PinBuf hostRecs, hostWork, hostOut; // hostOut: staging of the single-pair entry point's outputs
    int64_t total = -1;
    DevBuf rowMax;           // two-phase filtered passes: a float per row of a chunk (per buffer set)
    DevBuf candProbe;        // lists of the first chunk of a call that observes how far they fill
    int candObserved = 0;    // candidate slots per pair sized from such a first chunk (0 = none yet)
    int sixSkip = 0;            // calls with the same options that still skip the e2m3 pass, because a call before flooded it
    int64_t sixPairs = 0;       // pairs of the last call that took the e2m3 pass (its counter [4] says how many of them it flagged)
    bool sixFlooded = false;    // in that first chunk half the pairs or more flooded their e2m3 lists: later chunks start with int8
    bool floodObserved = false; // in that first chunk nine pairs in ten flooded their int8 lists and went on to the filtered passes
    MatchKey observedFor = {nullptr, 0, 0.0f, 0.0f, 0.0f, 0, 0};
    size_t packedBytes = 0; // single-pair calls: bytes of m1 before the list (count, per-row and per-column bests)
    hipEvent_t upDone = nullptr;  // the last upload of the last batch call from hostRecs / hostWork has been made
    bool upValid = false;         // ... and that call ran to its end
    hipEvent_t statsEv = nullptr; // the counters of the last batch call have arrived in hostStats
    bool statsPending = false;
    PinBuf hostStats;
    hipEvent_t done = nullptr; // recorded on the context's stream when a batch call has enqueued all its work
    ReadFence readFence;       // a gather on another stream still reads offsets / pairs up to this point
    // match lists sized as a call goes (option match_list_observe): the running total after every chunk, written
    // by the device into page-locked memory, an event per chunk in flight, and list buffers that a call outgrew
    // (freed when the next call starts or the result goes)
    PinBuf hostTotals;
    hipEvent_t listEv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> retired;
    // single-pair calls: the flag the last call's pair ended with (see pairOvf), for which options, and calls so far
    int pairFlag = 0;
    MatchKey pairKey = {nullptr, 0, 0.0f, 0.0f, 0.0f, 0, 0};
    unsigned pairCalls = 0;
};
int sfm_match_result_mark_done(sfm_ctx *ctx, sfm_match_result *res);

