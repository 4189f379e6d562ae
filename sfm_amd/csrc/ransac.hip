// placeholder until the RANSAC kernels land
#include "common.h"
struct sfm_ransac_batch { sfm_ctx *ctx; };
extern "C" int sfm_ransac_pair(sfm_ctx *ctx, const double *, const double *, int, const double *, const int32_t *, int, int,
                    double, double, int, int, int32_t *, int, int *, int *, double *, int32_t *, double *, int *) { SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "not built yet"); }
extern "C" int sfm_eight_point_fit(sfm_ctx *ctx, const double *, const double *, int, const int32_t *, int, double *) { SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "not built yet"); }
extern "C" int sfm_ransac_batch_run(sfm_ctx *ctx, int, const int64_t *, const double *, const double *, const double *, const int32_t *, int, double, double, int, int, sfm_ransac_batch **) { SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "not built yet"); }
extern "C" int sfm_ransac_batch_run_device(sfm_ctx *ctx, int, const int64_t *, const double *, const double *, const double *, const int32_t *, int, double, double, int, int, sfm_ransac_batch **) { SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "not built yet"); }
extern "C" int sfm_ransac_batch_best(sfm_ransac_batch *, int32_t *, int32_t *, double *, double *, int32_t *) { return SFM_ERR_UNSUPPORTED; }
extern "C" int sfm_ransac_batch_inlier_offsets(sfm_ransac_batch *, int64_t *) { return SFM_ERR_UNSUPPORTED; }
extern "C" int sfm_ransac_batch_inliers(sfm_ransac_batch *, int32_t *) { return SFM_ERR_UNSUPPORTED; }
extern "C" void sfm_ransac_batch_destroy(sfm_ransac_batch *) {}
