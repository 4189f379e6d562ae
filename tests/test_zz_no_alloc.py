"""sfm_ransac_pair_static, the RANSAC entry point that allocates nothing (its device work space is a fixed
array inside the library), against sfm_ransac_pair: identical outputs, bit for bit, on two calls in a row,
the second smaller than the first (the shared array keeps the first call's contents), and no change in the
device's free memory across the calls."""
import ctypes

import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu


def free_device_bytes():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def same(a, b):
    assert a["best_h"] == b["best_h"] and a["iterations"] == b["iterations"]
    assert np.array_equal(a["inliers"], b["inliers"])
    assert np.array_equal(a["pose"].view(np.uint64), b["pose"].view(np.uint64))
    if a["counts"] is not None:
        assert np.array_equal(a["counts"], b["counts"])
        assert np.array_equal(a["toterr"].view(np.uint64), b["toterr"].view(np.uint64))


@pytest.mark.parametrize("per_row", [True, False])
def test_static_workspace_equals_allocating_entry_point(ctx, per_row):
    cases = [(12000, 4096, 0.3, True), (700, 512, 0.6, True), (16384, 8192, 0.15, False), (8, 8, 1.0, False),
             (3000, 1000, 0.8, True)]
    inputs = []
    for k, (n, H, ratio, adaptive) in enumerate(cases):
        tv = synth.two_view(n, inlier_ratio=ratio, seed=900 + k)
        inputs.append((tv, synth.sample_table(n, H, seed=950 + k), adaptive))
    expected = [ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=ad,
                                per_row=per_row) for tv, S, ad in inputs]
    ctx.synchronize()
    before = free_device_bytes()
    for (tv, S, ad), exp in zip(inputs, expected):  # consecutive calls of different sizes share the array
        got = ctx.ransac_pair(tv["x"], tv["y"], tv["K"], S, max_sq_err=1.0, success_prob=0.999, adaptive=ad,
                              per_row=per_row, static_workspace=True)
        same(got, exp)
        assert len(exp["inliers"]) >= 1
    assert free_device_bytes() == before, "the static entry point changed the device's free memory"


def test_static_workspace_other_paths(ctx):
    """The instances without the pre-test (Sampson loss; intrinsics with skew) use the fp64 record slice."""
    tv = synth.two_view(5000, inlier_ratio=0.4, seed=77)
    S = synth.sample_table(5000, 1024, seed=78)
    K2 = tv["K"].copy()
    K2[0, 1] = 0.5
    for kw in (dict(loss=capi.LOSS_SAMPSON), dict()):
        K = tv["K"] if kw else K2
        exp = ctx.ransac_pair(tv["x"], tv["y"], K, S, max_sq_err=1.0, adaptive=False, **kw)
        got = ctx.ransac_pair(tv["x"], tv["y"], K, S, max_sq_err=1.0, adaptive=False, static_workspace=True, **kw)
        same(got, exp)


def test_static_workspace_limits(ctx):
    tv = synth.two_view(16385, inlier_ratio=0.5, seed=1)
    with pytest.raises(capi.SfmError) as e:
        ctx.ransac_pair(tv["x"], tv["y"], tv["K"], synth.sample_table(16385, 16, seed=2), static_workspace=True)
    assert e.value.code == capi.ERR_CAPACITY and "16384" in str(e.value)
    with pytest.raises(capi.SfmError) as e:
        ctx.ransac_pair(tv["x"][:100], tv["y"][:100], tv["K"], synth.sample_table(100, 8193, seed=2), static_workspace=True)
    assert e.value.code == capi.ERR_CAPACITY
    # fewer than minN correspondences: empty result, as the allocating entry point (RANSAC.cpp:148-149)
    r = ctx.ransac_pair(tv["x"][:7], tv["y"][:7], tv["K"], np.zeros((4, 8), np.int32), static_workspace=True)
    assert len(r["inliers"]) == 0 and r["best_h"] == -1
    # and the call after a refused one works
    S = synth.sample_table(100, 64, seed=3)
    same(ctx.ransac_pair(tv["x"][:100], tv["y"][:100], tv["K"], S, static_workspace=True),
         ctx.ransac_pair(tv["x"][:100], tv["y"][:100], tv["K"], S))
