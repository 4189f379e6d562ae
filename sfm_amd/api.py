"""Host-side mirror of the reference's operator interface for the two hot-path stages.

Names and argument meaning follow the reference (namespace SfM::match / SfM::solve):
  match(keypoints1, keypoints2, options)            submodules/match/match.hpp:70
  MATCHING_OPTIONS                                  submodules/match/match.hpp:55-61
  RANSAC(x, y, K, options, verbose)                 submodules/solve/solve.hpp:137
  RANSAC_OPTIONS                                    submodules/solve/solve.hpp:108-125
The two `std::function` plug-in points are mirrored by sentinel objects: only the
reference's own functors have kernels; anything else raises (there is no CPU fallback in this
package; the C++ adaptors under adaptors/ keep the reference's CPU loop for user functors).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import capi


class _Functor:
    def __init__(self, name: str, code: int):
        self.name, self.code = name, code

    def __repr__(self):
        return f"<{self.name}>"


# match.hpp:18-46
cosineSimilarity = _Functor("cosineSimilarity", capi.METRIC_COSINE)
dotProduct = _Functor("dotProduct", capi.METRIC_DOT)
# solve.hpp:78 and :119-124
eightPointAlgorithmFromSubset = _Functor("eightPointAlgorithmFromSubset", 0)
defaultLoss = _Functor("max(reprojectionError(cam1), reprojectionError(cam2))", capi.LOSS_REPROJECTION)
sampsonLoss = _Functor("sampsonDistance", capi.LOSS_SAMPSON)

TWO_SIDED_MATHICHNG = capi.MATCH_TWO_SIDED  # (sic) match.hpp:12


@dataclass
class Keypoint:
    """SfM.hpp:125-140 (the fields the matcher reads)."""
    descriptor: Any
    point: Any
    size: float = 0.0
    angle: float = 0.0
    response: float = 0.0
    octave: int = 0
    trackId: int = -1


@dataclass
class MATCHING_OPTIONS:
    """match.hpp:55-61, same defaults."""
    algorithm: int = TWO_SIDED_MATHICHNG
    similarityMetric: Any = cosineSimilarity
    threshold: float = 0.5
    maxDistancePxSquared: float = 10000.0 * 10000.0


@dataclass
class RANSAC_OPTIONS:
    """solve.hpp:108-125, same defaults."""
    minN: int = 8
    maxIter: int = 512
    maxTimeMs: int = 1000
    maxSquaredError: float = 10.0
    successProb: float = 0.99
    model: Any = eightPointAlgorithmFromSubset
    loss: Any = defaultLoss


_ctx: Optional[capi.Context] = None


def default_context() -> capi.Context:
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def _pack(keypoints: Sequence[Keypoint]):
    """Keypoint AoS -> contiguous descriptor and point tables (SfM.hpp:132-133)."""
    n = len(keypoints)
    if n == 0:
        return np.zeros((0, 128), np.float32), np.zeros((0, 2), np.float64)
    desc = np.stack([np.asarray(k.descriptor, np.float32) for k in keypoints])
    pts = np.stack([np.asarray(k.point, np.float64) for k in keypoints])
    return desc, pts


def match(keypoints1: Sequence[Keypoint], keypoints2: Sequence[Keypoint],
          options: MATCHING_OPTIONS = MATCHING_OPTIONS(), ctx: Optional[capi.Context] = None) -> List[Tuple[int, int]]:
    """SfM::match::match (match.cpp:9-22): list of (index in first, index in second)."""
    if options.algorithm != TWO_SIDED_MATHICHNG:
        raise ValueError(f"Match: Unknown matching algorithm. Was {options.algorithm}")  # match.cpp:18
    if not isinstance(options.similarityMetric, _Functor) or options.similarityMetric not in (cosineSimilarity, dotProduct):
        raise NotImplementedError("similarityMetric must be cosineSimilarity or dotProduct: a user functor has no kernel")
    ctx = ctx or default_context()
    dA, pA = _pack(keypoints1)
    dB, pB = _pack(keypoints2)
    if len(dA) == 0 and len(dB) != 0:
        dA = np.zeros((0, dB.shape[1]), np.float32)
    if len(dB) == 0 and len(dA) != 0:
        dB = np.zeros((0, dA.shape[1]), np.float32)
    pairs = ctx.match_pair(dA, pA, dB, pB, threshold=options.threshold, max_dist_sq=options.maxDistancePxSquared,
                           metric=options.similarityMetric.code, algorithm=options.algorithm)
    return [(int(i), int(j)) for i, j in pairs]


def safeCastToInt(value: float) -> int:
    """RANSAC.cpp:14-32."""
    if math.isnan(value):
        return 0
    if value >= 2147483647.0:
        return 2147483647
    if value <= -2147483648.0:
        return -2147483648
    return int(value)


def RANSAC(x, y, K, options: RANSAC_OPTIONS = RANSAC_OPTIONS(), verbose: bool = False, samples=None,
           seed: int = 0, ctx: Optional[capi.Context] = None) -> List[int]:
    """SfM::solve::RANSAC (RANSAC.cpp:146-270): indices of the largest inlier set.

    One library call: the rows of the sample table are fitted and scored in growing slices, the adaptive
    iteration cap (RANSAC.cpp:245-258) and the wall-clock cap (:189-196) are applied between slices on the
    device, and the winner's inlier list comes back.
    samples: optional [maxIter, minN] index table; by default the library draws one on the device from `seed`
    (0: from the clock; the reference seeds from the wall clock, RANSAC.cpp:170, so any table is a legal run)."""
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
    if len(x) != len(y) or len(x) < options.minN:  # RANSAC.cpp:148-149
        return []
    if options.model is not eightPointAlgorithmFromSubset:
        raise NotImplementedError("model must be eightPointAlgorithmFromSubset: a user functor has no kernel")
    if options.loss not in (defaultLoss, sampsonLoss):
        raise NotImplementedError("loss must be the default reprojection loss (or sampsonLoss)")
    if options.minN != 8:
        raise NotImplementedError("minN must be 8")
    ctx = ctx or default_context()
    start = time.monotonic()
    if samples is not None:
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 8)[:max(options.maxIter, 0)]
    ctx.set_option("ransac_seed", int(seed))
    ctx.set_option("ransac_max_time_ms", max(int(options.maxTimeMs), 0))
    try:
        r = ctx.ransac_pair(x, y, K, samples if samples is not None else max(int(options.maxIter), 0),
                            max_sq_err=options.maxSquaredError, success_prob=options.successProb, adaptive=True,
                            loss=options.loss.code, per_row=False)
    finally:
        ctx.set_option("ransac_seed", 0)
        ctx.set_option("ransac_max_time_ms", 0)
    if verbose:
        print(f"RANSAC: finished after {r['iterations']} iterations ({int((time.monotonic() - start) * 1000)}ms).")
    return [int(i) for i in r["inliers"]]
