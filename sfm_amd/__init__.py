"""sfm_amd: MI355X (gfx950) implementation of the SfM front end's two data-parallel stages,
pairwise descriptor matching and RANSAC two-view geometry, behind the reference's
MATCHING_OPTIONS / RANSAC_OPTIONS entry points.

The compute lives in sfm_amd/lib/libsfm_amd.so (hand-written HIP kernels behind the C-ABI of
include/sfm_amd.h); this package is the Python host-side mirror used by tests and benchmarks.
"""
from . import capi  # noqa: F401
from .api import MATCHING_OPTIONS, RANSAC_OPTIONS, Keypoint, match, RANSAC, cosineSimilarity, dotProduct  # noqa: F401

__all__ = ["capi", "MATCHING_OPTIONS", "RANSAC_OPTIONS", "Keypoint", "match", "RANSAC", "cosineSimilarity",
           "dotProduct"]
