"""Track ids for the correspondences of a match result.

Restates the rule of Scene::pushBackImageWithKeypoints (submodules/scene.cpp:80-120, splitTracks off):
walking a pair's matches in order, a feature of the older frame that has no track yet opens one and
its partner joins it; otherwise the partner inherits the older feature's track.  Host logic (one
gather/scatter per pair); the ids feed sfm_scale_chain_refit.
"""
import numpy as np


def assign_track_ids(pair_list, offsets, pairs, counts):
    """pair_list [P,2] image indices, offsets [P+1] / pairs [total,2] of a match result, counts [n_images]
    features per image.  Returns (ids [total] int32, one per match; n_tracks).  Pairs are taken in order."""
    track_of = [np.full(int(c), -1, np.int64) for c in counts]
    ids = np.zeros(len(pairs), np.int32)
    n_tracks = 0
    for k, (a, b) in enumerate(np.asarray(pair_list)):
        m = pairs[offsets[k]:offsets[k + 1]]
        t = track_of[a][m[:, 0]]
        new = t < 0
        t[new] = n_tracks + np.arange(int(new.sum()))
        n_tracks += int(new.sum())
        track_of[a][m[:, 0]] = t
        track_of[b][m[:, 1]] = t
        ids[offsets[k]:offsets[k + 1]] = t
    return ids, n_tracks
