"""ctypes binding of sfm_amd/lib/libsfm_amd.so (the C-ABI declared in include/sfm_amd.h).

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFM_AMD_LIB") or os.path.join(_HERE, "lib", "libsfm_amd.so")  # (the variable: A/B runs of other builds)

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_OOM, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_HIP, ERR_TIMEOUT = range(8)
METRIC_COSINE, METRIC_DOT, METRIC_L2 = 0, 1, 2
MATCH_TWO_SIDED = 0
LOSS_REPROJECTION, LOSS_SAMPSON = 0, 1

_STATUS_NAMES = {1: "invalid argument", 2: "no gfx950 device", 3: "out of memory", 4: "unsupported",
                 5: "output capacity too small", 6: "HIP runtime error", 7: "multi-GPU exchange timed out"}

# Every symbol include/sfm_amd.h declares.
EXPORTED_SYMBOLS = [
    "sfm_ctx_create", "sfm_ctx_destroy", "sfm_ctx_synchronize", "sfm_last_error", "sfm_version",
    "sfm_ctx_set_option", "sfm_ctx_last_kernel_ms", "sfm_host_alloc", "sfm_host_free",
    "sfm_match_pair", "sfm_match_pair_u8", "sfm_imageset_create", "sfm_imageset_create_u8", "sfm_imageset_create_device", "sfm_imageset_destroy",
    "sfm_imageset_device_bytes", "sfm_match_batch", "sfm_match_result_total", "sfm_match_result_offsets",
    "sfm_match_result_pairs", "sfm_match_result_device", "sfm_match_result_stats", "sfm_match_result_stats_ex",
    "sfm_match_result_device_bytes",
    "sfm_match_result_destroy",
    "sfm_ransac_pair", "sfm_ransac_pair_static", "sfm_eight_point_fit", "sfm_eight_point_from_subset", "sfm_eight_point_algorithm", "sfm_ransac_batch_run", "sfm_ransac_batch_run_device",
    "sfm_ransac_batch_best", "sfm_ransac_batch_inlier_offsets", "sfm_ransac_batch_inliers",
    "sfm_ransac_batch_destroy", "sfm_ransac_batch_refit", "sfm_ransac_batch_refit_fetch", "sfm_ransac_batch_handed_over",
    "sfm_corr_from_matches", "sfm_corr_offsets", "sfm_corr_device", "sfm_corr_fetch", "sfm_corr_sample_table",
    "sfm_ransac_batch_run_corr", "sfm_corr_destroy", "sfm_scale_chain", "sfm_scale_chain_refit",
    "sfm_scale_chain_colors", "sfm_pixel_bilinear_u8",
    "sfm_match_batch_ratio", "sfm_match_batch_l2", "sfm_binset_create", "sfm_binset_destroy", "sfm_match_batch_hamming",
    "sfm_shard_count", "sfm_shard_indices", "sfm_shard_layout", "sfm_comm_unique_id", "sfm_comm_create",
    "sfm_comm_create_local", "sfm_comm_rank", "sfm_comm_destroy", "sfm_match_gather", "sfm_match_gather_local",
    "sfm_match_batch_multi", "sfm_gathered_wait", "sfm_gathered_total", "sfm_gathered_layout", "sfm_gathered_pairs", "sfm_gathered_device",
    "sfm_gathered_destroy", "sfm_gathered_wait_timeout", "sfm_ransac_gather", "sfm_ransac_gather_local", "sfm_gathered_ransac", "sfm_shard_unpack_ransac",
]


class SfmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sfm_amd error {code} ({_STATUS_NAMES.get(code, '?')}): {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C sfm_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
    P = C.POINTER
    L.sfm_version.restype = C.c_char_p
    L.sfm_last_error.restype = C.c_char_p
    L.sfm_last_error.argtypes = [vp]
    L.sfm_ctx_create.argtypes = [i32, P(vp)]
    L.sfm_ctx_destroy.argtypes = [vp]
    L.sfm_ctx_destroy.restype = None
    L.sfm_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.sfm_host_free.argtypes = [vp]
    L.sfm_host_free.restype = None
    L.sfm_ctx_synchronize.argtypes = [vp]
    L.sfm_ctx_set_option.argtypes = [vp, C.c_char_p, i64]
    L.sfm_ctx_last_kernel_ms.argtypes = [vp, i32, P(f64), P(i32)]
    L.sfm_match_pair.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, f32, f32, vp, i32, P(i32), vp, vp]
    L.sfm_match_pair_u8.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, f32, f32, vp, i32, P(i32), vp, vp]
    L.sfm_imageset_create.argtypes = [vp, i32, vp, vp, vp, i32, P(vp)]
    L.sfm_imageset_create_device.argtypes = [vp, i32, vp, vp, vp, i32, P(vp)]
    L.sfm_imageset_create_u8.argtypes = [vp, i32, vp, vp, vp, i32, P(vp)]
    L.sfm_imageset_destroy.argtypes = [vp]
    L.sfm_imageset_destroy.restype = None
    L.sfm_imageset_device_bytes.argtypes = [vp]
    L.sfm_imageset_device_bytes.restype = i64
    L.sfm_match_batch.argtypes = [vp, vp, vp, i32, i32, f32, f32, P(vp)]
    L.sfm_match_batch_ratio.argtypes = [vp, vp, vp, i32, f32, f32, f32, P(vp)]
    if hasattr(L, "sfm_match_batch_l2"):
        L.sfm_match_batch_l2.argtypes = [vp, vp, vp, i32, f32, f32, f32, P(vp)]
    L.sfm_binset_create.argtypes = [vp, i32, vp, vp, vp, i32, P(vp)]
    L.sfm_binset_destroy.argtypes = [vp]
    L.sfm_binset_destroy.restype = None
    L.sfm_match_batch_hamming.argtypes = [vp, vp, vp, i32, i32, f32, P(vp)]
    L.sfm_match_result_total.argtypes = [vp, P(i64)]
    L.sfm_match_result_offsets.argtypes = [vp, vp]
    L.sfm_match_result_pairs.argtypes = [vp, vp]
    L.sfm_match_result_device.argtypes = [vp, P(vp), P(vp)]
    L.sfm_match_result_stats.argtypes = [vp, P(i64), P(i64)]
    L.sfm_match_result_stats_ex.argtypes = [vp, vp]
    if hasattr(L, "sfm_match_result_device_bytes"):
        L.sfm_match_result_device_bytes.argtypes = [vp]
        L.sfm_match_result_device_bytes.restype = i64
    L.sfm_match_result_destroy.argtypes = [vp]
    L.sfm_match_result_destroy.restype = None
    L.sfm_ransac_pair.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, f64, f64, i32, i32, vp, i32, P(i32),
                                  P(i32), vp, vp, vp, P(i32)]
    if hasattr(L, "sfm_ransac_pair_static"):  # (absent only from older builds used in A/B timing runs)
        L.sfm_ransac_pair_static.argtypes = L.sfm_ransac_pair.argtypes
    L.sfm_eight_point_fit.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    L.sfm_eight_point_algorithm.argtypes = [vp, vp, vp, i32, vp, vp]
    if hasattr(L, "sfm_eight_point_from_subset"):
        L.sfm_eight_point_from_subset.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp]
    L.sfm_ransac_batch_run.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, f64, f64, i32, i32, P(vp)]
    L.sfm_ransac_batch_run_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, f64, f64, i32, i32, P(vp)]
    L.sfm_ransac_batch_best.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sfm_ransac_batch_inlier_offsets.argtypes = [vp, vp]
    L.sfm_ransac_batch_inliers.argtypes = [vp, vp]
    L.sfm_ransac_batch_destroy.argtypes = [vp]
    L.sfm_ransac_batch_destroy.restype = None
    L.sfm_ransac_batch_refit.argtypes = [vp]
    L.sfm_ransac_batch_refit_fetch.argtypes = [vp, vp, vp, vp]
    L.sfm_ransac_batch_handed_over.argtypes = [vp, vp, vp, vp]
    L.sfm_corr_from_matches.argtypes = [vp, vp, vp, vp, P(vp)]
    L.sfm_corr_offsets.argtypes = [vp, vp]
    L.sfm_corr_device.argtypes = [vp, P(vp), P(vp)]
    L.sfm_corr_fetch.argtypes = [vp, vp, vp]
    L.sfm_corr_sample_table.argtypes = [vp, i32, C.c_uint64, P(vp), vp]
    L.sfm_ransac_batch_run_corr.argtypes = [vp, vp, vp, i32, C.c_uint64, f64, f64, i32, i32, P(vp)]
    L.sfm_corr_destroy.argtypes = [vp]
    L.sfm_scale_chain.argtypes = [vp, i32, vp, vp, vp, vp, vp, f64, i32, vp, vp, vp, vp]
    L.sfm_scale_chain_refit.argtypes = [vp, vp, vp, f64, i32, vp, vp, vp, vp]
    L.sfm_scale_chain_colors.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, f64, i32, vp, vp, vp, vp, vp]
    L.sfm_pixel_bilinear_u8.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    L.sfm_corr_destroy.restype = None
    if hasattr(L, "sfm_shard_count"):  # (absent only from older builds used in A/B timing runs)
        L.sfm_shard_count.argtypes = [i64, i32, i32]
        L.sfm_shard_count.restype = i64
        L.sfm_shard_indices.argtypes = [i64, i32, i32, vp]
        L.sfm_shard_layout.argtypes = [i64, i32, vp, i64, vp, vp]
        L.sfm_comm_unique_id.argtypes = [vp]
        L.sfm_comm_create.argtypes = [vp, i32, i32, vp, P(vp)]
        L.sfm_comm_create_local.argtypes = [i32, P(vp), P(vp)]
        L.sfm_comm_rank.argtypes = [vp, P(i32), P(i32)]
        L.sfm_comm_destroy.argtypes = [vp]
        L.sfm_comm_destroy.restype = None
        L.sfm_match_gather.argtypes = [vp, vp, i64, i32, P(vp)]
        L.sfm_match_gather_local.argtypes = [i32, P(vp), P(vp), i64, i32, P(vp)]
        L.sfm_match_batch_multi.argtypes = [i32, P(vp), P(vp), vp, i64, i32, f32, f32, i32, P(vp), P(vp)]
        L.sfm_gathered_wait.argtypes = [vp]
        L.sfm_gathered_total.argtypes = [vp, P(i64), vp]
        L.sfm_gathered_layout.argtypes = [vp, vp, vp]
        L.sfm_gathered_pairs.argtypes = [vp, vp]
        L.sfm_gathered_device.argtypes = [vp, P(vp), P(vp)]
        L.sfm_gathered_destroy.argtypes = [vp]
        L.sfm_gathered_destroy.restype = None
        L.sfm_gathered_wait_timeout.argtypes = [vp, i64]
        L.sfm_ransac_gather.argtypes = [vp, vp, i64, i32, P(vp)]
        L.sfm_ransac_gather_local.argtypes = [i32, P(vp), P(vp), i64, i32, P(vp)]
        L.sfm_gathered_ransac.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.sfm_shard_unpack_ransac.argtypes = [i64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


# ---- several GPUs: the library's static partition and RCCL gather ------------------------------------
def shard_count(n_pairs: int, rank: int, world: int) -> int:
    return int(lib().sfm_shard_count(int(n_pairs), int(rank), int(world)))


def shard_indices(n_pairs: int, rank: int, world: int) -> np.ndarray:
    """Ascending indices of the pairs rank owns (blocks of 64 consecutive pairs dealt round-robin)."""
    out = np.zeros(max(shard_count(n_pairs, rank, world), 1), np.int64)
    rc = lib().sfm_shard_indices(int(n_pairs), int(rank), int(world), _ptr(out))
    if rc != OK:
        raise SfmError(rc, "sfm_shard_indices")
    return out[:shard_count(n_pairs, rank, world)]


def shard_layout(n_pairs: int, world: int, counts: np.ndarray):
    """counts[world, max_owned] -> (starts int64 [n_pairs], counts int32 [n_pairs]) of the rank-major buffer."""
    counts = np.ascontiguousarray(counts, np.int32).reshape(world, -1)
    starts = np.zeros(max(n_pairs, 1), np.int64)
    cnt = np.zeros(max(n_pairs, 1), np.int32)
    rc = lib().sfm_shard_layout(int(n_pairs), int(world), _ptr(counts), counts.shape[1], _ptr(starts), _ptr(cnt))
    if rc != OK:
        raise SfmError(rc, "sfm_shard_layout")
    return starts[:n_pairs], cnt[:n_pairs]


RANSAC_RECORD = np.dtype([("best_h", np.int32), ("count", np.int32), ("iterations", np.int32), ("pad", np.int32),
                          ("toterr", np.float64), ("pose12", np.float64, (12,))])  # 120 bytes, as the gather moves them


def shard_unpack_ransac(n_pairs: int, world: int, counts, records, lists):
    """sfm_shard_unpack_ransac (host only): rank-major gather buffers -> results in the order of the whole list."""
    counts = np.ascontiguousarray(counts, np.int32).reshape(world, -1)
    records = np.ascontiguousarray(records, RANSAC_RECORD).reshape(world, -1)
    lists = np.ascontiguousarray(lists, np.int32)
    n = int(n_pairs)
    best = np.zeros(max(n, 1), np.int32)
    cnt = np.zeros(max(n, 1), np.int32)
    err = np.zeros(max(n, 1), np.float64)
    it = np.zeros(max(n, 1), np.int32)
    pose = np.zeros((max(n, 1), 12), np.float64)
    off = np.zeros(n + 1, np.int64)
    inl = np.zeros(max(len(lists), 1), np.int32)
    rc = lib().sfm_shard_unpack_ransac(n, int(world), _ptr(counts), counts.shape[1], _ptr(records), _ptr(lists) if len(lists) else None,
                                       _ptr(best), _ptr(cnt), _ptr(err), _ptr(it), _ptr(pose), _ptr(off), _ptr(inl))
    if rc != OK:
        raise SfmError(rc, "sfm_shard_unpack_ransac")
    return dict(best_h=best[:n], count=cnt[:n], toterr=err[:n], iterations=it[:n], pose12=pose[:n], inlier_offsets=off,
                inliers=inl[:int(off[n])])


class Comm:
    """One rank of an RCCL communicator (sfm_comm_*)."""

    def __init__(self, ctx: "Context", handle):
        _live.add(self)
        self.ctx = ctx
        self._h = handle
        r, w = C.c_int(0), C.c_int(0)
        ctx.check(lib().sfm_comm_rank(self._h, C.byref(r), C.byref(w)))
        self.rank, self.world = r.value, w.value

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib().sfm_comm_unique_id(buf)
        if rc != OK:
            raise SfmError(rc, "sfm_comm_unique_id")
        return buf.raw

    @classmethod
    def create(cls, ctx: "Context", rank: int, world: int, id128: bytes) -> "Comm":
        """One process per GPU; collective over all ranks."""
        h = C.c_void_p()
        ctx.check(lib().sfm_comm_create(ctx._h, int(rank), int(world), C.create_string_buffer(id128, 128), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def local(cls, ctxs) -> list:
        """One process that owns several devices: ctxs[i] becomes rank i."""
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c._h for c in ctxs])
        out = (C.c_void_p * n)()
        ctxs[0].check(lib().sfm_comm_create_local(n, hs, out))
        return [cls(ctxs[i], C.c_void_p(out[i])) for i in range(n)]

    def gather(self, result: "MatchResult", n_pairs_total: int, root: int = 0, gathered: "Gathered" = None) -> "Gathered":
        g = gathered if gathered is not None else Gathered(self)
        self.ctx.check(lib().sfm_match_gather(self._h, result._h, int(n_pairs_total), int(root), C.byref(g._h)))
        g.n_total, g.root = int(n_pairs_total), int(root)
        return g

    def gather_ransac(self, batch: "RansacBatch", n_pairs_total: int, root: int = 0, gathered: "Gathered" = None) -> "Gathered":
        """sfm_ransac_gather: the winners and inlier lists of this rank's share of the pair list to the root."""
        g = gathered if gathered is not None else Gathered(self)
        self.ctx.check(lib().sfm_ransac_gather(self._h, batch._h, int(n_pairs_total), int(root), C.byref(g._h)))
        g.n_total, g.root = int(n_pairs_total), int(root)
        return g

    def close(self):
        if self._h:
            lib().sfm_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gathered:
    """Match lists of a sharded pair list on the root rank (sfm_gathered_*): rank-major payload."""

    def __init__(self, comm: Comm):
        _live.add(self)
        self.comm = comm
        self.ctx = comm.ctx
        self._h = C.c_void_p()
        self.n_total = 0
        self.root = 0

    def wait(self):
        """Blocks until the lists are in place (root) and the local result may be overwritten."""
        if self._h:
            self.ctx.check(lib().sfm_gathered_wait(self._h))

    def wait_timeout(self, timeout_ms: int) -> bool:
        """True when the gather has finished, False when timeout_ms passed first (it stays in flight)."""
        if not self._h:
            return True
        rc = lib().sfm_gathered_wait_timeout(self._h, int(timeout_ms))
        if rc == ERR_TIMEOUT:
            return False
        self.ctx.check(rc)
        return True

    def ransac(self):
        """Gathered RANSAC results in the order of the whole pair list (root only): dict(best_h, count, toterr,
        iterations, pose12 [n,12], inlier_offsets [n+1], inliers)."""
        n = self.n_total
        tot = self.total()
        best = np.zeros(max(n, 1), np.int32)
        cnt = np.zeros(max(n, 1), np.int32)
        err = np.zeros(max(n, 1), np.float64)
        it = np.zeros(max(n, 1), np.int32)
        pose = np.zeros((max(n, 1), 12), np.float64)
        off = np.zeros(n + 1, np.int64)
        inl = np.zeros(max(tot, 1), np.int32)
        self.ctx.check(lib().sfm_gathered_ransac(self._h, _ptr(best), _ptr(cnt), _ptr(err), _ptr(it), _ptr(pose), _ptr(off),
                                                 _ptr(inl)))
        return dict(best_h=best[:n], count=cnt[:n], toterr=err[:n], iterations=it[:n], pose12=pose[:n],
                    inlier_offsets=off, inliers=inl[:tot])

    def total(self) -> int:
        t = C.c_int64(0)
        self.ctx.check(lib().sfm_gathered_total(self._h, C.byref(t), None))
        return t.value

    def per_rank_totals(self) -> np.ndarray:
        t = C.c_int64(0)
        out = np.zeros(self.comm.world, np.int64)
        self.ctx.check(lib().sfm_gathered_total(self._h, C.byref(t), _ptr(out)))
        return out

    def layout(self):
        """(starts int64 [n_total], counts int32 [n_total]): pair p owns pairs()[starts[p] : starts[p] + counts[p]]."""
        starts = np.zeros(max(self.n_total, 1), np.int64)
        counts = np.zeros(max(self.n_total, 1), np.int32)
        self.ctx.check(lib().sfm_gathered_layout(self._h, _ptr(starts), _ptr(counts)))
        return starts[:self.n_total], counts[:self.n_total]

    def pairs(self) -> np.ndarray:
        n = self.total()
        out = np.zeros((max(n, 1), 2), np.int32)
        self.ctx.check(lib().sfm_gathered_pairs(self._h, _ptr(out)))
        return out[:n]

    def close(self):
        if self._h:
            lib().sfm_gathered_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_batch_multi(comms, sets, pair_list, threshold=0.5, max_dist_sq=1e8, metric=METRIC_COSINE, root=0,
                      results=None, gathered=None):
    """sfm_match_batch_multi: one process, len(comms) devices.  Returns (results, gathered) lists (reusable)."""
    n = len(comms)
    pl = np.ascontiguousarray(pair_list, np.int32).reshape(-1, 2)
    results = results if results is not None else [MatchResult(c.ctx) for c in comms]
    gathered = gathered if gathered is not None else [Gathered(c) for c in comms]
    ch = (C.c_void_p * n)(*[c._h for c in comms])
    sh = (C.c_void_p * n)(*[s._h for s in sets])
    rh = (C.c_void_p * n)(*[r._h for r in results])
    gh = (C.c_void_p * n)(*[g._h for g in gathered])
    comms[0].ctx.check(lib().sfm_match_batch_multi(n, ch, sh, _ptr(pl), len(pl), metric, threshold, max_dist_sq, root, rh, gh))
    for i in range(n):
        results[i]._h = C.c_void_p(rh[i])
        results[i].n_pairs = shard_count(len(pl), i, n)
        gathered[i]._h = C.c_void_p(gh[i])
        gathered[i].n_total, gathered[i].root = len(pl), root
    return results, gathered


def ransac_gather_local(comms, batches, n_pairs_total, root=0, gathered=None):
    """sfm_ransac_gather_local: one process, len(comms) devices; batches[i] ran rank i's share."""
    n = len(comms)
    gathered = gathered if gathered is not None else [Gathered(c) for c in comms]
    ch = (C.c_void_p * n)(*[c._h for c in comms])
    bh = (C.c_void_p * n)(*[b._h for b in batches])
    gh = (C.c_void_p * n)(*[g._h for g in gathered])
    comms[0].ctx.check(lib().sfm_ransac_gather_local(n, ch, bh, int(n_pairs_total), int(root), gh))
    for i in range(n):
        gathered[i]._h = C.c_void_p(gh[i])
        gathered[i].n_total, gathered[i].root = int(n_pairs_total), int(root)
    return gathered


def _chain_outputs(S, n_tracks):
    return (np.zeros((S + 1, 16)), np.zeros((max(n_tracks, 1), 3)), np.zeros(max(n_tracks, 1), np.int32),
            np.zeros(max(S, 1)))


def _chain_result(out, S, n_tracks):
    return dict(extrinsics=out[0].reshape(S + 1, 4, 4).transpose(0, 2, 1).copy(), points3d=out[1][:n_tracks],
                counts=out[2][:n_tracks], scale=out[3][:S])


class _PinnedBlock:
    """Page-locked host bytes from sfm_host_alloc, exposed through the array interface; numpy arrays made from
    it keep it alive, and it returns the memory when the last of them has gone."""

    def __init__(self, addr, nbytes):
        self.addr = addr
        self.__array_interface__ = dict(shape=(max(nbytes, 1),), typestr="|u1", data=(addr, False), version=3)

    def __del__(self):
        try:
            lib().sfm_host_free(self.addr)
        except Exception:
            pass


def _ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


# Every live wrapper, so that interpreter exit releases device objects before their contexts and
# both before the HIP runtime unloads (a handle freed after that aborts the process).
_live = weakref.WeakSet()


def _close_all():
    objs = list(_live)
    for want_ctx in (False, True):
        for o in objs:
            if isinstance(o, Context) == want_ctx:
                try:
                    o.close()
                except Exception:
                    pass


atexit.register(_close_all)


class Context:
    """One device context (HIP stream + workspace)."""

    def __init__(self, device: int = 0):
        _live.add(self)
        self._h = C.c_void_p()
        rc = lib().sfm_ctx_create(int(device), C.byref(self._h))
        if rc != OK:
            raise SfmError(rc, f"sfm_ctx_create(device={device})")
        self.device = device

    def close(self):
        if self._h:
            lib().sfm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def check(self, rc: int):
        if rc != OK:
            raise SfmError(rc, lib().sfm_last_error(self._h).decode())

    def synchronize(self):
        self.check(lib().sfm_ctx_synchronize(self._h))

    def set_option(self, key: str, value: int):
        self.check(lib().sfm_ctx_set_option(self._h, key.encode(), int(value)))

    def last_kernel_ms(self, which: int = 0):
        ms, n = C.c_double(0), C.c_int(0)
        self.check(lib().sfm_ctx_last_kernel_ms(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- matching -------------------------------------------------------------------------
    def match_pair(self, descA, ptsA, descB, ptsB, threshold=0.5, max_dist_sq=1e8, metric=METRIC_COSINE,
                   algorithm=MATCH_TWO_SIDED, return_best=False):
        """SfM::match::match for one pair.  Tables of dtype uint8 (both) go through sfm_match_pair_u8.
        With the option "match_pair_reuse" the library recognises a table by its address: pass arrays that are
        already contiguous and of the call's dtype (float32 or uint8 descriptors, float64 points), so that no
        temporary copy stands in for them, and do not change them between calls."""
        descA, descB = np.asarray(descA), np.asarray(descB)
        as_bytes = descA.dtype == np.uint8 and descB.dtype == np.uint8
        descA = np.ascontiguousarray(descA, np.uint8 if as_bytes else np.float32)
        descB = np.ascontiguousarray(descB, np.uint8 if as_bytes else np.float32)
        ptsA = np.ascontiguousarray(ptsA, np.float64).reshape(-1, 2)
        ptsB = np.ascontiguousarray(ptsB, np.float64).reshape(-1, 2)
        nA, nB = len(ptsA), len(ptsB)
        dim = descA.shape[1] if descA.ndim == 2 and descA.shape[0] else (descB.shape[1] if descB.ndim == 2 else 128)
        cap = max(min(nA, nB), 1)
        pairs = np.zeros((cap, 2), np.int32)
        m1 = np.full(max(nA, 1), -1, np.int32)
        m2 = np.full(max(nB, 1), -1, np.int32)
        cnt = C.c_int(0)
        call = lib().sfm_match_pair_u8 if as_bytes else lib().sfm_match_pair
        self.check(call(self._h, algorithm, _ptr(descA), _ptr(ptsA), nA, _ptr(descB), _ptr(ptsB),
                        nB, dim, metric, threshold, max_dist_sq, _ptr(pairs), cap,
                        C.byref(cnt), _ptr(m1), _ptr(m2)))
        out = pairs[:cnt.value].copy()
        if return_best:
            return out, m1[:nA].copy(), m2[:nB].copy()
        return out

    def imageset(self, counts, desc, pts, dim=None):
        return ImageSet(self, counts, desc, pts, dim)

    def pinned_like(self, array):
        """A copy of `array` in page-locked host memory (sfm_host_alloc); freed with the array's last view."""
        array = np.ascontiguousarray(array)
        p = C.c_void_p()
        self.check(lib().sfm_host_alloc(self._h, array.nbytes, C.byref(p)))
        block = _PinnedBlock(p.value, array.nbytes)
        out = np.asarray(block)[:array.nbytes].view(array.dtype).reshape(array.shape)
        out[...] = array
        return out

    # ---- RANSAC ---------------------------------------------------------------------------
    def ransac_pair(self, x, y, K, samples, max_sq_err=10.0, success_prob=0.99, adaptive=True, min_n=8,
                    loss=LOSS_REPROJECTION, per_row=True, static_workspace=False):
        """solve::RANSAC for one pair.  per_row=False leaves out the per-hypothesis counts and error sums
        (the reference does not return them); the library then scores rows in growing slices and stops
        where the adaptive cap stops the reference's loop.  static_workspace=True calls the entry point
        that allocates nothing (sfm_ransac_pair_static)."""
        x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
        y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
        if len(x) != len(y):  # RANSAC.cpp:148
            return dict(inliers=np.zeros(0, np.int32), best_h=-1, pose=np.eye(4), counts=None, toterr=None,
                        iterations=0)
        Kc = np.asfortranarray(K, np.float64).reshape(-1, order="F").copy()
        if isinstance(samples, (int, np.integer)):  # a row count: the library draws the table on the device
            H, samples = int(samples), None
        else:
            samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 8 if min_n == 8 else min_n)
            H = samples.shape[0]
        n = len(x)
        inl = np.zeros(max(n, 1), np.int32)
        pose = np.eye(4).reshape(-1, order="F").copy()
        counts = np.zeros(max(H, 1), np.int32) if per_row else None
        toterr = np.zeros(max(H, 1), np.float64) if per_row else None
        cnt, best, iters = C.c_int(0), C.c_int(-1), C.c_int(0)
        entry = lib().sfm_ransac_pair_static if static_workspace else lib().sfm_ransac_pair
        self.check(entry(self._h, _ptr(x), _ptr(y), n, _ptr(Kc), _ptr(samples), H, min_n,
                         max_sq_err, success_prob, int(bool(adaptive)), loss, _ptr(inl), max(n, 1),
                         C.byref(cnt), C.byref(best), _ptr(pose), _ptr(counts), _ptr(toterr), C.byref(iters)))
        return dict(inliers=inl[:cnt.value].copy(), best_h=best.value, pose=pose.reshape(4, 4, order="F").copy(),
                    counts=counts[:H].copy() if per_row else None, toterr=toterr[:H].copy() if per_row else None,
                    iterations=iters.value)

    def eight_point_fit(self, x, y, samples):
        x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
        y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 8)
        H = samples.shape[0]
        pose12 = np.zeros((max(H, 1), 12))
        self.check(lib().sfm_eight_point_fit(self._h, _ptr(x), _ptr(y), len(x), _ptr(samples), H, _ptr(pose12)))
        return pose12[:H]

    def scale_chain(self, offsets, track_ids, points, poses, start_pose=None, max_translation=15.0, n_tracks=0):
        """Scale chaining + running point average over a frame sequence (scene.cpp:301-421).
        offsets [S+1], track_ids [total], points [total,3], poses [S,4,4].  Returns a dict with
        extrinsics [S+1,4,4], points3d [n_tracks,3], counts [n_tracks], scale [S]."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        S = len(offsets) - 1
        track_ids = np.ascontiguousarray(track_ids, np.int32)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        poses_c = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(S, 4, 4).transpose(0, 2, 1))
        start = np.eye(4) if start_pose is None else np.asarray(start_pose, np.float64)
        start_c = np.ascontiguousarray(start.T)
        out = _chain_outputs(S, n_tracks)
        self.check(lib().sfm_scale_chain(self._h, S, _ptr(offsets), _ptr(track_ids), _ptr(points), _ptr(poses_c),
                                         _ptr(start_c), float(max_translation), int(n_tracks), _ptr(out[0]),
                                         _ptr(out[1]), _ptr(out[2]), _ptr(out[3])))
        return _chain_result(out, S, n_tracks)

    def scale_chain_colors(self, offsets, track_ids, points, colors, poses, start_pose=None, max_translation=15.0, n_tracks=0):
        """scale_chain with the running colour average of scene.cpp:394-407 (colors uint8 [total, 3]); the result
        also holds colors [n_tracks, 3]."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        S = len(offsets) - 1
        track_ids = np.ascontiguousarray(track_ids, np.int32)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        colors = np.ascontiguousarray(colors, np.uint8).reshape(-1, 3)
        if len(colors) != len(points):
            raise ValueError("one colour per point")
        poses_c = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(S, 4, 4).transpose(0, 2, 1))
        start = np.eye(4) if start_pose is None else np.asarray(start_pose, np.float64)
        start_c = np.ascontiguousarray(start.T)
        out = _chain_outputs(S, n_tracks)
        cols = np.zeros((max(n_tracks, 1), 3), np.uint8)
        self.check(lib().sfm_scale_chain_colors(self._h, S, _ptr(offsets), _ptr(track_ids), _ptr(points), _ptr(colors),
                                                _ptr(poses_c), _ptr(start_c), float(max_translation), int(n_tracks),
                                                _ptr(out[0]), _ptr(out[1]), _ptr(cols), _ptr(out[2]), _ptr(out[3])))
        res = _chain_result(out, S, n_tracks)
        res["colors"] = cols[:n_tracks]
        return res

    def pixel_bilinear_u8(self, image_bgr, pixels):
        """util::getPixelBilinearUchar (util.cpp:40-78) at every (u, v) of pixels: RGB uint8 [n, 3]."""
        img = np.ascontiguousarray(image_bgr, np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("image must be rows x cols x 3 bytes")
        px = np.ascontiguousarray(pixels, np.float64).reshape(-1, 2)
        out = np.zeros((max(len(px), 1), 3), np.uint8)
        self.check(lib().sfm_pixel_bilinear_u8(self._h, _ptr(img), img.shape[0], img.shape[1], _ptr(px), len(px), _ptr(out)))
        return out[:len(px)]

    def eight_point_from_subset(self, calc1, calc2, all1, all2):
        """eightPointAlgorithmFromSubset (eightPoint.cpp:225-372): (pose 4x4, points [n_all,3])."""
        c1 = np.ascontiguousarray(calc1, np.float64).reshape(-1, 2)
        c2 = np.ascontiguousarray(calc2, np.float64).reshape(-1, 2)
        a1 = np.ascontiguousarray(all1, np.float64).reshape(-1, 2)
        a2 = np.ascontiguousarray(all2, np.float64).reshape(-1, 2)
        pose = np.zeros(16)
        pts = np.zeros((max(len(a1), 1), 3))
        self.check(lib().sfm_eight_point_from_subset(self._h, _ptr(c1), _ptr(c2), len(c1), _ptr(a1), _ptr(a2), len(a1),
                                                     _ptr(pose), _ptr(pts)))
        return pose.reshape(4, 4, order="F").copy(), pts[:len(a1)]

    def eight_point_algorithm(self, x, y):
        """eightPointAlgorithm (eightPoint.cpp:57-223): (pose 4x4, points [n,3])."""
        x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
        y = np.ascontiguousarray(y, np.float64).reshape(-1, 2)
        pose = np.zeros(16)
        pts = np.zeros((max(len(x), 1), 3))
        self.check(lib().sfm_eight_point_algorithm(self._h, _ptr(x), _ptr(y), len(x), _ptr(pose), _ptr(pts)))
        return pose.reshape(4, 4, order="F").copy(), pts[:len(x)]

    def ransac_batch(self, corr_offsets, x, y, K, samples, n_hyp, max_sq_err=10.0, success_prob=0.99,
                     adaptive=False, loss=LOSS_REPROJECTION, batch=None, device_ptrs=None):
        """x, y: concatenated correspondences (host arrays), or pass device_ptrs=(x_ptr, y_ptr, samples_ptr)."""
        off = np.ascontiguousarray(corr_offsets, np.int64)
        P = len(off) - 1
        Kc = np.asfortranarray(K, np.float64).reshape(-1, order="F").copy()
        b = batch if batch is not None else RansacBatch(self)
        if device_ptrs is not None:
            xp, yp, sp = device_ptrs
            self.check(lib().sfm_ransac_batch_run_device(self._h, P, _ptr(off), C.c_void_p(xp), C.c_void_p(yp),
                                                         _ptr(Kc), C.c_void_p(sp), n_hyp, max_sq_err, success_prob,
                                                         int(bool(adaptive)), loss, C.byref(b._h)))
        else:
            x = np.ascontiguousarray(x, np.float64)
            y = np.ascontiguousarray(y, np.float64)
            samples = np.ascontiguousarray(samples, np.int32)
            self.check(lib().sfm_ransac_batch_run(self._h, P, _ptr(off), _ptr(x), _ptr(y), _ptr(Kc), _ptr(samples),
                                                  n_hyp, max_sq_err, success_prob, int(bool(adaptive)), loss,
                                                  C.byref(b._h)))
        b.n_pairs = P
        return b


class ImageSet:
    """Descriptor/point table resident in HBM (sfm_imageset_*)."""

    def __init__(self, ctx: Context, counts, desc, pts, dim=None):
        _live.add(self)
        self.ctx = ctx
        counts = np.ascontiguousarray(counts, np.int32)
        self._h = C.c_void_p()
        self.n_images = len(counts)
        if isinstance(desc, np.ndarray) and desc.dtype == np.uint8:
            desc = np.ascontiguousarray(desc)
            pts = np.ascontiguousarray(pts, np.float64)
            if dim is None:
                dim = desc.shape[-1]
            ctx.check(lib().sfm_imageset_create_u8(ctx._h, len(counts), _ptr(counts), _ptr(desc), _ptr(pts), dim,
                                                   C.byref(self._h)))
        elif isinstance(desc, np.ndarray):
            desc = np.ascontiguousarray(desc, np.float32)
            pts = np.ascontiguousarray(pts, np.float64)
            if dim is None:
                dim = desc.shape[-1]
            ctx.check(lib().sfm_imageset_create(ctx._h, len(counts), _ptr(counts), _ptr(desc), _ptr(pts), dim,
                                                C.byref(self._h)))
        else:  # (device pointer ints); the caller keeps the memory alive
            ctx.check(lib().sfm_imageset_create_device(ctx._h, len(counts), _ptr(counts), C.c_void_p(int(desc)),
                                                       C.c_void_p(int(pts)), dim, C.byref(self._h)))
        self.counts = counts

    def device_bytes(self) -> int:
        return int(lib().sfm_imageset_device_bytes(self._h))

    def close(self):
        if self._h:
            lib().sfm_imageset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match(self, pair_list, threshold=0.5, max_dist_sq=1e8, metric=METRIC_COSINE, result=None, ratio=None):
        """ratio: Lowe's second-best test on top of the reference's matcher (sfm_match_batch_ratio)."""
        pl = np.ascontiguousarray(pair_list, np.int32).reshape(-1, 2)
        res = result if result is not None else MatchResult(self.ctx)
        if metric == METRIC_L2:  # threshold = largest admissible squared descriptor distance
            self.ctx.check(lib().sfm_match_batch_l2(self.ctx._h, self._h, _ptr(pl), len(pl), threshold, max_dist_sq,
                                                    float(ratio or 0.0), C.byref(res._h)))
        elif ratio is not None:
            if metric != METRIC_COSINE:
                raise SfmError(ERR_UNSUPPORTED, "the second-best ratio test is defined for the cosine metric only")
            self.ctx.check(lib().sfm_match_batch_ratio(self.ctx._h, self._h, _ptr(pl), len(pl), threshold,
                                                       max_dist_sq, float(ratio), C.byref(res._h)))
        else:
            self.ctx.check(lib().sfm_match_batch(self.ctx._h, self._h, _ptr(pl), len(pl), metric, threshold,
                                                 max_dist_sq, C.byref(res._h)))
        res.n_pairs = len(pl)
        return res


class BinarySet:
    """Binary descriptor table resident in HBM (sfm_binset_*): uint8 [rows, 16 | 32 | 64]."""

    def __init__(self, ctx: Context, counts, desc, pts):
        _live.add(self)
        self.ctx = ctx
        self._h = C.c_void_p()
        counts = np.ascontiguousarray(counts, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        nbytes = desc.shape[1] if desc.ndim == 2 else 32
        ctx.check(lib().sfm_binset_create(ctx._h, len(counts), _ptr(counts), _ptr(desc), _ptr(pts), nbytes,
                                          C.byref(self._h)))

    def match(self, pair_list, max_distance=64, max_dist_sq=1e8, result=None):
        pl = np.ascontiguousarray(pair_list, np.int32).reshape(-1, 2)
        res = result if result is not None else MatchResult(self.ctx)
        self.ctx.check(lib().sfm_match_batch_hamming(self.ctx._h, self._h, _ptr(pl), len(pl), int(max_distance),
                                                     max_dist_sq, C.byref(res._h)))
        res.n_pairs = len(pl)
        return res

    def close(self):
        if self._h:
            lib().sfm_binset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchResult:
    """CSR match lists of a batch, resident on the device until fetched."""

    def __init__(self, ctx: Context):
        _live.add(self)
        self.ctx = ctx
        self._h = C.c_void_p()
        self.n_pairs = 0

    def total(self) -> int:
        t = C.c_int64(0)
        self.ctx.check(lib().sfm_match_result_total(self._h, C.byref(t)))
        return t.value

    def offsets(self) -> np.ndarray:
        off = np.zeros(self.n_pairs + 1, np.int64)
        self.ctx.check(lib().sfm_match_result_offsets(self._h, _ptr(off)))
        return off

    def pairs(self) -> np.ndarray:
        out = np.zeros((max(self.total(), 1), 2), np.int32)
        self.ctx.check(lib().sfm_match_result_pairs(self._h, _ptr(out)))
        return out[:self.total()]

    def device_ptrs(self):
        o, p = C.c_void_p(), C.c_void_p()
        self.ctx.check(lib().sfm_match_result_device(self._h, C.byref(o), C.byref(p)))
        return o.value, p.value

    def stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx.check(lib().sfm_match_result_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats_ex(self):
        """(candidates re-scored, pairs scored cell by cell, pairs that needed the filtered passes,
        pairs settled by the fp16 bound after the int8 bound flooded)."""
        out = np.zeros(4, np.int64)
        self.ctx.check(lib().sfm_match_result_stats_ex(self._h, _ptr(out)))
        return tuple(int(v) for v in out)

    def device_bytes(self):
        return int(lib().sfm_match_result_device_bytes(self._h))

    def close(self):
        if self._h:
            lib().sfm_match_result_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RansacBatch:
    def __init__(self, ctx: Context):
        _live.add(self)
        self.ctx = ctx
        self._h = C.c_void_p()
        self.n_pairs = 0

    def best(self):
        P = max(self.n_pairs, 1)
        bh = np.zeros(P, np.int32)
        bc = np.zeros(P, np.int32)
        be = np.zeros(P, np.float64)
        pose = np.zeros((P, 12), np.float64)
        it = np.zeros(P, np.int32)
        self.ctx.check(lib().sfm_ransac_batch_best(self._h, _ptr(bh), _ptr(bc), _ptr(be), _ptr(pose), _ptr(it)))
        n = self.n_pairs
        return dict(best_h=bh[:n], count=bc[:n], toterr=be[:n], pose12=pose[:n], iterations=it[:n])

    def inlier_offsets(self):
        off = np.zeros(self.n_pairs + 1, np.int64)
        self.ctx.check(lib().sfm_ransac_batch_inlier_offsets(self._h, _ptr(off)))
        return off

    def handed_over(self):
        """(candidates, passed over, handed over): sample rows with many inliers, as the last run dealt with them."""
        c, p, h = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.ctx.check(lib().sfm_ransac_batch_handed_over(self._h, C.byref(c), C.byref(p), C.byref(h)))
        return int(c.value), int(p.value), int(h.value)

    def inliers(self):
        off = self.inlier_offsets()
        out = np.zeros(max(int(off[-1]), 1), np.int32)
        self.ctx.check(lib().sfm_ransac_batch_inliers(self._h, _ptr(out)))
        return off, out[:int(off[-1])]

    def refit(self, total: int):
        """eightPointAlgorithm on every pair's inliers (all points when fewer than 8 inliers).
        total = number of correspondences of the batch.  Returns (pose16 [P,4,4], used [P],
        points [total,3]; pair p's points start at its correspondence offset)."""
        self.ctx.check(lib().sfm_ransac_batch_refit(self._h))
        P = max(self.n_pairs, 1)
        pose = np.zeros((P, 16))
        used = np.zeros(P, np.int32)
        pts = np.zeros((max(total, 1), 3))
        self.ctx.check(lib().sfm_ransac_batch_refit_fetch(self._h, _ptr(pose), _ptr(used), _ptr(pts)))
        n = self.n_pairs
        return pose[:n].reshape(n, 4, 4).transpose(0, 2, 1).copy(), used[:n], pts[:total]

    def close(self):
        if self._h:
            lib().sfm_ransac_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _batch_scale_chain(self, corr_track_ids, start_pose=None, max_translation=15.0, n_tracks=0):
    """Scale chaining over the refitted pairs of this batch (pair p = step p), device to device.
    corr_track_ids: the track id of every correspondence of the batch."""
    ids = np.ascontiguousarray(corr_track_ids, np.int32)
    start = np.eye(4) if start_pose is None else np.asarray(start_pose, np.float64)
    start_c = np.ascontiguousarray(start.T)
    S = self.n_pairs
    out = _chain_outputs(S, n_tracks)
    self.ctx.check(lib().sfm_scale_chain_refit(self._h, _ptr(ids), _ptr(start_c), float(max_translation),
                                               int(n_tracks), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3])))
    return _chain_result(out, S, n_tracks)


RansacBatch.scale_chain = _batch_scale_chain


class Correspondences:
    """Normalised correspondences of every pair of a match result, resident on the device
    (sfm_corr_*): the hand-off from matching to RANSAC."""

    def __init__(self, ctx: Context, imageset: ImageSet, result: MatchResult, K, reuse: "Correspondences" = None):
        _live.add(self)
        self.ctx = ctx
        self._h = reuse._h if reuse is not None else C.c_void_p()
        if reuse is not None:
            reuse._h = C.c_void_p()
        Kinv = np.linalg.inv(np.asarray(K, np.float64))
        self.Kinv = Kinv
        Kc = np.asfortranarray(Kinv).reshape(-1, order="F").copy()
        ctx.check(lib().sfm_corr_from_matches(ctx._h, imageset._h, result._h, _ptr(Kc), C.byref(self._h)))
        self.n_pairs = result.n_pairs

    def offsets(self) -> np.ndarray:
        off = np.zeros(self.n_pairs + 1, np.int64)
        self.ctx.check(lib().sfm_corr_offsets(self._h, _ptr(off)))
        return off

    def fetch(self):
        n = int(self.offsets()[-1])
        x = np.zeros((max(n, 1), 2))
        y = np.zeros((max(n, 1), 2))
        self.ctx.check(lib().sfm_corr_fetch(self._h, _ptr(x), _ptr(y)))
        return x[:n], y[:n]

    def sample_table(self, n_hyp: int, seed: int) -> np.ndarray:
        out = np.zeros((max(self.n_pairs, 1), max(n_hyp, 1), 8), np.int32)
        dev = C.c_void_p()
        self.ctx.check(lib().sfm_corr_sample_table(self._h, n_hyp, C.c_uint64(seed), C.byref(dev), _ptr(out)))
        return out[:self.n_pairs, :n_hyp]

    def ransac(self, K, n_hyp: int, seed: int, max_sq_err=10.0, success_prob=0.99, adaptive=False,
               loss=LOSS_REPROJECTION, batch: "RansacBatch" = None) -> "RansacBatch":
        Kc = np.asfortranarray(K, np.float64).reshape(-1, order="F").copy()
        b = batch if batch is not None else RansacBatch(self.ctx)
        self.ctx.check(lib().sfm_ransac_batch_run_corr(self.ctx._h, self._h, _ptr(Kc), n_hyp, C.c_uint64(seed),
                                                       max_sq_err, success_prob, int(bool(adaptive)), loss,
                                                       C.byref(b._h)))
        b.n_pairs = self.n_pairs
        return b

    def close(self):
        if self._h:
            lib().sfm_corr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
