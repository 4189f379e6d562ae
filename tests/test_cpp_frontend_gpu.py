"""The C++ host mirror (sfm_amd/host: match() / RANSAC() with the reference's option structs)
driven end to end through a small binary, checked against the CPU restatement."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sfm_amd", "host", "test_frontend")


def run_case(tmp_path, dA, pA, dB, pB, thr, gate, tv, max_iter, seed, tag):
    case, out = str(tmp_path / f"case{tag}.bin"), str(tmp_path / f"out{tag}.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<iiiff", len(dA), len(dB), dA.shape[1], thr, gate))
        for a in (dA.astype(np.float32), pA.astype(np.float64), dB.astype(np.float32), pB.astype(np.float64)):
            f.write(a.tobytes())
        f.write(struct.pack("<iiQdd", len(tv["x"]), max_iter, seed, 1.0, 0.999))
        f.write(np.asfortranarray(tv["K"]).reshape(-1, order="F").astype(np.float64).tobytes())
        f.write(tv["x"].astype(np.float64).tobytes())
        f.write(tv["y"].astype(np.float64).tobytes())
    r = subprocess.run([EXE, case, out], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr + r.stdout
    blob = open(out, "rb").read()
    raw = np.frombuffer(blob[:len(blob) // 4 * 4], np.int32)
    nm = raw[0]
    pairs = raw[1:1 + 2 * nm].reshape(-1, 2)
    ni = raw[1 + 2 * nm]
    inl = raw[2 + 2 * nm:2 + 2 * nm + ni]
    at = 4 * (2 + 2 * nm + ni)
    n3 = struct.unpack_from("<i", blob, at)[0]
    pose = np.frombuffer(blob, np.float64, 16, at + 4).reshape(4, 4, order="F")
    pts = np.frombuffer(blob, np.float64, 3 * n3, at + 4 + 128).reshape(-1, 3)
    return pairs, inl, r.stdout, pose, pts


def test_cpp_match_and_ransac(tmp_path, oracle):
    assert os.path.exists(EXE), "build with make -C sfm_amd/host"
    d, p, _ = synth.image_set(2, 700, seed=61)
    dA, pA, dB, pB = d[:650], p[:650], d[700:], p[700:]
    tv = synth.two_view(1500, inlier_ratio=0.7, seed=62, noise_px=0.3)
    pairs, inl, log, pose, pts = run_case(tmp_path, dA, pA, dB, pB, 0.95, 1e4, tv, 2048, 1234, "a")
    exp, _, _ = oracle.match(dA, pA, dB, pB, threshold=0.95, max_dist_sq=1e4)
    assert np.array_equal(pairs, exp)
    assert "RANSAC: finished after" in log
    assert len(inl) > 100 and np.all(np.diff(inl) > 0)
    assert tv["inlier_mask"][inl].mean() > 0.95
    pairs2, inl2, _, _, _ = run_case(tmp_path, dA, pA, dB, pB, 0.95, 1e4, tv, 2048, 1234, "b")
    assert np.array_equal(inl, inl2) and np.array_equal(pairs, pairs2)
    # eightPointAlgorithm on the inliers (scene.cpp:288) through the C++ mirror
    epose, epts, _, _ = oracle.eight_point(tv["x"][inl], tv["y"][inl])
    assert np.abs(pose - epose).max() < 1e-8
    assert len(pts) == len(inl) and np.abs(pts - epts).max() <= 1e-6 * np.abs(epts).max()
