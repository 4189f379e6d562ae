"""Handles may be destroyed in any order: every object holds a reference to its context, whose streams go
with the last of them.  Each case runs in a child process that must end with status 0 (an object destroyed
after its context used to synchronise on a destroyed stream: an abort at exit under a profiler, a
use-after-free without one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import os, sys
import numpy as np
sys.path.insert(0, %r)
from sfm_amd import capi, synth
ctx = capi.Context(0)
desc, pts, counts, _ = synth.frame_sequence(3, 500, seed=1)
s = ctx.imageset(counts, desc, pts)
res = s.match(np.array([[0, 1], [1, 2]], np.int32), threshold=0.95, max_dist_sq=1e4)
corr = capi.Correspondences(ctx, s, res, synth.default_K())
batch = corr.ransac(synth.default_K(), 64, 1, max_sq_err=1.0, success_prob=0.999, adaptive=True)
batch.refit(int(res.offsets()[-1]))
comm = capi.Comm.create(ctx, 0, 1, capi.Comm.unique_id())
g = comm.gather(res, 2)
tv = synth.two_view(200, inlier_ratio=0.6, seed=3)
ctx.ransac_pair(tv["x"], tv["y"], tv["K"], synth.sample_table(200, 64, seed=4))   # the context's own objects
ctx.match_pair(desc[:500], pts[:500], desc[500:1000], pts[500:1000], threshold=0.95)
n = res.total()
""" % ROOT

CASES = {
    "context_first": "ctx.close()\nfor o in (g, comm, batch, corr, res, s): o.close()\n",
    "context_in_the_middle": "batch.close(); comm.close(); ctx.close()\nassert g.total() == n\nfor o in (corr, g, res, s): o.close()\n",
    "nothing_closed": "",
    "no_atexit_hook": "import atexit; atexit.unregister(capi._close_all)\n",
    "results_outlive_the_context": "ctx.close()\nassert res.total() == n and len(res.pairs()) == n and len(g.pairs()) == n\n",
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_any_destruction_order_exits_cleanly(case):
    r = subprocess.run([sys.executable, "-c", PRELUDE + CASES[case] + "print('done', n)\n"], capture_output=True, text=True,
                       timeout=110)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "done" in r.stdout


def test_cpp_mirror_exits_with_a_live_thread_local_context(tmp_path):
    """sfm_amd/host/test_frontend ends without releasing the context its thread made on first use."""
    exe = os.path.join(ROOT, "sfm_amd", "host", "test_frontend")
    assert os.path.exists(exe), "build with make -C sfm_amd/host"
    r = subprocess.run([exe, "--lifetime"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "lifetime ok" in r.stdout
