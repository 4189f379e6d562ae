"""One rank of the two-process multi-GPU test (started by tests/test_multi_gpu.py, never collected by pytest):
python multi_rank_child.py RANK WORLD ID_FILE.  Rank 0 writes the communicator's id to ID_FILE, the others wait
for it; every rank matches and runs RANSAC over its share on device RANK and gathers to rank 0, which compares
the gathered results with the same work done on its own device alone and prints OK."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sfm_amd import capi, synth  # noqa: E402
import multi_cases  # noqa: E402


def main():
    rank, world, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    ctx = capi.Context(rank)
    ctx.set_option("comm_timeout_ms", 60000)
    if rank == 0:
        uid = capi.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise SystemExit("rank %d: no id after 60 s" % rank)
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    comm = capi.Comm.create(ctx, rank, world, uid)
    multi_cases.run_rank(ctx, comm, rank, world)
    comm.close()
    ctx.close()
    print(f"rank {rank}: OK")


if __name__ == "__main__":
    main()
