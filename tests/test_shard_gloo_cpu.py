"""The N > 1 path on CPU: two processes (gloo), each matches its share of an exhaustive pair list
(with the CPU restatement standing in for the GPU), the CSR lists are gathered to rank 0 and must
equal the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, block, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_api as oracle
    from sfm_amd import shard, synth
    oracle.set_threads(1)
    n_img, n_feat = 9, 96
    desc, pts, _ = synth.image_set(n_img, n_feat, seed=5)
    pl = synth.exhaustive_pairs(n_img)
    owned = shard.partition_pairs(len(pl), rank, world, block=block)
    lists = []
    for a, b in pl[owned]:
        lists.append(oracle.match(desc[a * n_feat:(a + 1) * n_feat], pts[a * n_feat:(a + 1) * n_feat],
                                  desc[b * n_feat:(b + 1) * n_feat], pts[b * n_feat:(b + 1) * n_feat],
                                  threshold=0.95, max_dist_sq=1e8)[0])
    counts = torch.tensor([len(l) for l in lists], dtype=torch.int64)
    payload = torch.from_numpy(np.concatenate(lists).astype(np.int32) if lists else np.zeros((0, 2), np.int32))
    got = shard.gather_csr(counts, payload, len(pl), rank, world, block=block)
    got2 = shard.gather_csr(counts, payload, len(pl), rank, world, block=block, ordered=False)
    if rank == 0:
        off, allp = got
        starts, cnts, raw = got2
        np.savez(out_path, off=off, pairs=allp.numpy(), starts=starts, counts=cnts, raw=raw.numpy())
    else:
        assert got is None and got2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("block", [4, 64])
def test_two_rank_gather_equals_unsharded(tmp_path, oracle, block):
    from sfm_amd import synth
    port = _free_port()
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, block, out), nprocs=2, join=True)
    g = np.load(out)
    n_img, n_feat = 9, 96
    desc, pts, _ = synth.image_set(n_img, n_feat, seed=5)
    pl = synth.exhaustive_pairs(n_img)
    assert len(g["off"]) == len(pl) + 1 and g["off"][0] == 0
    for k, (a, b) in enumerate(pl):
        exp = oracle.match(desc[a * n_feat:(a + 1) * n_feat], pts[a * n_feat:(a + 1) * n_feat],
                           desc[b * n_feat:(b + 1) * n_feat], pts[b * n_feat:(b + 1) * n_feat],
                           threshold=0.95, max_dist_sq=1e8)[0]
        assert np.array_equal(g["pairs"][g["off"][k]:g["off"][k + 1]], exp), k
        assert np.array_equal(g["raw"][g["starts"][k]:g["starts"][k] + g["counts"][k]], exp), k  # unordered layout
    assert g["off"][-1] == len(g["pairs"]) > 0
