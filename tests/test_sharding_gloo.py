"""N>1 path on CPU: two gloo ranks shard the SB rows of one picture (no data-path collective), run the oracle chain on
their shard, and the gathered me_results equal the single-process result."""
import os
import socket

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "svt-av1-1_amd", "python")); sys.path.insert(0, os.path.join(root, "tests"))
    from me_chain_util import oracle_me_picture
    from oracle.binding import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    mine = svtav1_hip.shard_sb_rows(w, h, world, rank)
    res, _ = oracle_me_picture(Oracle(), pics, P, True, True, 0, sb_subset=mine)
    # results only leave the rank at the end (the host gathers me_results); the ME itself exchanged nothing
    payload = torch.from_numpy(res.view(np.uint8).reshape(-1).copy())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([payload.numel()], dtype=torch.int64))
    bufs = [torch.zeros(int(s.item()), dtype=torch.uint8) for s in sizes]
    dist.all_gather(bufs, payload) if len(set(int(s.item()) for s in sizes)) == 1 else None
    if len(set(int(s.item()) for s in sizes)) != 1:
        # ragged shards: gather through point-to-point
        if rank == 0:
            bufs[0] = payload
            for r in range(1, world):
                dist.recv(bufs[r], src=r)
        else:
            dist.send(payload, dst=0)
    if rank == 0:
        np.save(out_path, np.concatenate([b.numpy() for b in bufs]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(256, 192), (256, 320)])  # 3 SB rows (ragged 2+1) and 5 SB rows (3+2)
def test_two_rank_sb_row_sharding(tmp_path, oracle, size):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from me_chain_util import oracle_me_picture
    w, h = size
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    got = np.load(out).view(svtav1_hip.ME_CU_RESULT_DTYPE).reshape(-1, 85)
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    ref, _ = oracle_me_picture(oracle, pics, svtav1_hip.default_me_params(w, h, 3, 1), True, True, 0)
    assert got.shape == ref.shape
    for f in ref.dtype.names:
        assert np.array_equal(got[f], ref[f]), f


def test_shard_partition_covers_everything():
    for (w, h, world) in [(1920, 1080, 8), (1920, 1080, 4), (3840, 2160, 8), (856, 480, 2), (256, 64, 4)]:
        nx, ny = (w + 63) // 64, (h + 63) // 64
        parts = [svtav1_hip.shard_sb_rows(w, h, world, r) for r in range(world)]
        allidx = np.concatenate(parts)
        assert np.array_equal(allidx, np.arange(nx * ny))
        rows = [len(p) // nx for p in parts]
        assert max(rows) - min(rows) <= 1
