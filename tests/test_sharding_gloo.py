"""N>1 path on CPU: two gloo ranks run the SAME partition / gather / reassembly code the GPU path runs over RCCL
(svtav1_hip.sharded), with the oracle standing in for the device compute (tests only):
  * frame-sharded ME: each rank searches its contiguous superblock range, the gathered me_results equal the single-process result;
  * recon exchange: each rank holds its SB-row slab of a reconstructed picture, one all-gather + generate_padding rebuilds the
    padded reference picture on every rank."""
import os
import socket

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import sys
    import torch.distributed as dist
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for p in (root, os.path.join(root, "svt-av1-1_amd", "python"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _me_worker(rank, world, port, w, h, granularity, out_path):
    import torch
    dist = _init(rank, world, port)
    from me_chain_util import oracle_me_picture
    from oracle.binding import Oracle
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    sme = sharded.ShardedMotionEstimation(w, h, granularity=granularity)
    first = sme.first

    def compute(sb_xy, n_local):  # CPU stand-in for svthip_motion_estimate_batch_dev on the shard's SB list
        res, _ = oracle_me_picture(Oracle(), pics, P, True, True, 0, sb_subset=np.arange(first, first + n_local))
        return torch.from_numpy(res.view(np.uint8).reshape(1, n_local, 85, 24).copy())

    full = sme.run(compute, gather=True)
    if rank == 1:  # every rank holds the complete result; check a non-zero rank's copy
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size,granularity", [((256, 192), "row"), ((256, 320), "row"), ((320, 192), "sb")])  # ragged 2+1 rows, 3+2 rows, 8+7 SBs
def test_two_rank_frame_sharded_me(tmp_path, oracle, size, granularity):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from me_chain_util import oracle_me_picture
    w, h = size
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_me_worker, args=(2, _free_port(), w, h, granularity, out), nprocs=2, join=True)
    got = np.load(out).reshape(-1).view(svtav1_hip.ME_CU_RESULT_DTYPE).reshape(-1, 85)
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    ref, _ = oracle_me_picture(oracle, pics, svtav1_hip.default_me_params(w, h, 3, 1), True, True, 0)
    assert got.shape == ref.shape
    for f in ref.dtype.names:
        assert np.array_equal(got[f], ref[f]), f


def _recon_worker(rank, world, port, w, h, pad, dtype_name, out_path):
    import torch
    dist = _init(rank, world, port)
    from oracle.binding import Oracle
    dt = np.dtype(dtype_name)
    rng = np.random.default_rng(99)
    picture = rng.integers(0, 256 if dt == np.uint8 else 1024, (h, w)).astype(dt)  # the same on every rank: the "true" reconstruction
    ex = sharded.ReconExchange(w, h, pad)
    y0, n = ex.my_rows
    plane = np.full((h + 2 * pad, w + 2 * pad), 0x77, dt)  # this rank has reconstructed only its slab
    plane[pad + y0:pad + y0 + n, pad:pad + w] = picture[y0:y0 + n]
    t = torch.from_numpy(plane.view(np.uint8) if dt == np.uint16 else plane)
    orc = Oracle()

    def pad_fn(tt):
        a = tt.numpy().view(dt) if dt == np.uint16 else tt.numpy()
        orc.generate_padding(a, w, h, pad, pad)

    ex.exchange(t, pad_fn)
    np.save(out_path.format(rank=rank), t.numpy().view(dt) if dt == np.uint16 else t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(256, 192, 160, "uint8"), (320, 328, 80, "uint8"), (192, 136, 160, "uint16")])
def test_two_rank_recon_exchange(tmp_path, case):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h, pad, dt = case
    out = str(tmp_path / "plane_{rank}.npy")
    mp.spawn(_recon_worker, args=(2, _free_port(), w, h, pad, dt, out), nprocs=2, join=True)
    rng = np.random.default_rng(99)
    picture = rng.integers(0, 256 if dt == "uint8" else 1024, (h, w)).astype(dt)
    want = np.pad(picture, pad, mode="edge")
    for r in range(2):
        assert np.array_equal(np.load(out.format(rank=r)), want), f"rank {r}"


def test_shard_partition_covers_everything():
    for (w, h, world) in [(1920, 1080, 8), (1920, 1080, 4), (3840, 2160, 8), (856, 480, 2), (256, 64, 4)]:
        nx, ny = (w + 63) // 64, (h + 63) // 64
        for gran in ("row", "sb"):
            parts = [sharded.shard_sb_indices(w, h, world, r, gran) for r in range(world)]
            assert np.array_equal(np.concatenate(parts), np.arange(nx * ny))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= (nx if gran == "row" else 1)
        assert np.array_equal(svtav1_hip.shard_sb_rows(w, h, world, 0), sharded.shard_sb_indices(w, h, world, 0, "row"))
    # 1080p over 8 ranks: 510 SBs -> 64 / 63 per rank (row granularity would be 90 / 60)
    assert [sharded.shard_sb_range(1920, 1080, 8, r)[1] for r in range(8)] == [64] * 6 + [63] * 2


def test_gather_rows_single_process_identity():
    torch = pytest.importorskip("torch")
    # world 1 never calls the collective
    sme = sharded.ShardedMotionEstimation(256, 128)
    assert sme.world == 1 and sme.count == 8 and sme.first == 0
    out = sme.run(lambda sb, n: torch.zeros((1, n, 85, 24), dtype=torch.uint8))
    assert out.shape == (1, 8, 85, 24)
