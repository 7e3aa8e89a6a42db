"""Multi-GPU data path on the HIP side, as far as one GPU box can show it:
  * the HIP ME of a SHARD (svthip_motion_estimate_batch_dev on a contiguous SB range) equals the same rows of the unsharded HIP
    result, for every shard of 2-, 3- and 8-way partitions, 85 and 209 PUs;
  * two real ranks (two processes sharing the one GPU, gloo for the collective because RCCL refuses two ranks on one device) run
    svtav1_hip.sharded -- the code the RCCL path runs -- with the HIP compute, and every rank ends with the unsharded result;
  * the recon exchange's device half: slabs written into a device plane, HIP border padding, equals the padded picture."""
import os
import socket

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import sharded, synth

pytestmark = pytest.mark.gpu


def _setup(w, h):
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5, 7, 3)]
    pool, descs = svtav1_hip.build_picture_pool(pics)
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    return pool, descs, P


@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_hip_shards_equal_unsharded_hip(hip_ctx, world, n_pu):
    torch = pytest.importorskip("torch")
    w, h = 448, 200   # 7 x 4 = 28 SBs, partial bottom row
    pool, descs, P = _setup(w, h)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    curs, r0, r1 = [descs[0], descs[3]], [descs[1], descs[4]], [descs[2], descs[0]]   # two B pictures per batch
    compute = sharded.device_me_compute(hip_ctx, d_pool.data_ptr(), curs, r0, r1, P, n_pu=n_pu)
    sb = svtav1_hip.sb_origins(w, h)
    full = compute(np.ascontiguousarray(sb), sb.shape[0]).cpu().numpy()
    assert full.shape == (2, 28, n_pu, 24) and full.any()
    for gran in ("sb", "row"):
        parts = []
        for r in range(world):
            first, count = sharded.shard_sb_range(w, h, world, r, gran)
            parts.append(compute(np.ascontiguousarray(sb[first:first + count]), count).cpu().numpy())
        assert np.array_equal(np.concatenate(parts, axis=1), full), (world, gran)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, w, h, out_path):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for p in (root, os.path.join(root, "svt-av1-1_amd", "python")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pool, descs, P = _setup(w, h)
    ctx = svtav1_hip.Context(0)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    dev_compute = sharded.device_me_compute(ctx, d_pool.data_ptr(), [descs[0]], [descs[1]], [descs[2]], P, n_pu=85)
    sme = sharded.ShardedMotionEstimation(w, h, granularity="sb")
    full = sme.run(lambda sb_xy, n: dev_compute(sb_xy, n).cpu(), gather=True)   # HIP compute, gloo gather (one GPU for both ranks)
    np.save(out_path.format(rank=rank), full.numpy())
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_hip_compute_shared_gather_code(hip_ctx, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h = 320, 192
    out = str(tmp_path / "full_{rank}.npy")
    mp.spawn(_rank_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    pool, descs, P = _setup(w, h)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    sb = svtav1_hip.sb_origins(w, h)
    want = sharded.device_me_compute(hip_ctx, d_pool.data_ptr(), [descs[0]], [descs[1]], [descs[2]], P)(np.ascontiguousarray(sb), sb.shape[0])
    want = want.cpu().numpy()
    for r in range(2):
        assert np.array_equal(np.load(out.format(rank=r)), want), f"rank {r}"


@pytest.mark.parametrize("case", [(1920, 1080, 160, 1, 8), (3840, 2160, 160, 2, 8), (960, 540, 80, 1, 3)])
def test_recon_slabs_then_hip_padding(hip_ctx, case):
    """What every rank does after the all-gather: the slabs of all ranks sit in its device plane, svthip_pad_plane_dev completes the
    reference picture (PadRefAndSetFlags).  The slab geometry is ReconExchange's (SB-row slabs)."""
    torch = pytest.importorskip("torch")
    w, h, pad, sb_, world = case
    dt = np.uint8 if sb_ == 1 else np.uint16
    rng = np.random.default_rng(w)
    picture = rng.integers(0, 256 if sb_ == 1 else 1024, (h, w)).astype(dt)
    stride = w + 2 * pad
    plane = torch.full(((h + 2 * pad) * stride * sb_,), 0x77, dtype=torch.uint8, device="cuda:0")
    host = np.full((h + 2 * pad, stride), 0x7777 if sb_ == 2 else 0x77, dt)
    covered = 0
    for r in range(world):
        ex = sharded.ReconExchange.__new__(sharded.ReconExchange)
        nx = (w + 63) // 64
        first, count = sharded.shard_sb_range(w, h, world, r, "row")
        y0 = (first // nx) * 64
        y1 = min(h, (first // nx + count // nx) * 64)
        host[pad + y0:pad + y1, pad:pad + w] = picture[y0:y1]
        covered += y1 - y0
    assert covered == h
    plane.copy_(torch.from_numpy(host.view(np.uint8).reshape(-1)))
    hip_ctx.pad_plane_dev(plane.data_ptr(), stride, w, h, pad, pad, sb_)
    hip_ctx.synchronize()
    got = plane.cpu().numpy().view(dt).reshape(h + 2 * pad, stride)
    assert np.array_equal(got, np.pad(picture, pad, mode="edge"))
