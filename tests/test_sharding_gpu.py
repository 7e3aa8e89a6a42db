"""Multi-GPU data path on the HIP side, as far as one GPU box can show it:
  * the HIP ME of a SHARD (svthip_motion_estimate_batch_dev on a contiguous SB range) equals the same rows of the unsharded HIP
    result, for every shard of 2-, 3- and 8-way partitions, 85 and 209 PUs;
  * two real ranks (two processes sharing the one GPU; the transfers of svthip_me_gather_plan go over gloo because RCCL refuses two
    ranks on one device) run svtav1_hip.sharded with the HIP compute, and every rank ends with the unsharded result;
  * the C ABI's exchange entries at world 1 (svthip_comm_create / svthip_recon_exchange_dev / svthip_me_gather_results_dev through
    sharded.Comm / ReconExchange): Y, Cb, Cr device planes holding every rank's slab come back padded like PadRefAndSetFlags."""
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
    sme = sharded.ShardedMotionEstimation(w, h)
    full = sme.run(lambda sb_xy, n: dev_compute(sb_xy, n).cpu(), gather=True)   # HIP compute, planned transfers over gloo (one GPU for both ranks)
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
def test_recon_exchange_entry_world1_all_slabs_then_padding(hip_ctx, case):
    """svthip_recon_exchange_dev through sharded.ReconExchange with a world-1 svthip_comm: the slabs of all (virtual) ranks -- geometry
    from ReconExchange's own row table for `world` ranks -- sit in the device planes Y, Cb, Cr; the entry pads all three like
    PadRefAndSetFlags.  (At world 1 the RCCL group is empty; the transfer lists themselves are executed by tests/test_sharding_gloo.py.)"""
    torch = pytest.importorskip("torch")
    w, h, pad, es, world = case
    dt = np.uint8 if es == 1 else np.uint16
    rng = np.random.default_rng(w)
    geo = [(w, h, pad), (w // 2, h // 2, pad // 2), (w // 2, h // 2, pad // 2)]
    truth = [rng.integers(0, 256 if es == 1 else 1024, (gh, gw)).astype(dt) for gw, gh, _ in geo]
    rows = [sharded.recon_slab_rows(h, world, r) for r in range(world)]
    assert sum(n for _, n in rows) == h and rows[0][0] == 0
    host = []
    for (gw, gh, gp), pic in zip(geo, truth):
        a = np.full((gh + 2 * gp, gw + 2 * gp), 0x7777 if es == 2 else 0x77, dt)
        sub = 0 if gw == w else 1
        for y0, n in rows:   # every rank's slab written where that rank would have reconstructed it
            a[gp + (y0 >> sub):gp + ((y0 + n) >> sub), gp:gp + gw] = pic[y0 >> sub:(y0 + n) >> sub]
        host.append(a)
    planes = [torch.from_numpy(a.view(np.uint8).copy()).to("cuda:0") for a in host]
    comm = sharded.Comm(hip_ctx)
    try:
        ex = sharded.ReconExchange(w, h, pad, comm=comm, sample_bytes=es)
        assert ex.world == 1 and ex.my_rows == (0, h)
        ex.exchange(planes)
        hip_ctx.synchronize()
    finally:
        comm.close()
    for p, (gw, gh, gp), pic in zip(planes, geo, truth):
        assert np.array_equal(p.cpu().numpy().view(dt), np.pad(pic, gp, mode="edge"))


def test_me_gather_entry_world1(hip_ctx):
    """svthip_me_gather_results_dev at world 1: the local [n_jobs][n][record] rows land in the full array (a strided device copy)."""
    torch = pytest.importorskip("torch")
    comm = sharded.Comm(hip_ctx)
    try:
        local = torch.randint(0, 255, (3, 28, 85, 24), dtype=torch.uint8, device="cuda:0")
        full = torch.zeros_like(local)
        comm.me_gather_results_dev(local.data_ptr(), full.data_ptr(), 3, 28, 85 * 24)
        hip_ctx.synchronize()
        assert torch.equal(local, full)
        sme = sharded.ShardedMotionEstimation(448, 200, comm=comm)
        assert (sme.first, sme.count, sme.n_total) == (0, 28, 28)
    finally:
        comm.close()
