"""N>1 path on CPU: gloo ranks execute the SAME partition and transfer lists the GPU path issues over RCCL -- they come from the C
library's host functions (svthip_shard_range, svthip_recon_slab_rows, svthip_recon_exchange_plan, svthip_me_gather_plan; no device
needed) -- with the oracle standing in for the device compute (tests only):
  * frame-sharded ME: each rank searches its contiguous superblock range, the gathered me_results equal the single-process result;
  * recon exchange: each rank holds its SB-row slab of a reconstructed picture (Y, Cb, Cr), the planned sends / receives +
    generate_padding rebuild the padded reference picture on every rank (2 and 3 ranks, ragged slabs, 8- and 16-bit planes)."""
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


def _me_worker(rank, world, port, w, h, out_path):
    import torch
    dist = _init(rank, world, port)
    from me_chain_util import oracle_me_picture
    from oracle.binding import Oracle
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    sme = sharded.ShardedMotionEstimation(w, h)
    first = sme.first

    def compute(sb_xy, n_local):  # CPU stand-in for svthip_motion_estimate_batch_dev on the shard's SB list (two "jobs": the same picture twice)
        res, _ = oracle_me_picture(Oracle(), pics, P, True, True, 0, sb_subset=np.arange(first, first + n_local))
        one = torch.from_numpy(res.view(np.uint8).reshape(1, n_local, 85, 24).copy())
        return torch.cat([one, one.flip(1)])

    full = sme.run(compute, gather=True)
    if rank == world - 1:  # every rank holds the complete result; check a non-zero rank's copy
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size,world", [((320, 192), 2), ((192, 192), 2), ((448, 200), 3)])  # 15 SBs -> 8+7, 9 -> 5+4, 28 -> 10+9+9
def test_frame_sharded_me_gather(tmp_path, oracle, size, world):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from me_chain_util import oracle_me_picture
    w, h = size
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_me_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got = np.load(out)
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (2, 0, 5)]
    ref, _ = oracle_me_picture(oracle, pics, svtav1_hip.default_me_params(w, h, 3, 1), True, True, 0)
    job0 = got[0].reshape(-1).view(svtav1_hip.ME_CU_RESULT_DTYPE).reshape(-1, 85)
    assert job0.shape == ref.shape
    for f in ref.dtype.names:
        assert np.array_equal(job0[f], ref[f]), f
    # job 1 = every rank's rows in reverse order inside its own range: the gather must keep jobs and ranges apart
    want1 = np.concatenate([got[0][f:f + c][::-1] for f, c in (sharded.shard_sb_range(w, h, world, r) for r in range(world))])
    assert np.array_equal(got[1], want1)


def _recon_worker(rank, world, port, w, h, pad, dtype_name, chroma, out_path):
    import torch
    dist = _init(rank, world, port)
    from oracle.binding import Oracle
    dt = np.dtype(dtype_name)
    es = dt.itemsize
    rng = np.random.default_rng(99)
    hi = 256 if dt == np.uint8 else 1024
    geo = [(w, h, pad)] + ([(w // 2, h // 2, pad // 2)] * 2 if chroma else [])
    truth = [rng.integers(0, hi, (gh, gw)).astype(dt) for gw, gh, _ in geo]  # the same on every rank: the "true" reconstruction
    ex = sharded.ReconExchange(w, h, pad, sample_bytes=es)
    y0, n = ex.my_rows
    planes = []
    for i, ((gw, gh, gp), pic) in enumerate(zip(geo, truth)):
        a = np.full((gh + 2 * gp, gw + 2 * gp), 0x77, dt)  # this rank has reconstructed only its slab
        r0, r1 = (y0, y0 + n) if i == 0 else (y0 // 2, (y0 + n) // 2)
        a[gp + r0:gp + r1, gp:gp + gw] = pic[r0:r1]
        planes.append(torch.from_numpy(a.view(np.uint8)))
    orc = Oracle()

    def pad_fn(i, tt):
        gw, gh, gp = geo[i]
        orc.generate_padding(tt.numpy().view(dt), gw, gh, gp, gp)

    ex.exchange(planes, pad_fn)
    np.savez(out_path.format(rank=rank), *[p.numpy().view(dt) for p in planes])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(256, 192, 160, "uint8", True, 2), (320, 328, 80, "uint8", True, 3), (192, 136, 160, "uint16", True, 2),
                                  (256, 64, 32, "uint8", False, 3)])   # the last: one SB row over 3 ranks -> two ranks own nothing
def test_recon_exchange_planned_transfers(tmp_path, case):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    w, h, pad, dt, chroma, world = case
    out = str(tmp_path / "planes_{rank}.npz")
    mp.spawn(_recon_worker, args=(world, _free_port(), w, h, pad, dt, chroma, out), nprocs=world, join=True)
    rng = np.random.default_rng(99)
    hi = 256 if dt == "uint8" else 1024
    geo = [(w, h, pad)] + ([(w // 2, h // 2, pad // 2)] * 2 if chroma else [])
    want = [np.pad(rng.integers(0, hi, (gh, gw)).astype(dt), gp, mode="edge") for gw, gh, gp in geo]
    for r in range(world):
        z = np.load(out.format(rank=r))
        for i, wv in enumerate(want):
            assert np.array_equal(z[f"arr_{i}"], wv), f"rank {r} plane {i}"


def test_plans_pair_up_and_cover_the_picture():
    """Every send of one rank's plan has the matching receive in its peer's plan (same plane, offset, size, in the same per-pair order)
    and the received slabs plus the rank's own tile the picture rows exactly -- for the BASELINE shapes at 8 ranks."""
    for (w, h, pad, es, world) in [(1920, 1080, 160, 1, 8), (3840, 2160, 160, 2, 8), (856, 480, 96, 1, 3)]:
        pic = sharded.ReconPicture()
        pic.y, pic.cb, pic.cr = 1, 1, 1
        pic.stride_y, pic.stride_cb, pic.stride_cr = w + 2 * pad, w // 2 + pad, w // 2 + pad
        pic.width, pic.height, pic.origin_x, pic.origin_y, pic.sample_bytes = w, h, pad, pad, es
        plans = [sharded.recon_exchange_plan(pic, world, r) for r in range(world)]
        for a in range(world):
            for b in range(world):
                if a == b:
                    continue
                sends = [(x.plane, x.offset, x.bytes) for x in plans[a] if x.send and x.peer == b]
                recvs = [(x.plane, x.offset, x.bytes) for x in plans[b] if not x.send and x.peer == a]
                assert sends == recvs and len(sends) == (3 if sharded.recon_slab_rows(h, world, a)[1] else 0)
        for r in range(world):
            for pl, (st, hh, pp) in enumerate([(pic.stride_y, h, pad), (pic.stride_cb, h // 2, pad // 2), (pic.stride_cr, h // 2, pad // 2)]):
                rows = sorted((x.offset // (st * es), x.bytes // (st * es)) for x in plans[r] if not x.send and x.plane == pl)
                y0, n = sharded.recon_slab_rows(h, world, r)
                rows = sorted(rows + [(pp + (y0 >> (pl > 0)), ((y0 + n) >> (pl > 0)) - (y0 >> (pl > 0)))])
                assert rows[0][0] == pp and sum(n_ for _, n_ in rows) == hh
                assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
        n_sb = ((w + 63) // 64) * ((h + 63) // 64)
        gp = [sharded.me_gather_plan(n_sb, 3, 85 * 24, world, r) for r in range(world)]
        for a in range(world):
            for b in range(world):
                if a != b:
                    assert [(x.plane, x.bytes) for x in gp[a] if x.send and x.peer == b] == [(x.plane, x.bytes) for x in gp[b] if not x.send and x.peer == a]


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


def test_single_process_needs_no_exchange():
    torch = pytest.importorskip("torch")
    # world 1 never calls a collective
    sme = sharded.ShardedMotionEstimation(256, 128)
    assert sme.world == 1 and sme.count == 8 and sme.first == 0
    out = sme.run(lambda sb, n: torch.zeros((1, n, 85, 24), dtype=torch.uint8))
    assert out.shape == (1, 8, 85, 24)
