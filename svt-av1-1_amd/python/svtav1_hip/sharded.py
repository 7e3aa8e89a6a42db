"""Multi-GPU host layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), SURVEY 8(e).

What shards and what is exchanged:
  * Open-loop ME of a picture: every superblock is independent given the read-only source planes, so every rank holds the planes
    and searches a contiguous range of the picture's superblocks.  NO data-path collective is needed; `gather_rows` exists for a
    consumer that wants all of me_results on every rank (one all_gather_into_tensor of 24 B per PU).
  * Transform / quantisation / reconstruction: TUs are independent, ranks take contiguous SB-row slabs of the picture.  The
    reconstructed slabs ARE exchanged -- the next picture's inter prediction reads the whole reference picture -- with ONE
    all_gather_into_tensor of the luma slabs followed by the border padding of PadRefAndSetFlags
    (Source/Lib/Codec/EbEncDecProcess.c:1135-1204) done redundantly on every rank (`ReconExchange`).

Everything here is backend-agnostic (it takes a process group): the GPU path runs it over RCCL with device tensors, the CPU tests run
the same partition / gather / reassembly code over gloo.  PyTorch is used for device memory and the collectives only.
"""
from __future__ import annotations

import numpy as np

from . import sb_origins


def sb_grid(width: int, height: int):
    return (width + 63) // 64, (height + 63) // 64


def shard_sb_range(width: int, height: int, world: int, rank: int, granularity: str = "sb"):
    """Contiguous share of a picture's superblocks (raster order) for `rank`: (first, count).

    granularity "row": whole SB rows (what a reconstructed-picture slab needs; 1080p = 17 rows -> 3/2/2/2/2/2/2/2 over 8 ranks);
    granularity "sb" : balanced to one superblock (1080p = 510 SBs -> 64/64/64/64/64/64/63/63), which is what the ME stage uses --
    its superblocks are independent, so there is no reason to pay the 3-vs-2-rows imbalance."""
    nx, ny = sb_grid(width, height)
    units, per = (ny, nx) if granularity == "row" else (nx * ny, 1)
    base, extra = divmod(units, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first * per, count * per


def shard_sb_indices(width: int, height: int, world: int, rank: int, granularity: str = "sb") -> np.ndarray:
    first, count = shard_sb_range(width, height, world, rank, granularity)
    return np.arange(first, first + count, dtype=np.int64)


def gather_rows(local, counts, group=None):
    """All-gather of per-rank row blocks of unequal length with ONE collective: every rank contributes a [max(counts), ...] block
    (its rows first, the rest untouched padding), the result is the [sum(counts), ...] concatenation in rank order.
    `local` is a torch tensor [counts[rank], ...] (device tensor under RCCL, CPU tensor under gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and local.shape[0] == counts[rank]
    mx = max(counts)
    tail = tuple(local.shape[1:])
    send = local if counts[rank] == mx else torch.cat([local, local.new_zeros((mx - counts[rank],) + tail)])
    recv = local.new_empty((world * mx,) + tail)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(c == mx for c in counts):
        return recv
    return torch.cat([recv[r * mx:r * mx + counts[r]] for r in range(world)])


class ShardedMotionEstimation:
    """Frame-sharded open-loop ME: this rank's contiguous superblock range of every picture of a batch.

    compute(sb_xy[n_local, 2] uint16 ndarray, n_local) -> tensor [n_jobs, n_local, n_pu, 24] uint8 (svthip_me_cu_result records)
    is the device call (svthip_motion_estimate_batch_dev on the shard's SB list) -- or, in the CPU tests, the oracle chain."""

    def __init__(self, width: int, height: int, group=None, granularity: str = "sb"):
        import torch.distributed as dist

        self.width, self.height, self.group = width, height, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.counts = [shard_sb_range(width, height, self.world, r, granularity)[1] for r in range(self.world)]
        self.first, self.count = shard_sb_range(width, height, self.world, self.rank, granularity)
        self.sb_xy = np.ascontiguousarray(sb_origins(width, height)[self.first:self.first + self.count])

    def run(self, compute, gather: bool = True):
        local = compute(self.sb_xy, self.count)  # [n_jobs, n_local, n_pu, 24]
        if not gather or self.world == 1:
            return local
        # gather along the SB axis: move it to the front, one collective for the whole batch
        moved = local.permute(1, 0, 2, 3).contiguous()
        full = gather_rows(moved, self.counts, self.group)
        return full.permute(1, 0, 2, 3).contiguous()


class ReconExchange:
    """Reassembles a reconstructed reference picture from per-rank SB-row slabs and pads its borders.

    Every rank owns a full padded plane buffer [height + 2 pad, stride] (device tensor, uint8 or uint16) and has written ITS slab
    (rows of its SB rows) into it.  exchange() = one all_gather_into_tensor of the slabs (staged to the size of the largest slab),
    the copy of the other ranks' slabs into place, then generate_padding on the whole picture (`pad_fn`: the HIP
    svthip_pad_plane_dev on the GPU path; the tests pass the oracle)."""

    def __init__(self, width: int, height: int, pad: int, group=None):
        import torch.distributed as dist

        self.width, self.height, self.pad, self.group = width, height, pad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        nx, _ = sb_grid(width, height)
        self.rows = []  # (first luma row, row count) per rank
        for r in range(self.world):
            first, count = shard_sb_range(width, height, self.world, r, "row")
            y0 = (first // nx) * 64
            y1 = min(height, (first // nx + count // nx) * 64)
            self.rows.append((y0, max(0, y1 - y0)))
        self.my_rows = self.rows[self.rank]

    def exchange(self, plane, pad_fn):
        """plane: [height + 2 pad, stride] tensor holding this rank's rows; returns it complete and padded (in place)."""
        import torch.distributed as dist

        if self.world > 1:
            p = self.pad
            mx = max(n for _, n in self.rows)
            stride = plane.shape[1]
            y0, n = self.my_rows
            send = plane.new_empty((mx, stride))
            send[:n] = plane[p + y0:p + y0 + n]
            recv = plane.new_empty((self.world * mx, stride))
            dist.all_gather_into_tensor(recv, send, group=self.group)
            for r, (ry, rn) in enumerate(self.rows):
                if r != self.rank and rn:
                    plane[p + ry:p + ry + rn] = recv[r * mx:r * mx + rn]
        pad_fn(plane)
        return plane


def device_me_compute(ctx, d_pool_ptr: int, curs, refs0, refs1, params, n_pu: int = 85, use_subpel: bool = True, cu8x8_mode: int = 0,
                      device="cuda:0"):
    """compute() for ShardedMotionEstimation on the HIP path: svthip_motion_estimate[209]_batch_dev on the shard's SB list."""
    import torch

    def compute(sb_xy, n_local):
        n_jobs = len(curs)
        out = torch.zeros((n_jobs, n_local, n_pu, 24), dtype=torch.uint8, device=device)
        if n_local == 0:
            return out
        d_sb = torch.from_numpy(sb_xy.view(np.int16).copy()).to(device)
        fn = ctx.motion_estimate209_batch_dev if n_pu == 209 else ctx.motion_estimate_batch_dev
        fn(d_pool_ptr, curs, refs0, refs1, params, d_sb.data_ptr(), n_local, out.data_ptr(), use_subpel, cu8x8_mode)
        ctx.synchronize()
        return out

    return compute
