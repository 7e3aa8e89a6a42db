"""Multi-GPU host layer: a thin Python driver over the C ABI's multi-GPU entries (include/svtav1_hip.h "Multi-GPU",
svt-av1-1_amd/csrc/svthip_comm.hip), one process per GPU.

What shards and what is exchanged:
  * Open-loop ME of a picture: every superblock is independent given the read-only source planes, so every rank holds the planes
    and searches a contiguous range of the picture's superblocks (svthip_shard_range, balanced to one SB).  NO data-path collective
    is needed; `ShardedMotionEstimation.run(gather=True)` exists for a consumer that wants all of me_results on every rank.
  * Transform / quantisation / reconstruction: TUs are independent, ranks take contiguous SB-row slabs of the picture
    (svthip_recon_slab_rows).  The reconstructed slabs ARE exchanged -- the next picture's inter prediction reads the whole reference
    picture -- and the borders of Y, Cb, Cr are padded as PadRefAndSetFlags does (Source/Lib/Codec/EbEncDecProcess.c:1135-1204).

Partition and transfer lists come from the C library (pure host functions, no device needed).  Two transports execute them:
  * `Comm` (svthip_comm over RCCL / xGMI) -> svthip_recon_exchange_dev / svthip_me_gather_results_dev: the product path, device buffers;
  * a torch.distributed process group (gloo) executing the SAME svthip_*_plan lists on CPU tensors: what the CPU tests run, so the
    partition, offsets and byte counts are covered without a multi-GPU node.
PyTorch is used for device memory, the launcher's rendezvous and the timing barrier only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib, sb_origins, _check


class ReconPicture(C.Structure):
    """svthip_recon_picture"""
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("stride_y", C.c_uint32), ("stride_cb", C.c_uint32),
                ("stride_cr", C.c_uint32), ("width", C.c_uint16), ("height", C.c_uint16), ("origin_x", C.c_uint16), ("origin_y", C.c_uint16),
                ("sample_bytes", C.c_uint8), ("reserved", C.c_uint8 * 3)]


class Xfer(C.Structure):
    """svthip_xfer"""
    _fields_ = [("peer", C.c_int32), ("plane", C.c_uint32), ("send", C.c_uint32), ("offset", C.c_uint64), ("bytes", C.c_uint64)]


def _bind():
    L = lib()
    if getattr(L, "_sharded_bound", False):
        return L
    u32p = C.POINTER(C.c_uint32)
    L.svthip_shard_range.restype = None
    L.svthip_shard_range.argtypes = [C.c_uint32, C.c_int32, C.c_int32, u32p, u32p]
    L.svthip_recon_slab_rows.restype = None
    L.svthip_recon_slab_rows.argtypes = [C.c_uint32, C.c_int32, C.c_int32, u32p, u32p]
    L.svthip_recon_exchange_plan.restype = C.c_int32
    L.svthip_recon_exchange_plan.argtypes = [C.POINTER(ReconPicture), C.c_int32, C.c_int32, C.POINTER(Xfer), C.c_uint32, u32p]
    L.svthip_me_gather_plan.restype = C.c_int32
    L.svthip_me_gather_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Xfer), C.c_uint32, u32p]
    L.svthip_comm_get_unique_id.restype = C.c_int32
    L.svthip_comm_get_unique_id.argtypes = [C.c_void_p]
    L.svthip_comm_create.restype = C.c_int32
    L.svthip_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.svthip_comm_destroy.restype = None
    L.svthip_comm_destroy.argtypes = [C.c_void_p]
    L.svthip_comm_last_error.restype = C.c_char_p
    L.svthip_recon_exchange_dev.restype = C.c_int32
    L.svthip_recon_exchange_dev.argtypes = [C.c_void_p, C.POINTER(ReconPicture), C.c_void_p]
    L.svthip_me_gather_results_dev.restype = C.c_int32
    L.svthip_me_gather_results_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L._sharded_bound = True
    return L


def _ccheck(rc: int):
    if rc != 0:
        raise RuntimeError(f"svthip multi-GPU error 0x{rc & 0xFFFFFFFF:08x}: {_bind().svthip_comm_last_error().decode()}")


def sb_grid(width: int, height: int):
    return (width + 63) // 64, (height + 63) // 64


def shard_range(n_units: int, world: int, rank: int):
    """svthip_shard_range: contiguous share of n_units, balanced to one unit -> (first, count)."""
    f, c = C.c_uint32(0), C.c_uint32(0)
    _bind().svthip_shard_range(n_units, world, rank, C.byref(f), C.byref(c))
    return f.value, c.value


def shard_sb_range(width: int, height: int, world: int, rank: int, granularity: str = "sb"):
    """Contiguous share of a picture's superblocks (raster order) for `rank`: (first, count).

    granularity "sb" : balanced to one superblock (1080p = 510 SBs -> 64/64/64/64/64/64/63/63) -- what the ME stage uses: its
    superblocks are independent, so there is no reason to pay a whole-row imbalance;
    granularity "row": whole SB rows (what a reconstructed-picture slab is; 1080p = 17 rows -> 3/2/2/2/2/2/2/2 over 8 ranks)."""
    nx, ny = sb_grid(width, height)
    if granularity == "row":
        f, c = shard_range(ny, world, rank)
        return f * nx, c * nx
    return shard_range(nx * ny, world, rank)


def shard_sb_indices(width: int, height: int, world: int, rank: int, granularity: str = "sb") -> np.ndarray:
    first, count = shard_sb_range(width, height, world, rank, granularity)
    return np.arange(first, first + count, dtype=np.int64)


def recon_slab_rows(height: int, world: int, rank: int):
    """svthip_recon_slab_rows: luma rows (first, count) of rank's SB-row slab."""
    f, c = C.c_uint32(0), C.c_uint32(0)
    _bind().svthip_recon_slab_rows(height, world, rank, C.byref(f), C.byref(c))
    return f.value, c.value


def recon_exchange_plan(pic: ReconPicture, world: int, rank: int):
    n = C.c_uint32(0)
    _ccheck(_bind().svthip_recon_exchange_plan(C.byref(pic), world, rank, None, 0, C.byref(n)))
    plan = (Xfer * max(1, n.value))()
    _ccheck(_bind().svthip_recon_exchange_plan(C.byref(pic), world, rank, plan, n.value, C.byref(n)))
    return [plan[i] for i in range(n.value)]


def me_gather_plan(n_sb_total: int, n_jobs: int, record_bytes: int, world: int, rank: int):
    n = C.c_uint32(0)
    _ccheck(_bind().svthip_me_gather_plan(n_sb_total, n_jobs, record_bytes, world, rank, None, 0, C.byref(n)))
    plan = (Xfer * max(1, n.value))()
    _ccheck(_bind().svthip_me_gather_plan(n_sb_total, n_jobs, record_bytes, world, rank, plan, n.value, C.byref(n)))
    return [plan[i] for i in range(n.value)]


def _run_plan_gloo(plan, send_bufs, recv_bufs, group=None):
    """Execute a svthip_xfer list over a torch.distributed group on flat uint8 CPU tensors (one per `plane`): the CPU-test transport."""
    import torch.distributed as dist

    ops = []
    for x in plan:
        if x.send:
            ops.append(dist.P2POp(dist.isend, send_bufs[x.plane][x.offset:x.offset + x.bytes], x.peer, group))
        else:
            ops.append(dist.P2POp(dist.irecv, recv_bufs[x.plane][x.offset:x.offset + x.bytes], x.peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


class Comm:
    """svthip_comm of this rank (RCCL).  With torch.distributed initialised the unique id travels through the launcher's store."""

    def __init__(self, ctx, rank: int = 0, world: int = 1, unique_id: bytes | None = None):
        L = _bind()
        self.ctx, self.rank, self.world = ctx, rank, world
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        _ccheck(L.svthip_comm_create(ctx._h, buf, rank, world, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _ccheck(_bind().svthip_comm_get_unique_id(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, ctx):
        """One communicator per rank of the default torch.distributed group; rank 0's id goes through the rendezvous store."""
        import torch.distributed as dist

        if not dist.is_initialized() or dist.get_world_size() == 1:
            return cls(ctx)
        rank, world = dist.get_rank(), dist.get_world_size()
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("svthip_comm_id", cls.unique_id())
        return cls(ctx, rank, world, bytes(store.get("svthip_comm_id")))

    def close(self):
        if self._h:
            _bind().svthip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def recon_exchange_dev(self, pic: ReconPicture, stream=None):
        _ccheck(_bind().svthip_recon_exchange_dev(self._h, C.byref(pic), stream))

    def me_gather_results_dev(self, d_local: int, d_full: int, n_jobs: int, n_sb_total: int, record_bytes: int, stream=None):
        _ccheck(_bind().svthip_me_gather_results_dev(self._h, d_local, d_full, n_jobs, n_sb_total, record_bytes, stream))


class ShardedMotionEstimation:
    """Frame-sharded open-loop ME: this rank's contiguous superblock range of every picture of a batch.

    compute(sb_xy[n_local, 2] uint16 ndarray, n_local) -> tensor [n_jobs, n_local, n_pu, 24] uint8 (svthip_me_cu_result records)
    is the device call (svthip_motion_estimate_batch_dev on the shard's SB list) -- or, in the CPU tests, the oracle chain.
    Transport of the optional gather: `comm` (RCCL, device tensors) or the torch.distributed group (gloo, CPU tensors)."""

    def __init__(self, width: int, height: int, group=None, granularity: str = "sb", comm: Comm | None = None):
        import torch.distributed as dist

        self.width, self.height, self.group, self.comm = width, height, group, comm
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if granularity != "sb" and self.world > 1:
            raise ValueError("the gather runs on the one-superblock partition (svthip_shard_range)")
        self.n_total = sb_grid(width, height)[0] * sb_grid(width, height)[1]
        self.first, self.count = shard_sb_range(width, height, self.world, self.rank, "sb")
        self.sb_xy = np.ascontiguousarray(sb_origins(width, height)[self.first:self.first + self.count])

    def run(self, compute, gather: bool = True):
        local = compute(self.sb_xy, self.count).contiguous()  # [n_jobs, n_local, n_pu, 24]
        if not gather or self.world == 1:
            return local
        n_jobs, _, n_pu, rec = local.shape
        record_bytes = n_pu * rec
        full = local.new_zeros((n_jobs, self.n_total, n_pu, rec))
        if self.comm is not None:
            self.comm.me_gather_results_dev(local.data_ptr(), full.data_ptr(), n_jobs, self.n_total, record_bytes)
            self.comm.ctx.synchronize()  # the gather ran on the context's stream; `local` may be released after this
            return full
        full[:, self.first:self.first + self.count] = local
        plan = me_gather_plan(self.n_total, n_jobs, record_bytes, self.world, self.rank)
        _run_plan_gloo(plan, [local.view(-1)] * n_jobs, [full.view(-1)] * n_jobs, self.group)
        return full


class ReconExchange:
    """Reassembles a reconstructed reference picture from per-rank SB-row slabs and pads its borders (Y, and Cb / Cr when given).

    Every rank owns full padded plane buffers and has reconstructed ITS slab (`my_rows`; chroma rows >> 1) in place.
      * `comm` given (RCCL): exchange(planes) = svthip_recon_exchange_dev on the device planes -- one group of direct sends / receives
        into the padded planes, then the three paddings; stream-ordered, nothing staged;
      * no `comm` (CPU tests, gloo): the same transfer list (svthip_recon_exchange_plan) over torch.distributed, then `pad_fn(plane_index,
        tensor)` for every plane (the tests pass the oracle's generate_padding)."""

    def __init__(self, width: int, height: int, pad: int, group=None, comm: Comm | None = None, sample_bytes: int = 1):
        import torch.distributed as dist

        self.width, self.height, self.pad, self.group, self.comm, self.sample_bytes = width, height, pad, group, comm, sample_bytes
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows = [recon_slab_rows(height, self.world, r) for r in range(self.world)]  # (first luma row, row count) per rank
        self.my_rows = self.rows[self.rank]

    def picture(self, y_ptr, stride_y, cb_ptr=None, cr_ptr=None, stride_c=0) -> ReconPicture:
        p = ReconPicture()
        p.y, p.cb, p.cr = y_ptr, cb_ptr, cr_ptr
        p.stride_y, p.stride_cb, p.stride_cr = stride_y, stride_c, stride_c
        p.width, p.height, p.origin_x, p.origin_y, p.sample_bytes = self.width, self.height, self.pad, self.pad, self.sample_bytes
        return p

    def exchange(self, planes, pad_fn=None, stream=None):
        """planes: [Y] or [Y, Cb, Cr] 2-D tensors [(rows + 2 pad), stride * sample_bytes] of bytes, each holding this rank's slab."""
        y = planes[0]
        es = self.sample_bytes
        cb = planes[1] if len(planes) == 3 else None
        cr = planes[2] if len(planes) == 3 else None
        pic = self.picture(y.data_ptr(), y.shape[1] // es, cb.data_ptr() if cb is not None else None, cr.data_ptr() if cr is not None else None,
                           cb.shape[1] // es if cb is not None else 0)
        if self.comm is not None:
            self.comm.recon_exchange_dev(pic, stream)
            return planes
        if self.world > 1:
            flat = [p.view(-1) for p in planes]
            _run_plan_gloo(recon_exchange_plan(pic, self.world, self.rank), flat, flat, self.group)
        for i, p in enumerate(planes):
            pad_fn(i, p)
        return planes


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
