#!/usr/bin/env python3
"""bench.py -- 64x64 full-pel SAD-search throughput (BASELINE.json metric) on N MI355X GPUs.

One "step" = one pass of the hot path over one batch: PICTURES_PER_STEP synthetic 1080p pictures
(510 superblocks each, one reference list) searched for all 85 square PUs, i.e. 510*P "blocks".
All planes, descriptors and result buffers are resident in HBM before the timed region.

Launch: `python bench.py --gpus 1` or, for N>1,
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU).
ME is open-loop (source pictures are the references), so ranks shard pictures with NO data-path
collective; torch.distributed is used only for the barrier and the max-over-ranks of the time.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))

import numpy as np  # noqa: E402

W, H = 1920, 1080
PICTURES_PER_STEP = 12
SEARCH_W = SEARCH_H = 64
ALGO_BYTES_PER_BLOCK = 4096 + 127 * 127 + 680  # src + ref window + results = 20905 B
ABSDIFF_PER_BLOCK = SEARCH_W * SEARCH_H * 2048  # 64 8x8 SADs x 32 abs-diffs per position
HBM_PEAK_GBPS = 8000.0
# measured packed-SAD issue ceiling (tools/ubench_valu.hip, profiles/ubench_valu_r01.txt):
# v_qsad_pk_u16_u8 = 16 abs-diff/lane per 16.15 SIMD-cycles, independent of occupancy (profiles/r01_ubench_occupancy.txt)
VALU_PEAK_ABSDIFF_PER_S = 256 * 4 * 64 * 16 / 16.15 * 2.4e9


def build_pool(n_pictures, rank_seed):
    """A picture pool: n_pictures+1 synthetic 1080p pictures (padded full-res + 1/4 + 1/16 planes each) stacked
    in one buffer; picture i+1 is searched in picture i.  Also returns zero-centred full-pel descriptors (used
    by the CPU baseline and by --no-hme)."""
    import svtav1_hip
    from svtav1_hip import synth

    pics = [synth.PaPicture(synth.synth_luma(W, H, t, seed=synth.SEED + 1000 * rank_seed)) for t in range(n_pictures + 1)]
    pool, pdesc = svtav1_hip.build_picture_pool(pics)
    descs = []
    for i in range(n_pictures):
        d = svtav1_hip.make_fullpel_desc(pics[i + 1], pics[i], None, SEARCH_W, SEARCH_H).astype(np.int64)
        d[:, 0] += pdesc[i + 1].full_offset  # current picture
        d[:, 1] += pdesc[i].full_offset  # its reference
        descs.append(d)
    desc = np.concatenate(descs).astype(np.int32)
    return pool, pics[0].stride, desc, pdesc


def cpu_baseline(pool, stride, desc, seconds=12.0):
    """Reference AVX2 kernels (oracle/_ref, timing baseline) driven like FullPelSearch_LCU on the host
    cores, one thread per core over disjoint SB ranges; falls back to the repo's C port."""
    from oracle.binding import Oracle, Reference

    pool2d = pool[: (pool.size // stride) * stride].reshape(-1, stride)   # flat pool viewed with the full-plane stride
    # the GPU box gives one GPU job a 16-CPU share; never start more workers than that
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))
    sample = desc[:510]
    if Reference.available():
        ref = Reference()
        kind = "reference"

        def run(chunk):
            ref.fullpel_search_batch(pool2d, pool2d, chunk, asm_type=1)
    else:
        orc = Oracle()
        kind = "port"

        def run(chunk):
            orc.fullpel_search_batch(pool2d, pool2d, chunk)
    # single thread
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < seconds * 0.3:
        run(sample[:128])
        n1 += 128
    single = n1 / (time.perf_counter() - t0)
    # all cores
    chunks = np.array_split(sample, ncores)
    done = [0] * ncores
    stop = time.perf_counter() + seconds * 0.7

    def worker(i):
        while time.perf_counter() < stop:
            run(chunks[i])
            done[i] += len(chunks[i])

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
    [t.start() for t in th]
    [t.join() for t in th]
    multi = sum(done) / (time.perf_counter() - t0)
    return {"value": round(multi, 1), "unit": "blocks/s", "cores": ncores, "kind": kind,
            "single_thread_value": round(single, 1),
            "sample": f"picture 0 (510 SBs, 64x64 search, 85 PUs) repeated for ~{seconds:.0f} s; "
                      f"{'reference ASM_AVX2 kernels driven like FullPelSearch_LCU' if kind == 'reference' else 'repo C port (oracle)'}; "
                      f"{ncores} threads over disjoint SB ranges"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", action="store_true", help="one fused search-centre + full-pel kernel instead of the two launches")
    ap.add_argument("--no-hme", action="store_true", help="time the full-pel search alone (zero-centred windows)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import svtav1_hip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    pool, stride, desc, pdesc = build_pool(PICTURES_PER_STEP, rank)
    n_blocks = desc.shape[0]
    n_sb = n_blocks // PICTURES_PER_STEP
    d_pool = torch.from_numpy(pool).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)  # zero-centred windows; overwritten by the HME kernel each step
    d_sad = torch.empty((n_blocks, 85), dtype=torch.int32, device=dev)
    d_mv = torch.empty((n_blocks, 85), dtype=torch.int32, device=dev)
    d_sb = torch.from_numpy(svtav1_hip.sb_origins(W, H).view(np.int16).copy()).to(dev)
    params = svtav1_hip.default_me_params(W, H, 3, 0)
    ctx = svtav1_hip.Context(local_rank)
    a = (d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n_blocks, SEARCH_W, SEARCH_H,
         d_sad.data_ptr(), d_mv.data_ptr())
    curs = [pdesc[i + 1] for i in range(PICTURES_PER_STEP)]
    refs = [pdesc[i] for i in range(PICTURES_PER_STEP)]

    def step():
        # hierarchical ME of every picture of the batch: ONE search-centre launch over the 510*P superblocks (its
        # descriptors land in d_desc), then ONE full-pel launch over the same superblocks, both on the context's stream
        if args.no_hme:
            ctx.fullpel_search_dev(*a)
        elif not args.fused:
            ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, refs, params, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr())
            ctx.fullpel_search_dev(*a)
        else:
            ctx.integer_search_batch_dev(d_pool.data_ptr(), curs, refs, params, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr(),
                                         d_sad.data_ptr(), d_mv.data_ptr())

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if distributed:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if distributed:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration, HIP events on the stream the kernel runs on
    kern_ms = ctx.fullpel_search_time_dev(*a, max(5, min(args.steps, 20)))

    # HBM traffic of the dominant kernel: measured offline with rocprofv3 --pmc (counters cannot be read from inside the
    # process); tools/run_pmc_traffic.sh -> profiles/r01_pmc_traffic.json, KB per launch of this same workload
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)["svthip::fullpel85_kernel"]
        # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM counters;
        # confirmed on the transform kernels, where 2 x FETCH_SIZE matches the algorithmic bytes: profiles/r01_pmc_traffic_tq.json)
        traffic = round((2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0)
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        total_blocks = n_blocks * world * args.steps
        value = total_blocks / elapsed
        achieved = ALGO_BYTES_PER_BLOCK * n_blocks / (kern_ms * 1e-3) / 1e9
        absdiff_rate = ABSDIFF_PER_BLOCK * n_blocks / (kern_ms * 1e-3)
        out = {
            "metric": "64x64 SAD-search blocks/sec",
            "value": round(value, 1),
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 64x64 full-pel SAD hierarchical ME (centre check + HME L0/L1/L2 + 64x64-area "
                                   "full-pel search of 85 square PUs), 1080p 8-bit, one reference list"
                       if not args.no_hme else "64x64 full-pel SAD search only (no HME), 85 square PUs, 1080p 8-bit",
                       "pictures_per_step_per_gpu": PICTURES_PER_STEP, "blocks_per_step_per_gpu": n_blocks,
                       "search_area": [SEARCH_W, SEARCH_H], "sharding": "pictures across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "traffic_note": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 from separate rocprofv3 --pmc passes "
                                         "(profiles/r01_pmc_traffic.json; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950); "
                                         "algorithmic bytes per launch = 20905 x blocks; the excess is 64/128-byte line granularity on "
                                         "unaligned 127-byte window rows, irrelevant at 3 % of the HBM peak (VALU-bound)",
                         "kernel": "fullpel85_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_block": ALGO_BYTES_PER_BLOCK,
                         "note": "search is ~400 abs-diff per compulsory byte: VALU-bound by construction (SURVEY 8d); see valu",
                         "valu": {"achieved_absdiff_per_s": round(absdiff_rate, 0),
                                  "peak_absdiff_per_s": round(VALU_PEAK_ABSDIFF_PER_S, 0),
                                  "frac": round(absdiff_rate / VALU_PEAK_ABSDIFF_PER_S, 4),
                                  "peak_source": "measured v_qsad_pk_u16_u8 issue rate, tools/ubench_valu.hip"}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pool, stride, desc)
        print(json.dumps(out), flush=True)
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
