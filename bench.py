#!/usr/bin/env python3
"""bench.py -- 64x64 full-pel SAD-search throughput (BASELINE.json metric) on N MI355X GPUs, plus one leg per BASELINE config.

Headline (configs[1]).  One "step" = hierarchical ME (search-centre chain + 64x64-area full-pel search of the 85 square PUs, one
reference list) of a batch of synthetic 1080p pictures whose planes live in HBM.  Frame sharding: the batch is
12 x N pictures and every rank searches ITS contiguous superblock range of EVERY picture (510 SBs -> 510/N per rank, balanced to
one SB), so per-GPU work is fixed as N grows ("weak") and no data-path collective is needed -- ME is open-loop, the references are
source pictures every rank already holds.  `value` = blocks searched by all ranks / max-over-ranks time.

Launch: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment the script starts the N
ranks itself (fresh children through torch.distributed.run, before this process touches a GPU); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.  RCCL (backend "nccl") carries
the barrier, the max-over-ranks of the time, the optional me_results all-gather and the reconstructed-picture all-gather leg.

Extra legs (same JSON line, under "legs"; rank 0 unless stated): whole-picture ME with sub-pel + bi-prediction (configs[2]),
transform/quantisation chains at the SURVEY 8d shapes with the reference's real quantiser rows (configs[3]), the 16x16 +-16
SadLoopKernel workload at 856x480 (configs[0]), a 3840x2160 / 10-bit pass (configs[4]) and the recon exchange (all ranks).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))

import numpy as np  # noqa: E402

W, H = 1920, 1080
PICTURES_PER_STEP = 12          # per GPU
SEARCH_W = SEARCH_H = 64
ALGO_BYTES_PER_BLOCK = 4096 + 127 * 127 + 680  # src + ref window + results = 20905 B
ABSDIFF_PER_BLOCK = SEARCH_W * SEARCH_H * 2048  # 64 8x8 SADs x 32 abs-diffs per position
HBM_PEAK_GBPS = 8000.0
# measured packed-SAD issue ceiling (tools/ubench_valu.hip, profiles/r01_ubench_valu.txt):
# v_qsad_pk_u16_u8 = 16 abs-diff/lane per 16.15 SIMD-cycles, independent of occupancy (profiles/r01_ubench_occupancy.txt)
VALU_PEAK_ABSDIFF_PER_S = 256 * 4 * 64 * 16 / 16.15 * 2.4e9
SEED = 20261004


# ----------------------------------------------------------------------------------------------------------------------
# synthetic pictures generated ON the device (SURVEY 8d formula), planes derived by the picture-analysis kernel
# ----------------------------------------------------------------------------------------------------------------------
def device_picture_pool(ctx, n_pictures, width, height, dev, seed=SEED):
    """n_pictures synthetic pictures in one device pool: the luma interiors are computed with torch on the GPU, the 68-sample
    borders and the 1/4 and 1/16 planes by svthip_pa_derive_planes_dev (one upload-free pass).  Returns (pool tensor, [PaPictureDesc])."""
    import torch

    import svtav1_hip

    fs, qs, ss = width + 136, (width >> 1) + 64, (width >> 2) + 32
    fb, qb, sb = fs * (height + 136), qs * ((height >> 1) + 64), ss * ((height >> 2) + 32)
    al = lambda v: (v + 15) & ~15  # noqa: E731
    per = al(fb) + al(qb) + al(sb)
    pool = torch.zeros(per * n_pictures + 256, dtype=torch.uint8, device=dev)
    descs = []
    xx = torch.arange(width, dtype=torch.int64, device=dev)[None, :]
    yy = torch.arange(height, dtype=torch.int64, device=dev)[:, None]
    for t in range(n_pictures):
        d = svtav1_hip.PaPictureDesc()
        d.full_offset, d.quarter_offset, d.sixteenth_offset = per * t, per * t + al(fb), per * t + al(fb) + al(qb)
        d.full_stride, d.quarter_stride, d.sixteenth_stride = fs, qs, ss
        d.width, d.height = width, height
        descs.append(d)
        x, y = xx + 3 * t, yy + 2 * t
        smooth = 48.0 * torch.sin(2 * np.pi * x.double() / 97.0) * torch.cos(2 * np.pi * y.double() / 61.0)
        quad = ((x * x + 3 * y * y) >> 9) & 63
        lcg = (1103515245 * (xx * 7919 + yy * 104729 + seed + 977 * t) + 12345) & 0xFFFFFFFF
        v = torch.floor(96.0 + smooth).long() + quad + ((lcg >> 16) & 31) - 16
        plane = pool[per * t: per * t + fb].view(height + 136, fs)
        plane[68:68 + height, 68:68 + width] = v.clamp_(0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    ctx.pa_derive_planes_dev(pool.data_ptr(), descs)
    ctx.synchronize()
    return pool, descs


def zero_centred_desc(descs_cur, descs_ref, sb_xy, width, height):
    """svthip_fullpel_desc rows (job-major) with zero search centres -- only for --no-hme and the CPU baseline."""
    from svtav1_hip import synth

    out = np.zeros((len(descs_cur), sb_xy.shape[0], 6), dtype=np.int64)
    for j, (c, r) in enumerate(zip(descs_cur, descs_ref)):
        for i, (ox, oy) in enumerate(sb_xy):
            xo, yo, sw, sh = synth.clamp_search_window(int(ox), int(oy), 0, 0, SEARCH_W, SEARCH_H, width, height)
            out[j, i] = [c.full_offset + (68 + int(oy)) * c.full_stride + 68 + int(ox),
                         r.full_offset + (68 + int(oy) + yo) * r.full_stride + 68 + int(ox) + xo, xo, yo, sw, sh]
    assert out.max() < 2 ** 31
    return out.reshape(-1, 6).astype(np.int32)


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the REFERENCE's own code (oracle/_ref, compiled from /root/reference in the build container; it
# travels to the GPU box as built libraries) on the host cores.  They run in a CHILD process that the parent starts before it touches a
# GPU (`--cpu-baseline-worker`): the child builds its own synthetic inputs with numpy, waits for one line on stdin, runs a bounded sample
# of every leg (~30 s in total) and prints one JSON line.  Reported baselines, not targets.
# ----------------------------------------------------------------------------------------------------------------------
def _host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # the GPU box gives one GPU job a 16-CPU share


def _threads_rate(run_chunk, chunks, seconds):
    """run_chunk(i) over len(chunks) threads (ctypes releases the GIL) until `seconds` have passed; returns units per second."""
    done = [0] * len(chunks)
    stop = time.perf_counter() + seconds

    def worker(i):
        while time.perf_counter() < stop:
            done[i] += run_chunk(i)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(chunks))]
    [t.start() for t in th]
    [t.join() for t in th]
    return sum(done) / (time.perf_counter() - t0)


def cpu_headline(ncores, seconds=8.0):
    """Reference AVX2 kernels (timing baseline) driven like FullPelSearch_LCU, one thread per core over disjoint SB ranges; falls back
    to the repo's C port when oracle/_ref is absent."""
    import svtav1_hip
    from oracle.binding import Oracle, Reference
    from svtav1_hip import synth

    pics = [synth.PaPicture(synth.synth_luma(W, H, t)) for t in (0, 1)]
    pool, pd = svtav1_hip.build_picture_pool(pics)
    stride = pd[0].full_stride
    pool2d = pool[: (pool.size // stride) * stride].reshape(-1, stride)
    sample = zero_centred_desc([pd[1]], [pd[0]], svtav1_hip.sb_origins(W, H), W, H)
    if Reference.available():
        eng, kind = Reference(), "reference"
        run = lambda chunk: eng.fullpel_search_batch(pool2d, pool2d, chunk, asm_type=1)  # noqa: E731
    else:
        eng, kind = Oracle(), "port"
        run = lambda chunk: eng.fullpel_search_batch(pool2d, pool2d, chunk)  # noqa: E731
    t0, n1 = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds * 0.3:
        run(sample[:128])
        n1 += 128
    single = n1 / (time.perf_counter() - t0)
    chunks = np.array_split(sample, ncores)

    def run_chunk(i):
        run(chunks[i])
        return len(chunks[i])

    multi = _threads_rate(run_chunk, chunks, seconds * 0.7)
    return {"value": round(multi, 1), "unit": "blocks/s", "cores": ncores, "kind": kind, "single_thread_value": round(single, 1),
            "sample": f"picture 1 searched in picture 0 (510 SBs, 64x64 search, 85 PUs) repeated for ~{seconds:.0f} s; "
                      f"{'reference ASM_AVX2 kernels driven like FullPelSearch_LCU' if kind == 'reference' else 'repo C port (oracle)'}; "
                      f"{ncores} threads over disjoint SB ranges"}


def cpu_sad_loop_baseline(seconds=3.0):
    """configs[0] on the host: the reference's C SadLoopKernel (C_DEFAULT/EbComputeSAD_C.c:73-119; oracle/_ref) on 16x16 blocks of an
    856x480 picture, +-16 search (33x33 positions), single thread -- "reference C path on host CPU"."""
    from oracle.binding import Oracle, Reference
    from svtav1_hip import synth

    cur = synth.PaPicture(synth.synth_luma(856, 480, 1))
    ref = synth.PaPicture(synth.synth_luma(856, 480, 0))
    use_ref = Reference.available()
    eng = Reference() if use_ref else Oracle()
    stride = cur.stride
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for by in range(0, 480 - 15, 16):
            for bx in range(0, 848, 16):
                so = (68 + by) * stride + 68 + bx
                ro = (68 + by - 16) * stride + 68 + bx - 16
                if use_ref:
                    eng.sad_loop("SadLoopKernel", cur.full, so, stride, ref.full, ro, stride, 16, 16, stride, 33, 33)
                else:
                    eng.sad_loop(cur.full, so, stride, ref.full, ro, stride, 16, 16, stride, 33, 33)
                n += 1
            if time.perf_counter() - t0 >= seconds:
                break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "16x16 blocks/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"{n} SadLoopKernel calls (16x16 block, 33x33 positions, 856x480 picture) in {dt:.1f} s through ctypes"}


_ME_PICS = None


def _me_one_picture(_):
    from oracle.binding import ReferenceME

    import svtav1_hip
    P = svtav1_hip.default_me_params(W, H, 3, 1)
    ReferenceME().run(_ME_PICS[0], _ME_PICS[1], _ME_PICS[2], P, two_lists=True, hierarchical_levels=3)
    return 1


def cpu_me_chain_baseline(ncores, seconds=6.0):
    """me_chain_subpel.85pu_B_fullpel_only on the host: the reference's own MotionEstimateLcu (sub-pel off -- the sub-pel arm needs a
    NASM-only symbol the image cannot assemble), every SB of a 1080p B picture, one PROCESS per core (the function keeps its
    allocations in process-wide tables), whole pictures per process."""
    global _ME_PICS
    import multiprocessing as mp

    from oracle.binding import ReferenceME
    from svtav1_hip import synth
    if not ReferenceME.available():
        return None
    _ME_PICS = [synth.PaPicture(synth.synth_luma(W, H, t)) for t in (1, 0, 2)]
    with mp.get_context("fork").Pool(ncores) as pool:
        pool.map(_me_one_picture, range(ncores))           # warm: libraries loaded, pages touched
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            n += sum(pool.map(_me_one_picture, range(ncores)))
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "1080p B pictures/s", "superblocks_per_s": round(n * 510 / dt, 1), "cores": ncores, "kind": "reference",
            "sample": f"{n} pictures in {dt:.1f} s: the reference's MotionEstimateLcu (asm_type 0, use_subpel_flag 0, 85 PUs, two lists, HME on) for all "
                      f"510 SBs of a 1080p picture, {ncores} processes x whole pictures"}


def cpu_tq_baseline(ncores, seconds_per_size=1.5):
    """tq_chain.frame_1080p on the host: the reference's C chain per TU (oracle/ref_bench_driver.c: ResidualKernel_c ->
    Av1TransformTwoD_NxN_c -> aom_quantize_b*_c_II -> av1_inv_txfm2d_add_NxN_c), the same 1920x1088 frame of TUs, rows and scans as the
    GPU leg, one thread per core over disjoint TU ranges."""
    import ctypes as C

    import svtav1_hip
    so = os.path.join(ROOT, "oracle", "_ref", "libsvtref_bench.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    f = lib.ref_bench_tq_chain
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
    qrows = np.ascontiguousarray(z["rows_bd8_inter"][[20, 120, 200], 0, :])
    rng = np.random.default_rng(3)
    pic_w, pic_h = 1920, 1088
    src = rng.integers(0, 256, pic_w * pic_h).astype(np.uint8)
    noise = rng.laplace(0, 6, pic_w * pic_h)
    pred = np.clip(src - noise, 0, 255).astype(np.uint8)
    recon = np.zeros_like(pred)
    out = {}
    for n in (4, 8, 16, 32, 64):
        ts = svtav1_hip.TX_SIZES_WH.index((n, n))
        types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(n, n)]
        isc = {t: int(z["scan_offsets"][int(z["scan_index"][ts, t])]) for t in types}
        d, nc = tile_tu_desc(svtav1_hip, n, n, pic_w, pic_h, isc, types, 3, rng)
        scan_ptrs, iscan_ptrs, keep = (C.c_void_p * 16)(), (C.c_void_p * 16)(), []
        for t in types:
            sc, isn = np.ascontiguousarray(z["scan_pool"][isc[t]:isc[t] + nc]), np.ascontiguousarray(z["iscan_pool"][isc[t]:isc[t] + nc])
            keep += [sc, isn]
            scan_ptrs[t], iscan_ptrs[t] = sc.ctypes.data, isn.ctypes.data
        offs = np.ascontiguousarray(d["src_offset"].astype(np.uint32))
        tt = np.ascontiguousarray(d["tx_type"].astype(np.uint8))
        qi = np.ascontiguousarray(d["qparam_index"].astype(np.uint8))
        bounds = np.linspace(0, len(d), ncores + 1).astype(int)

        def run_chunk(i):
            a, b = int(bounds[i]), int(bounds[i + 1])
            f(src.ctypes.data, pred.ctypes.data, recon.ctypes.data, pic_w, offs[a:b].ctypes.data, tt[a:b].ctypes.data, qi[a:b].ctypes.data, b - a, n,
              qrows.ctypes.data, scan_ptrs, iscan_ptrs)
            return (b - a) * n * n

        rate = _threads_rate(run_chunk, list(range(ncores)), seconds_per_size)
        out[f"{n}x{n}"] = {"value": round(rate / 1e6, 2), "unit": "Mpixel/s", "cores": ncores, "kind": "reference",
                           "sample": f"one 1920x1088 frame of {len(d)} {n}x{n} TUs repeated for ~{seconds_per_size} s, reference C chain "
                                     "(ResidualKernel_c, Av1TransformTwoD_c, aom_quantize_b_c_II, av1_inv_txfm2d_add_c)"}
    return out


def cpu_convolve_baseline(ncores, seconds=3.0):
    """convolve_8tap on the host: the reference's av1_convolve_2d_sr_c (and the x / y / copy forms) driven like av1_inter_prediction on 64x64
    blocks at the 225 fractional phases, one thread per core."""
    import ctypes as C

    from svtav1_hip import synth
    so = os.path.join(ROOT, "oracle", "_ref", "libsvtref_bench.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    f = lib.ref_bench_convolve
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
    pic = synth.PaPicture(synth.synth_luma(W, H, 1))
    S = pic.stride
    per = 450                                     # blocks per thread and call: two SB columns x 225 phases
    i = np.arange(per * ncores)
    blk, ph = i // 225, i % 225
    src_off = ((68 + (blk // 30) * 64) * S + 68 + (blk % 30) * 64).astype(np.uint32)
    mode = ((1 + ph % 15) | ((1 + ph // 15) << 4) | (((blk + ph) % 4) << 8) | (((blk // 3 + ph // 5) % 4) << 12)).astype(np.uint16)
    dst = np.zeros((ncores, per * 4096), np.uint8)
    dst_off = (np.arange(per) * 4096).astype(np.uint32)

    def run_chunk(k):
        f(pic.full.ctypes.data, S, dst[k].ctypes.data, 64, src_off[k * per:(k + 1) * per].ctypes.data, dst_off.ctypes.data,
          mode[k * per:(k + 1) * per].ctypes.data, per, 64, 64)
        return per * 4096

    rate = _threads_rate(run_chunk, list(range(ncores)), seconds)
    return {"value": round(rate / 1e6, 2), "unit": "Mpixel/s", "cores": ncores, "kind": "reference",
            "sample": f"64x64 blocks of a 1080p frame at the 15 x 15 fractional phases, {per} blocks per call and thread for ~{seconds} s, reference C "
                      "av1_convolve_2d_sr_c through oracle/ref_convolve_driver.c"}


def cpu_baseline_worker_main():
    """Child process of the N = 1 run (started before the parent initialises a GPU): waits for one line on stdin, then times the
    reference on the host cores and prints one JSON line {leg: cpu_baseline}."""
    if not sys.stdin.readline():
        return 0
    ncores = _host_cores()
    out = {"headline": cpu_headline(ncores)}
    for name, fn in (("sad_loop_480p", cpu_sad_loop_baseline), ("me_chain_subpel.85pu_B_fullpel_only", lambda: cpu_me_chain_baseline(ncores)),
                     ("tq_chain", lambda: cpu_tq_baseline(ncores)), ("convolve_8tap", lambda: cpu_convolve_baseline(ncores))):
        try:
            out[name] = fn()
        except Exception as e:  # a missing reference library must not cost the GPU numbers
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------------------
def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_traffic_bytes(entry):
    """(2 x FETCH_SIZE + WRITE_SIZE) KB -> bytes per launch; FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes."""
    if not entry or "FETCH_SIZE" not in entry or "WRITE_SIZE" not in entry:
        return None
    return round((2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0)


_PMC = None


def pmc_entry(kernel, which=-1, workgroup=None, grid=None):
    """Counter record of one kernel from profiles/r03_pmc_traffic.json (tools/run_r03_pmc_traffic.sh: separate rocprofv3 --pmc passes over
    this very bench command).  Records of a kernel are sorted by grid size: which = -1 takes the largest launch, 0 the smallest;
    `workgroup` filters on the workgroup size.  None when the profile has no such kernel."""
    global _PMC
    if _PMC is None:
        _PMC = load_profile_json("r03_pmc_traffic.json") or {}
    recs = _PMC.get(kernel)
    if not recs:   # rocprofv3 prints namespaces and template arguments: take the first kernel whose name contains the given text
        recs = next((v for k, v in sorted(_PMC.items()) if kernel in k), None)
    if not recs:
        return None
    if workgroup is not None:
        recs = [r for r in recs if r.get("workgroup") == workgroup]
    if grid is not None:   # the launch with exactly this many threads
        recs = [r for r in recs if r.get("grid") == grid]
    if which == "longest":   # the launch shape that kept the GPU busy longest (the persistent kernels launch one grid for every large batch)
        return max(recs, key=lambda r: r.get("GRBM_GUI_ACTIVE", 0.0)) if recs else None
    return recs[which] if recs else None


def counters(entry, ms, algorithmic_bytes):
    """The counter-backed view of one launch: algorithmic bytes (stated in DESIGN.md), HBM-side traffic from the counters, what fraction of the
    8 TB/s peak that traffic is at the duration measured LIVE in this run, and how busy the vector units were (SQ_ACTIVE_INST_VALU
    quad-cycles x 4 over the 1024 SIMDs' share of GRBM_GUI_ACTIVE / 8 cycles)."""
    out = {"algorithmic_bytes": int(algorithmic_bytes), "traffic_bytes": None, "frac_hbm_by_traffic": None, "valu_busy": None}
    if not entry:
        return out
    t = pmc_traffic_bytes(entry)
    if t is not None:
        out["traffic_bytes"] = t
        out["frac_hbm_by_traffic"] = round(t / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    if entry.get("GRBM_GUI_ACTIVE") and entry.get("SQ_ACTIVE_INST_VALU"):
        out["valu_busy"] = round(4.0 * entry["SQ_ACTIVE_INST_VALU"] / (1024.0 * entry["GRBM_GUI_ACTIVE"] / 8.0), 3)
    return out


class EventTimer:
    """HIP events on the stream the kernels are launched on (a torch side stream handed to every svthip call)."""

    def __init__(self, torch):
        self.torch = torch
        self.tstream = torch.cuda.Stream()
        self.stream = self.tstream.cuda_stream

    def ms(self, fn, iters, warm=2):
        torch = self.torch
        with torch.cuda.stream(self.tstream):
            for _ in range(warm):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.tstream)
            for _ in range(iters):
                fn()
            e1.record(self.tstream)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters


def tile_tu_desc(svtav1_hip, w, h, pic_w, pic_h, iscan_offset_by_type, tx_types, n_qrows, rng):
    """svthip_tu_desc rows tiling a pic_w x pic_h plane completely with w x h TUs, in place (recon = pred)."""
    n = min(w, 32) * min(h, 32)
    cols, rows = pic_w // w, pic_h // h
    n_tu = cols * rows
    idx = np.arange(n_tu)
    off = (idx // cols) * h * pic_w + (idx % cols) * w
    d = np.zeros(n_tu, dtype=svtav1_hip.TU_DESC_DTYPE)
    d["src_offset"] = d["pred_offset"] = d["recon_offset"] = off
    d["coeff_offset"] = idx * n
    tt = np.asarray(tx_types)[rng.integers(0, len(tx_types), n_tu)]
    d["tx_type"] = tt
    d["iscan_offset"] = np.asarray([iscan_offset_by_type[int(t)] for t in tt])
    d["src_stride"] = d["pred_stride"] = d["recon_stride"] = pic_w
    d["qparam_index"] = rng.integers(0, n_qrows, n_tu)
    # descriptors ordered by transform type, as svthip_tu_batcher_flush emits them: a wave owns 64 / n consecutive TUs and its lanes
    # branch on their TU's transform kinds, so mixed waves would run the DCT and the ADST network in every pass (TUs are independent)
    if min(w, h) >= 8:
        d = d[np.argsort(d["tx_type"], kind="stable")]
    return d, n


# ----------------------------------------------------------------------------------------------------------------------
# legs
# ----------------------------------------------------------------------------------------------------------------------
def leg_me_chain(ctx, torch, svtav1_hip, timer, pool, descs, params_b, d_sb, n_sb, dev):
    """configs[2] (ME half): whole-picture ME of 12 B pictures = per list { centres, full-pel, sub-pel }, bi-prediction, packing."""
    out = {}
    n_jobs = PICTURES_PER_STEP
    curs = [descs[i + 1] for i in range(n_jobs)]
    r0 = [descs[i] for i in range(n_jobs)]
    r1 = [descs[(i + 2) % len(descs)] for i in range(n_jobs)]
    for n_pu, fn in ((85, ctx.motion_estimate_batch_dev), (209, ctx.motion_estimate209_batch_dev)):
        d_out = torch.zeros((n_jobs * n_sb, n_pu, 24), dtype=torch.uint8, device=dev)
        for name, refs1, subpel in (("B_subpel", r1, True), ("P_subpel", None, True), ("B_fullpel_only", r1, False)):
            ms = timer.ms(lambda: fn(pool.data_ptr(), curs, r0, refs1, params_b, d_sb.data_ptr(), n_sb, d_out.data_ptr(), subpel, 0, None, None,
                                     timer.stream), 5)
            out[f"{n_pu}pu_{name}"] = {"ms_per_picture": round(ms / n_jobs, 4), "superblocks_per_s": round(n_jobs * n_sb / ms * 1e3, 0)}
    # the kernels of these chains one by one: live duration (HIP events on the launch stream), algorithmic bytes, counter traffic, VALU busy
    n = n_jobs * n_sb
    stride = descs[0].full_stride
    d_desc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    hme = lambda: ctx.hme_search_center_batch_dev(pool.data_ptr(), curs, r0, params_b, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr(), None, None,  # noqa: E731
                                                  timer.stream)
    P = params_b
    l0w = [int(P.hme_level0_search_area_in_width_array[k] * P.hme_level0_multiplier_x // 100) for k in range(2)]
    l0h = [int(P.hme_level0_search_area_in_height_array[k] * P.hme_level0_multiplier_y // 100) for k in range(2)]
    # per SB: the three source blocks + per region the level-0 / 1 / 2 windows + five 64 x 32-row centre-check blocks + the descriptor
    hme_bytes = 4096 + 1024 + 256 + sum((l0w[a] + 15) * (l0h[b] + 15) + (16 + 31) * (16 + 31) + (8 + 63) * (8 + 63) for a in range(2) for b in range(2)) \
        + 5 * 2048 + 24
    kern = {"hme_center_kernel": dict(ms=round(timer.ms(hme, 10), 4), **counters(pmc_entry("hme_center_kernel"), timer.ms(hme, 10), n * hme_bytes))}
    for n_pu, fp, sub, wg, kfp in ((85, ctx.fullpel_search_dev, ctx.subpel_refine_dev, 512, "fullpel85_kernel"),
                                   (209, ctx.fullpel_search209_dev, ctx.subpel_refine209_dev, 448, "fullpel209_kernel")):
        d_sad = torch.empty((n, n_pu), dtype=torch.int32, device=dev)
        d_mv = torch.empty_like(d_sad)
        a = (pool.data_ptr(), stride, pool.data_ptr(), stride, d_desc.data_ptr(), n, SEARCH_W, SEARCH_H, d_sad.data_ptr(), d_mv.data_ptr())
        run_fp = lambda: fp(*a, timer.stream)  # noqa: E731
        ms_fp = timer.ms(run_fp, 10)
        kern[kfp] = dict(ms=round(ms_fp, 4), frac_sad_ceiling=round(ABSDIFF_PER_BLOCK * n / (ms_fp * 1e-3) / VALU_PEAK_ABSDIFF_PER_S, 4),
                         **counters(pmc_entry(kfp), ms_fp, n * (4096 + 127 * 127 + 8 * n_pu)))
        torch.cuda.synchronize()
        s0, m0 = d_sad.clone(), d_mv.clone()
        with torch.cuda.stream(timer.tstream):
            def restore():
                d_sad.copy_(s0, non_blocking=True)
                d_mv.copy_(m0, non_blocking=True)

            def run_sub():   # the refinement is in place: restore the full-pel results first (timed separately and subtracted)
                restore()
                sub(pool.data_ptr(), stride, pool.data_ptr(), stride, d_desc.data_ptr(), n, SEARCH_W, SEARCH_H, d_sad.data_ptr(), d_mv.data_ptr(), False,
                    timer.stream)
            ms_sub = timer.ms(run_sub, 10) - timer.ms(restore, 10)
        kern[f"subpel_planes_kernel_{n_pu}pu"] = dict(ms=round(ms_sub, 4), **counters(pmc_entry("subpel_planes_kernel", -1, wg), ms_sub,
                                                                                       n * (4096 + 135 * 135 + 16 * n_pu)))
    out["kernels_6120_superblocks"] = kern
    out["workload"] = "12 x 1080p pictures per call, 64x64 search area, HME on, sub-pel = half + quarter pel of every PU (SSD metric), bi-prediction"
    return out


def leg_ois(ctx, torch, svtav1_hip, timer, pool, descs, params_b, d_sb, n_sb, dev):
    """OpenLoopIntraSearchLcu of 12 x 1080p pictures per call, after the ME of the same pictures (its distortions feed
    the general branch).  Algorithmic pixel work per SB and mode: 3 CU levels x 4096 predicted pixels + SAD."""
    n_jobs = PICTURES_PER_STEP
    curs = [descs[i + 1] for i in range(n_jobs)]
    r0 = [descs[i] for i in range(n_jobs)]
    r1 = [descs[(i + 2) % len(descs)] for i in range(n_jobs)]
    d_me = torch.zeros((n_jobs * n_sb, 85, 24), dtype=torch.uint8, device=dev)
    ctx.motion_estimate_batch_dev(pool.data_ptr(), curs, r0, r1, params_b, d_sb.data_ptr(), n_sb, d_me.data_ptr(), True, 0, None, None, timer.stream)
    d_cand = torch.zeros((n_jobs * n_sb, 85, 18), dtype=torch.int32, device=dev)
    d_total = torch.zeros((n_jobs * n_sb, 85), dtype=torch.uint8, device=dev)
    out = {}
    for name, modes, kw in (("intra_picture", 7, dict(slice_is_intra=1)), ("base_layer_35_modes", 35, dict(temporal_layer_index=0)),
                            ("general_branch", 10, dict(temporal_layer_index=2, is_used_as_reference_flag=1)),
                            ("dc_only", 1, dict(temporal_layer_index=3, limit_ois_to_dc_mode_flag=1))):
        prm = svtav1_hip.OisParams()
        for k, v in kw.items():
            setattr(prm, k, v)
        ms = timer.ms(lambda: ctx.open_loop_intra_search_batch_dev(pool.data_ptr(), curs, prm, d_sb.data_ptr(), n_sb, d_me.data_ptr(), 85,
                                                                   d_cand.data_ptr(), d_total.data_ptr(), timer.stream), 5)
        out[name] = {"ms_per_picture": round(ms / n_jobs, 4), "modes_evaluated": modes,
                     "gpix_predicted_per_s": round(n_jobs * n_sb * 3 * 4096 * modes / ms * 1e-6, 1)}
    out["workload"] = "12 x 1080p pictures per call; 84 CUs (4 + 16 + 64) per SB; every mode of the branch's list for every CU, then the per-CU decision"
    return out


def leg_tq(ctx, torch, svtav1_hip, timer, dev, rng):
    """configs[3]: fused encode chain (residual -> fwd txfm -> quant -> dequant -> inv txfm -> recon) per square size, (a) on one
    1080p luma frame of TUs (SURVEY 8d config 4 shapes) and (b) on a 64 M-pixel plane for the roofline fraction (working set beyond
    the 256 MB infinity cache); real quantiser rows at qindex 20 / 120 / 200, real scan orders, DCT_DCT / ADST_ADST / IDTX."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))   # data produced by the reference's av1_build_quantizer
    qrows = np.ascontiguousarray(z["rows_bd8_inter"][[20, 120, 200], 0, :])
    d_qp = torch.from_numpy(qrows).to(dev)
    d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
    out = {"workload": "fused per-TU encode chain, 8-bit; quantiser rows = av1_build_quantizer luma rows at qindex 20/120/200; "
                       "bytes/pixel algorithmic = 1 src + 1 pred + 1 recon + 4 qcoeff + 2 iscan = 9; descriptors ordered by transform type like the "
                       "batcher's flush (a random order costs sizes <= 16x16 about a third more: mixed waves run both networks)",
           "sizes": {}}
    for n in (4, 8, 16, 32, 64):
        ts = svtav1_hip.TX_SIZES_WH.index((n, n))
        types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(n, n)]
        isc = {t: int(z["scan_offsets"][int(z["scan_index"][ts, t])]) for t in types}
        res = {}
        for label, pic_w, pic_h in (("frame_1080p", 1920, 1088), ("plane_64Mpx", 8192, 8192)):
            d, nc = tile_tu_desc(svtav1_hip, n, n, pic_w, pic_h, isc, types, 3, rng)
            n_tu = len(d)
            src = torch.randint(0, 256, (pic_w * pic_h,), dtype=torch.uint8, device=dev)
            noise = torch.empty(pic_w * pic_h, device=dev).exponential_(1 / 6.0) * (torch.randint(0, 2, (pic_w * pic_h,), device=dev) * 2 - 1)
            pred = (src.float() - noise).clamp_(0, 255).to(torch.uint8)   # Laplacian(b = 6) residual
            recon = torch.empty_like(pred)                                 # out of place: every timed launch sees the same residual
            d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
            d_q = torch.empty(n_tu * nc, dtype=torch.int32, device=dev)
            d_eob = torch.empty(n_tu, dtype=torch.int16, device=dev)

            def run():
                ctx.encode_tu_batch_dev(src.data_ptr(), pred.data_ptr(), recon.data_ptr(), d_desc.data_ptr(), n_tu, n, n, d_qp.data_ptr(),
                                        d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None, timer.stream)
            ms = timer.ms(run, 20 if label == "frame_1080p" else 5)
            px = n_tu * n * n
            algo = px * 3 + n_tu * nc * 6 + n_tu * 2
            # rocprofv3 names the instantiation by log2 of the sides, "encode_tu_kernel<4, 4, unsigned char>" = 16x16; other legs launch
            # the same kernel with other grids: a wave owns 64 / n TUs, four waves per workgroup -> n_tu * n threads in whole workgroups,
            # capped at 256 x 64 workgroups (grid-stride beyond that); the trailing "false" = no distortion sums asked for
            lg = n.bit_length() - 1
            if label == "frame_1080p":
                rec = pmc_entry(f"encode_tu_kernel<{lg}, {lg}, unsigned char, false>", grid=min((n_tu * n + 255) // 256, 256 * 64) * 256)
            else:   # square sizes up to 32 walk the plane with as many workgroups as the chip holds: its longest launch
                rec = pmc_entry(f"encode_tu_kernel<{lg}, {lg}, unsigned char, false>", which="longest")
            res[label] = {"n_tu": n_tu, "ms": round(ms, 4), "gpix_per_s": round(px / ms / 1e6, 2), "algorithmic_gbps": round(algo / ms / 1e6, 1),
                          "frac_hbm_algorithmic": round(algo / ms / 1e6 / HBM_PEAK_GBPS, 4), **counters(rec, ms, algo)}
            del src, pred, recon, noise, d_q
        out["sizes"][f"{n}x{n}"] = res
        torch.cuda.empty_cache()
    return out


def leg_convolve(ctx, torch, svtav1_hip, timer, pool, pdesc, dev):
    """configs[2] (8-tap half): av1_convolve_2d_sr on 64x64 blocks of one 1080p frame at all 15 x 15 fractional phases
    (SURVEY 8d config 3): 510 blocks x 225 phases = 114 750 blocks per launch, filters cycling REGULAR / SMOOTH / SHARP / BILINEAR."""
    S = pdesc[0].full_stride
    nbx, nby, n_ph = 30, 17, 225
    n = nbx * nby * n_ph
    d = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    i = np.arange(n)
    blk, ph = i // n_ph, i % n_ph
    d["src_offset"] = pdesc[1].full_offset + (68 + (blk // nbx) * 64) * S + 68 + (blk % nbx) * 64
    d["dst_offset"] = i * 4096                       # dense 64 x 64 tiles
    d["subpel_x"], d["subpel_y"] = 1 + ph % 15, 1 + ph // 15
    d["filter_x"], d["filter_y"] = (blk + ph) % 4, (blk // 3 + ph // 5) % 4
    d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
    d_dst = torch.empty(n * 4096, dtype=torch.uint8, device=dev)
    ms = timer.ms(lambda: ctx.av1_convolve_sr_batch_dev(pool.data_ptr(), S, d_dst.data_ptr(), 64, d_desc.data_ptr(), n, 64, 64, timer.stream), 10)
    algo = n * (71 * 71 + 4096 + 16)
    # The 225 phases of a block read the SAME 71 x 71 source samples: counted per (block, phase) the "algorithmic" bytes flatter the
    # kernel by the cache reuse (round 2 quoted 0.39 of the HBM peak that way); the counters see the source once per block and the
    # destination once per output, which is what frac_hbm_by_traffic reports.  unique_bytes = what has to cross HBM at least once.
    unique = 510 * 71 * 71 + n * (4096 + 16)
    out = {"blocks": n, "ms": round(ms, 4), "gpix_per_s": round(n * 4096 / ms / 1e6, 2), "algorithmic_gbps_per_block_phase": round(algo / ms / 1e6, 1),
           "unique_bytes": unique, "frac_hbm_unique": round(unique / ms / 1e6 / HBM_PEAK_GBPS, 4),
           **counters(pmc_entry("av1_convolve_mfma_kernel<2, false>"), ms, algo),
           "workload": "8-bit av1_convolve_2d_sr, 64x64 blocks x 225 phases over a 1080p frame; algorithmic bytes = 71 x 71 read + 4096 written per "
                       "(block, phase); the source of a block is shared by its 225 phases (cache hits), so the HBM fraction that counts is "
                       "frac_hbm_by_traffic"}
    # the BI_PRED form of the same blocks: list 0 from picture 1, list 1 from picture 2 with the phases swapped (av1_jnt_convolve_2d pair)
    c = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
    c["src0_offset"] = d["src_offset"]
    c["src1_offset"] = pdesc[2].full_offset + (68 + (blk // nbx) * 64) * S + 68 + (blk % nbx) * 64
    c["dst_offset"] = d["dst_offset"]
    c["subpel0"] = d["subpel_x"] | (d["subpel_y"] << 4)
    c["subpel1"] = d["subpel_y"] | (d["subpel_x"] << 4)
    c["filter_x"], c["filter_y"] = d["filter_x"], d["filter_y"]
    d_cdesc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).to(dev)
    msc = timer.ms(lambda: ctx.av1_convolve_compound_batch_dev(pool.data_ptr(), S, pool.data_ptr(), S, d_dst.data_ptr(), 64, d_cdesc.data_ptr(), n, 64, 64,
                                                               timer.stream), 10)
    algoc = n * (2 * 71 * 71 + 4096 + 16)
    out["compound_bipred"] = {"ms": round(msc, 4), "gpix_per_s": round(n * 4096 / msc / 1e6, 2),
                              "algorithmic_gbps_per_block_phase": round(algoc / msc / 1e6, 1),
                              **counters(pmc_entry("av1_convolve_mfma_kernel<2, true>"), msc, algoc)}
    return out


def leg_sad_loop(ctx, torch, svtav1_hip, timer, dev):
    """configs[0] on the GPU: SadLoopKernel semantics, every 16x16 block of 12 synthetic 856x480 pictures (53 x 30 = 1590 per picture),
    +-16 search = 33 x 33 positions, full rows."""
    w, h, n_pic = 856, 480, 12
    pool, descs = device_picture_pool(ctx, n_pic + 1, w, h, dev)
    S = descs[0].full_stride
    rows = []
    for j in range(n_pic):
        for by in range(0, 480, 16):
            for bx in range(0, 848, 16):
                rows.append((descs[j + 1].full_offset + (68 + by) * S + 68 + bx, descs[j].full_offset + (68 + by - 16) * S + 68 + bx - 16))
    d = np.asarray(rows, dtype=np.uint32)
    n = d.shape[0]
    d_desc = torch.from_numpy(d.view(np.int32).reshape(-1).copy()).to(dev)
    d_sad = torch.empty(n, dtype=torch.int32, device=dev)
    d_xy = torch.empty((n, 2), dtype=torch.int16, device=dev)
    ms = timer.ms(lambda: ctx.sad_loop_batch_dev(pool.data_ptr(), S, pool.data_ptr(), S, S, d_desc.data_ptr(), n, 16, 16, 33, 33, d_sad.data_ptr(),
                                                 d_xy.data_ptr(), timer.stream), 10)
    absdiff = n * 33 * 33 * 256
    return {"blocks": n, "ms": round(ms, 4), "blocks_per_s": round(n / ms * 1e3, 0), "absdiff_per_s": round(absdiff / ms * 1e3, 0),
            "frac_sad_ceiling": round(absdiff / ms * 1e3 / VALU_PEAK_ABSDIFF_PER_S, 4),
            **counters(pmc_entry("sad_loop_qsad_kernel"), ms, n * (256 + 48 * 48 + 8)),
            "workload": "16x16 blocks, 33x33 positions (+-16), 856x480 8-bit, 12 pictures per launch, SadLoopKernel semantics (packed-SAD kernel: 12 positions per lane, five blocks per workgroup, v_qsad_pk_u16_u8)"}


def leg_4k(ctx, torch, svtav1_hip, timer, dev, rng):
    """configs[4] on one GPU: 3840x2160 -- hierarchical ME on the 8-bit MSB plane (2040 SBs per picture) and the 10-bit fused TU chain
    (svthip_encode_tu16_batch_dev, bd 10 rows) over one 4K luma frame."""
    w, h = 3840, 2160
    pool, descs = device_picture_pool(ctx, 4, w, h, dev)
    sb = svtav1_hip.sb_origins(w, h)
    n_sb = sb.shape[0]
    d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
    params = svtav1_hip.default_me_params(w, h, 3, 0)
    curs, refs = [descs[1], descs[2], descs[3]], [descs[0], descs[1], descs[2]]
    n = len(curs) * n_sb
    d_desc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    d_sad = torch.empty((n, 85), dtype=torch.int32, device=dev)
    d_mv = torch.empty((n, 85), dtype=torch.int32, device=dev)
    fs = descs[0].full_stride

    def me():
        ctx.hme_search_center_batch_dev(pool.data_ptr(), curs, refs, params, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr(), None, None, timer.stream)
        ctx.fullpel_search_dev(pool.data_ptr(), fs, pool.data_ptr(), fs, d_desc.data_ptr(), n, SEARCH_W, SEARCH_H, d_sad.data_ptr(), d_mv.data_ptr(),
                               timer.stream)
    ms = timer.ms(me, 10)
    out = {"me_8bit_msb": {"pictures": len(curs), "blocks": n, "ms": round(ms, 4), "blocks_per_s": round(n / ms * 1e3, 0)}}
    z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
    qrows = np.ascontiguousarray(z["rows_bd10_inter"][[20, 120, 200], 0, :])
    d_qp = torch.from_numpy(qrows).to(dev)
    d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
    pic_w, pic_h = 3840, 2176
    for nn in (16, 32):
        ts = svtav1_hip.TX_SIZES_WH.index((nn, nn))
        types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(nn, nn)]
        isc = {t: int(z["scan_offsets"][int(z["scan_index"][ts, t])]) for t in types}
        d, nc = tile_tu_desc(svtav1_hip, nn, nn, pic_w, pic_h, isc, types, 3, rng)
        n_tu = len(d)
        src = torch.randint(0, 1024, (pic_w * pic_h,), dtype=torch.int16, device=dev)
        noise = (torch.empty(pic_w * pic_h, device=dev).exponential_(1 / 24.0) * (torch.randint(0, 2, (pic_w * pic_h,), device=dev) * 2 - 1))
        pred = (src.float() - noise).clamp_(0, 1023).to(torch.int16)
        d_desc_t = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
        d_q = torch.empty(n_tu * nc, dtype=torch.int32, device=dev)
        d_eob = torch.empty(n_tu, dtype=torch.int16, device=dev)
        recon = torch.empty_like(pred)

        def run():
            ctx.encode_tu_batch_dev(src.data_ptr(), pred.data_ptr(), recon.data_ptr(), d_desc_t.data_ptr(), n_tu, nn, nn, d_qp.data_ptr(),
                                    d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None, timer.stream, True)
        ms = timer.ms(run, 10)
        px = n_tu * nn * nn
        algo = px * 6 + n_tu * nc * 6
        out[f"encode_tu16_{nn}x{nn}_bd10"] = {"n_tu": n_tu, "ms": round(ms, 4), "gpix_per_s": round(px / ms / 1e6, 2),
                                            "algorithmic_gbps": round(algo / ms / 1e6, 1)}
    # 10-bit inter prediction of every 64x64 block of the 4K frame at 8 random phase pairs, uni- and bi-predicted
    S = 3840 + 160
    ref16 = torch.randint(0, 1024, (S * (2176 + 160),), dtype=torch.int16, device=dev)
    nbx, nby, n_ph = 60, 34, 8
    n = nbx * nby * n_ph
    i = np.arange(n)
    blk = i // n_ph
    d = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    d["src_offset"] = (80 + (blk // nbx) * 64) * S + 80 + (blk % nbx) * 64
    d["dst_offset"] = i * 4096
    d["subpel_x"], d["subpel_y"] = rng.integers(1, 16, n), rng.integers(1, 16, n)
    d["filter_x"], d["filter_y"] = rng.integers(0, 4, n), rng.integers(0, 4, n)
    c = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
    c["src0_offset"], c["src1_offset"], c["dst_offset"] = d["src_offset"], d["src_offset"] + 3 * S + 5, d["dst_offset"]
    c["subpel0"], c["subpel1"] = d["subpel_x"] | (d["subpel_y"] << 4), d["subpel_y"] | (d["subpel_x"] << 4)
    c["filter_x"], c["filter_y"] = d["filter_x"], d["filter_y"]
    d_d = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
    d_c = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).to(dev)
    d_dst = torch.empty(n * 4096, dtype=torch.int16, device=dev)
    for name, desc_t, comp in (("inter_pred_64x64_bd10", d_d, False), ("inter_pred_64x64_bd10_bipred", d_c, True)):
        ms = timer.ms(lambda: ctx.av1_highbd_convolve_batch_dev(ref16.data_ptr(), S, ref16.data_ptr(), S, d_dst.data_ptr(), 64, desc_t.data_ptr(), comp, n,
                                                                64, 64, 10, timer.stream), 5)
        out[name] = {"blocks": n, "ms": round(ms, 4), "gpix_per_s": round(n * 4096 / ms / 1e6, 2)}
    out["workload"] = ("3840x2160: ME = search centres + 64x64 full-pel, 85 PUs, one list, 3 pictures per launch; T/Q = one 10-bit luma frame of TUs; "
                       "inter prediction = every 64x64 block x 8 phase pairs, 10-bit samples")
    return out


def leg_host_threads(svtav1_hip, seconds=2.0):
    """What the reference's ME process would see through the HOST-pointer entry: T host threads (its ME process runs max(20, cores / 3)
    of them, Codec/EbEncHandle.c:436-439), one context each, one 1080p P picture per call of svthip_motion_estimate_picture (upload of
    two luma planes from pageable memory, planes derived on the device, the whole chain with sub-pel, MeCuResults_t rows back).
    pictures/s for T = 1, 4, 20 beside the device-resident figure of the same chain."""
    import ctypes as C

    from svtav1_hip import synth
    L = svtav1_hip.lib()

    class HostPicture(C.Structure):
        _fields_ = [("buffer_y", C.c_void_p), ("stride_y", C.c_uint32), ("origin_x", C.c_uint16), ("origin_y", C.c_uint16), ("width", C.c_uint16),
                    ("height", C.c_uint16)]

    L.svthip_motion_estimate_picture.restype = C.c_int32
    L.svthip_motion_estimate_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p]
    pics = [synth.PaPicture(synth.synth_luma(W, H, t)) for t in (1, 0)]
    P = svtav1_hip.default_me_params(W, H, 3, 0)
    n_sb = 510
    out = {}
    for T in (1, 4, 20):
        ctxs = [svtav1_hip.Context(0) for _ in range(T)]
        for c in ctxs:
            c.reserve(W, H, 85, 1, True)   # scratch sized once, as MeContextCtor sizes the reference's buffers
        bufs = []
        for _ in range(T):   # every thread owns its pictures and its result rows, like a picture control set
            cur, ref = pics[0].full.copy(), pics[1].full.copy()
            rows = np.zeros((n_sb, 85, 40), np.uint8)
            ptrs = (C.c_void_p * n_sb)(*[rows.ctypes.data + i * 85 * 40 for i in range(n_sb)])
            bufs.append((cur, ref, rows, ptrs, HostPicture(cur.ctypes.data, pics[0].stride, 68, 68, W, H),
                         HostPicture(ref.ctypes.data, pics[1].stride, 68, 68, W, H)))
        done = [0] * T
        err = []

        def worker(i, stop):
            _, _, _, ptrs, hc, hr = bufs[i]
            while time.perf_counter() < stop[0]:
                rc = L.svthip_motion_estimate_picture(ctxs[i]._h, C.byref(hc), C.byref(hr), None, C.byref(P), 1, 0, 85, ptrs)
                if rc:
                    err.append(rc)
                    return
                done[i] += 1

        for phase_s in (0.3, seconds):   # warm-up, then the timed phase
            done[:] = [0] * T
            stop = [time.perf_counter() + phase_s]
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i, stop)) for i in range(T)]
            [t.start() for t in th]
            [t.join() for t in th]
            dt = time.perf_counter() - t0
        assert not err, f"svthip_motion_estimate_picture failed: 0x{err[0] & 0xffffffff:08x}"
        out[f"threads_{T}"] = {"pictures_per_s": round(sum(done) / dt, 1), "ms_per_picture_per_thread": round(dt / max(1, max(done)) * 1e3, 3)}
        for c in ctxs:
            c.close()
    up, down = 2 * W * H, n_sb * 85 * 40
    out["pcie_bytes_per_picture"] = {"host_to_device": up, "device_to_host": down}
    out["workload"] = ("svthip_motion_estimate_picture: one 1080p P picture per call (85 PUs, sub-pel on), host pointers, pageable memory, one context and "
                       "one stream per host thread; compare legs.me_chain_subpel.85pu_P_subpel (12 pictures per call, device resident)")
    return out


def leg_batcher(ctx, torch, svtav1_hip, dev, rng):
    """The batching layer where it will be used: the tx-type search of mode decision adds every (TU, tx_type) candidate of a
    superblock's partition tree -- 64x64 (DCT), 4 x 32x32 (DCT, IDTX), 16 x 16x16, 64 x 8x8 and 256 x 4x4 with three types each: 1 017
    candidates per SB -- and flushes once per G superblocks.  One flush = one descriptor upload, one fused launch per size present,
    one download; G = 1 (what a serial SB loop gives), 8, 30 (an SB row of 1080p), 510 (a frame).  Adds are host-side appends in C and are
    not timed (the Python ctypes loop that drives them here would dominate)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
    qrows = np.ascontiguousarray(z["rows_bd8_inter"][[20, 120, 200], 0, :])
    d_qp = torch.from_numpy(qrows).to(dev)
    d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
    pic_w, pic_h = 1920, 1088
    src = torch.randint(0, 256, (pic_w * pic_h,), dtype=torch.uint8, device=dev)
    pred = (src.short() + torch.randint(-6, 7, src.shape, dtype=torch.int16, device=dev)).clamp_(0, 255).to(torch.uint8)
    per_sb = []   # (tx_size index, tx_type, x, y) of one superblock's candidates
    for n, types in ((64, (0,)), (32, (0, 9)), (16, (0, 3, 9)), (8, (0, 3, 9)), (4, (0, 3, 9))):
        ts = svtav1_hip.TX_SIZES_WH.index((n, n))
        for y in range(0, 64, n):
            for x in range(0, 64, n):
                for t in types:
                    per_sb.append((ts, t, x, y, int(z["scan_offsets"][int(z["scan_index"][ts, t])])))
    cand_per_sb = len(per_sb)
    coeff_per_sb = sum(min(svtav1_hip.TX_SIZES_WH[c[0]][0], 32) ** 2 for c in per_sb)
    out = {"candidates_per_superblock": cand_per_sb}
    SCR = 0xffffffff
    for G in (1, 8, 30, 510):
        b = svtav1_hip.TuBatcher(ctx, G * cand_per_sb, G * coeff_per_sb)
        reps = 1 if G >= 510 else 3 if G >= 30 else 6
        t_flush = 0.0
        for _ in range(reps + 1):
            b.begin(src.data_ptr(), pred.data_ptr(), None, 0, d_qp.data_ptr(), d_iscan.data_ptr())
            for g in range(G):
                ox, oy = (g % 30) * 64, (g // 30) * 64
                for ts, t, x, y, isc in per_sb:
                    off = (oy + y) * pic_w + ox + x
                    b.add(ts, t, off, pic_w, off, pic_w, SCR, 0, g % 3, isc)
            ctx.synchronize()
            t0 = time.perf_counter()
            b.flush()
            dt = time.perf_counter() - t0
            if _:
                t_flush += dt
        ms = t_flush / reps * 1e3
        px = G * sum(svtav1_hip.TX_SIZES_WH[c[0]][0] ** 2 for c in per_sb)
        out[f"flush_per_{G}_sb"] = {"ms_per_flush": round(ms, 4), "candidates_per_s": round(G * cand_per_sb / ms * 1e3, 0), "gpix_per_s": round(px / ms / 1e6, 3),
                                    "ms_per_superblock": round(ms / G, 4)}
        b.close()
    out["workload"] = ("svthip_tu_batcher: 1 017 (TU, tx_type) candidates per superblock (1 x 64x64, 8 x 32x32, 48 x 16x16, 192 x 8x8, 768 x 4x4), real "
                       "quantiser rows and scans, reconstruction into batcher scratch; wall time of flush = upload + five fused launches + download")
    return out


def leg_recon_exchange(ctx, comm, torch, dist, svtav1_hip, dev, world, steps=10):
    """The real exchange step of the encode pass (north-star): every rank reconstructs ITS SB-row slab of a picture with the fused TU
    chain, then svthip_recon_exchange_dev (C ABI, RCCL over xGMI: one group of direct sends / receives straight into the padded planes
    of Y, Cb, Cr) assembles the picture on every rank and pads the three planes like PadRefAndSetFlags.  All ranks take part; times are
    max over ranks.  Everything is enqueued on the context's stream; one host synchronisation per picture."""
    from svtav1_hip import sharded

    z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
    rng = np.random.default_rng(5)
    out = {}
    for label, w, h, bd in (("1080p_8bit", 1920, 1080, 8), ("2160p_10bit", 3840, 2160, 10)):
        pad, nn = 160, 16
        wide = bd == 10
        es = 2 if wide else 1
        stride = w + 2 * pad
        ex = sharded.ReconExchange(w, h, pad, comm=comm, sample_bytes=es)
        y0, nrows = ex.my_rows
        ts = svtav1_hip.TX_SIZES_WH.index((nn, nn))
        types = [0, 3, 9]
        isc = {t: int(z["scan_offsets"][int(z["scan_index"][ts, t])]) for t in types}
        qrows = np.ascontiguousarray(z[f"rows_bd{bd}_inter"][[20, 120, 200], 0, :])
        d_qp = torch.from_numpy(qrows).to(dev)
        d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
        # TUs of this rank's slab, addressed inside the PADDED plane (offsets in samples)
        rows_tu, cols_tu = (nrows + nn - 1) // nn, w // nn
        n_tu = rows_tu * cols_tu
        idx = np.arange(n_tu)
        off = (pad + y0 + (idx // cols_tu) * nn) * stride + pad + (idx % cols_tu) * nn
        d = np.zeros(n_tu, dtype=svtav1_hip.TU_DESC_DTYPE)
        d["src_offset"] = d["pred_offset"] = d["recon_offset"] = off
        d["coeff_offset"] = idx * nn * nn
        tt = np.asarray(types)[rng.integers(0, 3, n_tu)]
        d["tx_type"] = tt
        d["iscan_offset"] = np.asarray([isc[int(t)] for t in tt])
        d["src_stride"] = d["pred_stride"] = d["recon_stride"] = stride
        d["qparam_index"] = rng.integers(0, 3, n_tu)
        d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
        rows_total = h + 2 * pad + nn  # room for a slab whose last TU row hangs below the picture (1080 = 67.5 x 16)
        hi = 1 << bd
        if wide:
            src = torch.randint(0, hi, (rows_total, stride), dtype=torch.int16, device=dev)
            pred = (src + torch.randint(-24, 25, src.shape, dtype=torch.int16, device=dev)).clamp_(0, hi - 1)
        else:
            src = torch.randint(0, hi, (rows_total, stride), dtype=torch.uint8, device=dev)
            pred = (src.short() + torch.randint(-6, 7, src.shape, dtype=torch.int16, device=dev)).clamp_(0, hi - 1).to(torch.uint8)
        recon = torch.zeros_like(pred)
        d_q = torch.empty(max(1, n_tu) * nn * nn, dtype=torch.int32, device=dev)
        d_eob = torch.empty(max(1, n_tu), dtype=torch.int16, device=dev)
        y_u8 = recon.view(torch.uint8).view(rows_total, stride * es)[: h + 2 * pad]
        # chroma planes (4:2:0, origin 80): carried through the exchange and the padding; their T/Q is not part of this leg
        c_rows, c_stride = h // 2 + pad, w // 2 + pad
        chroma = [torch.randint(0, 256, (c_rows, c_stride * es), dtype=torch.uint8, device=dev) for _ in range(2)]
        torch.cuda.synchronize()

        def step():
            if n_tu:
                ctx.encode_tu_batch_dev(src.data_ptr(), pred.data_ptr(), recon.data_ptr(), d_desc.data_ptr(), n_tu, nn, nn, d_qp.data_ptr(),
                                        d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None, None, wide)
            ex.exchange([y_u8, chroma[0], chroma[1]])
            ctx.synchronize()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        plan = sharded.recon_exchange_plan(ex.picture(1, stride, 1, 1, c_stride), world, ex.rank)
        out[label] = {"ms_per_picture": round(el / steps * 1e3, 4), "slab_rows_per_rank": [n for _, n in ex.rows],
                      "bytes_sent_per_rank": int(sum(x.bytes for x in plan if x.send)), "bytes_received_per_rank": int(sum(x.bytes for x in plan if not x.send)),
                      "tus_per_rank_16x16": n_tu, "planes": "Y + Cb + Cr",
                      "exchange": "svthip_recon_exchange_dev: one RCCL group of direct ncclSend/ncclRecv per picture" if world > 1 else "none (1 rank)"}
        del src, pred, recon, chroma
    out["workload"] = ("per picture: fused 16x16 TU chain on the rank's luma SB-row slab -> svthip_recon_exchange_dev (slabs of Y, Cb, Cr straight "
                       "into every rank's padded planes over RCCL) -> generate_padding of the three planes (160 / 80 samples) on every rank")
    return out


# ----------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """Start the N ranks as fresh children (this process has not touched a GPU) and relay their exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only")
    ap.add_argument("--no-hme", action="store_true", help="time the full-pel search alone (zero-centred windows)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--only-legs", default="", help="comma-separated leg names to run (development aid; default: all)")
    ap.add_argument("--no-host-threads", action="store_true", help="skip the multi-threaded host-pointer leg (counter collection under rocprofv3 "
                                                                    "crashed inside the profiler when 20 host threads issued copies)")
    ap.add_argument("--gather-results", action="store_true", help="also gather the (sad, mv) results of every step on every rank (svthip_me_gather_results_dev, RCCL)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        sys.exit(cpu_baseline_worker_main())

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
        world, rank, local_rank = 1, 0, 0
    else:
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)

    cpu_worker = None
    if world == 1 and not args.no_cpu_baseline:
        # the CPU baselines run in a child started NOW, before this process touches a GPU; it idles until the GPU numbers are in
        cpu_worker = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True)

    import torch
    import torch.distributed as dist

    import svtav1_hip
    from svtav1_hip import sharded

    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(3)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    ctx = svtav1_hip.Context(local_rank)
    try:
        comm = sharded.Comm.from_process_group(ctx)  # svthip_comm (RCCL); world 1 never touches RCCL
        comm_error = None
    except Exception as e:   # the headline needs no exchange: report it, and the exchange leg's failure, rather than nothing
        comm, comm_error = None, f"{type(e).__name__}: {e}"
        print(f"bench.py: rank {rank}: svthip_comm_create failed: {comm_error}", file=sys.stderr)

    # ---- headline: frame-sharded hierarchical ME ----
    n_jobs = PICTURES_PER_STEP * world
    pool, pdesc = device_picture_pool(ctx, n_jobs + 1, W, H, dev)          # the same pictures on every rank
    stride = pdesc[0].full_stride
    first, n_local = sharded.shard_sb_range(W, H, world, rank, "sb")
    sb_all = svtav1_hip.sb_origins(W, H)
    sb_local = np.ascontiguousarray(sb_all[first:first + n_local])
    d_sb = torch.from_numpy(sb_local.view(np.int16).copy()).to(dev)
    curs = [pdesc[i + 1] for i in range(n_jobs)]
    refs = [pdesc[i] for i in range(n_jobs)]
    n_blocks = n_jobs * n_local                                              # this rank's blocks per step
    total_blocks_per_step = n_jobs * sb_all.shape[0]                         # all ranks
    if args.no_hme:
        d_desc = torch.from_numpy(zero_centred_desc(curs, refs, sb_local, W, H)).to(dev)
    else:
        d_desc = torch.zeros((n_blocks, 6), dtype=torch.int32, device=dev)   # written by the search-centre kernel every step
    d_sad = torch.empty((n_blocks, 85), dtype=torch.int32, device=dev)
    d_mv = torch.empty((n_blocks, 85), dtype=torch.int32, device=dev)
    params = svtav1_hip.default_me_params(W, H, 3, 0)
    a = (pool.data_ptr(), stride, pool.data_ptr(), stride, d_desc.data_ptr(), n_blocks, SEARCH_W, SEARCH_H, d_sad.data_ptr(), d_mv.data_ptr())
    if args.gather_results:
        n_total = sb_all.shape[0]
        g_sad = torch.empty((n_jobs, n_total, 85), dtype=torch.int32, device=dev)
        g_mv = torch.empty((n_jobs, n_total, 85), dtype=torch.int32, device=dev)

    def step():
        # ONE search-centre launch per 32 pictures over this rank's superblocks (descriptors land in d_desc), then ONE full-pel launch
        if not args.no_hme:
            ctx.hme_search_center_batch_dev(pool.data_ptr(), curs, refs, params, 0, d_sb.data_ptr(), n_local, None, d_desc.data_ptr())
        ctx.fullpel_search_dev(*a)
        if args.gather_results and comm is not None:   # svthip_me_gather_results_dev on the same stream: every rank ends with all (SAD, MV) rows
            comm.me_gather_results_dev(d_sad.data_ptr(), g_sad.data_ptr(), n_jobs, n_total, 85 * 4)
            comm.me_gather_results_dev(d_mv.data_ptr(), g_mv.data_ptr(), n_jobs, n_total, 85 * 4)

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

    # dominant-kernel duration on this rank, HIP events on the stream the kernel runs on
    kern_ms = ctx.fullpel_search_time_dev(*a, max(5, min(args.steps, 20)))

    legs = {}
    if not args.no_legs:
        want = set(filter(None, args.only_legs.split(",")))
        on = lambda name: not want or name in want  # noqa: E731
        if on("recon_exchange"):
            # every rank takes part; all ranks agree on whether the communicator exists before entering the collective leg
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=dev)
            if distributed:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                legs["recon_exchange"] = leg_recon_exchange(ctx, comm, torch, dist, svtav1_hip, dev, world)
            else:
                legs["recon_exchange"] = {"error": comm_error or "svthip_comm_create failed on another rank"}
        if rank == 0:
            timer = EventTimer(torch)
            rng = np.random.default_rng(3)
            d_sb_all = torch.from_numpy(sb_all.view(np.int16).copy()).to(dev)
            params_b = svtav1_hip.default_me_params(W, H, 3, 1)
            if on("me_chain_subpel"):
                legs["me_chain_subpel"] = leg_me_chain(ctx, torch, svtav1_hip, timer, pool, pdesc, params_b, d_sb_all, sb_all.shape[0], dev)
            if on("open_loop_intra_search"):
                legs["open_loop_intra_search"] = leg_ois(ctx, torch, svtav1_hip, timer, pool, pdesc, params_b, d_sb_all, sb_all.shape[0], dev)
            if on("sad_loop_480p"):
                legs["sad_loop_480p"] = leg_sad_loop(ctx, torch, svtav1_hip, timer, dev)
            if on("convolve_8tap"):
                legs["convolve_8tap"] = leg_convolve(ctx, torch, svtav1_hip, timer, pool, pdesc, dev)
            if on("tq_chain"):
                legs["tq_chain"] = leg_tq(ctx, torch, svtav1_hip, timer, dev, rng)
            if on("uhd_10bit"):
                legs["uhd_10bit"] = leg_4k(ctx, torch, svtav1_hip, timer, dev, rng)
            if on("tu_batcher"):
                legs["tu_batcher"] = leg_batcher(ctx, torch, svtav1_hip, dev, rng)
            if on("host_pointer_threads") and not args.no_host_threads:
                legs["host_pointer_threads"] = leg_host_threads(svtav1_hip)

    if rank == 0:
        value = total_blocks_per_step * args.steps / elapsed
        achieved = ALGO_BYTES_PER_BLOCK * n_blocks / (kern_ms * 1e-3) / 1e9
        absdiff_rate = ABSDIFF_PER_BLOCK * n_blocks / (kern_ms * 1e-3)
        rec85 = pmc_entry("fullpel85_kernel")
        traffic = pmc_traffic_bytes(rec85)
        if traffic is None:   # no round-3 profile in the tree: the round-2 record of the same launch
            tr = load_profile_json("r02_pmc_traffic.json") or {}
            traffic = pmc_traffic_bytes(tr.get("svthip::fullpel85_kernel"))
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
                       "pictures_per_step": n_jobs, "pictures_per_step_per_gpu": PICTURES_PER_STEP,
                       "blocks_per_step": total_blocks_per_step, "blocks_per_step_rank0": n_blocks,
                       "search_area": [SEARCH_W, SEARCH_H],
                       "sharding": "frame-sharded: every rank searches its contiguous superblock range (510/N, balanced to one SB) of every "
                                   "picture of the 12 x N picture batch; no data-path collective"
                                   + ("; (sad, mv) all-gathered over RCCL every step" if args.gather_results and distributed else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "traffic_note": "bytes per launch of 6120 blocks = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 from separate rocprofv3 --pmc "
                                         "passes (profiles/*_pmc_traffic.json; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950); "
                                         "algorithmic bytes per launch = 20905 x blocks",
                         "valu_busy": counters(rec85, kern_ms, 0)["valu_busy"],
                         "kernel": "fullpel85_kernel", "kernel_ms": round(kern_ms, 4), "kernel_blocks": n_blocks,
                         "algorithmic_bytes_per_block": ALGO_BYTES_PER_BLOCK,
                         "note": "search is ~400 abs-diff per compulsory byte: VALU-bound by construction (SURVEY 8d); see valu",
                         "valu": {"achieved_absdiff_per_s": round(absdiff_rate, 0),
                                  "peak_absdiff_per_s": round(VALU_PEAK_ABSDIFF_PER_S, 0),
                                  "frac": round(absdiff_rate / VALU_PEAK_ABSDIFF_PER_S, 4),
                                  "peak_source": "measured v_qsad_pk_u16_u8 issue rate, tools/ubench_valu.hip"}},
            "legs": legs,
        }
        if cpu_worker is not None:
            try:
                res, _ = cpu_worker.communicate("go\n", timeout=300)
                cpu = json.loads(res.strip().splitlines()[-1])
            except Exception as e:
                cpu_worker.kill()
                cpu = {"headline": {"error": f"{type(e).__name__}: {e}"}}
            out["cpu_baseline"] = cpu.get("headline")
            if not args.no_legs:
                if cpu.get("sad_loop_480p"):
                    legs.setdefault("sad_loop_480p", {})["cpu_baseline"] = cpu["sad_loop_480p"]
                if cpu.get("me_chain_subpel.85pu_B_fullpel_only"):
                    legs.setdefault("me_chain_subpel", {}).setdefault("85pu_B_fullpel_only", {})["cpu_baseline"] = cpu["me_chain_subpel.85pu_B_fullpel_only"]
                if cpu.get("convolve_8tap"):
                    legs.setdefault("convolve_8tap", {})["cpu_baseline"] = cpu["convolve_8tap"]
                tq = cpu.get("tq_chain") or {}
                for size, v in tq.items():
                    if size in legs.get("tq_chain", {}).get("sizes", {}):
                        legs["tq_chain"]["sizes"][size]["frame_1080p"]["cpu_baseline"] = v
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
