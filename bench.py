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
# CPU baseline (rank 0, N = 1 only): the reference's AVX2 kernels on the host cores
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(pool_host, stride, desc, seconds=12.0):
    """Reference AVX2 kernels (oracle/_ref, timing baseline) driven like FullPelSearch_LCU on the host
    cores, one thread per core over disjoint SB ranges; falls back to the repo's C port."""
    from oracle.binding import Oracle, Reference

    pool2d = pool_host[: (pool_host.size // stride) * stride].reshape(-1, stride)   # flat pool viewed with the full-plane stride
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))  # the GPU box gives one GPU job a 16-CPU share
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
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < seconds * 0.3:
        run(sample[:128])
        n1 += 128
    single = n1 / (time.perf_counter() - t0)
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
            "sample": f"picture 1 searched in picture 0 (510 SBs, 64x64 search, 85 PUs) repeated for ~{seconds:.0f} s; "
                      f"{'reference ASM_AVX2 kernels driven like FullPelSearch_LCU' if kind == 'reference' else 'repo C port (oracle)'}; "
                      f"{ncores} threads over disjoint SB ranges"}


def cpu_sad_loop_baseline(seconds=4.0):
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
    if not entry:
        return None
    return round((2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0)


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
                       "bytes/pixel algorithmic = 1 src + 1 pred + 1 recon + 4 qcoeff + 2 iscan = 9",
           "sizes": {}}
    traffic = load_profile_json("r02_pmc_traffic_tq.json") or load_profile_json("r01_pmc_traffic_tq.json") or {}
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
            res[label] = {"n_tu": n_tu, "ms": round(ms, 4), "gpix_per_s": round(px / ms / 1e6, 2), "algorithmic_gbps": round(algo / ms / 1e6, 1),
                          "frac_hbm": round(algo / ms / 1e6 / HBM_PEAK_GBPS, 4)}
            del src, pred, recon, noise, d_q
        key = f"encode_tu_kernel_{n}x{n}"
        if key in traffic:
            res["plane_64Mpx"]["traffic_bytes"] = pmc_traffic_bytes(traffic[key])
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
    out = {"blocks": n, "ms": round(ms, 4), "gpix_per_s": round(n * 4096 / ms / 1e6, 2), "algorithmic_gbps": round(algo / ms / 1e6, 1),
           "frac_hbm": round(algo / ms / 1e6 / HBM_PEAK_GBPS, 4),
           "workload": "8-bit av1_convolve_2d_sr, 64x64 blocks x 225 phases over a 1080p frame; algorithmic bytes = 71 x 71 read + 4096 written per block"}
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
    out["compound_bipred"] = {"ms": round(msc, 4), "gpix_per_s": round(n * 4096 / msc / 1e6, 2), "algorithmic_gbps": round(algoc / msc / 1e6, 1),
                              "frac_hbm": round(algoc / msc / 1e6 / HBM_PEAK_GBPS, 4)}
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
            "workload": "16x16 blocks, 33x33 positions (+-16), 856x480 8-bit, 12 pictures per launch, SadLoopKernel semantics (packed-SAD kernel: 8 positions per lane, v_qsad_pk_u16_u8)"}


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
    ap.add_argument("--gather-results", action="store_true", help="also gather the (sad, mv) results of every step on every rank (svthip_me_gather_results_dev, RCCL)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
        world, rank, local_rank = 1, 0, 0
    else:
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)

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
    comm = sharded.Comm.from_process_group(ctx)      # svthip_comm (RCCL); world 1 never touches RCCL

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
        if args.gather_results:   # svthip_me_gather_results_dev on the same stream: every rank ends with all (SAD, MV) rows
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
        legs["recon_exchange"] = leg_recon_exchange(ctx, comm, torch, dist, svtav1_hip, dev, world)   # every rank takes part
        if rank == 0:
            timer = EventTimer(torch)
            rng = np.random.default_rng(3)
            d_sb_all = torch.from_numpy(sb_all.view(np.int16).copy()).to(dev)
            params_b = svtav1_hip.default_me_params(W, H, 3, 1)
            legs["me_chain_subpel"] = leg_me_chain(ctx, torch, svtav1_hip, timer, pool, pdesc, params_b, d_sb_all, sb_all.shape[0], dev)
            legs["open_loop_intra_search"] = leg_ois(ctx, torch, svtav1_hip, timer, pool, pdesc, params_b, d_sb_all, sb_all.shape[0], dev)
            legs["sad_loop_480p"] = leg_sad_loop(ctx, torch, svtav1_hip, timer, dev)
            legs["convolve_8tap"] = leg_convolve(ctx, torch, svtav1_hip, timer, pool, pdesc, dev)
            legs["tq_chain"] = leg_tq(ctx, torch, svtav1_hip, timer, dev, rng)
            legs["uhd_10bit"] = leg_4k(ctx, torch, svtav1_hip, timer, dev, rng)

    if rank == 0:
        value = total_blocks_per_step * args.steps / elapsed
        achieved = ALGO_BYTES_PER_BLOCK * n_blocks / (kern_ms * 1e-3) / 1e9
        absdiff_rate = ABSDIFF_PER_BLOCK * n_blocks / (kern_ms * 1e-3)
        tr = load_profile_json("r02_pmc_traffic.json") or load_profile_json("r01_pmc_traffic.json") or {}
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
                         "kernel": "fullpel85_kernel", "kernel_ms": round(kern_ms, 4), "kernel_blocks": n_blocks,
                         "algorithmic_bytes_per_block": ALGO_BYTES_PER_BLOCK,
                         "note": "search is ~400 abs-diff per compulsory byte: VALU-bound by construction (SURVEY 8d); see valu",
                         "valu": {"achieved_absdiff_per_s": round(absdiff_rate, 0),
                                  "peak_absdiff_per_s": round(VALU_PEAK_ABSDIFF_PER_S, 0),
                                  "frac": round(absdiff_rate / VALU_PEAK_ABSDIFF_PER_S, 4),
                                  "peak_source": "measured v_qsad_pk_u16_u8 issue rate, tools/ubench_valu.hip"}},
            "legs": legs,
        }
        if not args.no_cpu_baseline and world == 1:
            host_pool = pool[: pdesc[2].full_offset].cpu().numpy()            # pictures 0 and 1
            desc0 = zero_centred_desc([pdesc[1]], [pdesc[0]], sb_all, W, H)
            out["cpu_baseline"] = cpu_baseline(host_pool, stride, desc0)
            if not args.no_legs:
                legs.setdefault("sad_loop_480p", {})["cpu_baseline"] = cpu_sad_loop_baseline()
        print(json.dumps(out), flush=True)
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
