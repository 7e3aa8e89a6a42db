#!/usr/bin/env python3
"""Per-kernel launch times of the ME path on 12 x 1080p pictures (6120 superblocks per launch), HIP events on the launch stream.
GPU only.  usage: python tools/kernel_times.py [what ...]   what in: fp85 fp209 hme sub85 sub209 chain85 chain209 sad (default: all)
Set SVTAV1_HIP_LIB=<path to an experimental libsvtav1_hip.so> to time another build (one process per build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import svtav1_hip  # noqa: E402

what = sys.argv[1:] or ["fp85", "fp209", "hme", "sub85", "sub209", "chain85", "chain209"]
dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
N_PIC, W, H = 12, 1920, 1080
d_pool, pdesc = bench.device_picture_pool(ctx, N_PIC + 2, W, H, dev)
stride = pdesc[0].full_stride
sb = svtav1_hip.sb_origins(W, H)
n_sb = sb.shape[0]
n = N_PIC * n_sb
d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
curs = [pdesc[i + 1] for i in range(N_PIC)]
r0 = [pdesc[i] for i in range(N_PIC)]
r1 = [pdesc[i + 2] for i in range(N_PIC)]
P = svtav1_hip.default_me_params(W, H, 3, 1)
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
S = ts.cuda_stream


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        for _ in range(iters):
            fn()
        e1.record(ts)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


print(f"library: {svtav1_hip.LIB_PATH}", flush=True)
d_desc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, r0, P, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr(), None, None, S)
torch.cuda.synchronize()
out = {}
if "hme" in what:
    out["hme_center (6120 SBs)"] = timed(lambda: ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, r0, P, 0, d_sb.data_ptr(), n_sb, None, d_desc.data_ptr(),
                                                                               None, None, S))
for n_pu, key in ((85, "85"), (209, "209")):
    d_sad = torch.empty((n, n_pu), dtype=torch.int32, device=dev)
    d_mv = torch.empty_like(d_sad)
    a = (d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr(), S)
    fp = ctx.fullpel_search209_dev if n_pu == 209 else ctx.fullpel_search_dev
    if "fp" + key in what:
        out[f"fullpel{key} (6120 SBs)"] = timed(lambda: fp(*a))
    if "sub" + key in what:
        fp(*a)
        torch.cuda.synchronize()
        s0, m0 = d_sad.clone(), d_mv.clone()
        sub = ctx.subpel_refine209_dev if n_pu == 209 else ctx.subpel_refine_dev

        def run_sub():
            d_sad.copy_(s0, non_blocking=True)   # the refinement is in place: restore the full-pel results (two small copies, timed separately)
            d_mv.copy_(m0, non_blocking=True)
            sub(d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr(), False, S)

        def run_copies():
            d_sad.copy_(s0, non_blocking=True)
            d_mv.copy_(m0, non_blocking=True)

        out[f"subpel{key} (6120 SBs, one list)"] = timed(run_sub) - timed(run_copies)
    if "chain" + key in what:
        d_out = torch.zeros((n, n_pu, 24), dtype=torch.uint8, device=dev)
        entry = ctx.motion_estimate209_batch_dev if n_pu == 209 else ctx.motion_estimate_batch_dev
        for two in (True, False):
            out[f"chain{key} {'B' if two else 'P'} picture (ms per picture)"] = timed(
                lambda: entry(d_pool.data_ptr(), curs, r0, r1 if two else None, P, d_sb.data_ptr(), n_sb, d_out.data_ptr(), True, 0, None, None, S), 10) / N_PIC
for k, v in out.items():
    print(f"{k:48s} {v * 1e3:9.1f} us", flush=True)
