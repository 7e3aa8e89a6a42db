#!/usr/bin/env python3
"""Times the whole-picture ME entry (svthip_motion_estimate_picture_dev: HME -> full-pel -> sub-pel per list, bi-prediction,
packing) on one synthetic 1080p B picture; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import torch  # noqa: E402
import svtav1_hip  # noqa: E402
from svtav1_hip import synth  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    w, h = 1920, 1080
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (3, 0, 7)]
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(w, h)
    n = sb.shape[0]
    dev = torch.device("cuda:0")
    d_pool = torch.from_numpy(pool).to(dev)
    d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
    d_out = torch.zeros((n, 85, 24), dtype=torch.uint8, device=dev)
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    ctx = svtav1_hip.Context(0)
    for two_lists, subpel in ((True, True), (False, True), (True, False)):
        for _ in range(2):
            ctx.motion_estimate_picture_dev(d_pool.data_ptr(), descs[0], descs[1], descs[2] if two_lists else None, P, d_sb.data_ptr(), n,
                                            d_out.data_ptr(), subpel, 0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.motion_estimate_picture_dev(d_pool.data_ptr(), descs[0], descs[1], descs[2] if two_lists else None, P, d_sb.data_ptr(), n,
                                            d_out.data_ptr(), subpel, 0)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        print(f"1080p {'B' if two_lists else 'P'} picture, sub-pel {'on' if subpel else 'off'}: {ms:.3f} ms per picture "
              f"({n / ms * 1e3:.0f} SB/s)", flush=True)


def batch(n_pic=12, iters=10, n_pu=85):
    w, h = 1920, 1080
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in range(n_pic + 2)]
    pool, pd = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(w, h)
    n = sb.shape[0]
    dev = torch.device("cuda:0")
    d_pool = torch.from_numpy(pool).to(dev)
    d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
    d_out = torch.zeros((n_pic * n, n_pu, 24), dtype=torch.uint8, device=dev)
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    ctx = svtav1_hip.Context(0)
    entry = ctx.motion_estimate209_batch_dev if n_pu == 209 else ctx.motion_estimate_batch_dev
    curs = [pd[i + 1] for i in range(n_pic)]; r0 = [pd[i] for i in range(n_pic)]; r1 = [pd[i + 2] for i in range(n_pic)]
    for two in (True, False):
        for _ in range(2):
            entry(d_pool.data_ptr(), curs, r0, r1 if two else None, P, d_sb.data_ptr(), n, d_out.data_ptr(), True, 0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            entry(d_pool.data_ptr(), curs, r0, r1 if two else None, P, d_sb.data_ptr(), n, d_out.data_ptr(), True, 0)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        print(f"batch of {n_pic} 1080p {'B' if two else 'P'} pictures, {n_pu} PUs, sub-pel on: {ms / n_pic:.3f} ms per picture "
              f"({n_pic * n / ms * 1e3:.0f} SB/s)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] in ("batch", "batch209"):
        batch(12, int(sys.argv[1]), 209 if sys.argv[2] == "batch209" else 85)
        sys.exit(0)
    main()
