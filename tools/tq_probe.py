#!/usr/bin/env python3
"""Times the batched forward transform (and quantiser) per transform size on cuda:0.  Inputs resident in HBM; each launch
is timed with events on the stream the kernel is launched on.  Prints coefficient rate and the algorithmic-bytes rate
(2 B residual in + 4 B coefficient out per coefficient).  Usage: python tools/tq_probe.py [--coeffs N] [--iters K]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import torch  # noqa: E402
import svtav1_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--coeffs", type=int, default=1 << 23)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sizes", type=str, default="")
    args = ap.parse_args()
    ctx = svtav1_hip.Context(0)
    tstream = torch.cuda.Stream()            # a real (non-null) stream: a null stream argument means "the context's stream"
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    pic_w, pic_h = 1920, 1088
    rng = np.random.default_rng(1)
    res = torch.from_numpy(rng.integers(-255, 256, pic_w * pic_h).astype(np.int16)).to("cuda:0")
    sizes = svtav1_hip.TX_SIZES_WH
    if args.sizes:
        sizes = [tuple(int(v) for v in s.split("x")) for s in args.sizes.split(",")]
    for (w, h) in sizes:
        n_tu = args.coeffs // (w * h)
        # TUs tile the picture in raster order (wrapping), DCT_DCT, like an encode pass over a frame
        per_row = pic_w // w
        idx = np.arange(n_tu)
        x0 = (idx % per_row) * w
        y0 = ((idx // per_row) * h) % (pic_h - h + 1)
        desc = np.zeros(n_tu, dtype=svtav1_hip.TXFM_DESC_DTYPE)
        desc["in_offset"] = y0 * pic_w + x0
        desc["out_offset"] = idx * (w * h)
        desc["in_stride"] = pic_w
        desc["tx_type"] = 0
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        d_out = torch.empty(n_tu * w * h, dtype=torch.int32, device="cuda:0")
        for _ in range(3):
            ctx.fwd_txfm2d_batch_dev(res.data_ptr(), d_desc.data_ptr(), n_tu, w, h, 8, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ctx.fwd_txfm2d_batch_dev(res.data_ptr(), d_desc.data_ptr(), n_tu, w, h, 8, d_out.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        nc = n_tu * w * h
        print(f"fwd_txfm {w:2d}x{h:2d}  n_tu {n_tu:7d}  {ms:8.4f} ms  {nc / ms / 1e6:8.2f} Gcoeff/s  {nc * 6 / ms / 1e6:8.1f} GB/s algorithmic",
              flush=True)


if __name__ == "__main__":
    main()
