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
    ap.add_argument("--coeffs", type=int, default=1 << 26)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sizes", type=str, default="")
    args = ap.parse_args()
    ctx = svtav1_hip.Context(0)
    tstream = torch.cuda.Stream()            # a real (non-null) stream: a null stream argument means "the context's stream"
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    pic_w = 8192
    pic_h = max(64, (args.coeffs // pic_w) // 64 * 64)       # one residual / prediction plane holding every TU exactly once
    rng = np.random.default_rng(1)
    res = torch.from_numpy(rng.integers(-255, 256, pic_w * pic_h, dtype=np.int16)).to("cuda:0")
    pred = torch.from_numpy(rng.integers(0, 256, pic_w * pic_h, dtype=np.uint8)).to("cuda:0")
    src8 = torch.from_numpy(rng.integers(0, 256, pic_w * pic_h, dtype=np.uint8)).to("cuda:0")
    qp = np.zeros((1, 10), np.int16)
    for i, q in enumerate((40, 52)):   # one Quants / Dequants row, built like av1_build_quantizer
        l = int(q).bit_length() - 1
        qp[0, 4 + i] = np.int16(np.uint16((1 + (1 << (16 + l)) // q - (1 << 16)) & 0xffff)); qp[0, 6 + i] = 1 << (16 - l)
        qp[0, 0 + i] = (84 * q + 64) >> 7; qp[0, 2 + i] = (64 * q) >> 7; qp[0, 8 + i] = q
    d_qp = torch.from_numpy(qp).to("cuda:0")
    d_iscan = torch.from_numpy(np.arange(1024, dtype=np.int16)).to("cuda:0")
    sizes = svtav1_hip.TX_SIZES_WH
    if args.sizes:
        sizes = [tuple(int(v) for v in s.split("x")) for s in args.sizes.split(",")]

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    for (w, h) in sizes:
        per_row = pic_w // w
        n_tu = per_row * (pic_h // h)
        idx = np.arange(n_tu)
        off = (idx // per_row) * h * pic_w + (idx % per_row) * w
        desc = np.zeros(n_tu, dtype=svtav1_hip.TXFM_DESC_DTYPE)
        desc["in_offset"], desc["out_offset"], desc["in_stride"] = off, idx * (w * h), pic_w
        win, hin = min(w, 32), min(h, 32)
        idesc = np.zeros(n_tu, dtype=svtav1_hip.ITXFM_DESC_DTYPE)
        idesc["coeff_offset"], idesc["recon_offset"], idesc["recon_stride"] = idx * (win * hin), off, pic_w
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        d_idesc = torch.from_numpy(idesc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        d_out = torch.empty(n_tu * w * h, dtype=torch.int32, device="cuda:0")
        nc = n_tu * w * h
        ms = timed(lambda: ctx.fwd_txfm2d_batch_dev(res.data_ptr(), d_desc.data_ptr(), n_tu, w, h, 8, d_out.data_ptr(), stream))
        print(f"fwd_txfm {w:2d}x{h:2d}  n_tu {n_tu:8d}  {ms:8.4f} ms  {nc / ms / 1e6:8.2f} Gcoeff/s  {nc * 6 / ms / 1e6:8.1f} GB/s algorithmic",
              flush=True)
        d_out >>= 3                                           # keep the inverse input in a plausible dequantised range
        ms = timed(lambda: ctx.inv_txfm2d_add_batch_dev(d_out.data_ptr(), d_idesc.data_ptr(), n_tu, w, h, 8, False, pred.data_ptr(),
                                                        stream))
        ib = n_tu * (win * hin * 4 + 2 * w * h)               # coefficients in + prediction read + reconstruction written
        print(f"inv_txfm {w:2d}x{h:2d}  n_tu {n_tu:8d}  {ms:8.4f} ms  {nc / ms / 1e6:8.2f} Gpix/s    {ib / ms / 1e6:8.1f} GB/s algorithmic",
              flush=True)
        # fused residual -> transform -> quantise -> dequantise -> inverse -> reconstruct (8-bit), in place on the prediction plane
        tdesc = np.zeros(n_tu, dtype=svtav1_hip.TU_DESC_DTYPE)
        tdesc["src_offset"], tdesc["pred_offset"], tdesc["recon_offset"] = off, off, off
        tdesc["coeff_offset"] = idx * (win * hin)
        tdesc["src_stride"], tdesc["pred_stride"], tdesc["recon_stride"] = pic_w, pic_w, pic_w
        d_tdesc = torch.from_numpy(tdesc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        d_q = torch.empty(n_tu * win * hin, dtype=torch.int32, device="cuda:0")
        d_eob = torch.empty(n_tu, dtype=torch.int16, device="cuda:0")
        ms = timed(lambda: ctx.encode_tu_batch_dev(src8.data_ptr(), pred.data_ptr(), pred.data_ptr(), d_tdesc.data_ptr(), n_tu, w, h,
                                                   d_qp.data_ptr(), d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(),
                                                   None, None, stream))
        eb = n_tu * (2 * w * h + w * h + win * hin * (2 + 4))    # source + prediction in, reconstruction out, iscan in, qcoeff out
        print(f"encode_tu {w:2d}x{h:2d} n_tu {n_tu:8d}  {ms:8.4f} ms  {nc / ms / 1e6:8.2f} Gpix/s    {eb / ms / 1e6:8.1f} GB/s algorithmic",
              flush=True)
        del d_out, d_q
    # stand-alone quantiser: 1024-coefficient TUs (32x32 class), 14 B per coefficient (4 in, 2 iscan, 4 + 4 out)
    n_tu = args.coeffs // 1024
    qd = np.zeros(n_tu, dtype=svtav1_hip.QUANT_DESC_DTYPE)
    qd["coeff_offset"] = np.arange(n_tu) * 1024
    qd["n_coeffs"] = 1024
    qd["log_scale"] = 1
    d_qd = torch.from_numpy(qd.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    d_c = torch.from_numpy(rng.integers(-2000, 2001, n_tu * 1024, dtype=np.int32)).to("cuda:0")
    d_qo = torch.empty_like(d_c); d_dq = torch.empty_like(d_c)
    d_eob = torch.empty(n_tu, dtype=torch.int16, device="cuda:0")
    ms = timed(lambda: ctx.quantize_b_batch_dev(d_c.data_ptr(), d_qd.data_ptr(), n_tu, d_qp.data_ptr(), d_iscan.data_ptr(), d_qo.data_ptr(),
                                                d_dq.data_ptr(), d_eob.data_ptr(), stream))
    nc = n_tu * 1024
    print(f"quantize_b 1024-coeff TUs  n_tu {n_tu:8d}  {ms:8.4f} ms  {nc / ms / 1e6:8.2f} Gcoeff/s  {nc * 14 / ms / 1e6:8.1f} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
