#!/usr/bin/env python3
"""Fused T/Q chain on the 64 M-pixel plane under different descriptor orders / transform-type mixes (GPU only):
   raster_dct     raster order, every TU DCT_DCT            (no divergence, every line fetched once)
   raster_mixed   raster order, random DCT / ADST / IDTX    (mixed waves, every line fetched once)
   type_sorted    sorted by transform type                   (uniform waves, a line is fetched once per type bucket)
   tile_sorted:T  raster chunks of T TUs, sorted by type inside a chunk
usage: python tools/tq_order_probe.py [sizes...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import svtav1_hip  # noqa: E402

dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
timer = bench.StreamTimer(torch) if hasattr(bench, "StreamTimer") else None
z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
qrows = np.ascontiguousarray(z["rows_bd8_inter"][[20, 120, 200], 0, :])
d_qp = torch.from_numpy(qrows).to(dev)
d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
sizes = [int(s) for s in sys.argv[1:]] or [8, 16, 32]
pic_w = pic_h = 8192
ts_ = torch.cuda.Stream()
torch.cuda.set_stream(ts_)
S = ts_.cuda_stream
src = torch.randint(0, 256, (pic_w * pic_h,), dtype=torch.uint8, device=dev)
noise = torch.empty(pic_w * pic_h, device=dev).exponential_(1 / 6.0) * (torch.randint(0, 2, (pic_w * pic_h,), device=dev) * 2 - 1)
pred = (src.float() - noise).clamp_(0, 255).to(torch.uint8)
recon = torch.empty_like(pred)
del noise
for n in sizes:
    tsz = svtav1_hip.TX_SIZES_WH.index((n, n))
    types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(n, n)]
    isc = {t: int(z["scan_offsets"][int(z["scan_index"][tsz, t])]) for t in types}
    for order in ("raster_dct", "raster_mixed", "type_sorted", "tile_sorted:64", "tile_sorted:256", "tile_sorted:1024"):
        rng = np.random.default_rng(5)
        d, nc = bench.tile_tu_desc(svtav1_hip, n, n, pic_w, pic_h, isc, types if order != "raster_dct" else [0], 3, rng)
        d = d[np.argsort(d["coeff_offset"], kind="stable")]   # back to raster
        if order == "type_sorted":
            d = d[np.argsort(d["tx_type"], kind="stable")]
        elif order.startswith("tile_sorted"):
            T = int(order.split(":")[1])
            key = (np.arange(len(d)) // T).astype(np.int64) * 16 + d["tx_type"]
            d = d[np.argsort(key, kind="stable")]
        n_tu = len(d)
        d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
        d_q = torch.empty(n_tu * nc, dtype=torch.int32, device=dev)
        d_eob = torch.empty(n_tu, dtype=torch.int16, device=dev)

        def run():
            ctx.encode_tu_batch_dev(src.data_ptr(), pred.data_ptr(), recon.data_ptr(), d_desc.data_ptr(), n_tu, n, n, d_qp.data_ptr(),
                                    d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None, S)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts_)
        for _ in range(5):
            run()
        e1.record(ts_)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{n:2d}x{n:<2d} {order:18s} {ms:7.4f} ms   {n_tu * n * n * 9 / ms / 1e9:6.2f} TB/s algorithmic", flush=True)
        del d_q, d_desc
