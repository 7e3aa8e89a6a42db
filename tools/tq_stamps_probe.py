#!/usr/bin/env python3
"""Phase time stamps of the fused T/Q chain (square size n, 64 M-pixel plane), from a build with -DSVTHIP_TQ_STAMPS
(tools/build_variant.sh stamps csrc/tq_encode_tu.hip -DSVTHIP_TQ_STAMPS; SVTAV1_HIP_LIB=.../variants/libsvtav1_hip_stamps.so).
Per group of a wave: top (operands of the group taken over, next descriptor requested) + A0 | A | next rows requested | B (forward row, quantiser,
inverse row) | staged coefficient stores + C | D.
usage: python tools/tq_stamps_probe.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import svtav1_hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
z = np.load(os.path.join(ROOT, "tests", "golden", "quant_tables.npz"))
d_qp = torch.from_numpy(np.ascontiguousarray(z["rows_bd8_inter"][[20, 120, 200], 0, :])).to(dev)
d_iscan = torch.from_numpy(z["iscan_pool"]).to(dev)
pic_w = pic_h = 8192
tsz = svtav1_hip.TX_SIZES_WH.index((n, n))
types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(n, n)]
isc = {t: int(z["scan_offsets"][int(z["scan_index"][tsz, t])]) for t in types}
d, nc = bench.tile_tu_desc(svtav1_hip, n, n, pic_w, pic_h, isc, types, 3, np.random.default_rng(5))
n_tu = len(d)
src = torch.randint(0, 256, (pic_w * pic_h,), dtype=torch.uint8, device=dev)
pred = (src.float() + torch.randn(pic_w * pic_h, device=dev) * 6).clamp_(0, 255).to(torch.uint8)
recon = torch.empty_like(pred)
d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)
d_q = torch.empty(n_tu * nc, dtype=torch.int32, device=dev)
d_c = torch.zeros(n_tu * nc, dtype=torch.int32, device=dev)
d_eob = torch.empty(n_tu, dtype=torch.int16, device=dev)
for _ in range(3):
    ctx.encode_tu_batch_dev(src.data_ptr(), pred.data_ptr(), recon.data_ptr(), d_desc.data_ptr(), n_tu, n, n, d_qp.data_ptr(), d_iscan.data_ptr(),
                            d_c.data_ptr(), d_q.data_ptr(), None, d_eob.data_ptr(), None, None)
ctx.synchronize()
G = 64 // n
groups = (n_tu + G - 1) // G
st = d_c.cpu().numpy().view(np.uint64)[:groups * 8].reshape(groups, 8)
ok = st[:, 0] > 0
st = st[ok]
print(f"{n}x{n}: {groups} groups, {ok.sum()} stamped; kernel span {(st[:, 6].max() - st[:, 0].min())} ticks")
names = ["top + A0", "A", "prefetch", "B", "flush + C", "D"]
dt = np.diff(st[:, :7].astype(np.int64), axis=1)
for k in range(6):
    print(f"  {names[k]:10s} mean {dt[:, k].mean():9.0f}  p10 {np.percentile(dt[:, k], 10):9.0f}  p50 {np.percentile(dt[:, k], 50):9.0f}  p90 {np.percentile(dt[:, k], 90):9.0f}")
tot = (st[:, 6] - st[:, 0]).astype(np.int64)
print(f"  group     mean {tot.mean():9.0f}  p10 {np.percentile(tot, 10):9.0f}  p50 {np.percentile(tot, 50):9.0f}  p90 {np.percentile(tot, 90):9.0f}")
wid = st[:, 7]
order = np.lexsort((st[:, 0], wid))
s2 = st[order]
same = s2[1:, 7] == s2[:-1, 7]
gap = (s2[1:, 0].astype(np.int64) - s2[:-1, 6].astype(np.int64))[same]
print(f"  between groups of a wave: mean {gap.mean():9.0f} p50 {np.percentile(gap, 50):9.0f} p90 {np.percentile(gap, 90):9.0f}; groups per wave {len(st) / len(np.unique(wid)):.1f}")
w0 = np.array([s2[s2[:, 7] == w][0, 0] for w in np.unique(wid)[:2000]]).astype(np.int64)
w1 = np.array([s2[s2[:, 7] == w][-1, 6] for w in np.unique(wid)[:2000]]).astype(np.int64)
print(f"  first start spread {w0.max() - w0.min()}  last end spread {w1.max() - w1.min()}  mean wave span {(w1 - w0).mean():.0f}")
