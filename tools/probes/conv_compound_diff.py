#!/usr/bin/env python3
"""Where do the matrix-core and the VALU compound predictions differ (development probe, GPU only)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import torch
import svtav1_hip
from svtav1_hip import synth
ctx = svtav1_hip.Context(0)
w = h = 64
pics = [synth.PaPicture(synth.synth_luma(1920, 1080, t)) for t in (1, 4)]
S = pics[0].full.shape[1]
nbx, nby, n_ph = 30, 16, 24
n = nbx * nby * n_ph
rng = np.random.default_rng(12)
i = np.arange(n)
blk, ph = i // n_ph, i % n_ph
base = (68 + (blk // nbx) * 64) * S + 68 + (blk % nbx) * 64
mvx, mvy = rng.integers(-30, 31, n), rng.integers(-30, 31, n)
c = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
sx, sy = rng.integers(0, 16, n), rng.integers(0, 16, n)
sx[ph == 0] = 0; sy[ph == 0] = 0
fx, fy = rng.integers(0, 4, n), rng.integers(0, 4, n)
c["src0_offset"], c["src1_offset"], c["dst_offset"] = base + mvy * S + mvx, base - mvy * S - mvx, i * 4096
c["subpel0"] = sx | (sy << 4)
c["subpel1"] = rng.integers(0, 256, n)
c["filter_x"], c["filter_y"] = fx, fy
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")
pad = np.zeros(256, np.uint8)
d_s0, d_s1 = dev(np.concatenate([pics[0].full.reshape(-1), pad])), dev(np.concatenate([pics[1].full.reshape(-1), pad]))
d_c = dev(c)
def run():
    out = torch.zeros(n * 4096, dtype=torch.uint8, device="cuda:0")
    ctx.av1_convolve_compound_batch_dev(d_s0.data_ptr(), S, d_s1.data_ptr(), S, out.data_ptr(), 64, d_c.data_ptr(), n, w, h)
    ctx.synchronize()
    return out.cpu().numpy().reshape(n, 64, 64)
a = run()
ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 1)
b = run()
bad = np.argwhere(a != b)
print("mismatching pixels", len(bad), "blocks", len(np.unique(bad[:, 0])))
if len(bad):
    ub = np.unique(bad[:, 0])[:8]
    for k in ub:
        m = (a[k] != b[k])
        rows, cols = np.where(m)
        print("block", k, "subpel0", hex(int(c['subpel0'][k])), "subpel1", hex(int(c['subpel1'][k])), "rows", rows.min(), rows.max(), "cols", cols.min(), cols.max(), "count", m.sum(),
              "sample mfma/valu", a[k][rows[0], cols[0]], b[k][rows[0], cols[0]])
    s1 = c['subpel1'][np.unique(bad[:, 0])]
    s0 = c['subpel0'][np.unique(bad[:, 0])]
    print("subpel1 x==0:", np.mean((s1 & 15) == 0), "y==0:", np.mean((s1 >> 4) == 0), " subpel0 x==0:", np.mean((s0 & 15) == 0), "y==0:", np.mean((s0 >> 4) == 0))
