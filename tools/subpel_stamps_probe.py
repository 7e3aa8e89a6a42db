#!/usr/bin/env python3
"""Where does a sub-pel workgroup spend its time?  Needs a variant built with
    tools/build_variant.sh stamps csrc/me_subpel_planes.hip -DSVTHIP_SUBPEL_STAMPS
and SVTAV1_HIP_LIB pointing at it.  One 12 x 1080p launch per PU mode; per-phase medians of wave 0 in shader-clock ticks.  GPU only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import torch  # noqa: E402

import bench  # noqa: E402
import svtav1_hip  # noqa: E402

dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
d_pool, pdesc = bench.device_picture_pool(ctx, 14, 1920, 1080, dev)
stride = pdesc[0].full_stride
sb = svtav1_hip.sb_origins(1920, 1080)
d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
curs = [pdesc[i + 1] for i in range(12)]
refs = [pdesc[i] for i in range(12)]
P = svtav1_hip.default_me_params(1920, 1080, 3, 1)
n = 6120
d_desc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, refs, P, 0, d_sb.data_ptr(), 510, None, d_desc.data_ptr())
f = svtav1_hip.lib().svthip_debug_subpel_stamps
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_size_t]
names = ["vectors -> bounding box", "window A (global -> LDS) + barrier", "b, h + barrier", "j + barrier", "PU phase (wave 0's tasks)"]
for n_pu, fp, sub in ((85, ctx.fullpel_search_dev, ctx.subpel_refine_dev), (209, ctx.fullpel_search209_dev, ctx.subpel_refine209_dev)):
    d_sad = torch.empty((n, n_pu), dtype=torch.int32, device=dev)
    d_mv = torch.empty_like(d_sad)
    fp(d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr())
    ctx.synchronize()
    s0, m0 = d_sad.clone(), d_mv.clone()
    for _ in range(2):
        d_sad.copy_(s0); d_mv.copy_(m0)
        torch.cuda.synchronize()
        sub(d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr(), False)
        ctx.synchronize()
    st = np.zeros((n, 8), np.uint64)
    assert f(st.ctypes.data, st.nbytes) == 0
    d = np.diff(st[:, :6].astype(np.int64), axis=1)
    print(f"== {n_pu} PUs: launch span {int(st[:, 5].max() - st[:, 0].min())} ticks, {n} workgroups")
    for i, nm in enumerate(names):
        print(f"  {nm:38s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
    fw = svtav1_hip.lib().svthip_debug_subpel_wave_end
    fw.restype = C.c_int
    fw.argtypes = [C.c_void_p, C.c_size_t]
    we = np.zeros((n, 8), np.uint64)
    assert fw(we.ctypes.data, we.nbytes) == 0
    nw = 7 if n_pu == 209 else 8
    pu = we[:, :nw].astype(np.int64) - st[:, 4:5].astype(np.int64)
    print("  PU phase per wave (median ticks from the end of the plane phases): " + " ".join(f"{np.median(pu[:, w]):.0f}" for w in range(nw)))
    print(f"  {'whole workgroup (wave 0)':38s} median {np.median(st[:, 5].astype(np.int64) - st[:, 0].astype(np.int64)):8.0f}")
