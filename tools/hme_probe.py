#!/usr/bin/env python3
"""Times the batched search-centre kernel with HME levels switched off one by one (where does its time go)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import numpy as np, torch
import svtav1_hip, bench

dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
d_pool, pdesc = bench.device_picture_pool(ctx, 13, 1920, 1080, dev)
d_desc = torch.zeros(12 * 510 * 6, dtype=torch.int32, device=dev)
d_sb = torch.from_numpy(svtav1_hip.sb_origins(1920, 1080).view(np.int16).copy()).to(dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
curs = [pdesc[i + 1] for i in range(12)]; refs = [pdesc[i] for i in range(12)]
for name, flags in (("all levels", (1, 1, 1, 1)), ("L0 + L1", (1, 1, 1, 0)), ("L0 only", (1, 1, 0, 0)), ("centre check only", (0, 0, 0, 0))):
    P = svtav1_hip.default_me_params(1920, 1080, 3, 0)
    P.enable_hme_flag, P.enable_hme_level0_flag, P.enable_hme_level1_flag, P.enable_hme_level2_flag = flags
    f = lambda: ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, refs, P, 0, d_sb.data_ptr(), 510, None, d_desc.data_ptr(), stream=ts.cuda_stream)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per 6120 SBs", flush=True)
