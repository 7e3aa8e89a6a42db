#!/usr/bin/env python3
"""Where does a search-centre workgroup spend its time?  Builds a copy of the library with -DSVTHIP_HME_STAMPS (s_memtime of every region
wave at the phase boundaries of hme_center_sb), runs one 12 x 1080p launch and prints the per-phase medians in microseconds.
GPU box only:  python tools/hme_stamps_probe.py"""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
SO = "/tmp/libsvtav1_hip_stamps.so"
if not os.path.exists(SO):
    srcs = sorted(glob.glob(os.path.join(ROOT, "svt-av1-1_amd", "csrc", "*.hip")))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DSVTHIP_HME_STAMPS", "-shared", *srcs, "-o", SO])
import torch  # noqa: E402
import svtav1_hip  # noqa: E402
svtav1_hip.LIB_PATH = SO
import bench  # noqa: E402

dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
d_pool, pdesc = bench.device_picture_pool(ctx, 13, 1920, 1080, dev)
d_desc = torch.zeros(12 * 510 * 6, dtype=torch.int32, device=dev)
d_sb = torch.from_numpy(svtav1_hip.sb_origins(1920, 1080).view(np.int16).copy()).to(dev)
curs = [pdesc[i + 1] for i in range(12)]; refs = [pdesc[i] for i in range(12)]
P = svtav1_hip.default_me_params(1920, 1080, 3, 0)
for _ in range(3):
    ctx.hme_search_center_batch_dev(d_pool.data_ptr(), curs, refs, P, 0, d_sb.data_ptr(), 510, None, d_desc.data_ptr())
ctx.synchronize()
n = 6120
st = np.zeros((n, 4, 8), np.uint64)
f = svtav1_hip.lib().svthip_debug_hme_stamps
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_size_t]
assert f(st.ctypes.data, st.nbytes) == 0
clk = 100e6   # s_memtime ticks at the 100 MHz reference clock on this part when it is not the shader clock: report both readings
d = np.diff(st[:, :, :7].astype(np.int64), axis=2)
names = ["source staging", "centre check", "level 0", "level 1", "level 2", "region pick .. end"]
full = (st[:, 0, 6] > 0)
print("workgroups with stamps:", int(full.sum()))
for i, nm in enumerate(names):
    v = d[full][:, :, i].reshape(-1)
    print(f"{nm:22s} median {np.median(v):9.0f} ticks   p10 {np.percentile(v, 10):9.0f}   p90 {np.percentile(v, 90):9.0f}")
tot = (st[full][:, :, 6].astype(np.int64) - st[full][:, :, 0].astype(np.int64)).reshape(-1)
print(f"{'whole chain':22s} median {np.median(tot):9.0f} ticks")
span = int(st[full][:, :, 6].max() - st[full][:, :, 0].min())
print("launch span", span, "ticks")

# sub-phases of the level-1 / level-2 SAD loop (wave_sad_loop_lds), averaged per call
ph = np.zeros(8, np.uint64)
g = svtav1_hip.lib().svthip_debug_hme_loop_phases
g.restype = C.c_int
g.argtypes = [C.c_void_p]
assert g(ph.ctypes.data) == 0
for lvl, o in (("level 1 (W = 32)", 0), ("level 2 (W = 64)", 4)):
    calls = max(int(ph[o + 3]), 1)
    print(f"{lvl}: per call: window staging {int(ph[o]) / calls:8.0f}  search loop {int(ph[o + 1]) / calls:8.0f}  wave minimum + decode {int(ph[o + 2]) / calls:8.0f} ticks ({calls} calls)")
