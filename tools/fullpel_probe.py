"""Times the full-pel kernel at several batch sizes (tail / occupancy effects).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import numpy as np, torch
import svtav1_hip
import bench

dev = torch.device("cuda:0")
ctx = svtav1_hip.Context(0)
d_pool, pdesc = bench.device_picture_pool(ctx, 13, 1920, 1080, dev)
stride = pdesc[0].full_stride
desc = bench.zero_centred_desc(pdesc[1:13], pdesc[0:12], svtav1_hip.sb_origins(1920, 1080), 1920, 1080)
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1536, 3072, 4080, 4608, 6120]
for n in sizes:
    d = desc[:n]
    d_desc = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    d_sad = torch.empty((n, 85), dtype=torch.int32, device=dev); d_mv = torch.empty_like(d_sad)
    a = (d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr())
    ctx.fullpel_search_time_dev(*a, 3)
    ms = min(ctx.fullpel_search_time_dev(*a, 20) for _ in range(3))
    print(f"n_sb {n:5d}: {ms*1e3:8.1f} us  {n/ms/1e3:7.2f} M blocks/s  {ms*1e3/n*768:7.1f} us per 768 blocks", flush=True)

# 209-PU mode (squares + rectangles) on the same descriptors
n = 6120
d_desc = torch.from_numpy(np.ascontiguousarray(desc[:n])).to(dev)
d_sad = torch.empty((n, 209), dtype=torch.int32, device=dev); d_mv = torch.empty_like(d_sad)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
args = (d_pool.data_ptr(), stride, d_pool.data_ptr(), stride, d_desc.data_ptr(), n, 64, 64, d_sad.data_ptr(), d_mv.data_ptr(), ts.cuda_stream)
for _ in range(3):
    ctx.fullpel_search209_dev(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ctx.fullpel_search209_dev(*args)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"209-PU mode n_sb {n}: {ms*1e3:8.1f} us  {n/ms/1e3:7.2f} M blocks/s", flush=True)
