import sys, os, ctypes as C
sys.path.insert(0,'svt-av1-1_amd/python'); sys.path.insert(0,'.')
import numpy as np, torch, svtav1_hip
from oracle.binding import Oracle
o = Oracle(); ctx = svtav1_hip.Context(0)
rng = np.random.default_rng(1)
S=256; R=128
src = rng.integers(0,256,(R,S),dtype=np.uint8)
for D, dx in ((512,0),(515,0),(512,1),(512,2)):
    w=h=32
    desc = np.zeros(4, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    for i in range(4):
        desc[i] = ((20+i)*S+40+i, (i*40)*D + dx + 64*0, 5, 7, 0, 0, 0)
    dst = np.zeros((200,D),np.uint8); want = dst.copy()
    f = o.lib.orc_av1_convolve_sr_batch; f.restype=None
    f.argtypes=[C.c_void_p,C.c_int32,C.c_void_p,C.c_int32,C.c_void_p,C.c_uint32,C.c_int32,C.c_int32]
    dd = np.zeros((4,4),np.uint32); dd[:,0]=desc["src_offset"]; dd[:,1]=desc["dst_offset"]; dd[:,2]=5|(7<<8)
    f(src.ctypes.data,S,want.ctypes.data,D,dd.ctypes.data,4,w,h)
    d_src=torch.from_numpy(np.concatenate([src.reshape(-1),np.zeros(64,np.uint8)])).to("cuda:0")
    d_dst=torch.from_numpy(dst.reshape(-1).copy()).to("cuda:0"); d_desc=torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    ctx.av1_convolve_sr_batch_dev(d_src.data_ptr(),S,d_dst.data_ptr(),D,d_desc.data_ptr(),4,w,h); ctx.synchronize()
    got=d_dst.cpu().numpy().reshape(dst.shape)
    bad=np.argwhere(got!=want)
    print("D",D,"dx",dx,"bad",len(bad), bad[:6].tolist())
    if len(bad):
        y,x=bad[0]; print(" got",got[y,max(0,x-4):x+8].tolist()," want",want[y,max(0,x-4):x+8].tolist())
