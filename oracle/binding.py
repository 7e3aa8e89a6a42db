"""ctypes bindings for the CPU oracle (oracle/libsvtoracle.so) and, when present, the reference's own
kernels (oracle/_ref/libsvtref_kernels.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsvtoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsvtref_kernels.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


def build_oracle():
    subprocess.check_call(["make", "-C", HERE, "-s"])


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.orc_nxm_sad.restype = C.c_uint32
        L.orc_nxm_sad.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_sad_loop_kernel.restype = None
        L.orc_sad_loop_kernel.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                                          C.c_uint32, C.c_int16, C.c_int16]
        L.orc_fullpel_search_85pu.restype = None
        L.orc_fullpel_search_85pu.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int16, C.c_int16, C.c_uint32,
                                              C.c_uint32, u32p, u32p]
        L.orc_fullpel_search_batch.restype = None
        L.orc_fullpel_search_batch.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, i32p, C.c_uint32, u32p, u32p]

    def fullpel_search_batch(self, src_plane, ref_plane, desc):
        """desc: int32 [n,6] = src_offset, ref_offset, x_origin, y_origin, sw, sh -> (sad[n,85], mv[n,85])."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.empty((n, 85), dtype=np.uint32)
        mv = np.empty((n, 85), dtype=np.uint32)
        self.lib.orc_fullpel_search_batch(_ptr(src_plane, u8p), src_plane.shape[1], _ptr(ref_plane, u8p),
                                          ref_plane.shape[1], _ptr(desc, i32p), n, _ptr(sad, u32p), _ptr(mv, u32p))
        return sad, mv

    def sad_loop(self, src, src_off, src_stride, ref, ref_off, ref_stride, height, width, ref_stride_raw, sw, sh):
        best = C.c_uint64(0)
        x = C.c_int16(-12345)
        y = C.c_int16(-12345)
        sp = C.cast(C.c_void_p(src.ctypes.data + src_off), u8p)
        rp = C.cast(C.c_void_p(ref.ctypes.data + ref_off), u8p)
        self.lib.orc_sad_loop_kernel(sp, src_stride, rp, ref_stride, height, width, C.byref(best), C.byref(x),
                                     C.byref(y), ref_stride_raw, sw, sh)
        return best.value, x.value, y.value


class Reference:
    """The reference's kernels built by oracle/build_ref.sh (absent => available() is False)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ref_fullpel_search_batch.restype = None
        L.ref_fullpel_search_batch.argtypes = [C.c_int, u8p, C.c_uint32, u8p, C.c_uint32, i32p, C.c_uint32, u32p, u32p]
        for name in ("SadLoopKernel", "SadLoopKernel_SSE4_1_INTRIN", "SadLoopKernel_AVX2_INTRIN",
                     "SadLoopKernel_SSE4_1_HmeL0_INTRIN", "SadLoopKernel_AVX2_HmeL0_INTRIN"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                          C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.c_uint32, C.c_int16, C.c_int16]

    def fullpel_search_batch(self, src_plane, ref_plane, desc, asm_type=0):
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.empty((n, 85), dtype=np.uint32)
        mv = np.empty((n, 85), dtype=np.uint32)
        self.lib.ref_fullpel_search_batch(asm_type, _ptr(src_plane, u8p), src_plane.shape[1], _ptr(ref_plane, u8p),
                                          ref_plane.shape[1], _ptr(desc, i32p), n, _ptr(sad, u32p), _ptr(mv, u32p))
        return sad, mv

    def sad_loop(self, name, src, src_off, src_stride, ref, ref_off, ref_stride, height, width, ref_stride_raw, sw, sh):
        best = C.c_uint64(0)
        x = C.c_int16(-12345)
        y = C.c_int16(-12345)
        sp = C.cast(C.c_void_p(src.ctypes.data + src_off), u8p)
        rp = C.cast(C.c_void_p(ref.ctypes.data + ref_off), u8p)
        getattr(self.lib, name)(sp, src_stride, rp, ref_stride, height, width, C.byref(best), C.byref(x),
                                C.byref(y), ref_stride_raw, sw, sh)
        return best.value, x.value, y.value
