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

    def fullpel_search_batch(self, src_plane, ref_plane, desc, src_stride=None, ref_stride=None):
        """desc: int32 [n,6] = src_offset, ref_offset, x_origin, y_origin, sw, sh -> (sad[n,85], mv[n,85]).
        Planes are 2-D arrays (stride = row length) or a flat pool with explicit strides."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.empty((n, 85), dtype=np.uint32)
        mv = np.empty((n, 85), dtype=np.uint32)
        ss = src_stride if src_stride is not None else src_plane.shape[1]
        rs = ref_stride if ref_stride is not None else ref_plane.shape[1]
        self.lib.orc_fullpel_search_batch(_ptr(src_plane, u8p), ss, _ptr(ref_plane, u8p), rs, _ptr(desc, i32p), n,
                                          _ptr(sad, u32p), _ptr(mv, u32p))
        return sad, mv

    def fullpel_search209_batch(self, src_plane, ref_plane, desc, src_stride=None, ref_stride=None):
        """209-PU mode (squares + rectangles): -> (sad[n,209], mv[n,209]) in ME-buffer order."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.empty((n, 209), dtype=np.uint32)
        mv = np.empty((n, 209), dtype=np.uint32)
        ss = src_stride if src_stride is not None else src_plane.shape[1]
        rs = ref_stride if ref_stride is not None else ref_plane.shape[1]
        f = self.lib.orc_fullpel_search209_batch
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        f(src_plane.ctypes.data, ss, ref_plane.ctypes.data, rs, desc.ctypes.data, n, sad.ctypes.data, mv.ctypes.data)
        return sad, mv

    def hme_search_center_batch(self, pool, cur, ref, params, list_index, sb, l0_best_mv64=None, hme_state=None):
        """-> (desc int32 [n,6], center int16 [n,2]); cur/ref are svtav1_hip.PaPictureDesc, params MeParams."""
        sb = np.ascontiguousarray(sb, dtype=np.uint16)
        n = sb.shape[0]
        desc = np.zeros((n, 6), dtype=np.int32)
        center = np.zeros((n, 2), dtype=np.int16)
        mvp = None
        if l0_best_mv64 is not None:
            l0_best_mv64 = np.ascontiguousarray(l0_best_mv64, dtype=np.uint32)
            mvp = l0_best_mv64.ctypes.data_as(C.c_void_p)
        f = self.lib.orc_hme_search_center_batch
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p]
        if hme_state is not None:
            assert hme_state.dtype == np.int16 and hme_state.shape == (n, 25) and hme_state.flags.c_contiguous
        f(pool.ctypes.data, C.byref(cur), C.byref(ref), C.byref(params), list_index, sb.ctypes.data, n, mvp,
          desc.ctypes.data, center.ctypes.data, hme_state.ctypes.data if hme_state is not None else None)
        return desc, center

    def interp_planes(self, ref_plane, ref_off, x0, y0, w, h):
        """b/h/j tiles (h x w) whose (0,0) is search-region coordinate (x0,y0); search position (0,0) = ref_plane.flat[ref_off]."""
        b = np.zeros((h, w), np.uint8); hh = np.zeros((h, w), np.uint8); j = np.zeros((h, w), np.uint8)
        f = self.lib.orc_interp_planes
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f(ref_plane.ctypes.data + ref_off, ref_plane.shape[1], x0, y0, w, h, b.ctypes.data, hh.ctypes.data, j.ctypes.data)
        return b, hh, j

    def ssd_wrapped(self, a, b):
        f = self.lib.orc_ssd_wrapped
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        return f(a.ctypes.data, a.shape[1], b.ctypes.data, b.shape[1], a.shape[1], a.shape[0])

    def subpel_refine_batch(self, src_plane, ref_plane, desc, sad, mv, disable_8x8=False, src_stride=None, ref_stride=None):
        """In: full-pel (sad, mv) [n,85]; out: refined copies + (ssd [n,85], dir [n,85])."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.ascontiguousarray(sad, dtype=np.uint32).copy()
        mv = np.ascontiguousarray(mv, dtype=np.uint32).copy()
        ssd = np.zeros((n, 85), np.uint32)
        dr = np.zeros((n, 85), np.uint8)
        f = self.lib.orc_subpel_refine_batch
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p]
        ss = src_stride if src_stride is not None else src_plane.shape[1]
        rs = ref_stride if ref_stride is not None else ref_plane.shape[1]
        f(src_plane.ctypes.data, ss, ref_plane.ctypes.data, rs, desc.ctypes.data, n, int(disable_8x8), sad.ctypes.data,
          mv.ctypes.data, ssd.ctypes.data, dr.ctypes.data)
        return sad, mv, ssd, dr

    def pu_geometry209(self):
        """[209,5] = w, h, px, py, ME-buffer index by raster PU index."""
        g = np.zeros((209, 5), np.uint8)
        self.lib.orc_pu_geometry209.restype = None
        self.lib.orc_pu_geometry209.argtypes = [C.c_void_p]
        self.lib.orc_pu_geometry209(g.ctypes.data)
        return g

    def subpel_refine209_batch(self, src_plane, ref_plane, desc, sad, mv, disable_8x8=False, src_stride=None, ref_stride=None):
        """In: full-pel (sad, mv) [n,209] (ME-buffer order); out: refined copies."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.ascontiguousarray(sad, dtype=np.uint32).copy()
        mv = np.ascontiguousarray(mv, dtype=np.uint32).copy()
        assert sad.shape == (n, 209) and mv.shape == (n, 209)
        f = self.lib.orc_subpel_refine209_batch
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        ss = src_stride if src_stride is not None else src_plane.shape[1]
        rs = ref_stride if ref_stride is not None else ref_plane.shape[1]
        f(src_plane.ctypes.data, ss, ref_plane.ctypes.data, rs, desc.ctypes.data, n, int(disable_8x8), sad.ctypes.data, mv.ctypes.data)
        return sad, mv

    def subpel_refine_method(self, src_plane, ref_plane, desc, sad, mv, method, all_pu, disable_8x8=False, src_stride=None,
                             ref_stride=None):
        """The refinement under one of the reference's fractional search methods (0 SUB_SAD_SEARCH, 1 FULL_SAD_SEARCH,
        2 SSD_SEARCH).  In: full-pel (sad, mv) [n, 85 | 209] in ME-buffer order; out: refined copies + half-pel direction [n, npu]."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n, npu = desc.shape[0], 209 if all_pu else 85
        sad = np.ascontiguousarray(sad, dtype=np.uint32).copy()
        mv = np.ascontiguousarray(mv, dtype=np.uint32).copy()
        assert sad.shape == (n, npu) and mv.shape == (n, npu)
        dr = np.zeros((n, npu), np.uint8)
        f = self.lib.orc_subpel_refine_method
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int16, C.c_int16, C.c_int, C.c_int, C.c_int, C.c_void_p,
                      C.c_void_p, C.c_void_p]
        ss = src_stride if src_stride is not None else src_plane.shape[1]
        rs = ref_stride if ref_stride is not None else ref_plane.shape[1]
        for i in range(n):
            d = desc[i]
            f(src_plane.ctypes.data + int(d[0]), ss, ref_plane.ctypes.data + int(d[1]), rs, int(d[2]), int(d[3]), int(disable_8x8),
              int(all_pu), int(method), sad[i].ctypes.data, mv[i].ctypes.data, dr[i].ctypes.data)
        return sad, mv, dr

    def bipred_pack_batch(self, src_plane, src_stride, ref0_plane, ref0_stride, desc0, sad0, mv0, ref1_plane=None,
                          ref1_stride=0, desc1=None, sad1=None, mv1=None, bipred_8x8=True, n_pu=85):
        """-> structured array [n,n_pu] of svtav1_hip.ME_CU_RESULT_DTYPE (raster PU order); n_pu = 85 or 209."""
        import svtav1_hip
        n = desc0.shape[0]
        n_lists = 2 if desc1 is not None else 1
        out = np.zeros((n, n_pu), dtype=svtav1_hip.ME_CU_RESULT_DTYPE)
        assert np.asarray(sad0).shape == (n, n_pu)
        g = self.lib.orc_bipred_pack_batch_npu
        g.restype = None
        g.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        f = lambda *a: g(*a[:-1], n_pu, a[-1])
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        d0 = c(desc0, np.int32); s0 = c(sad0, np.uint32); m0 = c(mv0, np.uint32)
        if n_lists == 2:
            d1 = c(desc1, np.int32); s1 = c(sad1, np.uint32); m1 = c(mv1, np.uint32)
            f(src_plane.ctypes.data, src_stride, ref0_plane.ctypes.data, ref0_stride, d0.ctypes.data, ref1_plane.ctypes.data,
              ref1_stride, d1.ctypes.data, n, s0.ctypes.data, m0.ctypes.data, s1.ctypes.data, m1.ctypes.data, 2, int(bipred_8x8),
              out.ctypes.data)
        else:
            f(src_plane.ctypes.data, src_stride, ref0_plane.ctypes.data, ref0_stride, d0.ctypes.data, None, 0, None, n,
              s0.ctypes.data, m0.ctypes.data, None, None, 1, int(bipred_8x8), out.ctypes.data)
        return out

    def generate_padding(self, plane, width, height, pad_w, pad_h):
        """In place on a 2-D uint8 / uint16 array whose row length is the stride (samples)."""
        f = self.lib.orc_generate_padding if plane.dtype == np.uint8 else self.lib.orc_generate_padding16
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        f(plane.ctypes.data, plane.shape[1], width, height, pad_w, pad_h)

    def pa_derive_planes(self, luma, full_stride=None, quarter_stride=None, sixteenth_stride=None):
        """luma [h,w] uint8 -> (full, quarter, sixteenth) padded planes as Picture Analysis leaves them."""
        h, w = luma.shape
        fs = full_stride or w + 136
        qs = quarter_stride or (w >> 1) + 64
        ss = sixteenth_stride or (w >> 2) + 32
        full = np.full((h + 136, fs), 0xA5, np.uint8)
        full[68:68 + h, 68:68 + w] = luma
        quarter = np.full(((h >> 1) + 64, qs), 0x5A, np.uint8)
        sixteenth = np.full(((h >> 2) + 32, ss), 0x3C, np.uint8)
        f = self.lib.orc_pa_derive_planes
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        f(full.ctypes.data, fs, w, h, quarter.ctypes.data, qs, sixteenth.ctypes.data, ss)
        return full, quarter, sixteenth

    # ---- open-loop intra search (oracle/svt_ois_oracle.c) ----
    def ois_neighbours(self, plane, origin, width, height, cu_x, cu_y, size):
        refs = np.zeros(4 * size + 1, np.uint8)
        self.lib.orc_ois_neighbours.restype = None
        self.lib.orc_ois_neighbours(C.c_void_p(plane.ctypes.data), plane.shape[1], origin, width, height, cu_x, cu_y, size,
                                    C.c_void_p(refs.ctypes.data))
        return refs

    def ois_predict(self, refs, size, mode):
        pred = np.zeros((size, size), np.uint8)
        self.lib.orc_ois_predict.restype = None
        self.lib.orc_ois_predict(C.c_void_p(refs.ctypes.data), size, mode, C.c_void_p(pred.ctypes.data))
        return pred

    def ois_sad_table(self, plane, origin, width, height):
        n_sb = ((width + 63) // 64) * ((height + 63) // 64)
        out = np.zeros((n_sb, 85, 35), np.uint32)
        self.lib.orc_ois_sad_table(C.c_void_p(plane.ctypes.data), plane.shape[1], origin, width, height,
                                   C.c_void_p(out.ctypes.data))
        return out

    def ois_search_picture(self, plane, origin, width, height, op, me_dist=None):
        """op = OIS_PARAMS-ordered int32[7]; returns (cand [n_sb][85][18] u32, total [n_sb][85] u8)."""
        n_sb = ((width + 63) // 64) * ((height + 63) // 64)
        cand = np.zeros((n_sb, 85, 18), np.uint32)
        total = np.zeros((n_sb, 85), np.uint8)
        op = np.ascontiguousarray(op, np.int32)
        md = None if me_dist is None else np.ascontiguousarray(me_dist, np.uint32)
        self.lib.orc_ois_search_picture.restype = C.c_int
        rc = self.lib.orc_ois_search_picture(C.c_void_p(plane.ctypes.data), plane.shape[1], origin, width, height,
                                             C.c_void_p(op.ctypes.data), C.c_void_p(None if md is None else md.ctypes.data),
                                             C.c_void_p(cand.ctypes.data), C.c_void_p(total.ctypes.data))
        assert rc == 0
        return cand, total

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


REF_ME_SO = os.path.join(HERE, "_ref", "libsvtref_me.so")
REF_SUBPEL_SO = os.path.join(HERE, "_ref", "libsvtref_subpel.so")


class ReferenceSubpel:
    """The reference's own HalfPelSearch_LCU + QuarterPelSearch_LCU run with fractionalSearchMethod = SUB_SAD_SEARCH (0) or
    FULL_SAD_SEARCH (1) (oracle/_ref/libsvtref_subpel.so; oracle/ref_subpel_search_driver.c says why SSD_SEARCH cannot run here)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SUBPEL_SO)

    def __init__(self):
        self.lib = C.CDLL(REF_SUBPEL_SO, mode=os.RTLD_LAZY)  # NASM-only symbols stay unresolved by design and are never reached
        self.lib.ref_subpel_search.restype = C.c_int
        self.lib.ref_subpel_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p]

    def subpel_search(self, src_plane, ref_plane, desc, sad, mv, method, all_pu, disable_8x8=False, asm_type=0):
        """desc rows as for the oracle: (src_offset, ref_offset, x_origin, y_origin, search_w, search_h); -> (sad, mv, dir)."""
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n, npu = desc.shape[0], 209 if all_pu else 85
        sad = np.ascontiguousarray(sad, dtype=np.uint32).copy()
        mv = np.ascontiguousarray(mv, dtype=np.uint32).copy()
        assert sad.shape == (n, npu) and mv.shape == (n, npu)
        dr = np.zeros((n, npu), np.uint8)
        for i in range(n):
            d = desc[i]
            # ref_offset points at search position (0,0); the driver wants the sample co-located with the SB origin
            ref00 = ref_plane.ctypes.data + int(d[1]) - int(d[2]) - int(d[3]) * ref_plane.shape[1]
            geo = np.array([d[2], d[3], d[4], d[5]], np.int32)
            rc = self.lib.ref_subpel_search(src_plane.ctypes.data + int(d[0]), src_plane.shape[1], ref00, ref_plane.shape[1],
                                            geo.ctypes.data, int(method), int(all_pu), int(disable_8x8), int(asm_type),
                                            sad[i].ctypes.data, mv[i].ctypes.data, dr[i].ctypes.data)
            if rc != 0:
                raise RuntimeError(f"ref_subpel_search failed: {rc}")
        return sad, mv, dr



class ReferenceME:
    """The reference's own MotionEstimateLcu (oracle/_ref/libsvtref_me.so, sub-pel disabled: see
    oracle/ref_me_lcu_driver.c for why)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_ME_SO)

    def __init__(self):
        # lazy binding: three NASM-only symbols stay unresolved by design and are never reached
        self.lib = C.CDLL(REF_ME_SO, mode=os.RTLD_LAZY)
        self.lib.ref_me_lcu_run.restype = C.c_int
        self.lib.ref_me_lcu_run.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]

    def ois_predict(self, plane, origin, width, height, cu_x, cu_y, size, mode):
        pred = np.zeros((size, size), np.uint8)
        refs = np.zeros(4 * size + 1, np.uint8)
        self.lib.ref_ois_predict.restype = C.c_int
        rc = self.lib.ref_ois_predict(C.c_void_p(plane.ctypes.data), plane.shape[1], origin, width, height, cu_x, cu_y, size,
                                      mode, C.c_void_p(pred.ctypes.data), C.c_void_p(refs.ctypes.data))
        assert rc == 0
        return pred, refs

    def ois_search_picture(self, plane, origin, width, height, op, me_dist=None):
        n_sb = ((width + 63) // 64) * ((height + 63) // 64)
        cand = np.zeros((n_sb, 85, 18), np.uint32)
        total = np.zeros((n_sb, 85), np.uint8)
        op = np.ascontiguousarray(op, np.int32)
        md = None if me_dist is None else np.ascontiguousarray(me_dist, np.uint32)
        self.lib.ref_ois_search_picture.restype = C.c_int
        rc = self.lib.ref_ois_search_picture(C.c_void_p(plane.ctypes.data), plane.shape[1], origin, width, height,
                                             C.c_void_p(op.ctypes.data), C.c_void_p(None if md is None else md.ctypes.data),
                                             C.c_void_p(cand.ctypes.data), C.c_void_p(total.ctypes.data))
        assert rc == 0
        return cand, total

    def fullpel_search209_batch(self, src_plane, ref_plane, desc):
        """The reference's ExtSadCalculation* functions driven like open_loop_me_fullpel_search_sblock
        (oracle/ref_fullpel209_driver.c) -> (sad[n,209], mv[n,209])."""
        f = self.lib.ref_fullpel_search_209pu
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int16, C.c_int16, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        desc = np.ascontiguousarray(desc, dtype=np.int32)
        n = desc.shape[0]
        sad = np.full((n, 209), 128 * 128 * 255, dtype=np.uint32)
        mv = np.zeros((n, 209), dtype=np.uint32)
        for i in range(n):
            d = desc[i]
            f(src_plane.ctypes.data + int(d[0]), src_plane.shape[1], ref_plane.ctypes.data + int(d[1]), ref_plane.shape[1], int(d[2]), int(d[3]),
              int(d[4]), int(d[5]), sad[i].ctypes.data, mv[i].ctypes.data)
        return sad, mv

    def interp_region(self, ref_plane, ref_off, sw, sh, rows, cols):
        """The reference's InterpolateSearchRegionAVC on the region whose position (0,0) is ref_plane.flat[ref_off];
        returns the consumed b/h/j planes cropped to rows x cols."""
        f = self.lib.ref_interp_region
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        b = np.zeros((rows, cols), np.uint8); h = np.zeros((rows, cols), np.uint8); j = np.zeros((rows, cols), np.uint8)
        rc = f(ref_plane.ctypes.data + ref_off, ref_plane.shape[1], sw, sh, rows, cols, b.ctypes.data, h.ctypes.data, j.ctypes.data)
        assert rc == 0
        return b, h, j

    def run(self, cur, ref0, ref1, params, two_lists=False, hierarchical_levels=3, asm_type=0, all_pu=False, resolution_4k=False):
        """cur/ref0/ref1: svtav1_hip.synth.PaPicture; params: svtav1_hip.MeParams.
        all_pu: the 209-PU mode (NPU = 209, else 85)
        -> dict(sad [n,2,NPU], mv [n,2,NPU], origin [n,2,2], res [n,NPU,11])"""
        w, h = cur.width, cur.height
        planes = (C.c_void_p * 9)()
        keep = []
        for i, p in enumerate((cur, ref0, ref1)):
            for j, a in enumerate((p.full, p.quarter, p.sixteenth)):
                a = np.ascontiguousarray(a)
                keep.append(a)
                planes[3 * i + j] = a.ctypes.data
        P = params
        ip = np.array([P.search_area_width, P.search_area_height, P.number_hme_search_region_in_width,
                       P.number_hme_search_region_in_height, P.hme_level0_total_search_area_width,
                       P.hme_level0_total_search_area_height,
                       P.hme_level0_search_area_in_width_array[0], P.hme_level0_search_area_in_width_array[1],
                       P.hme_level0_search_area_in_height_array[0], P.hme_level0_search_area_in_height_array[1],
                       P.hme_level1_search_area_in_width_array[0], P.hme_level1_search_area_in_width_array[1],
                       P.hme_level1_search_area_in_height_array[0], P.hme_level1_search_area_in_height_array[1],
                       P.hme_level2_search_area_in_width_array[0], P.hme_level2_search_area_in_width_array[1],
                       P.hme_level2_search_area_in_height_array[0], P.hme_level2_search_area_in_height_array[1],
                       P.enable_hme_flag, P.enable_hme_level0_flag, P.enable_hme_level1_flag, P.enable_hme_level2_flag,
                       int(two_lists), P.temporal_layer_index, hierarchical_levels, P.is_used_as_reference_flag, 0,
                       10, 10 if P.ref_poc_equal else 20, asm_type, int(all_pu), int(resolution_4k)], dtype=np.int32)
        n = ((w + 63) // 64) * ((h + 63) // 64)
        npu = 209 if all_pu else 85
        sad = np.zeros((n, 2, npu), np.uint32)
        mv = np.zeros((n, 2, npu), np.uint32)
        origin = np.zeros((n, 2, 4), np.int32)
        res = np.zeros((n, npu, 11), np.int32)
        rc = self.lib.ref_me_lcu_run(planes, w, h, ip.ctypes.data, sad.ctypes.data, mv.ctypes.data, origin.ctypes.data,
                                     res.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"ref_me_lcu_run failed: {rc}")
        return {"sad": sad, "mv": mv, "origin": origin[:, :, :2], "res": res}
