"""ctypes binding of libsvtav1_hip.so (the C ABI in include/svtav1_hip.h) for tests and bench.py.

The product is the shared library; this module only marshals numpy / torch device pointers into it.
There is no Python or CPU fallback: if the library is missing, import of `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
# SVTAV1_HIP_LIB: A/B tooling only (tools/kernel_times.py loads experimental builds of the library side by side)
LIB_PATH = os.environ.get("SVTAV1_HIP_LIB") or os.path.join(_PKG_ROOT, "libsvtav1_hip.so")

NUM_SQ_PU = 85
MAX_SAD_VALUE = 128 * 128 * 255

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)


class FullpelDesc(C.Structure):
    _fields_ = [("src_offset", C.c_int32), ("ref_offset", C.c_int32), ("x_search_area_origin", C.c_int32),
                ("y_search_area_origin", C.c_int32), ("search_area_width", C.c_int32), ("search_area_height", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    """Load libsvtav1_hip.so (built in-tree by `make -C svt-av1-1_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `make -C svt-av1-1_amd` (there is no CPU fallback)")
    # The PyTorch wheel bundles its own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one process
    # cannot both own the GPU (measured: loading this library first makes torch report "No HIP GPUs"),
    # so when torch is installed it is imported first and this library binds to the runtime torch loaded.
    # A C host (the real integration) links the system ROCm runtime and never sees torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.svthip_last_error.restype = C.c_char_p
    L.svthip_create.restype = C.c_int32
    L.svthip_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    L.svthip_destroy.restype = None
    L.svthip_destroy.argtypes = [C.c_void_p]
    L.svthip_stream.restype = C.c_void_p
    L.svthip_stream.argtypes = [C.c_void_p]
    L.svthip_synchronize.restype = C.c_int32
    L.svthip_synchronize.argtypes = [C.c_void_p]
    L.svthip_set_option.restype = C.c_int32
    L.svthip_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.svthip_reserve.restype = C.c_int32
    L.svthip_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32]
    L.svthip_me_fullpel_search.restype = C.c_int32
    L.svthip_me_fullpel_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t,
                                           C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.svthip_me_fullpel_search_dev.restype = C.c_int32
    L.svthip_me_fullpel_search_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_me_fullpel_search_time_dev.restype = C.c_int32
    L.svthip_me_fullpel_search_time_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                    C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                    C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    L.svthip_me_hme_search_center_dev.restype = C.c_int32
    L.svthip_me_hme_search_center_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                  C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]
    L.svthip_me_hme_search_center_batch_dev.restype = C.c_int32
    L.svthip_me_hme_search_center_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_motion_estimate_picture_dev.restype = C.c_int32
    L.svthip_motion_estimate_picture_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]
    L.svthip_me_fullpel_search209_dev.restype = C.c_int32
    L.svthip_me_fullpel_search209_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_motion_estimate_batch_dev.restype = C.c_int32
    L.svthip_motion_estimate_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                   C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
    L.svthip_me_subpel_refine_dev.restype = C.c_int32
    L.svthip_me_subpel_refine_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_me_bipred_pack_dev.restype = C.c_int32
    L.svthip_me_bipred_pack_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
    L.svthip_me_subpel_refine209_dev.restype = C.c_int32
    L.svthip_me_subpel_refine209_dev.argtypes = L.svthip_me_subpel_refine_dev.argtypes
    L.svthip_me_subpel_search_dev.restype = C.c_int32
    L.svthip_me_subpel_search_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_me_bipred_pack209_dev.restype = C.c_int32
    L.svthip_me_bipred_pack209_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                               C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.svthip_motion_estimate209_batch_dev.restype = C.c_int32
    L.svthip_motion_estimate209_batch_dev.argtypes = L.svthip_motion_estimate_batch_dev.argtypes
    L.svthip_quantize_b_batch_dev.restype = C.c_int32
    L.svthip_quantize_b_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_fwd_txfm2d_batch_dev.restype = C.c_int32
    L.svthip_fwd_txfm2d_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p]
    L.svthip_inv_txfm2d_add_batch_dev.restype = C.c_int32
    L.svthip_inv_txfm2d_add_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_uint32, C.c_void_p, C.c_void_p]
    L.svthip_encode_tu16_batch_dev.restype = C.c_int32
    L.svthip_encode_tu16_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.c_uint32] + [C.c_void_p] * 9
    L.svthip_encode_tu_batch_dev.restype = C.c_int32
    L.svthip_encode_tu_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                             C.c_uint32] + [C.c_void_p] * 9
    L.svthip_pa_derive_planes_dev.restype = C.c_int32
    L.svthip_pa_derive_planes_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p]
    L.svthip_open_loop_intra_search_batch_dev.restype = C.c_int32
    L.svthip_open_loop_intra_search_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                          C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_pad_plane_dev.restype = C.c_int32
    L.svthip_pad_plane_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.svthip_av1_convolve_sr_batch_dev.restype = C.c_int32
    L.svthip_av1_convolve_sr_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                                   C.c_uint32, C.c_void_p]
    L.svthip_av1_convolve_compound_batch_dev.restype = C.c_int32
    L.svthip_av1_convolve_compound_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.svthip_av1_highbd_convolve_batch_dev.restype = C.c_int32
    L.svthip_av1_highbd_convolve_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                       C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.svthip_sad_loop_batch_dev.restype = C.c_int32
    L.svthip_sad_loop_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.svthip_tu_batcher_create.restype = C.c_int32
    L.svthip_tu_batcher_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.svthip_tu_batcher_destroy.restype = None
    L.svthip_tu_batcher_destroy.argtypes = [C.c_void_p]
    L.svthip_tu_batcher_begin.restype = C.c_int32
    L.svthip_tu_batcher_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.svthip_tu_batcher_add.restype = C.c_int32
    L.svthip_tu_batcher_add.argtypes = [C.c_void_p] + [C.c_uint32] * 10 + [C.POINTER(C.c_uint32)]
    L.svthip_tu_batcher_flush.restype = C.c_int32
    L.svthip_tu_batcher_flush.argtypes = [C.c_void_p]
    L.svthip_tu_batcher_result.restype = C.c_int32
    L.svthip_tu_batcher_result.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.svthip_tu_batcher_read_coeffs.restype = C.c_int32
    L.svthip_tu_batcher_read_coeffs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.svthip_tu_batcher_pools.restype = C.c_int32
    L.svthip_tu_batcher_pools.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _lib = L
    return L


OPT_SADLOOP_GENERIC = 0
OPT_CONVOLVE_VALU = 1
OPT_TQ_MAX_WORKGROUPS = 2


class SvtHipError(RuntimeError):
    pass


def _check(rc: int):
    if rc != 0:
        raise SvtHipError(f"svthip error 0x{rc & 0xFFFFFFFF:08x}: {lib().svthip_last_error().decode()}")


class Context:
    """One svthip_ctx (stream + scratch), the analogue of one MeContext_t."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().svthip_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().svthip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return lib().svthip_stream(self._h)

    def synchronize(self):
        _check(lib().svthip_synchronize(self._h))

    def set_option(self, option: int, value: int):
        """svthip_set_option: kernel-selection override of this context (OPT_SADLOOP_GENERIC / OPT_CONVOLVE_VALU / OPT_TQ_MAX_WORKGROUPS)."""
        _check(lib().svthip_set_option(self._h, option, value))

    def reserve(self, width: int, height: int, n_pu: int = 85, n_jobs: int = 1, host_forms: bool = False):
        _check(lib().svthip_reserve(self._h, width, height, n_pu, n_jobs, int(host_forms)))

    # -- host-pointer form (numpy in / numpy out) ------------------------------------------------
    def fullpel_search(self, src_plane: np.ndarray, ref_plane: np.ndarray, desc: np.ndarray):
        """desc int32 [n,6] -> (best_sad [n,85] uint32, best_mv [n,85] uint32)."""
        assert src_plane.dtype == np.uint8 and ref_plane.dtype == np.uint8
        src_plane = np.ascontiguousarray(src_plane)
        ref_plane = np.ascontiguousarray(ref_plane)
        desc = np.ascontiguousarray(desc, dtype=np.int32).reshape(-1, 6)
        n = desc.shape[0]
        sad = np.zeros((n, NUM_SQ_PU), dtype=np.uint32)
        mv = np.zeros((n, NUM_SQ_PU), dtype=np.uint32)
        _check(lib().svthip_me_fullpel_search(self._h, src_plane.ctypes.data, src_plane.nbytes, src_plane.shape[1],
                                               ref_plane.ctypes.data, ref_plane.nbytes, ref_plane.shape[1],
                                               desc.ctypes.data, n, sad.ctypes.data, mv.ctypes.data))
        return sad, mv

    # -- device-pointer form (raw addresses, e.g. torch tensors' data_ptr()) -----------------------
    def fullpel_search_dev(self, d_src: int, src_stride: int, d_ref: int, ref_stride: int, d_desc: int, n_sb: int,
                           max_sw: int, max_sh: int, d_sad: int, d_mv: int, stream: int | None = None):
        _check(lib().svthip_me_fullpel_search_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw,
                                                  max_sh, d_sad, d_mv, stream))

    def fullpel_search_time_dev(self, d_src: int, src_stride: int, d_ref: int, ref_stride: int, d_desc: int, n_sb: int,
                                max_sw: int, max_sh: int, d_sad: int, d_mv: int, iters: int) -> float:
        ms = C.c_float(0)
        _check(lib().svthip_me_fullpel_search_time_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb,
                                                       max_sw, max_sh, d_sad, d_mv, iters, C.byref(ms)))
        return ms.value


def _hme_search_center_dev(self, d_pool, cur, ref, params, list_index, d_sb, n_sb, d_l0_mv64, d_desc, d_center=None,
                           d_state=None, stream=None, l0_mv_stride=1):
    """cur/ref: PaPictureDesc, params: MeParams (host structs); the rest are device addresses."""
    _check(lib().svthip_me_hme_search_center_dev(self._h, d_pool, C.byref(cur), C.byref(ref), C.byref(params), list_index,
                                                 d_sb, n_sb, d_l0_mv64, l0_mv_stride, d_desc, d_center, d_state, stream))


def _hme_search_center_batch_dev(self, d_pool, curs, refs, params, list_index, d_sb, n_sb, d_l0_mv64, d_desc, d_center=None,
                                 d_state=None, stream=None, l0_mv_stride=1):
    """curs/refs: sequences of PaPictureDesc (one pair per job); per-SB device arrays hold the jobs back to back."""
    n = len(curs)
    ca = (PaPictureDesc * n)(*curs)
    ra = (PaPictureDesc * n)(*refs)
    _check(lib().svthip_me_hme_search_center_batch_dev(self._h, d_pool, ca, ra, n, C.byref(params), list_index, d_sb, n_sb, d_l0_mv64,
                                                       l0_mv_stride, d_desc, d_center, d_state, stream))


def _motion_estimate_picture_dev(self, d_pool, cur, ref0, ref1, params, d_sb, n_sb, d_out, use_subpel=True, cu8x8_mode=0,
                                 d_list_sad=None, d_list_mv=None, stream=None):
    """Whole-picture ME (MotionEstimateLcu over all SBs): ref1=None for P pictures."""
    _check(lib().svthip_motion_estimate_picture_dev(self._h, d_pool, C.byref(cur), C.byref(ref0),
                                                    C.byref(ref1) if ref1 is not None else None, C.byref(params),
                                                    int(use_subpel), int(cu8x8_mode), d_sb, n_sb, d_out, d_list_sad, d_list_mv,
                                                    stream))


Context.motion_estimate_picture_dev = _motion_estimate_picture_dev


def _motion_estimate_batch_dev(self, d_pool, curs, refs0, refs1, params, d_sb, n_sb, d_out, use_subpel=True, cu8x8_mode=0,
                               d_list_sad=None, d_list_mv=None, stream=None):
    """Whole-picture ME of len(curs) pictures in one call; refs1=None for P pictures."""
    n = len(curs)
    ca, r0 = (PaPictureDesc * n)(*curs), (PaPictureDesc * n)(*refs0)
    r1 = (PaPictureDesc * n)(*refs1) if refs1 is not None else None
    _check(lib().svthip_motion_estimate_batch_dev(self._h, d_pool, ca, r0, r1, n, C.byref(params), int(use_subpel), int(cu8x8_mode),
                                                  d_sb, n_sb, d_out, d_list_sad, d_list_mv, stream))


Context.motion_estimate_batch_dev = _motion_estimate_batch_dev


def _fullpel_search209_dev(self, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh, d_sad, d_mv, stream=None):
    """209-PU full-pel search (squares + rectangles); d_sad / d_mv: [n_sb][209] uint32 device arrays."""
    _check(lib().svthip_me_fullpel_search209_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh, d_sad, d_mv,
                                                 stream))


Context.fullpel_search209_dev = _fullpel_search209_dev
Context.hme_search_center_dev = _hme_search_center_dev
Context.hme_search_center_batch_dev = _hme_search_center_batch_dev


def _subpel_refine_dev(self, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh, d_sad, d_mv, disable_8x8=False,
                       stream=None):
    _check(lib().svthip_me_subpel_refine_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh,
                                             int(disable_8x8), d_sad, d_mv, stream))


Context.subpel_refine_dev = _subpel_refine_dev


def _bipred_pack_dev(self, d_src, src_stride, d_ref0, ref0_stride, d_desc0, d_ref1, ref1_stride, d_desc1, n_sb, max_sw, max_sh,
                     d_sad0, d_mv0, d_sad1, d_mv1, n_lists, d_out, bipred_8x8=True, stream=None):
    _check(lib().svthip_me_bipred_pack_dev(self._h, d_src, src_stride, d_ref0, ref0_stride, d_desc0, d_ref1, ref1_stride, d_desc1,
                                           n_sb, max_sw, max_sh, d_sad0, d_mv0, d_sad1, d_mv1, n_lists, int(bipred_8x8), d_out,
                                           stream))


Context.bipred_pack_dev = _bipred_pack_dev


def _subpel_refine209_dev(self, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh, d_sad, d_mv, disable_8x8=False,
                          stream=None):
    """Sub-pel refinement of all 209 PUs; d_sad / d_mv: [n_sb][209] uint32 device arrays (ME-buffer order), in place."""
    _check(lib().svthip_me_subpel_refine209_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh,
                                                int(disable_8x8), d_sad, d_mv, stream))


FRACTIONAL_SUB_SAD_SEARCH, FRACTIONAL_FULL_SAD_SEARCH, FRACTIONAL_SSD_SEARCH = 0, 1, 2  # MeContext_t::fractionalSearchMethod


def _subpel_search_dev(self, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh, d_sad, d_mv, method, all_pu,
                       disable_8x8=False, stream=None):
    """The refinement under one of the reference's fractional search methods; d_sad / d_mv: [n_sb][209 if all_pu else 85], in place."""
    _check(lib().svthip_me_subpel_search_dev(self._h, d_src, src_stride, d_ref, ref_stride, d_desc, n_sb, max_sw, max_sh,
                                             int(disable_8x8), int(all_pu), int(method), d_sad, d_mv, stream))


Context.subpel_search_dev = _subpel_search_dev


def _bipred_pack209_dev(self, d_src, src_stride, d_ref0, ref0_stride, d_desc0, d_ref1, ref1_stride, d_desc1, n_sb, max_sw, max_sh,
                        d_sad0, d_mv0, d_sad1, d_mv1, n_lists, d_out, stream=None):
    """Bi-prediction + packing over all 209 PUs; d_out: [n_sb][209] ME_CU_RESULT_DTYPE in raster PU order."""
    _check(lib().svthip_me_bipred_pack209_dev(self._h, d_src, src_stride, d_ref0, ref0_stride, d_desc0, d_ref1, ref1_stride, d_desc1,
                                              n_sb, max_sw, max_sh, d_sad0, d_mv0, d_sad1, d_mv1, n_lists, d_out, stream))


def _motion_estimate209_batch_dev(self, d_pool, curs, refs0, refs1, params, d_sb, n_sb, d_out, use_subpel=True, cu8x8_mode=0,
                                  d_list_sad=None, d_list_mv=None, stream=None):
    """Whole-picture ME in the 209-PU mode of len(curs) pictures in one call; refs1=None for P pictures."""
    n = len(curs)
    ca, r0 = (PaPictureDesc * n)(*curs), (PaPictureDesc * n)(*refs0)
    r1 = (PaPictureDesc * n)(*refs1) if refs1 is not None else None
    _check(lib().svthip_motion_estimate209_batch_dev(self._h, d_pool, ca, r0, r1, n, C.byref(params), int(use_subpel), int(cu8x8_mode),
                                                     d_sb, n_sb, d_out, d_list_sad, d_list_mv, stream))


Context.subpel_refine209_dev = _subpel_refine209_dev
Context.bipred_pack209_dev = _bipred_pack209_dev
Context.motion_estimate209_batch_dev = _motion_estimate209_batch_dev


def _quantize_b_batch_dev(self, d_coeff, d_desc, n_tu, d_qparams, d_iscan, d_qcoeff, d_dqcoeff, d_eob, stream=None):
    _check(lib().svthip_quantize_b_batch_dev(self._h, d_coeff, d_desc, n_tu, d_qparams, d_iscan, d_qcoeff, d_dqcoeff, d_eob, stream))


Context.quantize_b_batch_dev = _quantize_b_batch_dev


def _fwd_txfm2d_batch_dev(self, d_residual, d_desc, n_tu, tx_width, tx_height, bit_depth, d_coeff, stream=None):
    _check(lib().svthip_fwd_txfm2d_batch_dev(self._h, d_residual, d_desc, n_tu, tx_width, tx_height, bit_depth, d_coeff, stream))


Context.fwd_txfm2d_batch_dev = _fwd_txfm2d_batch_dev


def _inv_txfm2d_add_batch_dev(self, d_coeff, d_desc, n_tu, tx_width, tx_height, bit_depth, recon_16bit, d_recon, stream=None):
    _check(lib().svthip_inv_txfm2d_add_batch_dev(self._h, d_coeff, d_desc, n_tu, tx_width, tx_height, bit_depth, int(recon_16bit),
                                                 d_recon, stream))


Context.inv_txfm2d_add_batch_dev = _inv_txfm2d_add_batch_dev


def _encode_tu_batch_dev(self, d_src, d_pred, d_recon, d_desc, n_tu, tx_width, tx_height, d_qparams, d_iscan, d_coeff, d_qcoeff,
                         d_dqcoeff, d_eob, d_energy=None, d_dist=None, stream=None, planes_16bit=False):
    fn = lib().svthip_encode_tu16_batch_dev if planes_16bit else lib().svthip_encode_tu_batch_dev
    _check(fn(self._h, d_src, d_pred, d_recon, d_desc, n_tu, tx_width, tx_height, d_qparams, d_iscan,
              d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_energy, d_dist, stream))


Context.encode_tu_batch_dev = _encode_tu_batch_dev


def _pa_derive_planes_dev(self, d_pool, pics, want_quarter=True, want_sixteenth=True, stream=None):
    """Pad the full-resolution planes and build the 1/4 and 1/16 planes of `pics` (PaPictureDesc list) inside the device pool."""
    n = len(pics)
    _check(lib().svthip_pa_derive_planes_dev(self._h, d_pool, (PaPictureDesc * n)(*pics), n, int(want_quarter), int(want_sixteenth), stream))


def _pad_plane_dev(self, d_plane, stride, width, height, pad_w, pad_h, sample_bytes=1, stream=None):
    """generate_padding / generate_padding16_bit of one device plane in place (all quantities in samples)."""
    _check(lib().svthip_pad_plane_dev(self._h, d_plane, stride, width, height, pad_w, pad_h, sample_bytes, stream))


HME_MAX_JOBS = 32  # SVTHIP_HME_MAX_JOBS: pictures per kernel launch of the batch entries


class OisParams(C.Structure):
    _fields_ = [("slice_is_intra", C.c_uint8), ("temporal_layer_index", C.c_uint8), ("is_used_as_reference_flag", C.c_uint8),
                ("input_resolution_4k", C.c_uint8), ("limit_ois_to_dc_mode_flag", C.c_uint8), ("cu8x8_mode", C.c_uint8),
                ("enc_mode", C.c_uint8), ("reserved", C.c_uint8)]


def _open_loop_intra_search_batch_dev(self, d_pool, curs, params, d_sb, n_sb, d_me, me_pu_stride, d_cand, d_total, stream=None):
    """OpenLoopIntraSearchLcu over len(curs) pictures: d_cand [n][n_sb][85][18] u32 OisCandidate_t words, d_total [n][n_sb][85] u8."""
    n = len(curs)
    _check(lib().svthip_open_loop_intra_search_batch_dev(self._h, d_pool, (PaPictureDesc * n)(*curs), n, C.byref(params), d_sb, n_sb,
                                                         d_me, me_pu_stride, d_cand, d_total, stream))


Context.open_loop_intra_search_batch_dev = _open_loop_intra_search_batch_dev
Context.pa_derive_planes_dev = _pa_derive_planes_dev
Context.pad_plane_dev = _pad_plane_dev


class TuResult(C.Structure):
    _fields_ = [("distortion", C.c_uint64 * 2), ("three_quad_energy", C.c_uint64), ("coeff_offset", C.c_uint32), ("eob", C.c_uint16),
                ("tx_size", C.c_uint8), ("tx_type", C.c_uint8)]


TU_RECON_SCRATCH = 0xffffffff


class TuBatcher:
    """svthip_tu_batcher: host-side gather / scatter of (TU, tx_type) candidates for the fused T/Q chain."""

    def __init__(self, ctx: "Context", max_candidates: int, max_coeff_samples: int):
        self._h = C.c_void_p()
        self._ctx = ctx
        _check(lib().svthip_tu_batcher_create(ctx._h, max_candidates, max_coeff_samples, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().svthip_tu_batcher_destroy(self._h)
            self._h = C.c_void_p()

    def begin(self, d_src, d_pred, d_recon, planes_16bit, d_qparams, d_iscan):
        _check(lib().svthip_tu_batcher_begin(self._h, d_src, d_pred, d_recon, int(planes_16bit), d_qparams, d_iscan))

    def add(self, tx_size, tx_type, src_offset, src_stride, pred_offset, pred_stride, recon_offset, recon_stride, qparam_index, iscan_offset) -> int:
        h = C.c_uint32(0)
        _check(lib().svthip_tu_batcher_add(self._h, tx_size, tx_type, src_offset, src_stride, pred_offset, pred_stride, recon_offset, recon_stride,
                                           qparam_index, iscan_offset, C.byref(h)))
        return h.value

    def flush(self):
        _check(lib().svthip_tu_batcher_flush(self._h))

    def result(self, handle) -> TuResult:
        r = TuResult()
        _check(lib().svthip_tu_batcher_result(self._h, handle, C.byref(r)))
        return r

    def read_coeffs(self, handle, n):
        q = np.zeros(n, np.int32); dq = np.zeros(n, np.int32)
        _check(lib().svthip_tu_batcher_read_coeffs(self._h, handle, q.ctypes.data, dq.ctypes.data))
        return q, dq

    def pools(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().svthip_tu_batcher_pools(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


CONVOLVE_COMPOUND_DESC_DTYPE = np.dtype([("src0_offset", "<u4"), ("src1_offset", "<u4"), ("dst_offset", "<u4"), ("subpel0", "u1"), ("subpel1", "u1"),
                                         ("filter_x", "u1"), ("filter_y", "u1")])
assert CONVOLVE_COMPOUND_DESC_DTYPE.itemsize == 16


def _av1_convolve_compound_batch_dev(self, d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, n_blocks, width, height, stream=None):
    """BI_PRED luma prediction of n_blocks blocks of one size (CONVOLVE_COMPOUND_DESC_DTYPE descriptors; subpel = x | y << 4)."""
    _check(lib().svthip_av1_convolve_compound_batch_dev(self._h, d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, n_blocks, width,
                                                        height, stream))


Context.av1_convolve_compound_batch_dev = _av1_convolve_compound_batch_dev


def _av1_highbd_convolve_batch_dev(self, d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, compound, n_blocks, width, height,
                                   bit_depth=10, stream=None):
    """10-bit inter prediction in 16-bit planes (offsets / strides in samples); compound selects the descriptor type."""
    _check(lib().svthip_av1_highbd_convolve_batch_dev(self._h, d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, int(compound),
                                                      n_blocks, width, height, bit_depth, stream))


Context.av1_highbd_convolve_batch_dev = _av1_highbd_convolve_batch_dev


def _sad_loop_batch_dev(self, d_src, src_stride, d_ref, ref_stride, ref_stride_raw, d_desc, n_blocks, width, height, sw, sh, d_best_sad, d_best_xy,
                        stream=None):
    """SadLoopKernel over n_blocks (src_offset, ref_offset) uint32 pairs; d_best_sad uint32 [n], d_best_xy int16 [n][2] = (x, y) index."""
    _check(lib().svthip_sad_loop_batch_dev(self._h, d_src, src_stride, d_ref, ref_stride, ref_stride_raw, d_desc, n_blocks, width, height, sw, sh,
                                           d_best_sad, d_best_xy, stream))


Context.sad_loop_batch_dev = _sad_loop_batch_dev


CONVOLVE_DESC_DTYPE = np.dtype([("src_offset", "<u4"), ("dst_offset", "<u4"), ("subpel_x", "u1"), ("subpel_y", "u1"), ("filter_x", "u1"),
                                ("filter_y", "u1"), ("reserved", "<u4")])
assert CONVOLVE_DESC_DTYPE.itemsize == 16
# the 22 AV1 block sizes (width, height)
AV1_BLOCK_SIZES_WH = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64),
                      (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def _av1_convolve_sr_batch_dev(self, d_src, src_stride, d_dst, dst_stride, d_desc, n_blocks, width, height, stream=None):
    """8-bit single-reference AV1 convolution (2-D / x / y / copy by phase) of n_blocks blocks of one size."""
    _check(lib().svthip_av1_convolve_sr_batch_dev(self._h, d_src, src_stride, d_dst, dst_stride, d_desc, n_blocks, width, height, stream))


Context.av1_convolve_sr_batch_dev = _av1_convolve_sr_batch_dev


def make_fullpel_desc(cur, ref, centers=None, search_w=64, search_h=64) -> np.ndarray:
    """Descriptors for every SB of a picture the way MotionEstimateLcu derives them
    (Codec/EbMotionEstimation.c:6667-6738): window centred on `centers[sb] = (x, y)` (default 0,0),
    clipped to the picture, source block at the SB origin."""
    from . import synth

    nx, ny = cur.sb_grid()
    out = np.zeros((nx * ny, 6), dtype=np.int32)
    for sy in range(ny):
        for sx in range(nx):
            i = sy * nx + sx
            ox, oy = sx * 64, sy * 64
            cx, cy = (0, 0) if centers is None else centers[i]
            xo, yo, sw, sh = synth.clamp_search_window(ox, oy, int(cx), int(cy), search_w, search_h, cur.width, cur.height)
            out[i] = [(synth.PAD_FULL + oy) * cur.stride + synth.PAD_FULL + ox,
                      (synth.PAD_FULL + oy + yo) * ref.stride + synth.PAD_FULL + ox + xo, xo, yo, sw, sh]
    return out


# ------------------------------------------------------------------------------------------------
# Hierarchical ME: ctypes mirrors of the parameter blocks in include/svtav1_hip.h
# ------------------------------------------------------------------------------------------------
class PaPictureDesc(C.Structure):
    _fields_ = [("full_offset", C.c_int64), ("quarter_offset", C.c_int64), ("sixteenth_offset", C.c_int64),
                ("full_stride", C.c_uint32), ("quarter_stride", C.c_uint32), ("sixteenth_stride", C.c_uint32),
                ("width", C.c_uint16), ("height", C.c_uint16)]


class MeParams(C.Structure):
    _fields_ = [("search_area_width", C.c_uint16), ("search_area_height", C.c_uint16),
                ("number_hme_search_region_in_width", C.c_uint16), ("number_hme_search_region_in_height", C.c_uint16),
                ("hme_level0_total_search_area_width", C.c_uint16), ("hme_level0_total_search_area_height", C.c_uint16),
                ("hme_level0_search_area_in_width_array", C.c_uint16 * 2), ("hme_level0_search_area_in_height_array", C.c_uint16 * 2),
                ("hme_level1_search_area_in_width_array", C.c_uint16 * 2), ("hme_level1_search_area_in_height_array", C.c_uint16 * 2),
                ("hme_level2_search_area_in_width_array", C.c_uint16 * 2), ("hme_level2_search_area_in_height_array", C.c_uint16 * 2),
                ("hme_level0_multiplier_x", C.c_uint32), ("hme_level0_multiplier_y", C.c_uint32),
                ("enable_hme_flag", C.c_uint8), ("enable_hme_level0_flag", C.c_uint8), ("enable_hme_level1_flag", C.c_uint8),
                ("enable_hme_level2_flag", C.c_uint8), ("temporal_layer_index", C.c_uint8),
                ("is_used_as_reference_flag", C.c_uint8), ("ref_poc_equal", C.c_uint8), ("reserved", C.c_uint8)]


class MeCuResult(C.Structure):
    _fields_ = [("xMvL0", C.c_int16), ("yMvL0", C.c_int16), ("xMvL1", C.c_int16), ("yMvL1", C.c_int16),
                ("distortion", C.c_uint32 * 3), ("direction", C.c_uint8 * 3), ("totalMeCandidateIndex", C.c_uint8)]


ME_CU_RESULT_DTYPE = np.dtype([("xMvL0", "<i2"), ("yMvL0", "<i2"), ("xMvL1", "<i2"), ("yMvL1", "<i2"), ("distortion", "<u4", 3),
                               ("direction", "u1", 3), ("totalMeCandidateIndex", "u1")])
assert ME_CU_RESULT_DTYPE.itemsize == C.sizeof(MeCuResult) == 24


QUANT_DESC_DTYPE = np.dtype([("coeff_offset", "<u4"), ("iscan_offset", "<u4"), ("qparam_index", "<u4"), ("n_coeffs", "<u2"),
                             ("log_scale", "u1"), ("highbd", "u1")])
assert QUANT_DESC_DTYPE.itemsize == 16


TXFM_DESC_DTYPE = np.dtype([("in_offset", "<u4"), ("out_offset", "<u4"), ("in_stride", "<u2"), ("tx_type", "u1"), ("reserved", "u1")])
assert TXFM_DESC_DTYPE.itemsize == 12
ITXFM_DESC_DTYPE = np.dtype([("coeff_offset", "<u4"), ("recon_offset", "<u4"), ("recon_stride", "<u2"), ("tx_type", "u1"),
                             ("reserved", "u1")])
assert ITXFM_DESC_DTYPE.itemsize == 12
TU_DESC_DTYPE = np.dtype([("src_offset", "<u4"), ("pred_offset", "<u4"), ("recon_offset", "<u4"), ("coeff_offset", "<u4"),
                          ("iscan_offset", "<u4"), ("src_stride", "<u2"), ("pred_stride", "<u2"), ("recon_stride", "<u2"),
                          ("qparam_index", "<u2"), ("tx_type", "u1"), ("reserved", "u1", (3,))])
assert TU_DESC_DTYPE.itemsize == 32

# the 19 AV1 transform sizes (width, height), TxSize order (Codec/EbDefinitions.h)
TX_SIZES_WH = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
               (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
_VTX = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3]  # vtx_tab / htx_tab (Codec/EbTransforms.h:88-97): 0 DCT 1 ADST 2 FLIPADST 3 IDTX
_HTX = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2]


def valid_tx_types(w: int, h: int):
    """TxTypes for which the reference has 1-D networks at this size (ADST up to 16 points, identity up to 32, 64 DCT only)."""
    out = []
    for t in range(16):
        kc, kr = _VTX[t], _HTX[t]
        if (kc in (1, 2) and h > 16) or (kr in (1, 2) and w > 16) or (kc == 3 and h > 32) or (kr == 3 and w > 32):
            continue
        out.append(t)
    return out


class SbOrigin(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16)]


# HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_X / _Y [hierarchical_levels][temporal_layer_index]
# (Codec/EbDefinitions.h:2980-2996); X and Y tables are identical in the reference.
HME_LEVEL0_MULTIPLIER = [[100], [100, 100], [100, 100, 100], [200, 140, 100, 70], [350, 200, 100, 100, 100],
                         [525, 350, 200, 100, 100, 100]]


def default_me_params(width: int, height: int, hierarchical_levels: int = 3, temporal_layer_index: int = 0,
                      is_ref: bool = True, ref_poc_equal: bool = False) -> MeParams:
    """set_me_hme_params_oq() for enc modes M0..M3 (column 0 of the tables, Codec/EbDefinitions.h:3332-3465;
    resolution class per Codec/EbMotionEstimationProcess.c:104-110)."""
    px = width * height
    ratio = width // height
    if px < 1280 * 720 * 0.75:  # INPUT_SIZE_576p_RANGE_OR_LOWER (Codec/EbDefinitions.h input-size thresholds)
        ri = 0
    elif px < 1920 * 1080 * 0.75 and ratio < 2:
        ri = 1
    elif px <= 1920 * 1088 * 1.5:
        ri = 3
    else:
        ri = 4
    tot_w = [48, 64, 96, 96, 128][ri]
    tot_h = [40, 48, 48, 48, 80][ri]
    p = MeParams()
    p.search_area_width, p.search_area_height = 64, 64
    p.number_hme_search_region_in_width = p.number_hme_search_region_in_height = 2
    p.hme_level0_total_search_area_width, p.hme_level0_total_search_area_height = tot_w, tot_h
    for k in range(2):
        p.hme_level0_search_area_in_width_array[k] = tot_w // 2
        p.hme_level0_search_area_in_height_array[k] = tot_h // 2
        p.hme_level1_search_area_in_width_array[k] = 16
        p.hme_level1_search_area_in_height_array[k] = 16
        p.hme_level2_search_area_in_width_array[k] = 8
        p.hme_level2_search_area_in_height_array[k] = 8
    m = HME_LEVEL0_MULTIPLIER[hierarchical_levels][temporal_layer_index]
    p.hme_level0_multiplier_x = p.hme_level0_multiplier_y = m
    p.enable_hme_flag = p.enable_hme_level0_flag = p.enable_hme_level1_flag = p.enable_hme_level2_flag = 1
    p.temporal_layer_index = temporal_layer_index
    p.is_used_as_reference_flag = int(is_ref)
    p.ref_poc_equal = int(ref_poc_equal)
    return p


def build_picture_pool(pictures):
    """Stack the three planes of each PaPicture into one uint8 pool (4-byte aligned planes).
    Returns (pool, [PaPictureDesc])."""
    chunks, descs, off = [], [], 0
    for p in pictures:
        d = PaPictureDesc()
        for name, arr in (("full", p.full), ("quarter", p.quarter), ("sixteenth", p.sixteenth)):
            flat = arr.reshape(-1)
            padn = (-flat.size) % 16
            setattr(d, name + "_offset", off)
            setattr(d, name + "_stride", arr.shape[1])
            chunks.append(flat)
            if padn:
                chunks.append(np.zeros(padn, np.uint8))
            off += flat.size + padn
        d.width, d.height = p.width, p.height
        descs.append(d)
    chunks.append(np.zeros(256, np.uint8))   # the kernels' aligned-group loads may run up to 64 bytes past the last plane
    return np.concatenate(chunks), descs


def sb_origins(width: int, height: int) -> np.ndarray:
    nx, ny = (width + 63) // 64, (height + 63) // 64
    out = np.zeros((nx * ny, 2), dtype=np.uint16)
    for sy in range(ny):
        for sx in range(nx):
            out[sy * nx + sx] = (sx * 64, sy * 64)
    return out


def shard_sb_rows(width: int, height: int, world: int, rank: int) -> np.ndarray:
    """Contiguous SB-ROW partition of one picture across `world` ranks (1080p -> 17 rows -> 3/2/2/...): the indices
    (into sb_origins(width, height)) of the SBs rank `rank` owns.  See svtav1_hip.sharded for the multi-GPU layer built on it."""
    from .sharded import shard_sb_indices

    return shard_sb_indices(width, height, world, rank, "row")
