"""Every same-signature drop-in of include/svtav1_hip_rtcd.h called through ctypes exactly as a C host would call the pointer it
replaces, against the oracle (which is pinned to the reference's own C functions, tests/test_tq_vs_ref.py): the 19 forward transforms,
the 19 inverse transforms on 16-bit planes at bd 8 and 10 (the three reference signatures), av1_inv_txfm_add on 8-bit planes with a
TxfmParam, the six quantisers at the coefficient counts and log-scales their callers use, and the two leaf SAD pointers."""
import ctypes as C

import numpy as np
import pytest

import svtav1_hip

pytestmark = pytest.mark.gpu

SQUARE = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)]
RECT4 = [(4, 8), (8, 4), (4, 16), (16, 4)]


def _orc(oracle):
    L = oracle.lib
    L.orc_fwd_txfm2d.restype = None
    L.orc_fwd_txfm2d.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_inv_txfm2d_add.restype = None
    L.orc_inv_txfm2d_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_quantize_b.restype = None
    L.orc_quantize_b.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("size", svtav1_hip.TX_SIZES_WH)
def test_forward_shims_every_size_and_type(hip_ctx, oracle, size):
    w, h = size
    L, O = svtav1_hip.lib(), _orc(oracle)
    fn = getattr(L, f"svthip_av1_fwd_txfm2d_{w}x{h}")
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint8]
    rng = np.random.default_rng(w * 100 + h)
    for bd in (8, 10):
        for t in svtav1_hip.valid_tx_types(w, h):
            stride = w + 8
            res = rng.integers(-(1 << bd) + 1, 1 << bd, (h, stride)).astype(np.int16)
            got = np.zeros(w * h, np.int32)
            want = np.zeros(w * h, np.int32)
            fn(res.ctypes.data, got.ctypes.data, stride, t, bd)
            O.orc_fwd_txfm2d(res.ctypes.data, stride, w, h, t, want.ctypes.data)
            assert np.array_equal(got, want), (size, t, bd)


@pytest.mark.parametrize("size", svtav1_hip.TX_SIZES_WH)
def test_inverse_shims_every_size_and_type(hip_ctx, oracle, size):
    w, h = size
    L, O = svtav1_hip.lib(), _orc(oracle)
    ts = svtav1_hip.TX_SIZES_WH.index(size)
    fn = getattr(L, f"svthip_av1_inv_txfm2d_add_{w}x{h}")
    fn.restype = None
    if size in SQUARE:       # (input, output, stride, tx_type, bd)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_int32]
        call = lambda c, r, s, t, bd: fn(c, r, s, t, bd)                    # noqa: E731
    elif size in RECT4:      # (..., tx_type, tx_size, bd)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32]
        call = lambda c, r, s, t, bd: fn(c, r, s, t, ts, bd)                # noqa: E731
    else:                    # (..., tx_type, tx_size, eob, bd)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32]
        call = lambda c, r, s, t, bd: fn(c, r, s, t, ts, 1024, bd)          # noqa: E731
    add8 = L.svthip_av1_inv_txfm_add
    add8.restype = None
    add8.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    rng = np.random.default_rng(w * 131 + h)
    n = min(w, 32) * min(h, 32)   # 64-point dimensions read the packed 32-wide block, like the reference
    for bd in (8, 10):
        for t in svtav1_hip.valid_tx_types(w, h):
            coeff = (rng.laplace(0, 40 << (bd - 8), n) * (rng.random(n) < 0.4)).astype(np.int32)
            coeff[0] = 700 << (bd - 8)
            stride = w + 6
            pred = rng.integers(0, 1 << bd, (h, stride)).astype(np.uint16)
            got, want = pred.copy(), pred.copy()
            call(coeff.ctypes.data, got.ctypes.data, stride, t, bd)
            O.orc_inv_txfm2d_add(coeff.ctypes.data, want.ctypes.data, stride, w, h, t, bd)
            assert np.array_equal(got, want), (size, t, bd)
            assert not np.array_equal(got[:, :w], pred[:, :w])
            if bd == 8:   # the 8-bit pointer: TxfmParam {tx_type u8, tx_size u8, lossless, bd, is_hbd, tx_set_type u8, eob}
                prm = np.zeros(24, np.uint8)
                prm[0], prm[1] = t, ts
                prm[8:12] = np.frombuffer(np.int32(8).tobytes(), np.uint8)
                prm[12:16] = np.frombuffer(np.int32(1).tobytes(), np.uint8)
                prm[20:24] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
                g8 = pred.astype(np.uint8)
                add8(coeff.ctypes.data, g8.ctypes.data, stride, prm.ctypes.data)
                assert np.array_equal(g8, want.astype(np.uint8)), (size, t, "av1_inv_txfm_add")


@pytest.mark.parametrize("name,log_scale,highbd,counts", [
    ("svthip_aom_quantize_b", 0, 0, (16, 32, 64, 128, 256)), ("svthip_aom_quantize_b_32x32", 1, 0, (256, 512, 1024)),
    ("svthip_aom_quantize_b_64x64", 2, 0, (512, 1024)), ("svthip_aom_highbd_quantize_b", 0, 1, (16, 64, 256)),
    ("svthip_aom_highbd_quantize_b_32x32", 1, 1, (512, 1024)), ("svthip_aom_highbd_quantize_b_64x64", 2, 1, (1024,))])
def test_quantiser_shims(hip_ctx, oracle, name, log_scale, highbd, counts):
    from tq_util import RealTables
    L, O = svtav1_hip.lib(), _orc(oracle)
    fn = getattr(L, name)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int32] + [C.c_void_p] * 10
    tables = RealTables()
    rng = np.random.default_rng(len(name) + log_scale)
    for n in counts:
        for q in (20, 120, 200):
            row = np.ascontiguousarray(tables.rows(10 if highbd else 8, "inter")[q, 0])   # zbin[2] round[2] quant[2] quant_shift[2] dequant[2]
            scan = rng.permutation(n).astype(np.int16)
            iscan = np.zeros(n, np.int16)
            iscan[scan] = np.arange(n, dtype=np.int16)
            coeff = rng.laplace(0, 250 * (4 if highbd else 1), n).astype(np.int32)
            coeff[rng.integers(0, n)] = 30000 * (8 if highbd else 1)                        # exercises the clamp of the 8-bit path
            gq, gdq, ge = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(999)
            fn(coeff.ctypes.data, n, 0, row[0:].ctypes.data, row[2:].ctypes.data, row[4:].ctypes.data, row[6:].ctypes.data, gq.ctypes.data, gdq.ctypes.data,
               row[8:].ctypes.data, C.addressof(ge), scan.ctypes.data, iscan.ctypes.data)
            wq, wdq, we = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(0)
            O.orc_quantize_b(coeff.ctypes.data, n, row.ctypes.data, scan.ctypes.data, log_scale, highbd, wq.ctypes.data, wdq.ctypes.data, C.addressof(we))
            assert np.array_equal(gq, wq) and np.array_equal(gdq, wdq) and ge.value == we.value and we.value > 0, (name, n, q)


def test_leaf_sad_shims(hip_ctx, oracle):
    L = svtav1_hip.lib()
    sad = L.svthip_nxm_sad_kernel
    sad.restype = C.c_uint32
    sad.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    loop = L.svthip_sad_loop_kernel
    loop.restype = None
    loop.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int16, C.c_int16]
    u8p = C.POINTER(C.c_uint8)
    rng = np.random.default_rng(77)
    S = 200
    plane = rng.integers(0, 256, (200, S)).astype(np.uint8)
    src = rng.integers(0, 256, (64, 96)).astype(np.uint8)
    P = lambda a, off=0: C.cast(a.ctypes.data + off, u8p)   # noqa: E731
    for (w, h) in [(4, 4), (8, 8), (8, 4), (16, 16), (24, 32), (32, 32), (48, 64), (64, 64), (64, 32), (16, 5)]:
        assert sad(src.ctypes.data, 96, plane.ctypes.data + 3 * S + 5, S, h, w) == oracle.lib.orc_nxm_sad(P(src), 96, P(plane, 3 * S + 5), S, h, w), (w, h)
    for (w, h, sw, sh, k) in [(16, 16, 33, 33, 1), (8, 8, 16, 16, 1), (64, 32, 8, 8, 2), (32, 16, 48, 24, 2), (4, 4, 1, 1, 1), (16, 8, 64, 64, 1)]:
        gb, gx, gy = C.c_uint64(0), C.c_int16(-1), C.c_int16(-1)
        wb, wx, wy = C.c_uint64(0), C.c_int16(-1), C.c_int16(-1)
        loop(src.ctypes.data, 96, plane.ctypes.data + 2 * S + 7, k * S, h, w, C.addressof(gb), C.addressof(gx), C.addressof(gy), S, sw, sh)
        oracle.lib.orc_sad_loop_kernel(P(src), 96, P(plane, 2 * S + 7), k * S, h, w, C.byref(wb), C.byref(wx), C.byref(wy), S, sw, sh)
        assert (gb.value, gx.value, gy.value) == (wb.value, wx.value, wy.value), (w, h, sw, sh, k)
