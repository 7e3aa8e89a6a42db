"""Row a19: the quantiser on the reference's REAL tables.  tests/golden/quant_tables.npz holds the rows av1_build_quantizer
(Codec/EbModeDecisionConfigurationProcess.c:417-506) builds and the scan orders of av1_scan_orders (Codec/EbTransforms.h:3336), both
produced by the reference's own code (tests/golden/make_golden.py).  Here: the fixture is what the reference produces now (when
oracle/_ref is present), and the oracle quantiser equals the reference's aom_quantize_b* / aom_highbd_quantize_b* on those rows and
scans at the SURVEY 8d config-4 qindices 20 / 120 / 200.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

import svtav1_hip
from tq_util import RealTables

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TQ_SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref_tq.so")
ME_SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")


@pytest.fixture(scope="module")
def tables():
    return RealTables()


def test_fixture_is_what_the_reference_builds(tables):
    if not os.path.exists(ME_SO):
        pytest.skip("oracle/_ref not built")
    me = C.CDLL(ME_SO, mode=os.RTLD_LAZY)
    for bd in (8, 10):
        for kind, dq in (("inter", -20), ("intra", -10), ("flat", 0)):
            rows = np.zeros((256, 3, 10), np.int16)
            assert me.ref_build_quantizer_rows(bd, 0, dq, dq, dq, dq, C.c_void_p(rows.ctypes.data)) == 0
            assert np.array_equal(rows, tables.rows(bd, kind))
    for ts, (w, h) in enumerate(svtav1_hip.TX_SIZES_WH):
        n = min(w, 32) * min(h, 32)
        for tt in range(16):
            scan = np.zeros(n, np.int16); iscan = np.zeros(n, np.int16)
            got = me.ref_scan_order(ts, tt, C.c_void_p(scan.ctypes.data), C.c_void_p(iscan.ctypes.data))
            assert got == n
            o = tables.scan_offset(ts, tt)
            assert np.array_equal(tables.scan_pool[o:o + n], scan) and np.array_equal(tables.iscan_pool[o:o + n], iscan)


def test_real_rows_have_the_documented_structure(tables):
    """dequant = the AV1 quantiser step (monotone in qindex), zbin / round derive from it as av1_build_quantizer states, and every scan
    / iscan pair is a permutation and its inverse."""
    for bd in (8, 10):
        r = tables.rows(bd, "flat").astype(np.int32)
        dq = r[:, 0, 8:10]
        assert (np.diff(dq[:, 0]) >= 0).all() and (np.diff(dq[:, 1]) >= 0).all() and dq.min() >= 4
        q = np.arange(256)
        rounding = np.where(q == 0, 64, 48)[:, None]
        assert np.array_equal(r[:, 0, 2:4], (rounding * dq) >> 7)
    for k in range(len(tables.scan_offsets) - 1):
        a, b = int(tables.scan_offsets[k]), int(tables.scan_offsets[k + 1])
        sc, isc = tables.scan_pool[a:b], tables.iscan_pool[a:b]
        assert np.array_equal(np.sort(sc), np.arange(b - a)) and np.array_equal(isc[sc], np.arange(b - a))


QUANT_FN = {(0, 0): "aom_quantize_b_c_II", (1, 0): "aom_quantize_b_32x32_c_II", (2, 0): "aom_quantize_b_64x64_c_II",
            (0, 1): "aom_highbd_quantize_b_c", (1, 1): "aom_highbd_quantize_b_32x32_c", (2, 1): "aom_highbd_quantize_b_64x64_c"}


def _log_scale(w, h):  # av1_get_tx_scale: pixels > 256 -> 1, > 1024 -> 2
    return int(w * h > 256) + int(w * h > 1024)


@pytest.mark.parametrize("qindex", [20, 120, 200])
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_oracle_quantiser_on_real_rows_and_scans(oracle, tables, qindex, bit_depth):
    if not os.path.exists(TQ_SO):
        pytest.skip("oracle/_ref/libsvtref_tq.so not built")
    reftq = C.CDLL(TQ_SO)
    orc = oracle.lib.orc_quantize_b
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(qindex * 16 + bit_depth)
    highbd = int(bit_depth == 10)
    for ts, (w, h) in enumerate(svtav1_hip.TX_SIZES_WH):
        n = min(w, 32) * min(h, 32)
        f = getattr(reftq, QUANT_FN[(_log_scale(w, h), highbd)])
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int32] + [C.c_void_p] * 10
        for plane in (0, 1):
            qp = np.ascontiguousarray(tables.rows(bit_depth, "inter")[qindex, plane])
            for tt in (0, 10, 11):  # default zig-zag, and the two 1-D classes (row / column scans)
                o = tables.scan_offset(ts, tt)
                scan = np.ascontiguousarray(tables.scan_pool[o:o + n]); iscan = np.ascontiguousarray(tables.iscan_pool[o:o + n])
                for kind in range(3):
                    dq_ac = int(qp[9])
                    if kind == 0:
                        coeff = rng.laplace(0, 6 << (bit_depth - 8 + 3), n).astype(np.int32)   # Laplacian residual after an 8x gain
                    elif kind == 1:
                        coeff = rng.integers(-3 * dq_ac, 3 * dq_ac + 1, n).astype(np.int32)    # around the dead zone
                    else:
                        coeff = rng.integers(-(1 << (bit_depth + 9)), 1 << (bit_depth + 9), n).astype(np.int32)
                    rq = np.full(n, 7, np.int32); rdq = np.full(n, 7, np.int32); reob = C.c_uint16(999)
                    f(coeff.ctypes.data, n, 0, qp[0:2].ctypes.data, qp[2:4].ctypes.data, qp[4:6].ctypes.data, qp[6:8].ctypes.data,
                      rq.ctypes.data, rdq.ctypes.data, qp[8:10].ctypes.data, C.addressof(reob), scan.ctypes.data, iscan.ctypes.data)
                    oq = np.full(n, 9, np.int32); odq = np.full(n, 9, np.int32); oeob = C.c_uint16(0)
                    orc(coeff.ctypes.data, n, qp.ctypes.data, scan.ctypes.data, _log_scale(w, h), highbd, oq.ctypes.data, odq.ctypes.data,
                        C.addressof(oeob))
                    assert np.array_equal(oq, rq) and np.array_equal(odq, rdq) and oeob.value == reob.value, (w, h, plane, tt, kind)
