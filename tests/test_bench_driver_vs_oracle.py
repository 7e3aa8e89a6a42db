"""bench.py's per-leg CPU baselines run the REFERENCE's own functions in batch loops (oracle/ref_bench_driver.c ->
oracle/_ref/libsvtref_bench.so).  This checks that those loops compute the same thing the GPU legs compute: the reference chain
(ResidualKernel_c -> Av1TransformTwoD_NxN_c -> aom_quantize_b*_c_II -> av1_inv_txfm_add_c) reconstructs exactly what the oracle chain
(= the HIP fused kernel's checker) reconstructs, and the batched convolution equals the oracle's.  CPU only; skipped without oracle/_ref."""
import ctypes as C
import os

import numpy as np
import pytest

import svtav1_hip
from tq_util import RealTables, frame_encode_batch, oracle_encode_batch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref_bench.so")


@pytest.fixture(scope="module")
def refbench():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libsvtref_bench.so not built")
    return C.CDLL(SO)


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_reference_tq_chain_loop_equals_the_oracle_chain(oracle, refbench, n):
    tables = RealTables()
    rng = np.random.default_rng(n)
    b = frame_encode_batch(rng, n, n, 256, 128, tables)
    want = oracle_encode_batch(oracle, b)
    d = b["desc"]
    ts = svtav1_hip.TX_SIZES_WH.index((n, n))
    nc = min(n, 32) ** 2
    scan_ptrs, iscan_ptrs, keep = (C.c_void_p * 16)(), (C.c_void_p * 16)(), []
    for t in svtav1_hip.valid_tx_types(n, n):
        if int(tables.scan_index[ts, t]) < 0:
            continue
        o = tables.scan_offset(ts, t)
        s, i = np.ascontiguousarray(tables.scan_pool[o:o + nc]), np.ascontiguousarray(tables.iscan_pool[o:o + nc])
        keep += [s, i]
        scan_ptrs[t], iscan_ptrs[t] = s.ctypes.data, i.ctypes.data
    recon = np.zeros_like(b["pred"])
    f = refbench.ref_bench_tq_chain
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    offs = np.ascontiguousarray(d["src_offset"].astype(np.uint32))
    tt = np.ascontiguousarray(d["tx_type"].astype(np.uint8))
    qi = np.ascontiguousarray(d["qparam_index"].astype(np.uint8))
    src = b["src"].copy()
    pred = b["pred"].copy()
    f(src.ctypes.data, pred.ctypes.data, recon.ctypes.data, 256, offs.ctypes.data, tt.ctypes.data, qi.ctypes.data, len(d), n, b["qparams"].ctypes.data,
      scan_ptrs, iscan_ptrs)
    assert np.array_equal(recon, want["recon"]), n
    assert np.array_equal(pred, b["pred"]) and want["eob"].sum() > 0


def test_reference_convolve_loop_equals_the_oracle(oracle, refbench):
    rng = np.random.default_rng(5)
    S, D, n = 200, 64, 24
    plane = rng.integers(0, 256, (160, S)).astype(np.uint8)
    src_off = (rng.integers(8, 80, n) * S + rng.integers(8, 120, n)).astype(np.uint32)
    dst_off = (np.arange(n) * 4096).astype(np.uint32)
    sx, sy, fx, fy = rng.integers(0, 16, n), rng.integers(0, 16, n), rng.integers(0, 4, n), rng.integers(0, 4, n)
    mode = (sx | (sy << 4) | (fx << 8) | (fy << 12)).astype(np.uint16)
    got = np.zeros(n * 4096, np.uint8)
    f = refbench.ref_bench_convolve
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
    f(plane.ctypes.data, S, got.ctypes.data, D, src_off.ctypes.data, dst_off.ctypes.data, mode.ctypes.data, n, 64, 64)
    want = np.zeros(n * 4096, np.uint8)
    desc = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    desc["src_offset"], desc["dst_offset"] = src_off, dst_off
    desc["subpel_x"], desc["subpel_y"], desc["filter_x"], desc["filter_y"] = sx, sy, fx, fy
    g = oracle.lib.orc_av1_convolve_sr_batch
    g.restype = None
    g.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
    g(plane.ctypes.data, S, want.ctypes.data, D, desc.ctypes.data, n, 64, 64)
    assert np.array_equal(got, want)
