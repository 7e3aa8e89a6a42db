"""Pins the transform / quantisation oracle against the reference's own C functions (oracle/_ref/libsvtref_tq.so). CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TQ_SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref_tq.so")


@pytest.fixture(scope="module")
def reftq():
    if not os.path.exists(TQ_SO):
        pytest.skip("oracle/_ref/libsvtref_tq.so not built")
    return C.CDLL(TQ_SO)


def make_qparams(rng, qstep_dc, qstep_ac):
    """A plausible quantiser row built the way av1_build_quantizer does (Codec/EbModeDecisionConfigurationProcess.c:417-506):
    quant = (1 << 16) / q via invert_quant, zbin = 84 * q / 128 rounded, round = 64 * q / 128, dequant = q."""
    out = np.zeros(10, np.int16)
    for i, q in enumerate((qstep_dc, qstep_ac)):
        l = int(q).bit_length() - 1
        m = 1 + (1 << (16 + l)) // q
        out[4 + i] = np.int16(np.uint16((m - (1 << 16)) & 0xffff))  # quant
        out[6 + i] = 1 << (16 - l)                                 # quant_shift
        out[0 + i] = (84 * q + 64) >> 7                            # zbin
        out[2 + i] = (64 * q) >> 7                                 # round
        out[8 + i] = q                                             # dequant
    return out


QUANT_FUNCS = [("aom_quantize_b_c_II", 0, 0), ("aom_quantize_b_32x32_c_II", 1, 0), ("aom_quantize_b_64x64_c_II", 2, 0),
               ("aom_highbd_quantize_b_c", 0, 1), ("aom_highbd_quantize_b_32x32_c", 1, 1), ("aom_highbd_quantize_b_64x64_c", 2, 1)]


@pytest.mark.parametrize("fn", QUANT_FUNCS)
def test_quantize_matches_reference(oracle, reftq, fn):
    name, log_scale, highbd = fn
    f = getattr(reftq, name)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int32] + [C.c_void_p] * 10
    orc = oracle.lib.orc_quantize_b
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(100 + log_scale * 7 + highbd)
    for trial in range(60):
        n = int(rng.choice([16, 64, 256, 1024]))
        q_dc, q_ac = int(rng.integers(4, 1337)), int(rng.integers(4, 1829))
        qp = make_qparams(rng, q_dc, q_ac)
        kind = trial % 4
        if kind == 0:
            coeff = rng.integers(-40, 41, n)
        elif kind == 1:
            coeff = (rng.laplace(0, 6 * q_ac, n)).astype(np.int64)
        elif kind == 2:
            coeff = rng.integers(-(1 << 17), 1 << 17, n)   # beyond int16: exercises the 8-bit clamp
        else:
            coeff = np.zeros(n, np.int64); coeff[rng.integers(0, n, 3)] = rng.integers(-3000, 3000, 3)
        coeff = coeff.astype(np.int32)
        scan = rng.permutation(n).astype(np.int16)
        iscan = np.zeros(n, np.int16); iscan[scan] = np.arange(n, dtype=np.int16)
        rq = np.full(n, 7, np.int32); rdq = np.full(n, 7, np.int32); reob = C.c_uint16(999)
        f(coeff.ctypes.data, n, 0, qp[0:2].ctypes.data, qp[2:4].ctypes.data, qp[4:6].ctypes.data, qp[6:8].ctypes.data,
          rq.ctypes.data, rdq.ctypes.data, qp[8:10].ctypes.data, C.addressof(reob), scan.ctypes.data, iscan.ctypes.data)
        oq = np.full(n, 9, np.int32); odq = np.full(n, 9, np.int32); oeob = C.c_uint16(0)
        orc(coeff.ctypes.data, n, qp.ctypes.data, scan.ctypes.data, log_scale, highbd, oq.ctypes.data, odq.ctypes.data, C.addressof(oeob))
        assert np.array_equal(oq, rq), (name, trial)
        assert np.array_equal(odq, rdq), (name, trial)
        assert oeob.value == reob.value, (name, trial)
