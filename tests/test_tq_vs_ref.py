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


# ---------------------------------------------------------------------------------------------------------------------
# forward transforms
# ---------------------------------------------------------------------------------------------------------------------
def ref_fwd_name(w, h):
    return f"Av1TransformTwoD_{w}x{h}_c" if w == h else f"av1_fwd_txfm2d_{w}x{h}_c"


FWD_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
             (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def test_trig_tables_match_reference(oracle, reftq):
    """The oracle computes cos tables from the closed form; the reference holds them as data (EbTransforms.c:1242-1311)."""
    ref_cos = (C.c_int32 * (7 * 64)).in_dll(reftq, "av1_cospi_arr_data")
    ref_sin = (C.c_int32 * (7 * 5)).in_dll(reftq, "av1_sinpi_arr_data")
    oracle.lib.orc_cospi_table.restype = C.POINTER(C.c_int32)
    oracle.lib.orc_sinpi_table.restype = C.POINTER(C.c_int32)
    for bit in range(10, 17):
        oc = oracle.lib.orc_cospi_table(bit)
        os_ = oracle.lib.orc_sinpi_table(bit)
        assert [oc[j] for j in range(64)] == list(ref_cos[(bit - 10) * 64:(bit - 9) * 64])
        assert [os_[j] for j in range(5)] == list(ref_sin[(bit - 10) * 5:(bit - 9) * 5])


def fwd_inputs(rng, w, h, kind, bd):
    lim = (1 << bd) - 1
    if kind == 0:
        return rng.integers(-lim, lim + 1, (h, w))
    if kind == 1:
        return rng.choice([-lim, lim], (h, w))                      # worst-case magnitude
    if kind == 2:
        x = np.zeros((h, w), np.int64); x[rng.integers(0, h), rng.integers(0, w)] = int(rng.integers(-lim, lim + 1)); return x
    if kind == 3:
        yy, xx = np.mgrid[0:h, 0:w]
        return np.clip((40 * np.sin(xx / 3.0) + 25 * np.cos(yy / 5.0) + rng.integers(-4, 5, (h, w))).astype(np.int64), -lim, lim)
    return np.zeros((h, w), np.int64)


@pytest.mark.parametrize("size", FWD_SIZES)
def test_fwd_txfm2d_matches_reference(oracle, reftq, size):
    w, h = size
    f = getattr(reftq, ref_fwd_name(w, h))
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint8]
    orc = oracle.lib.orc_fwd_txfm2d
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    oracle.lib.orc_fwd_txfm2d_valid.restype = C.c_int
    rng = np.random.default_rng(w * 131 + h)
    n_checked = 0
    for tx_type in range(16):
        if oracle.lib.orc_fwd_txfm2d_valid(w, h, tx_type) != 0:
            continue
        for trial in range(10):
            bd = 10 if trial >= 5 else 8
            stride = w + int(rng.integers(0, 9))
            x = np.zeros((h, stride), np.int16)
            x[:, :w] = fwd_inputs(rng, w, h, trial % 5, bd)
            ro = np.full(h * w, 12345, np.int32); oo = np.full(h * w, 54321, np.int32)
            f(x.ctypes.data, ro.ctypes.data, stride, tx_type, bd)
            orc(x.ctypes.data, stride, w, h, tx_type, oo.ctypes.data)
            assert np.array_equal(oo, ro), (w, h, tx_type, trial, np.flatnonzero(oo != ro)[:8])
            n_checked += 1
    assert n_checked >= 10


# ---------------------------------------------------------------------------------------------------------------------
# inverse transforms + reconstruction
# ---------------------------------------------------------------------------------------------------------------------
INV_SIG5 = {(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)}            # (input, output, stride, tx_type, bd)
INV_SIG6 = {(4, 8), (8, 4), (4, 16), (16, 4)}                          # (..., tx_type, tx_size, bd)
TX_SIZE_ENUM = {s: i for i, s in enumerate(FWD_SIZES)}                 # FWD_SIZES is in TxSize order


def call_ref_inv(reftq, w, h, coeff, recon, stride, tx_type, bd):
    f = getattr(reftq, f"av1_inv_txfm2d_add_{w}x{h}_c")
    f.restype = None
    if (w, h) in INV_SIG5:
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int32]
        f(coeff.ctypes.data, recon.ctypes.data, stride, tx_type, bd)
    elif (w, h) in INV_SIG6:
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int32]
        f(coeff.ctypes.data, recon.ctypes.data, stride, tx_type, TX_SIZE_ENUM[(w, h)], bd)
    else:
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int32]
        f(coeff.ctypes.data, recon.ctypes.data, stride, tx_type, TX_SIZE_ENUM[(w, h)], 1, bd)


def inv_inputs(rng, n, kind, bd):
    """Dequantised-coefficient-like inputs; kinds 3/4 are far out of range and exercise every clamp of the networks."""
    if kind == 0:
        return rng.laplace(0, 40 << (bd - 8), n).astype(np.int64)
    if kind == 1:
        x = np.zeros(n, np.int64); k = rng.integers(0, n, 4); x[k] = rng.integers(-(1 << (bd + 6)), 1 << (bd + 6), 4); return x
    if kind == 2:
        return rng.integers(-(1 << (bd + 7)), 1 << (bd + 7), n)
    if kind == 3:
        return rng.integers(-(1 << 20), 1 << 20, n)
    if kind == 4:
        return rng.choice([-(1 << (bd + 7)), (1 << (bd + 7)) - 1], n)
    return np.zeros(n, np.int64)


@pytest.mark.parametrize("size", FWD_SIZES)
def test_inv_txfm2d_add_matches_reference(oracle, reftq, size):
    w, h = size
    orc = oracle.lib.orc_inv_txfm2d_add
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    oracle.lib.orc_fwd_txfm2d_valid.restype = C.c_int
    rng = np.random.default_rng(w * 977 + h)
    n_in = min(w, 32) * min(h, 32)
    n_checked = 0
    for tx_type in range(16):
        if oracle.lib.orc_fwd_txfm2d_valid(w, h, tx_type) != 0:
            continue
        for trial in range(12):
            bd = 10 if trial >= 6 else 8
            stride = w + int(rng.integers(0, 9))
            coeff = inv_inputs(rng, n_in, trial % 6, bd).astype(np.int32)
            pred = rng.integers(0, 1 << bd, (h, stride)).astype(np.uint16)
            r_rec = pred.copy(); o_rec = pred.copy()
            call_ref_inv(reftq, w, h, coeff, r_rec, stride, tx_type, bd)
            orc(coeff.ctypes.data, o_rec.ctypes.data, stride, w, h, tx_type, bd)
            assert np.array_equal(o_rec, r_rec), (w, h, tx_type, trial, np.argwhere(o_rec != r_rec)[:6])
            n_checked += 1
    assert n_checked >= 12


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_inv_txfm2d_residual_matches_reference(oracle, reftq, n):
    """Av1InverseTransformTwoD_NxN_c returns the residual before the pixel clip, so every clamp inside the networks shows."""
    f = getattr(reftq, f"Av1InverseTransformTwoD_{n}x{n}_c")
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint8]
    orc = oracle.lib.orc_inv_txfm2d
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int32]
    oracle.lib.orc_fwd_txfm2d_valid.restype = C.c_int
    rng = np.random.default_rng(n)
    for tx_type in range(16):
        if oracle.lib.orc_fwd_txfm2d_valid(n, n, tx_type) != 0:
            continue
        for trial in range(12):
            bd = 10 if trial >= 6 else 8
            coeff = inv_inputs(rng, n * n, trial % 6, bd).astype(np.int32)
            ro = np.full(n * n, 7, np.int32); oo = np.full(n * n, 9, np.int32)
            f(coeff.ctypes.data, n, ro.ctypes.data, n, tx_type, bd)
            orc(coeff.ctypes.data, n, n, n, tx_type, bd, oo.ctypes.data, n)
            assert np.array_equal(oo, ro), (n, tx_type, trial, np.flatnonzero(oo != ro)[:6])


# ---------------------------------------------------------------------------------------------------------------------
# glue: 64-point packing / energy, distortion, residual
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(64, 64), (64, 32), (32, 64), (64, 16), (16, 64)])
def test_pack_transform_matches_reference(oracle, reftq, size):
    w, h = size
    f = getattr(reftq, f"HandleTransform{w}x{h}_c")
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_uint32]
    orc = oracle.lib.orc_pack_transform
    orc.restype = C.c_uint64
    orc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(w + h)
    win, hin = min(w, 32), min(h, 32)
    for trial in range(8):
        full = rng.integers(-(1 << 20), 1 << 20, w * h).astype(np.int32)
        ref = full.copy()
        e_ref = f(ref.ctypes.data, w)
        packed = np.zeros(win * hin, np.int32)
        e_orc = orc(full.ctypes.data, w, h, packed.ctypes.data)
        assert e_orc == e_ref
        kept = ref.reshape(h, w)[:hin, :win]                           # what Av1EstimateTransform's memcpy loop re-packs
        assert np.array_equal(packed.reshape(hin, win), kept)
        assert not ref.reshape(h, w)[:, win:].any() and not ref.reshape(h, w)[hin:].any()


def test_full_distortion_and_residual_match_reference(oracle, reftq):
    fd = reftq.FullDistortionKernel32Bits
    fd.restype = None
    fd.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    od = oracle.lib.orc_full_distortion
    od.restype = None
    od.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(3)
    for n in (4, 8, 16, 32):
        a = rng.integers(-(1 << 22), 1 << 22, n * n).astype(np.int32)
        b = rng.integers(-(1 << 22), 1 << 22, n * n).astype(np.int32)
        r = np.zeros(2, np.uint64); o = np.zeros(2, np.uint64)
        fd(a.ctypes.data, n, b.ctypes.data, n, r.ctypes.data, n, n)
        od(a.ctypes.data, b.ctypes.data, n * n, o.ctypes.data)
        assert np.array_equal(r, o)
    rk = reftq.ResidualKernel_c
    rk.restype = None
    rk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    for n in (4, 8, 16, 32, 64):
        s = rng.integers(0, 256, (n, n + 5)).astype(np.uint8)
        p = rng.integers(0, 256, (n, n + 3)).astype(np.uint8)
        out = np.zeros((n, n), np.int16)
        rk(s.ctypes.data, n + 5, p.ctypes.data, n + 3, out.ctypes.data, n, n, n)
        assert np.array_equal(out, s[:, :n].astype(np.int16) - p[:, :n].astype(np.int16))
