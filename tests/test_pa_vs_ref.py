"""Pins the picture-analysis oracle (oracle/svt_pa_oracle.c: generate_padding, generate_padding16_bit, Decimation2D and the
DecimateInputPicture sequence) against the reference's own functions compiled into oracle/_ref/libsvtref_me.so, and checks that
the numpy planes the tests build (svtav1_hip.synth.PaPicture) are the same thing.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from svtav1_hip import synth


@pytest.fixture(scope="module")
def refme():
    from oracle.binding import ReferenceME
    if not ReferenceME.available():
        pytest.skip("oracle/_ref/libsvtref_me.so not built")
    return ReferenceME()


@pytest.mark.parametrize("geom", [(64, 48, 68, 68), (856, 480, 68, 68), (214, 120, 16, 16), (40, 24, 8, 3), (16, 8, 160, 160)])
def test_generate_padding_matches_reference(oracle, refme, geom):
    w, h, pw, ph = geom
    rng = np.random.default_rng(w * 7 + h)
    stride = w + 2 * pw + 4  # a few spare bytes per row, like strides rounded up
    a = rng.integers(0, 256, (h + 2 * ph, stride), dtype=np.uint8)
    b = a.copy()
    f = refme.lib.generate_padding
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    f(a.ctypes.data, stride, w, h, pw, ph)
    oracle.generate_padding(b, w, h, pw, ph)
    assert np.array_equal(a, b)
    assert np.array_equal(b[:, :w + 2 * pw], np.pad(b[ph:ph + h, pw:pw + w], ((ph, ph), (pw, pw)), mode="edge"))


@pytest.mark.parametrize("geom", [(64, 48, 80, 80), (120, 72, 160, 160), (60, 36, 80, 80)])
def test_generate_padding16_matches_reference(oracle, refme, geom):
    """generate_padding16_bit takes BYTE quantities (PadRefAndSetFlags passes stride << 1, width << 1, origin_x << 1,
    Codec/EbEncDecProcess.c:1176-1201); the oracle takes samples."""
    w, h, pw, ph = geom
    rng = np.random.default_rng(w + h)
    stride = w + 2 * pw
    a = rng.integers(0, 1024, (h + 2 * ph, stride), dtype=np.uint16)
    b = a.copy()
    f = refme.lib.generate_padding16_bit
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    f(a.ctypes.data, stride << 1, w << 1, h, pw << 1, ph)
    oracle.generate_padding(b, w, h, pw, ph)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("size", [(64, 64), (856, 480), (328, 200), (1920, 1080)])
def test_pa_planes_match_reference_sequence(oracle, refme, size):
    """The whole DecimateInputPicture sequence with the reference's Decimation2D + generate_padding vs the oracle, and vs the
    numpy planes of synth.PaPicture that every other test feeds to the kernels."""
    w, h = size
    luma = synth.synth_luma(w, h, 2)
    full, quarter, sixteenth = oracle.pa_derive_planes(luma)
    gp = refme.lib.generate_padding
    gp.restype = None
    gp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    dec = refme.lib.Decimation2D
    dec.restype = None
    dec.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    rf = np.full_like(full, 0xA5); rf[68:68 + h, 68:68 + w] = luma
    rq = np.full_like(quarter, 0x5A); rs = np.full_like(sixteenth, 0x3C)
    gp(rf.ctypes.data, rf.shape[1], w, h, 68, 68)
    pic = rf.ctypes.data + 68 * rf.shape[1] + 68
    dec(pic, rf.shape[1], w, h, rq.ctypes.data + 32 * rq.shape[1] + 32, rq.shape[1], 2)
    gp(rq.ctypes.data, rq.shape[1], w >> 1, h >> 1, 32, 32)
    dec(pic, rf.shape[1], w, h, rs.ctypes.data + 16 * rs.shape[1] + 16, rs.shape[1], 4)
    gp(rs.ctypes.data, rs.shape[1], w >> 2, h >> 2, 16, 16)
    assert np.array_equal(full, rf) and np.array_equal(quarter, rq) and np.array_equal(sixteenth, rs)
    p = synth.PaPicture(luma)
    assert np.array_equal(p.full, full) and np.array_equal(p.quarter, quarter) and np.array_equal(p.sixteenth, sixteenth)
