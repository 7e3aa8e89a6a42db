"""Pins the AV1 single-reference convolution oracle (oracle/svt_interp_oracle.c) and its filter tables against the reference's own
av1_convolve_2d_sr_c / _x_sr_c / _y_sr_c / _2d_copy_sr_c driven like av1_inter_prediction (oracle/ref_convolve_driver.c): every
interpolation filter, all 16 x 16 phase pairs, every AV1 block size.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ME_SO = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")

# the 22 AV1 block sizes (width, height)
BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64), (64, 128),
               (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(ME_SO):
        pytest.skip("oracle/_ref not built")
    lib = C.CDLL(ME_SO, mode=os.RTLD_LAZY)
    lib.ref_av1_convolve_sr.restype = None
    lib.ref_av1_convolve_sr.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    return lib


def _orc(oracle):
    f = oracle.lib.orc_av1_convolve_sr
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    return f


@pytest.mark.parametrize("filters", [(0, 0), (1, 1), (2, 2), (3, 3), (0, 2), (1, 0)])
def test_every_phase_pair_matches_reference(oracle, ref, filters):
    fx, fy = filters
    orc = _orc(oracle)
    rng = np.random.default_rng(fx * 4 + fy)
    for (w, h) in ((64, 64), (4, 4), (8, 16)):
        S = w + 16
        src = rng.integers(0, 256, (h + 16, S), dtype=np.uint8)
        src[:4] = 255; src[4:8, ::2] = 0   # extremes: both clips
        for sx in range(16):
            for sy in range(16):
                a = np.full((h, w), 7, np.uint8); b = np.full((h, w), 9, np.uint8)
                p = src.ctypes.data + 8 * S + 8
                ref.ref_av1_convolve_sr(p, S, a.ctypes.data, w, w, h, fx, fy, sx, sy)
                orc(p, S, b.ctypes.data, w, w, h, fx, fy, sx, sy)
                assert np.array_equal(a, b), (w, h, fx, fy, sx, sy)


@pytest.mark.parametrize("size", BLOCK_SIZES)
def test_every_block_size_matches_reference(oracle, ref, size):
    w, h = size
    orc = _orc(oracle)
    rng = np.random.default_rng(w * 131 + h)
    S = w + 24
    for kind in range(3):
        if kind == 0:
            src = rng.integers(0, 256, (h + 16, S), dtype=np.uint8)
        elif kind == 1:
            yy, xx = np.mgrid[0:h + 16, 0:S]
            src = (((xx // 2 + yy // 3) & 1) * 255).astype(np.uint8)
        else:
            src = np.clip(128 + 90 * np.sin(np.arange(S) / 3.0)[None, :] + rng.normal(0, 8, (h + 16, S)), 0, 255).astype(np.uint8)
        for trial in range(12):
            fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
            if trial < 3:
                sx = 0 if trial != 1 else sx
                sy = 0 if trial != 2 else sy
            a = np.zeros((h, w + 5), np.uint8); b = np.zeros((h, w + 5), np.uint8)
            p = src.ctypes.data + 8 * S + 8
            ref.ref_av1_convolve_sr(p, S, a.ctypes.data, w + 5, w, h, fx, fy, sx, sy)
            orc(p, S, b.ctypes.data, w + 5, w, h, fx, fy, sx, sy)
            assert np.array_equal(a, b), (w, h, fx, fy, sx, sy)


@pytest.mark.parametrize("size", BLOCK_SIZES)
def test_compound_matches_reference(oracle, ref, size):
    """BI_PRED luma: both lists through the reference's av1_jnt_convolve_* exactly as av1_inter_prediction drives them, against the
    oracle's compound restatement: every combination of copy / x-only / y-only / 2-D between the two lists."""
    w, h = size
    ref.ref_av1_convolve_compound.restype = None
    ref.ref_av1_convolve_compound.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_int] * 6
    orc = oracle.lib.orc_av1_convolve_compound
    orc.restype = None
    orc.argtypes = ref.ref_av1_convolve_compound.argtypes
    rng = np.random.default_rng(w * 17 + h)
    S0, S1 = w + 24, w + 40
    for kind in range(2):
        if kind == 0:
            s0 = rng.integers(0, 256, (h + 16, S0), dtype=np.uint8); s1 = rng.integers(0, 256, (h + 16, S1), dtype=np.uint8)
        else:
            yy, xx = np.mgrid[0:h + 16, 0:S0]
            s0 = (((xx // 2 + yy // 3) & 1) * 255).astype(np.uint8)
            s1 = np.full((h + 16, S1), 255, np.uint8); s1[::3] = 0
        for trial in range(20):
            fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            ph = [int(v) for v in rng.integers(0, 16, 4)]
            if trial < 16:   # all 4 x 4 case combinations
                a, b = trial & 3, trial >> 2
                ph[0] = ph[0] or 5 if a & 1 else 0; ph[1] = (ph[1] or 9) if a & 2 else 0
                ph[2] = (ph[2] or 3) if b & 1 else 0; ph[3] = (ph[3] or 12) if b & 2 else 0
                ph[0] = (ph[0] or 5) if a & 1 else 0
            x = np.zeros((h, w + 3), np.uint8); y = np.zeros((h, w + 3), np.uint8)
            args = (s0.ctypes.data + 8 * S0 + 8, S0, s1.ctypes.data + 8 * S1 + 8, S1)
            ref.ref_av1_convolve_compound(*args, x.ctypes.data, w + 3, w, h, fx, fy, *ph)
            orc(*args, y.ctypes.data, w + 3, w, h, fx, fy, *ph)
            assert np.array_equal(x, y), (w, h, fx, fy, ph)


@pytest.mark.parametrize("size", BLOCK_SIZES)
def test_highbd_matches_reference(oracle, ref, size):
    """10-bit video in 16-bit planes: av1_highbd_convolve_*_sr_c and the av1_highbd_jnt_convolve_* pair (bd = 10) against the oracle:
    all four cases per list, random and extreme (0 / 1023) pictures."""
    w, h = size
    A = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_int] * 5
    B = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_int] * 7
    for f, at in ((ref.ref_av1_highbd_convolve_sr, A), (oracle.lib.orc_av1_highbd_convolve_sr, A), (ref.ref_av1_highbd_convolve_compound, B),
                  (oracle.lib.orc_av1_highbd_convolve_compound, B)):
        f.restype = None
        f.argtypes = at
    rng = np.random.default_rng(w * 19 + h)
    S0, S1 = w + 24, w + 40
    for kind in range(2):
        if kind == 0:
            s0 = rng.integers(0, 1024, (h + 16, S0), dtype=np.uint16); s1 = rng.integers(0, 1024, (h + 16, S1), dtype=np.uint16)
        else:
            yy, xx = np.mgrid[0:h + 16, 0:S0]
            s0 = (((xx // 2 + yy // 3) & 1) * 1023).astype(np.uint16)
            s1 = np.full((h + 16, S1), 1023, np.uint16); s1[::3] = 0
        p0, p1 = s0.ctypes.data + 2 * (8 * S0 + 8), s1.ctypes.data + 2 * (8 * S1 + 8)
        for trial in range(20):
            fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            ph = [int(v) for v in rng.integers(0, 16, 4)]
            if trial < 16:
                a, b = trial & 3, trial >> 2
                ph = [(ph[0] or 5) if a & 1 else 0, (ph[1] or 9) if a & 2 else 0, (ph[2] or 3) if b & 1 else 0, (ph[3] or 12) if b & 2 else 0]
            x = np.zeros((h, w + 3), np.uint16); y = np.zeros((h, w + 3), np.uint16)
            ref.ref_av1_highbd_convolve_sr(p0, S0, x.ctypes.data, w + 3, w, h, fx, fy, ph[0], ph[1], 10)
            oracle.lib.orc_av1_highbd_convolve_sr(p0, S0, y.ctypes.data, w + 3, w, h, fx, fy, ph[0], ph[1], 10)
            assert np.array_equal(x, y), ("sr", w, h, fx, fy, ph)
            ref.ref_av1_highbd_convolve_compound(p0, S0, p1, S1, x.ctypes.data, w + 3, w, h, fx, fy, *ph, 10)
            oracle.lib.orc_av1_highbd_convolve_compound(p0, S0, p1, S1, y.ctypes.data, w + 3, w, h, fx, fy, *ph, 10)
            assert np.array_equal(x, y), ("compound", w, h, fx, fy, ph)
