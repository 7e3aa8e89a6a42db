"""Pins the FRACTIONAL half of the bi-prediction path (row a13) against the reference's own BiPredictionSearch ->
BiPredictionCompensation -> BiPredAverging -> SelectBuffer / QuarterPelCompensation (oracle/ref_bipred_driver.c in
oracle/_ref/libsvtref_me.so): for arbitrary quarter-pel vector pairs of every PU, on half-pel planes made by the reference's own
InterpolateSearchRegionAVC.  None of these functions needs a NASM-only symbol, so this part of the sub-pel stage IS pinned although the
refinement that produces the vectors is not.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth


@pytest.fixture(scope="module")
def refme():
    from oracle.binding import ReferenceME
    if not ReferenceME.available():
        pytest.skip("oracle/_ref/libsvtref_me.so not built")
    r = ReferenceME()
    r.lib.ref_bipred_search.restype = C.c_int
    r.lib.ref_bipred_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p]
    return r


def _vectors(rng, desc, n_pu, mode):
    """Quarter-pel vector words [n_pu] whose integer part lies in the search area and whose fraction is anything in -3 .. 3."""
    xo, yo, sw, sh = (int(v) for v in desc[2:6])
    ix = rng.integers(0, sw, n_pu)
    iy = rng.integers(0, sh, n_pu)
    if mode == "all_fracs":
        fx = rng.integers(-3, 4, n_pu); fy = rng.integers(-3, 4, n_pu)
    elif mode == "integer":
        fx = np.zeros(n_pu, np.int64); fy = np.zeros(n_pu, np.int64)
    else:  # half-pel positions only (SelectBuffer without averaging)
        fx = rng.choice([-2, 0, 2], n_pu); fy = rng.choice([-2, 0, 2], n_pu)
    x = 4 * (xo + ix) + fx
    y = 4 * (yo + iy) + fy
    return ((y.astype(np.int64) & 0xffff) << 16 | (x.astype(np.int64) & 0xffff)).astype(np.uint32)


@pytest.mark.parametrize("asm_type", [0, 1])
@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("mode", ["all_fracs", "half", "integer"])
@pytest.mark.parametrize("kind", ["synth", "random"])
def test_bipred_fractional_vectors_match_reference(oracle, refme, kind, mode, n_pu, asm_type):
    w, h = 256, 192
    if kind == "synth":
        lumas = [synth.synth_luma(w, h, t) for t in (3, 0, 7)]
    else:
        rng0 = np.random.default_rng(9)
        lumas = [rng0.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    cur, r0, r1 = (synth.PaPicture(x) for x in lumas)
    rng = np.random.default_rng(n_pu * 7 + asm_type)
    n_sb = ((w + 63) // 64) * ((h + 63) // 64)
    d0 = svtav1_hip.make_fullpel_desc(cur, r0, rng.integers(-20, 21, (n_sb, 2)), 64, 64)
    d1 = svtav1_hip.make_fullpel_desc(cur, r1, rng.integers(-20, 21, (n_sb, 2)), 40, 23)
    S = cur.full.shape[1]
    mv0 = np.stack([_vectors(rng, d0[i], n_pu, mode) for i in range(n_sb)])
    mv1 = np.stack([_vectors(rng, d1[i], n_pu, mode) for i in range(n_sb)])
    big = np.full((n_sb, n_pu), 0x00ffffff, np.uint32)   # uni-prediction candidates that never win: entry 0 of every PU is the bi-prediction
    res = oracle.bipred_pack_batch(cur.full, S, r0.full, S, d0, big, mv0, r1.full, S, d1, big, mv1, bipred_8x8=True, n_pu=n_pu)
    assert (res["direction"][:, :, 0] == 2).all()
    for i in range(n_sb):
        geo = np.array([d0[i][2], d0[i][3], d0[i][4], d0[i][5], d1[i][2], d1[i][3], d1[i][4], d1[i][5]], np.int32)
        out = np.zeros(n_pu, np.uint32)
        src_off = int(d0[i][0])
        # the reference plane pointers at the SB's co-located sample: desc ref_off points at the search-area origin
        ref0_00 = int(d0[i][1]) - (int(d0[i][3]) * S + int(d0[i][2]))
        ref1_00 = int(d1[i][1]) - (int(d1[i][3]) * S + int(d1[i][2]))
        m0, m1 = np.ascontiguousarray(mv0[i]), np.ascontiguousarray(mv1[i])
        rc = refme.lib.ref_bipred_search(cur.full.ctypes.data + src_off, S, r0.full.ctypes.data + ref0_00, S, r1.full.ctypes.data + ref1_00, S,
                                         geo.ctypes.data, m0.ctypes.data, m1.ctypes.data, n_pu, asm_type, out.ctypes.data)
        assert rc == 0
        bad = np.nonzero(out != res["distortion"][i, :, 0])[0]
        assert bad.size == 0, (kind, mode, n_pu, i, bad[:5], out[bad[:5]], res["distortion"][i, bad[:5], 0])
