"""Pins the CPU oracle (oracle/svt_me_oracle.c) against the reference's own kernels compiled from
/root/reference (oracle/_ref).  CPU only."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth


def _pictures(w, h, kind):
    if kind == "synth":
        return synth.PaPicture(synth.synth_luma(w, h, 1)), synth.PaPicture(synth.synth_luma(w, h, 0))
    if kind == "flat":  # maximal ties: first position in raster order must win everywhere
        a = np.full((h, w), 77, np.uint8)
        return synth.PaPicture(a), synth.PaPicture(a.copy())
    rng = np.random.default_rng(1234)
    return (synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)),
            synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)))


@pytest.mark.parametrize("kind", ["synth", "flat", "random"])
@pytest.mark.parametrize("search", [(64, 64), (16, 16), (23, 9), (127, 127), (7, 5), (1, 1)])
def test_fullpel_oracle_equals_reference(oracle, reference, kind, search):
    w, h = 192, 136  # 3x3 SBs, last row partial (8 rows) -> exercises edge clipping
    cur, ref = _pictures(w, h, kind)
    rng = np.random.default_rng(7)
    nx, ny = cur.sb_grid()
    centers = rng.integers(-40, 41, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, *search)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s1, m1 = reference.fullpel_search_batch(cur.full, ref.full, desc, asm_type=0)
    assert np.array_equal(s0, s1)
    assert np.array_equal(m0, m1)


def test_flat_picture_first_position_wins(oracle):
    cur, ref = _pictures(128, 128, "flat")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    sad, mv = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    assert (sad == 0).all()
    for i in range(desc.shape[0]):
        xo, yo = int(desc[i, 2]), int(desc[i, 3])
        expect = ((yo * 4) & 0xFFFF) << 16 | ((xo * 4) & 0xFFFF)
        assert (mv[i] == expect).all()


def test_avx2_row_differs_only_in_32x32_mv(oracle, reference):
    """the reference's AVX2 8-position 32x32 kernel returns wrong MVs under GCC; the
    parity target is the ASM_NON_AVX2 row.  Recorded here so the choice of oracle is evidenced."""
    cur, ref = _pictures(192, 128, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s2, m2 = reference.fullpel_search_batch(cur.full, ref.full, desc, asm_type=1)
    assert np.array_equal(s0, s2)
    diff_pus = set(np.nonzero(m0 != m2)[1].tolist())
    assert diff_pus <= {1, 2, 3, 4}


@pytest.mark.parametrize("shape", [(16, 8, 48, 24), (32, 16, 16, 16), (64, 32, 8, 8), (16, 16, 33, 33), (8, 4, 5, 3)])
def test_sad_loop_kernel(oracle, reference, shape):
    bw, bh, sw, sh = shape
    rng = np.random.default_rng(bw * 1000 + sw)
    src = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, (160, 256), dtype=np.uint8)
    for trial in range(20):
        ro = int(rng.integers(0, 40)) * 256 + int(rng.integers(0, 60))
        a = oracle.sad_loop(src, 0, 64, ref, ro, 256, bh, bw, 256, sw, sh)
        b = reference.sad_loop("SadLoopKernel", src, 0, 64, ref, ro, 256, bh, bw, 256, sw, sh)
        assert a == b
        if sw >= 8:
            c = reference.sad_loop("SadLoopKernel_SSE4_1_INTRIN", src, 0, 64, ref, ro, 256, bh, bw, 256, sw, sh)
            assert a == c


# ---------------------------------------------------------------------------------------------------------------------
# 209-PU mode (row a9): squares + rectangular PUs, incl. the stale-variable update of 32x16[5]
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(192, 136, "synth", 64, 64), (192, 136, "random", 23, 9), (136, 128, "flat", 16, 16),
                                  (192, 136, "pan", 127, 40), (192, 136, "extreme", 33, 33)])
def test_fullpel_209pu_matches_reference(oracle, case):
    from oracle.binding import ReferenceME
    if not ReferenceME.available():
        pytest.skip("oracle/_ref/libsvtref_me.so not built")
    import svtav1_hip
    from svtav1_hip import synth
    w, h, kind, sw, sh = case
    rng = np.random.default_rng(sw * 100 + sh)
    if kind == "synth":
        f = [synth.synth_luma(w, h, 3), synth.synth_luma(w, h, 0)]
    elif kind == "pan":
        big = synth.synth_luma(w + 64, h + 64, 0); f = [big[20:20 + h, 30:30 + w], big[17:17 + h, 21:21 + w]]
    elif kind == "flat":
        f = [np.full((h, w), 77, np.uint8), np.full((h, w), 77, np.uint8)]        # all SADs zero: every update is a tie
    elif kind == "extreme":
        f = [rng.choice([0, 255], (h, w)).astype(np.uint8), rng.choice([0, 255], (h, w)).astype(np.uint8)]
    else:
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h, w), dtype=np.uint8)]
    cur, ref = synth.PaPicture(np.ascontiguousarray(f[0])), synth.PaPicture(np.ascontiguousarray(f[1]))
    nx, ny = cur.sb_grid()
    centers = rng.integers(-20, 21, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, sw, sh)
    s_o, m_o = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    s_r, m_r = ReferenceME().fullpel_search209_batch(cur.full, ref.full, desc)
    assert np.array_equal(s_o, s_r), np.argwhere(s_o != s_r)[:5]
    assert np.array_equal(m_o, m_r), np.argwhere(m_o != m_r)[:5]
    s85, m85 = oracle.fullpel_search_batch(cur.full, ref.full, desc)          # the squares agree with the 85-PU mode
    assert np.array_equal(s_o[:, :85], s85) and np.array_equal(m_o[:, :85], m85)
