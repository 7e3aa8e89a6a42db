"""Pins the leaf arithmetic and plane geometry of the sub-pel oracle (oracle/svt_subpel_oracle.c) against the
reference's own code in oracle/_ref: InterpolateSearchRegionAVC (b/h/j planes), the wrapped-SSD kernels and the
averaging SAD.  The refinement control flow itself cannot be executed from the reference here (it calls a
NASM-only symbol) and is covered by hand-computed invariants instead.  CPU only."""
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
    return ReferenceME()


@pytest.mark.parametrize("kind", ["synth", "random", "extreme"])
@pytest.mark.parametrize("search", [(64, 64), (16, 9), (37, 21)])
def test_half_pel_planes_match_reference_interpolation(oracle, refme, kind, search):
    sw, sh = search
    if kind == "synth":
        img = synth.synth_luma(320, 256, 1)
    elif kind == "random":
        img = np.random.default_rng(3).integers(0, 256, (256, 320), dtype=np.uint8)
    else:  # checkerboard of 0/255: exercises the clip on both sides
        yy, xx = np.mgrid[0:256, 0:320]
        img = (((xx // 3 + yy // 2) & 1) * 255).astype(np.uint8)
    plane = synth.pad_plane(img, synth.PAD_FULL)
    stride = plane.shape[1]
    off = (synth.PAD_FULL + 70) * stride + synth.PAD_FULL + 90
    rows, cols = sh + 63 + 1, sw + 63 + 1  # every sample the refinement can touch
    rb, rh, rj = refme.interp_region(plane, off, sw, sh, rows, cols)
    ob, oh, oj = oracle.interp_planes(plane, off, 0, 0, cols, rows)
    assert np.array_equal(ob, rb), "b plane"
    assert np.array_equal(oh, rh), "h plane"
    assert np.array_equal(oj, rj), "j plane"


def test_wrapped_ssd_matches_reference_kernels(oracle, reference):
    rng = np.random.default_rng(8)
    L = reference.lib
    for name, w, h in (("SpatialFullDistortionKernel8x8_SSSE3_INTRIN", 8, 8), ("SpatialFullDistortionKernel16MxN_SSSE3_INTRIN", 16, 16),
                       ("SpatialFullDistortionKernel16MxN_SSSE3_INTRIN", 32, 32), ("SpatialFullDistortionKernel16MxN_SSSE3_INTRIN", 64, 64)):
        f = getattr(L, name)
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        for trial in range(10):
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
            b = rng.integers(0, 256, (h, w), dtype=np.uint8)
            if trial == 0:
                a[:] = 255; b[:] = 0      # |a-b| = 255 wraps to 1
            if trial == 1:
                a[:] = 200; b[:] = 72     # difference 128 stays 128
            ref = f(a.ctypes.data, w, b.ctypes.data, w, w, h)
            assert oracle.ssd_wrapped(a, b) == ref


def test_averaging_sad_matches_reference(oracle, reference):
    """(p1 + p2 + 1) >> 1 then SAD: the quarter-pel candidate metric's SAD side (CombinedAveragingSAD)."""
    rng = np.random.default_rng(9)
    f = reference.lib.CombinedAveragingSAD
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    for w in (8, 16, 32):
        s = rng.integers(0, 256, (w, w), dtype=np.uint8)
        a = rng.integers(0, 256, (w, w), dtype=np.uint8)
        b = rng.integers(0, 256, (w, w), dtype=np.uint8)
        ref = f(s.ctypes.data, w, a.ctypes.data, w, b.ctypes.data, w, w, w)
        avg = ((a.astype(np.int32) + b + 1) >> 1)
        assert int(np.abs(s.astype(np.int32) - avg).sum()) == ref


def _setup(kind="synth"):
    w, h = 256, 192
    if kind == "synth":
        cur = synth.PaPicture(synth.synth_luma(w, h, 1)); ref = synth.PaPicture(synth.synth_luma(w, h, 0))
    else:
        rng = np.random.default_rng(21)
        base = rng.integers(0, 256, (h + 8, w + 8), dtype=np.uint8)
        cur = synth.PaPicture(np.ascontiguousarray(base[4:4 + h, 4:4 + w])); ref = synth.PaPicture(np.ascontiguousarray(base[3:3 + h, 2:2 + w]))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    return cur, ref, desc


def test_subpel_invariants(oracle):
    """Properties the refinement must satisfy whatever the control flow: MVs move by at most 3 quarter-pels per
    axis from the full-pel MV; a PU whose MV did not move keeps its full-pel SAD; the reported SSD never exceeds
    the SSD at the full-pel position; identical pictures stay at their zero-distortion full-pel MV."""
    cur, ref, desc = _setup()
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s1, m1, ssd, dr = oracle.subpel_refine_batch(cur.full, ref.full, desc, s0, m0)
    x0 = (m0 & 0xffff).astype(np.int16).astype(np.int32); y0 = (m0 >> 16).astype(np.int16).astype(np.int32)
    x1 = (m1 & 0xffff).astype(np.int16).astype(np.int32); y1 = (m1 >> 16).astype(np.int16).astype(np.int32)
    assert (np.abs(x1 - x0) <= 3).all() and (np.abs(y1 - y0) <= 3).all()
    same = (m1 == m0)
    assert np.array_equal(s1[same], s0[same])
    assert (dr <= 7).all()
    sz, mz = oracle.fullpel_search_batch(cur.full, cur.full, desc)
    s2, m2, ssd2, _ = oracle.subpel_refine_batch(cur.full, cur.full, desc, sz, mz)
    assert np.array_equal(m2, mz) and (s2 == 0).all() and (ssd2 == 0).all()


def test_subpel_finds_true_half_pel_shift(oracle):
    """Reference picture = current picture shifted by exactly half a pixel horizontally (built with the same 4-tap
    filter): the refined 64x64 MV must land on the half-pel position and its SSD must drop to ~0."""
    w, h = 256, 192
    yy, xx = np.mgrid[0:h + 16, 0:w + 16]
    base = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 40 * np.sin((xx + 2 * yy) / 23.0), 0, 255).astype(np.int32)
    # cur(x) = half-pel sample of base between x+7 and x+8
    a, b, c, d = base[8:8 + h, 6:6 + w], base[8:8 + h, 7:7 + w], base[8:8 + h, 8:8 + w], base[8:8 + h, 9:9 + w]
    cur_img = np.clip((-2 * a + 18 * b + 18 * c - 2 * d + 16) >> 5, 0, 255).astype(np.uint8)
    ref_img = base[8:8 + h, 8:8 + w].astype(np.uint8)
    cur = synth.PaPicture(np.ascontiguousarray(cur_img)); ref = synth.PaPicture(np.ascontiguousarray(ref_img))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 16, 16)
    inner = [i for i in range(desc.shape[0]) if desc[i, 4] == 16 and desc[i, 5] == 16 and desc[i, 2] == -8 and desc[i, 3] == -8]
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s1, m1, ssd, _ = oracle.subpel_refine_batch(cur.full, ref.full, desc, s0, m0)
    nx = w // 64
    inner = [i for i in inner if 0 < i % nx < nx - 1]  # SBs whose filter taps stay inside the picture
    assert inner
    for i in inner:
        for pu in (0, 1, 2, 3, 4):
            fx = int(np.int16(m0[i, pu] & 0xffff)); fy = int(np.int16(m0[i, pu] >> 16))
            if fy != 0 or fx not in (0, -4):
                continue  # the true position is not among the 8 half-pel neighbours of this PU's full-pel MV
            x = int(np.int16(m1[i, pu] & 0xffff)); y = int(np.int16(m1[i, pu] >> 16))
            assert (x, y) == (-2, 0), f"SB {i} PU {pu}: MV ({x},{y}) instead of the half-pel position (-2,0)"
            assert ssd[i, pu] == 0


# every (width, height) pair HalfPelSearch_LCU / QuarterPelSearch_LCU pass to the PU refinement functions
# (Codec/EbMotionEstimation.c:2278-2786, :3369-4114): the 4 square and 10 rectangular shape classes of the 209-PU mode
PU_SHAPES = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 16), (16, 8), (32, 64), (16, 32), (8, 16), (32, 8), (8, 32),
             (64, 16), (16, 64)]


def _leaf_fn(refme, name, n_ptr):
    f = getattr(refme.lib, name)
    f.restype = C.c_uint32
    return f


@pytest.mark.parametrize("asm_type", [0, 1])
@pytest.mark.parametrize("shape", PU_SHAPES)
def test_half_pel_leaf_dispatch_per_shape(oracle, refme, shape, asm_type):
    """The half-pel stage's two metrics as the reference DISPATCHES them for a (w,h) PU: the SSD leaf is picked by width
    only -- SpatialFullDistortionKernel_funcPtrArray[asm][Log2f(pu_width) - 2] -- and the width-8 leaf always runs 8 rows,
    so 8x16 / 8x32 PUs are compared on their top 8 rows; the stored SAD (NxMSadKernel_funcPtrArray[asm][pu_width >> 3])
    covers every row.  The table index is computed here as floor(log2(w)) - 2 (what Log2f means; the NASM Log2f_SSE2
    itself cannot be assembled in this image and no stand-in is linked)."""
    w, h = shape
    ssd_ref = refme.lib.ref_halfpel_ssd_leaf; ssd_ref.restype = C.c_uint32
    sad_ref = refme.lib.ref_halfpel_sad_leaf; sad_ref.restype = C.c_uint32
    ssd_orc = oracle.lib.orc_halfpel_ssd_dispatched; ssd_orc.restype = C.c_uint32
    sad_orc = oracle.lib.orc_halfpel_sad_dispatched; sad_orc.restype = C.c_uint32
    for f in (ssd_ref, sad_ref):
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    for f in (ssd_orc, sad_orc):
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ssd_index = int(np.log2(w)) - 2
    rng = np.random.default_rng(100 * w + h)
    ss, rs = 200, 1350  # source stride of a padded plane / interpolated_stride
    for trial in range(6):
        src = rng.integers(0, 256, (h, ss), dtype=np.uint8)
        rec = rng.integers(0, 256, (h, rs), dtype=np.uint8)
        if trial == 0:
            src[:] = 255; rec[:] = 0
        if trial == 1:
            src[:] = 200; rec[:] = 72
        want_ssd = ssd_ref(asm_type, ssd_index, src.ctypes.data, ss, rec.ctypes.data, rs, w, h)
        want_sad = sad_ref(asm_type, w >> 3, src.ctypes.data, ss, rec.ctypes.data, rs, w, h)
        assert ssd_orc(src.ctypes.data, ss, rec.ctypes.data, rs, w, h) == want_ssd, f"SSD as dispatched for {w}x{h}"
        assert sad_orc(src.ctypes.data, ss, rec.ctypes.data, rs, w, h) == want_sad, f"SAD as dispatched for {w}x{h}"
        if w == 8 and h > 8 and trial >= 2:
            full = oracle.ssd_wrapped(np.ascontiguousarray(src[:, :w]), np.ascontiguousarray(rec[:, :w]))
            assert full != want_ssd, "the 8-row leaf and the full-height sum must differ on random data"


@pytest.mark.parametrize("asm_type", [0, 1])
@pytest.mark.parametrize("shape", PU_SHAPES)
def test_quarter_pel_leaf_dispatch_per_shape(oracle, refme, shape, asm_type):
    """Quarter-pel stage: CombinedAveragingSSD (true SSD) for the comparison and NxMSadAveragingKernel_funcPtrArray[asm][w >> 3]
    for the stored SAD, both over every row of the PU, source read at stride 64 (sb_buffer)."""
    w, h = shape
    ssd_ref = refme.lib.ref_quarterpel_ssd_leaf; ssd_ref.restype = C.c_uint32
    ssd_ref.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    sad_ref = refme.lib.ref_quarterpel_sad_leaf; sad_ref.restype = C.c_uint32
    sad_ref.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ssd_orc = oracle.lib.orc_quarterpel_ssd_dispatched; ssd_orc.restype = C.c_uint32
    sad_orc = oracle.lib.orc_quarterpel_sad_dispatched; sad_orc.restype = C.c_uint32
    for f in (ssd_orc, sad_orc):
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(7 * w + h)
    for trial in range(4):
        src = rng.integers(0, 256, (h, 64), dtype=np.uint8)
        r1 = rng.integers(0, 256, (h, 1350), dtype=np.uint8)
        r2 = rng.integers(0, 256, (h, 200), dtype=np.uint8)
        a = (src.ctypes.data, 64, r1.ctypes.data, 1350, r2.ctypes.data, 200, w, h)
        assert ssd_orc(*a) == ssd_ref(*a)
        assert sad_orc(*a) == sad_ref(asm_type, w >> 3, *a)


def test_8_wide_rectangles_use_the_8_row_metric(oracle):
    """The 209-PU refinement with the reference's dispatch differs from the all-rows restatement (round 1) in the 8x16 and
    8x32 classes and ONLY there."""
    cur, ref, desc = _setup("synth")  # textured pictures with noise: no exact full-pel match, sub-pel candidates do win
    s0, m0 = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    s_ref, m_ref = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    oracle.lib.orc_set_halfpel_dispatch_exact(0)
    try:
        s_all, m_all = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    finally:
        oracle.lib.orc_set_halfpel_dispatch_exact(1)
    g = oracle.pu_geometry209()
    narrow = np.zeros(209, bool)
    for pu in range(209):
        if g[pu, 0] == 8 and g[pu, 1] > 8:
            narrow[g[pu, 4]] = True
    diff = (m_ref != m_all) | (s_ref != s_all)
    assert diff[:, narrow].any(), "the 8-row metric must change some 8x16 / 8x32 result on random pictures"
    assert not diff[:, ~narrow].any(), "every other shape class is independent of the fix"


# ------------------------------------------------------------------------------------------------------------------------------
# The refinement's CONTROL FLOW against the reference executing it.  MeContext_t::fractionalSearchMethod selects the distortion at run
# time; with SUB_SAD_SEARCH (0) / FULL_SAD_SEARCH (1) the reference's HalfPelSearch_LCU + QuarterPelSearch_LCU run here (Log2f_SSE2 is
# only evaluated by the SSD_SEARCH branch of the conditional operator), statement for statement the code the SSD method runs too.
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def refsubpel():
    from oracle.binding import ReferenceSubpel
    if not ReferenceSubpel.available():
        pytest.skip("oracle/_ref/libsvtref_subpel.so not built")
    return ReferenceSubpel()


def _search_case(kind, w=256, h=192, seed=5):
    rng = np.random.default_rng(seed)
    if kind == "synth":
        f = [synth.synth_luma(w, h, 2), synth.synth_luma(w, h, 0)]
    elif kind == "pan":  # a sub-pel pan: the refinement really moves
        big = synth.synth_luma(2 * w + 64, 2 * h + 64, 0).astype(np.int32)
        f = [((big[10:10 + 2 * h:2, 12:12 + 2 * w:2] + big[11:11 + 2 * h:2, 12:12 + 2 * w:2] + 1) >> 1).astype(np.uint8),
             big[7:7 + 2 * h:2, 9:9 + 2 * w:2].astype(np.uint8)]
    elif kind == "coarse":  # few grey levels: exact ties between candidates, everywhere
        f = [(rng.integers(0, 3, (h, w)) * 100).astype(np.uint8), (rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)]
    elif kind == "flat":
        f = [np.full((h, w), 90, np.uint8), np.full((h, w), 90, np.uint8)]
    else:
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h, w), dtype=np.uint8)]
    return synth.PaPicture(np.ascontiguousarray(f[0])), synth.PaPicture(np.ascontiguousarray(f[1]))


@pytest.mark.parametrize("all_pu", [False, True])
@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("kind", ["synth", "pan", "coarse", "flat", "random"])
def test_refinement_control_flow_matches_reference_execution(oracle, refsubpel, kind, method, all_pu):
    cur, ref = _search_case(kind)
    rng = np.random.default_rng(17)
    nx, ny = cur.sb_grid()
    centers = rng.integers(-12, 13, size=(nx * ny, 2))
    sw, sh = (24, 16) if kind in ("random", "coarse") else (40, 32)
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, sw, sh)
    if all_pu:
        s0, m0 = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    else:
        s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    for asm_type in (0, 1):
        s_r, m_r, d_r = refsubpel.subpel_search(cur.full, ref.full, desc, s0, m0, method, all_pu, asm_type=asm_type)
        s_o, m_o, d_o = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, method, all_pu)
        bad = np.argwhere((s_r != s_o) | (m_r != m_o) | (d_r != d_o))
        assert bad.size == 0, (f"asm {asm_type}: {len(bad)} mismatches, first (sb, ME index) {bad[0]}: reference sad/mv/dir "
                               f"{s_r[tuple(bad[0])]}/{m_r[tuple(bad[0])]:#x}/{d_r[tuple(bad[0])]} oracle {s_o[tuple(bad[0])]}/{m_o[tuple(bad[0])]:#x}/{d_o[tuple(bad[0])]}")
    if kind in ("synth", "pan"):
        assert (m_o != m0).mean() > 0.05, "the case must actually move vectors to sub-pel positions"


def test_refinement_with_8x8_disabled_matches_reference_execution(oracle, refsubpel):
    """cu8x8_mode == CU_8x8_MODE_1: the 8x8 PUs keep their full-pel result (:2375, :3470)."""
    cur, ref = _search_case("pan")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 32, 32)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s_r, m_r, d_r = refsubpel.subpel_search(cur.full, ref.full, desc, s0, m0, 0, False, disable_8x8=True)
    s_o, m_o, d_o = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, 0, False, disable_8x8=True)
    assert np.array_equal(s_r, s_o) and np.array_equal(m_r, m_o)
    assert np.array_equal(m_o[:, 21:], m0[:, 21:])


def test_ssd_method_runs_the_same_functions(oracle):
    """The SSD method (what MotionEstimateLcu hard-wires, :6254) goes through the very functions pinned above -- pu_half_pel /
    pu_quarter_pel of oracle/svt_subpel_oracle.c take the method as an argument and differ only in the distortion lines, whose leaves and
    per-shape dispatch are pinned separately (test_half_pel_leaf_dispatch_per_shape, test_quarter_pel_leaf_dispatch_per_shape).  Here:
    method 2 through the general entry equals the 85- / 209-PU entries the GPU tests and golden fixtures use."""
    cur, ref = _search_case("pan")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 32, 32)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s_a, m_a, _, d_a = oracle.subpel_refine_batch(cur.full, ref.full, desc, s0, m0)
    s_b, m_b, d_b = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, 2, False)
    assert np.array_equal(s_a, s_b) and np.array_equal(m_a, m_b) and np.array_equal(d_a, d_b)
    s0, m0 = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    s_a, m_a = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    s_b, m_b, _ = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, 2, True)
    assert np.array_equal(s_a, s_b) and np.array_equal(m_a, m_b)
    # and the three methods really are different searches on this content
    s_c, m_c, _ = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, 0, True)
    assert (m_c != m_b).any() or (s_c != s_b).any()


@pytest.mark.parametrize("all_pu", [False, True])
def test_refinement_control_flow_480p_all_sbs(oracle, refsubpel, all_pu):
    """BASELINE config-1 picture size (856x480: partial right column and bottom row), the default 64x64 area, every superblock: the
    reference's own refinement (SUB_SAD_SEARCH) against the oracle on the full-pel results of the whole picture."""
    cur, ref = synth.PaPicture(synth.synth_luma(856, 480, 1)), synth.PaPicture(synth.synth_luma(856, 480, 0))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    s0, m0 = (oracle.fullpel_search209_batch if all_pu else oracle.fullpel_search_batch)(cur.full, ref.full, desc)
    s_r, m_r, d_r = refsubpel.subpel_search(cur.full, ref.full, desc, s0, m0, 0, all_pu)
    s_o, m_o, d_o = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, 0, all_pu)
    assert np.array_equal(s_r, s_o) and np.array_equal(m_r, m_o) and np.array_equal(d_r, d_o)
