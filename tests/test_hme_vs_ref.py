"""Pins the oracle's hierarchical-ME chain (centre check -> HME L0/L1/L2 -> region pick -> zero-centre check
-> window clipping -> full-pel 85-PU search) against the reference's own MotionEstimateLcu, run standalone from
oracle/_ref/libsvtref_me.so with sub-pel refinement disabled (the reference's sub-pel path needs a NASM-only
symbol, see oracle/ref_me_lcu_driver.c).  CPU only."""
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


def _pics(w, h, kind):
    if kind == "synth":
        f = [synth.synth_luma(w, h, t) for t in (3, 0, 7)]
    elif kind == "pan":  # large global motion so the HME centres move far from (0,0)
        big = synth.synth_luma(w + 128, h + 96, 0)
        f = [big[40:40 + h, 50:50 + w], big[30:30 + h, 14:14 + w], big[70:70 + h, 100:100 + w]]
    elif kind == "flat":
        f = [np.full((h, w), 90, np.uint8)] * 3
    else:
        rng = np.random.default_rng(99)
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    return [synth.PaPicture(np.ascontiguousarray(x)) for x in f]


def oracle_chain(oracle, pics, P, two_lists):
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
    out = {}
    state = np.zeros((sb.shape[0], 25), np.int16)
    d0, c0 = oracle.hme_search_center_batch(pool, descs[0], descs[1], P, 0, sb, None, state)
    s0, m0 = oracle.fullpel_search_batch(pool, pool, d0, descs[0].full_stride, descs[1].full_stride)
    out[0] = (d0, s0, m0)
    if two_lists:
        d1, c1 = oracle.hme_search_center_batch(pool, descs[0], descs[2], P, 1, sb, m0[:, 0], state)
        s1, m1 = oracle.fullpel_search_batch(pool, pool, d1, descs[0].full_stride, descs[2].full_stride)
        out[1] = (d1, s1, m1)
    return out


CASES = [
    # (w, h, kind, hierarchical_levels, temporal_layer, two_lists, is_ref, poc_equal)
    (448, 320, "synth", 3, 0, False, True, False),
    (448, 320, "pan", 3, 0, False, True, False),
    (448, 320, "pan", 3, 1, True, True, False),
    (448, 320, "pan", 4, 0, True, True, False),     # multiplier 350 -> large L0 areas, clipped at the borders
    (448, 320, "pan", 5, 0, True, False, False),    # multiplier 525, non-reference picture (no zero-centre check)
    (448, 320, "pan", 3, 3, True, True, True),      # multiplier 70, same-POC list 1 takes the 2nd best region
    (448, 320, "random", 3, 2, True, True, False),
    (448, 320, "flat", 3, 1, True, True, False),
    (456, 328, "pan", 3, 1, True, True, False),     # partial right column (8 px) and bottom row (8 rows)
    (856, 480, "synth", 3, 0, False, True, False),  # BASELINE config-1 picture size
]


@pytest.mark.parametrize("case", CASES)
def test_chain_matches_reference_motion_estimate_lcu(oracle, refme, case):
    w, h, kind, hier, tl, two, is_ref, poc_eq = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl, is_ref, poc_eq)
    ref = refme.run(pics[0], pics[1], pics[2], P, two_lists=two, hierarchical_levels=hier)
    mine = oracle_chain(oracle, pics, P, two)
    for l in mine:
        d, s, m = mine[l]
        assert np.array_equal(d[:, 2:4], ref["origin"][:, l]), f"list {l}: search-area origins differ"
        assert np.array_equal(s, ref["sad"][:, l]), f"list {l}: SADs differ"
        assert np.array_equal(m, ref["mv"][:, l]), f"list {l}: MVs differ"


@pytest.mark.parametrize("case", [c for c in CASES if c[2] != "flat"])
def test_bipred_and_packing_match_reference(oracle, refme, case):
    """me_results of the reference (bi-prediction SAD at integer MVs + Sort3Elements ordering) vs the oracle's
    orc_bipred_pack_* fed with the oracle chain's per-list results."""
    w, h, kind, hier, tl, two, is_ref, poc_eq = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl, is_ref, poc_eq)
    ref = refme.run(pics[0], pics[1], pics[2], P, two_lists=two, hierarchical_levels=hier)
    mine = oracle_chain(oracle, pics, P, two)
    pool, descs = svtav1_hip.build_picture_pool(pics)
    fs = descs[0].full_stride
    if two:
        res = oracle.bipred_pack_batch(pool, fs, pool, fs, mine[0][0], mine[0][1], mine[0][2], pool, fs, mine[1][0], mine[1][1], mine[1][2])
    else:
        res = oracle.bipred_pack_batch(pool, fs, pool, fs, mine[0][0], mine[0][1], mine[0][2])
    r = ref["res"]  # [n,85,9] = xMvL0,yMvL0,xMvL1,yMvL1, dist0,dir0, dist1,dir1, total
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    assert np.array_equal(res["xMvL0"], r[:, :, 0]) and np.array_equal(res["yMvL0"], r[:, :, 1])
    assert np.array_equal(res["distortion"][:, :, 0], r[:, :, 4].astype(np.uint32)) and np.array_equal(res["direction"][:, :, 0], r[:, :, 5])
    if two:
        assert np.array_equal(res["xMvL1"], r[:, :, 2]) and np.array_equal(res["yMvL1"], r[:, :, 3])
        assert np.array_equal(res["distortion"][:, :, 1], r[:, :, 6].astype(np.uint32)) and np.array_equal(res["direction"][:, :, 1], r[:, :, 7])


def test_hme_disabled_levels(oracle, refme):
    pics = _pics(448, 320, "pan")
    for flags in [(1, 0, 0), (1, 1, 0), (0, 1, 1), (0, 0, 0)]:
        P = svtav1_hip.default_me_params(448, 320, 3, 1)
        P.enable_hme_level0_flag, P.enable_hme_level1_flag, P.enable_hme_level2_flag = flags
        if flags == (0, 0, 0):
            P.enable_hme_flag = 0
        ref = refme.run(pics[0], pics[1], pics[2], P, two_lists=True)
        mine = oracle_chain(oracle, pics, P, True)
        for l in mine:
            d, s, m = mine[l]
            assert np.array_equal(d[:, 2:4], ref["origin"][:, l]), (flags, l)
            assert np.array_equal(s, ref["sad"][:, l]) and np.array_equal(m, ref["mv"][:, l]), (flags, l)


# ---------------------------------------------------------------------------------------------------------------------
# 209-PU (all-partition) mode: the reference's MotionEstimateLcu with pic_depth_mode = PIC_ALL_DEPTH_MODE
# ---------------------------------------------------------------------------------------------------------------------
def oracle_chain209(oracle, pics, P, two_lists):
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
    out = {}
    state = np.zeros((sb.shape[0], 25), np.int16)
    d0, _ = oracle.hme_search_center_batch(pool, descs[0], descs[1], P, 0, sb, None, state)
    s0, m0 = oracle.fullpel_search209_batch(pool, pool, d0, descs[0].full_stride, descs[1].full_stride)
    out[0] = (d0, s0, m0)
    if two_lists:
        d1, _ = oracle.hme_search_center_batch(pool, descs[0], descs[2], P, 1, sb, m0[:, 0], state)
        s1, m1 = oracle.fullpel_search209_batch(pool, pool, d1, descs[0].full_stride, descs[2].full_stride)
        out[1] = (d1, s1, m1)
    return out


CASES209 = [CASES[0], CASES[2], CASES[5], CASES[6], CASES[8]]


@pytest.mark.parametrize("case", CASES209)
def test_209pu_mode_matches_reference_motion_estimate_lcu(oracle, refme, case):
    """HME + open_loop_me_fullpel_search_sblock (the reference's static 209-PU caller, :1556) + bi-prediction at integer MVs
    over all 209 PUs + packing through tab32x16 .. tab8x32, against the oracle chain."""
    w, h, kind, hier, tl, two, is_ref, poc_eq = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl, is_ref, poc_eq)
    ref = refme.run(pics[0], pics[1], pics[2], P, two_lists=two, hierarchical_levels=hier, all_pu=True)
    mine = oracle_chain209(oracle, pics, P, two)
    for l in mine:
        d, s, m = mine[l]
        assert np.array_equal(d[:, 2:4], ref["origin"][:, l]), f"list {l}: search-area origins differ"
        assert np.array_equal(s, ref["sad"][:, l]), f"list {l}: SADs differ"
        assert np.array_equal(m, ref["mv"][:, l]), f"list {l}: MVs differ"
    pool, descs = svtav1_hip.build_picture_pool(pics)
    fs = descs[0].full_stride
    if two:
        res = oracle.bipred_pack_batch(pool, fs, pool, fs, mine[0][0], mine[0][1], mine[0][2], pool, fs, mine[1][0], mine[1][1], mine[1][2],
                                       n_pu=209)
    else:
        res = oracle.bipred_pack_batch(pool, fs, pool, fs, mine[0][0], mine[0][1], mine[0][2], n_pu=209)
    r = ref["res"]  # [n,209,11] = xMvL0,yMvL0,xMvL1,yMvL1, dist0,dir0, dist1,dir1, total, dist2,dir2
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    assert np.array_equal(res["xMvL0"], r[:, :, 0]) and np.array_equal(res["yMvL0"], r[:, :, 1])
    ncand = 3 if two else 1
    for k, (di, dr) in enumerate(((4, 5), (6, 7), (9, 10))[:ncand]):
        assert np.array_equal(res["distortion"][:, :, k], r[:, :, di].astype(np.uint32)), f"distortion[{k}]"
        assert np.array_equal(res["direction"][:, :, k], r[:, :, dr]), f"direction[{k}]"
    if two:
        assert np.array_equal(res["xMvL1"], r[:, :, 2]) and np.array_equal(res["yMvL1"], r[:, :, 3])


def test_pu_geometry209_consistent_with_fullpel(oracle):
    """The derived raster -> ME-buffer map and PU rectangles: at a single search position the 209-PU full-pel SAD of buffer entry
    n must be the (row-subsampled) SAD over exactly the rectangle the geometry table gives for it."""
    g = oracle.pu_geometry209()
    assert sorted(g[:, 4].tolist()) == list(range(209))
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    desc = np.array([[0, 0, 0, 0, 1, 1]], np.int32)
    sad, _ = oracle.fullpel_search209_batch(src, ref, desc)
    ad = np.abs(src.astype(np.int32) - ref.astype(np.int32))
    for pu in range(209):
        w, h, px, py, n = (int(v) for v in g[pu])
        blk = ad[py:py + h, px:px + w]
        # 8x8 SADs are computed on the even rows and doubled (ExtSadCalculation_8x8_16x16); everything else is a sum of them
        want = 2 * int(blk[0::2].sum())
        assert int(sad[0, n]) == want, (pu, n, w, h, px, py)
