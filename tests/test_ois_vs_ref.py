"""Pins the open-loop intra search oracle (oracle/svt_ois_oracle.c) against the reference's own functions compiled into
oracle/_ref/libsvtref_me.so (oracle/ref_ois_driver.c): UpdateNeighborSamplesArrayOpenLoop + IntraPredictionOpenLoop for every
mode, CU size and neighbour availability, and OpenLoopIntraSearchLcu over whole pictures through every branch.  CPU only."""
import numpy as np
import pytest

from svtav1_hip import synth

OP = dict(slice_i=0, temporal_layer=1, is_ref=2, res_4k=3, limit_dc=4, cu8x8_mode=5, enc_mode=6)


def make_op(**kw):
    op = np.zeros(7, np.int32)
    for k, v in kw.items():
        op[OP[k]] = v
    return op


@pytest.fixture(scope="module")
def refme():
    from oracle.binding import ReferenceME
    if not ReferenceME.available():
        pytest.skip("oracle/_ref/libsvtref_me.so not built")
    return ReferenceME()


def padded(kind, w, h, seed):
    if kind == "synth":
        luma = synth.synth_luma(w, h, seed)
    elif kind == "random":
        luma = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "extreme":
        luma = (np.random.default_rng(seed).integers(0, 2, (h, w)) * 255).astype(np.uint8)
    else:
        luma = np.full((h, w), 77, np.uint8)
    return np.ascontiguousarray(np.pad(luma, 68, mode="edge"))


@pytest.mark.parametrize("kind", ["synth", "random", "extreme"])
@pytest.mark.parametrize("size", [8, 16, 32])
def test_predictors_match_reference(oracle, refme, kind, size):
    """All 35 modes at CU positions covering: interior, picture left / top edges (neighbours 128), the right / bottom edges
    (top-right and bottom-left samples beyond the picture stay 128)."""
    w, h = 128, 96
    plane = padded(kind, w, h, size)
    pos = [(0, 0), (size, 0), (0, size), (size, size), (w - size, size), (w - 2 * size, h - size), (size, h - size),
           (w - size, h - size), (64, 32)]
    for cu_x, cu_y in pos:
        refs = oracle.ois_neighbours(plane, 68, w, h, cu_x, cu_y, size)
        for mode in range(35):
            rp, rr = refme.ois_predict(plane, 68, w, h, cu_x, cu_y, size, mode)
            assert np.array_equal(rr, refs), (cu_x, cu_y)
            op = oracle.ois_predict(refs, size, mode)
            assert np.array_equal(rp, op), (kind, size, cu_x, cu_y, mode)


def me_dist_for(oracle, plane, w, h, seed):
    """Synthetic ME distortions spread around the DC SADs so that every OIS point (very fast .. very complex) occurs."""
    tab = oracle.ois_sad_table(plane, 68, w, h)
    rng = np.random.default_rng(seed)
    scale = rng.choice([0.0, 0.1, 0.5, 0.9, 1.0, 1.5, 2.2, 3.5, 6.0], size=tab.shape[:2])
    d = (tab[:, :, 1].astype(np.float64) * scale).astype(np.uint32)
    d[rng.random(d.shape) < 0.03] = 0
    return d


CASES = [
    dict(slice_i=1),
    dict(temporal_layer=0),                                   # all 35 modes, sorted best 18
    dict(temporal_layer=0, res_4k=1, is_ref=1),               # 4K base layer takes the general path, heavy thresholds
    dict(temporal_layer=1, is_ref=1),
    dict(temporal_layer=2, is_ref=1),
    dict(temporal_layer=3, is_ref=0),
    dict(temporal_layer=3, is_ref=0, res_4k=1),               # default thresholds
    dict(temporal_layer=2, is_ref=1, res_4k=1),
    dict(temporal_layer=3, limit_dc=1),
    dict(temporal_layer=2, cu8x8_mode=1),
    dict(temporal_layer=2, enc_mode=3, res_4k=1),             # vertical winner's valid flag depends on enc_mode
    dict(temporal_layer=0, cu8x8_mode=1),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("pic", [("synth", 200, 136), ("random", 128, 64), ("flat", 72, 72)])
def test_search_matches_reference(oracle, refme, case, pic):
    kind, w, h = pic
    plane = padded(kind, w, h, 5 + case)
    op = make_op(**CASES[case])
    md = me_dist_for(oracle, plane, w, h, case)
    rc, rt = refme.ois_search_picture(plane, 68, w, h, op, md)
    oc, ot = oracle.ois_search_picture(plane, 68, w, h, op, md)
    assert np.array_equal(rt, ot)
    bad = np.argwhere(rc != oc)
    assert bad.size == 0, (bad[:5], [hex(rc[tuple(b)]) for b in bad[:5]], [hex(oc[tuple(b)]) for b in bad[:5]])
    if not CASES[case].get("slice_i") and not CASES[case].get("limit_dc") and CASES[case].get("temporal_layer"):
        assert len(np.unique(ot)) >= 4  # several OIS points occurred
