"""GPU parity: HIP sub-pel refinement (half + quarter pel, 85 PUs) through the C ABI vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu


def _pictures(w, h, kind, seed=4):
    if kind == "synth":
        return synth.PaPicture(synth.synth_luma(w, h, 1)), synth.PaPicture(synth.synth_luma(w, h, 0))
    if kind == "smooth":
        yy, xx = np.mgrid[0:h + 16, 0:w + 16]
        base = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 40 * np.sin((xx + 2 * yy) / 23.0), 0, 255).astype(np.uint8)
        return synth.PaPicture(np.ascontiguousarray(base[5:5 + h, 7:7 + w])), synth.PaPicture(np.ascontiguousarray(base[8:8 + h, 8:8 + w]))
    if kind == "flat":
        a = np.full((h, w), 77, np.uint8)
        return synth.PaPicture(a), synth.PaPicture(a.copy())
    if kind == "extreme":  # 0/255 stripes: wrapped SSD differs from true SSD, filter clips on both sides
        yy, xx = np.mgrid[0:h, 0:w]
        a = (((xx // 3 + yy // 5) & 1) * 255).astype(np.uint8)
        b = (((xx // 4 + yy // 3) & 1) * 255).astype(np.uint8)
        return synth.PaPicture(a), synth.PaPicture(b)
    rng = np.random.default_rng(seed)
    return (synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)),
            synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)))


def _run_device(ctx, cur, ref, desc, sad0, mv0, disable_8x8=False):
    import torch
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(cur.full).to(dev)
    d_ref = torch.from_numpy(ref.full).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)
    d_sad = torch.from_numpy(sad0.view(np.int32).copy()).to(dev)
    d_mv = torch.from_numpy(mv0.view(np.int32).copy()).to(dev)
    torch.cuda.synchronize()
    ctx.subpel_refine_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), desc.shape[0],
                          int(desc[:, 4].max()), int(desc[:, 5].max()), d_sad.data_ptr(), d_mv.data_ptr(), disable_8x8)
    ctx.synchronize()
    return d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("kind", ["synth", "smooth", "flat", "extreme", "random"])
@pytest.mark.parametrize("search", [(64, 64), (16, 16), (23, 9), (127, 127)])
def test_subpel_matches_oracle(hip_ctx, oracle, kind, search):
    pytest.importorskip("torch")
    w, h = 192, 136
    cur, ref = _pictures(w, h, kind)
    rng = np.random.default_rng(13)
    nx, ny = cur.sb_grid()
    centers = rng.integers(-30, 31, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, *search)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s_o, m_o, _, _ = oracle.subpel_refine_batch(cur.full, ref.full, desc, s0, m0)
    s_h, m_h = _run_device(hip_ctx, cur, ref, desc, s0, m0)
    bad = np.argwhere((s_h != s_o) | (m_h != m_o))
    assert bad.size == 0, f"{len(bad)} mismatches, first (sb,pu)={bad[0]}: hip {s_h[tuple(bad[0])]}/{m_h[tuple(bad[0])]:#x} oracle {s_o[tuple(bad[0])]}/{m_o[tuple(bad[0])]:#x}"


def test_subpel_8x8_disabled(hip_ctx, oracle):
    pytest.importorskip("torch")
    cur, ref = _pictures(256, 128, "smooth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 32, 32)
    s0, m0 = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    s_o, m_o, _, _ = oracle.subpel_refine_batch(cur.full, ref.full, desc, s0, m0, disable_8x8=True)
    s_h, m_h = _run_device(hip_ctx, cur, ref, desc, s0, m0, disable_8x8=True)
    assert np.array_equal(s_h, s_o) and np.array_equal(m_h, m_o)
    assert np.array_equal(m_h[:, 21:], m0[:, 21:])


def test_subpel_1080p_after_fullpel(hip_ctx, oracle):
    """Full BASELINE size: device full-pel then device sub-pel on all 510 SBs; a sample against the oracle and the
    size-independent properties over everything (MVs within 3 quarter-pels of the full-pel MV, unchanged PUs keep
    their full-pel SAD)."""
    pytest.importorskip("torch")
    cur, ref = _pictures(1920, 1080, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    s0, m0 = hip_ctx.fullpel_search(cur.full, ref.full, desc)
    s_h, m_h = _run_device(hip_ctx, cur, ref, desc, s0, m0)
    sample = np.random.default_rng(2).choice(desc.shape[0], 24, replace=False)
    s_o, m_o, _, _ = oracle.subpel_refine_batch(cur.full, ref.full, desc[sample], s0[sample], m0[sample])
    assert np.array_equal(s_h[sample], s_o) and np.array_equal(m_h[sample], m_o)
    x0 = (m0 & 0xffff).astype(np.int16).astype(np.int32); y0 = (m0 >> 16).astype(np.int16).astype(np.int32)
    x1 = (m_h & 0xffff).astype(np.int16).astype(np.int32); y1 = (m_h >> 16).astype(np.int16).astype(np.int32)
    assert (np.abs(x1 - x0) <= 3).all() and (np.abs(y1 - y0) <= 3).all()
    same = m_h == m0
    assert np.array_equal(s_h[same], s0[same])


@pytest.mark.parametrize("kind,search", [("smooth", (64, 64)), ("extreme", (23, 9)), ("random", (127, 127)), ("synth", (16, 16))])
def test_subpel209_matches_oracle(hip_ctx, oracle, kind, search):
    """svthip_me_subpel_refine209_dev: the 85 squares and the 124 rectangular PUs of the all-partition mode."""
    torch = pytest.importorskip("torch")
    w, h = 192, 136
    cur, ref = _pictures(w, h, kind)
    rng = np.random.default_rng(21)
    nx, ny = cur.sb_grid()
    centers = rng.integers(-30, 31, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, *search)
    s0, m0 = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    s_o, m_o = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(cur.full).to(dev)
    d_ref = torch.from_numpy(ref.full).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)
    d_sad = torch.from_numpy(s0.view(np.int32).copy()).to(dev)
    d_mv = torch.from_numpy(m0.view(np.int32).copy()).to(dev)
    torch.cuda.synchronize()
    hip_ctx.subpel_refine209_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), desc.shape[0],
                                 int(desc[:, 4].max()), int(desc[:, 5].max()), d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    s_h, m_h = d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)
    bad = np.argwhere((s_h != s_o) | (m_h != m_o))
    assert bad.size == 0, f"{len(bad)} mismatches, first (sb,pu)={bad[0]}, PUs {sorted(set(bad[:, 1].tolist()))[:24]}"
    if kind == "smooth":
        assert (m_o[:, 85:] != m0[:, 85:]).any()  # the refinement moved some rectangular PUs


def test_subpel209_8_wide_rectangles_follow_the_8_row_ssd(hip_ctx, oracle):
    """The reference dispatches the half-pel SSD by width only and its width-8 leaf runs 8 rows (8x16 / 8x32 PUs are compared on
    their top 8 rows; tests/test_subpel_vs_ref.py pins that against the reference's own tables).  The device result must equal the
    oracle with that dispatch AND differ from the all-rows restatement in the 8-wide rectangular classes -- otherwise this test
    could not tell the two metrics apart."""
    torch = pytest.importorskip("torch")
    cur, ref = _pictures(256, 192, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    s0, m0 = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    s_o, m_o = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    oracle.lib.orc_set_halfpel_dispatch_exact(0)
    try:
        s_all, m_all = oracle.subpel_refine209_batch(cur.full, ref.full, desc, s0, m0)
    finally:
        oracle.lib.orc_set_halfpel_dispatch_exact(1)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(cur.full).to(dev); d_ref = torch.from_numpy(ref.full).to(dev); d_desc = torch.from_numpy(desc).to(dev)
    d_sad = torch.from_numpy(s0.view(np.int32).copy()).to(dev); d_mv = torch.from_numpy(m0.view(np.int32).copy()).to(dev)
    torch.cuda.synchronize()
    hip_ctx.subpel_refine209_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), desc.shape[0],
                                 64, 64, d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    s_h, m_h = d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)
    assert np.array_equal(s_h, s_o) and np.array_equal(m_h, m_o)
    g = oracle.pu_geometry209()
    narrow = np.zeros(209, bool)
    narrow[[int(g[pu, 4]) for pu in range(209) if g[pu, 0] == 8 and g[pu, 1] > 8]] = True
    assert narrow.sum() == 48
    differs = (m_h != m_all) | (s_h != s_all)
    assert differs[:, narrow].any() and not differs[:, ~narrow].any()


# ------------------------------------------------------------------------------------------------------------------------------
# The reference's three fractional search methods through svthip_me_subpel_search_dev.  Methods 0 / 1 (SUB_SAD / FULL_SAD) run the same
# control flow as SSD_SEARCH with SAD leaves, and for them the expected values are pinned against the reference EXECUTING its own
# HalfPelSearch_LCU + QuarterPelSearch_LCU (tests/test_subpel_vs_ref.py, tests/golden/subpel_search.npz).
# ------------------------------------------------------------------------------------------------------------------------------
def _run_search(ctx, cur, ref, desc, sad0, mv0, method, all_pu, disable_8x8=False):
    import torch
    dev = torch.device("cuda:0")
    d_src, d_ref, d_desc = torch.from_numpy(cur.full).to(dev), torch.from_numpy(ref.full).to(dev), torch.from_numpy(desc).to(dev)
    d_sad = torch.from_numpy(sad0.view(np.int32).copy()).to(dev)
    d_mv = torch.from_numpy(mv0.view(np.int32).copy()).to(dev)
    torch.cuda.synchronize()
    ctx.subpel_search_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), desc.shape[0],
                          int(desc[:, 4].max()), int(desc[:, 5].max()), d_sad.data_ptr(), d_mv.data_ptr(), method, all_pu, disable_8x8)
    ctx.synchronize()
    return d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("all_pu", [False, True])
@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("kind,search", [("smooth", (64, 64)), ("synth", (40, 24)), ("extreme", (23, 9)), ("flat", (16, 16)), ("random", (64, 64))])
def test_subpel_search_methods_match_oracle(hip_ctx, oracle, kind, search, method, all_pu):
    pytest.importorskip("torch")
    cur, ref = _pictures(192, 136, kind)
    rng = np.random.default_rng(29)
    nx, ny = cur.sb_grid()
    desc = svtav1_hip.make_fullpel_desc(cur, ref, rng.integers(-20, 21, size=(nx * ny, 2)), *search)
    s0, m0 = (oracle.fullpel_search209_batch if all_pu else oracle.fullpel_search_batch)(cur.full, ref.full, desc)
    s_o, m_o, _ = oracle.subpel_refine_method(cur.full, ref.full, desc, s0, m0, method, all_pu)
    s_h, m_h = _run_search(hip_ctx, cur, ref, desc, s0, m0, method, all_pu)
    bad = np.argwhere((s_h != s_o) | (m_h != m_o))
    assert bad.size == 0, f"{len(bad)} mismatches, first (sb,pu)={bad[0]}: hip {s_h[tuple(bad[0])]}/{m_h[tuple(bad[0])]:#x} oracle {s_o[tuple(bad[0])]}/{m_o[tuple(bad[0])]:#x}"


@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("method", [0, 1])
def test_subpel_search_matches_reference_execution(hip_ctx, n_pu, method):
    """HIP against what the REFERENCE ITSELF computed (tests/golden/subpel_search.npz, generated by running HalfPelSearch_LCU +
    QuarterPelSearch_LCU of /root/reference with the SAD search methods): no oracle in between."""
    import os
    pytest.importorskip("torch")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "subpel_search.npz"))
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    s_h, m_h = _run_search(hip_ctx, cur, ref, g["desc"], g[f"sad0_{n_pu}"], g[f"mv0_{n_pu}"], method, n_pu == 209)
    assert np.array_equal(s_h, g[f"sad_{n_pu}_m{method}"]) and np.array_equal(m_h, g[f"mv_{n_pu}_m{method}"])


def test_subpel_search_rejects_bad_methods(hip_ctx):
    import torch
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
    out = torch.zeros(1 << 12, dtype=torch.int32, device="cuda:0")
    p = buf.data_ptr()
    with pytest.raises(svtav1_hip.SvtHipError, match="fractional_search_method"):
        hip_ctx.subpel_search_dev(p, 256, p, 256, p, 1, 64, 64, out.data_ptr(), out.data_ptr(), 3, False)
    with pytest.raises(svtav1_hip.SvtHipError, match="fractional_search_method"):
        hip_ctx.subpel_search_dev(p, 256, p, 256, p, 1, 64, 64, out.data_ptr(), out.data_ptr(), -1, True)


@pytest.mark.parametrize("method", [0, 1])
def test_subpel_search_sad_methods_1080p(hip_ctx, oracle, method):
    """Full BASELINE size, all 209 PUs of the 510 SBs under the SAD search methods: a sample against the (reference-pinned) oracle, and over
    everything the properties that hold whatever the content: vectors within 3 quarter-pels of the full-pel vector, unchanged PUs keep
    their full-pel SAD, and the stored distortion never grows (a candidate only replaces a strictly larger best)."""
    pytest.importorskip("torch")
    cur, ref = _pictures(1920, 1080, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    sample = np.random.default_rng(6).choice(desc.shape[0], 12, replace=False)
    s0 = np.zeros((desc.shape[0], 209), np.uint32); m0 = np.zeros_like(s0)
    # full-pel results of the whole picture from the device (pinned elsewhere), 209-PU mode
    import torch
    dev = torch.device("cuda:0")
    d_src, d_ref, d_desc = torch.from_numpy(cur.full).to(dev), torch.from_numpy(ref.full).to(dev), torch.from_numpy(desc).to(dev)
    d_sad = torch.zeros((desc.shape[0], 209), dtype=torch.int32, device=dev); d_mv = torch.zeros_like(d_sad)
    hip_ctx.fullpel_search209_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), desc.shape[0], 64, 64,
                                  d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    s0, m0 = d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)
    s_h, m_h = _run_search(hip_ctx, cur, ref, desc, s0, m0, method, True)
    s_o, m_o, _ = oracle.subpel_refine_method(cur.full, ref.full, desc[sample], s0[sample], m0[sample], method, True)
    assert np.array_equal(s_h[sample], s_o) and np.array_equal(m_h[sample], m_o)
    x0 = (m0 & 0xffff).astype(np.int16).astype(np.int32); y0 = (m0 >> 16).astype(np.int16).astype(np.int32)
    x1 = (m_h & 0xffff).astype(np.int16).astype(np.int32); y1 = (m_h >> 16).astype(np.int16).astype(np.int32)
    assert (np.abs(x1 - x0) <= 3).all() and (np.abs(y1 - y0) <= 3).all()
    same = m_h == m0
    assert np.array_equal(s_h[same], s0[same])
    assert (s_h <= s0).all()
    assert (~same).any()
