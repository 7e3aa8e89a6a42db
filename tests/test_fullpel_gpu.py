"""GPU parity: the HIP full-pel 85-PU search (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu


def _pictures(w, h, kind, seed=1234):
    if kind == "synth":
        return synth.PaPicture(synth.synth_luma(w, h, 1)), synth.PaPicture(synth.synth_luma(w, h, 0))
    if kind == "flat":
        a = np.full((h, w), 77, np.uint8)
        return synth.PaPicture(a), synth.PaPicture(a.copy())
    if kind == "coarse":  # few grey levels -> many exact SAD ties at non-trivial positions
        rng = np.random.default_rng(seed)
        return (synth.PaPicture((rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)),
                synth.PaPicture((rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)))
    rng = np.random.default_rng(seed)
    return (synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)),
            synth.PaPicture(rng.integers(0, 256, (h, w), dtype=np.uint8)))


def _compare(hip_ctx, oracle, cur, ref, desc):
    s_h, m_h = hip_ctx.fullpel_search(cur.full, ref.full, desc)
    s_o, m_o = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    bad = np.argwhere((s_h != s_o) | (m_h != m_o))
    assert bad.size == 0, f"{len(bad)} mismatches, first (sb,pu)={bad[0]}: hip sad/mv {s_h[tuple(bad[0])]}/{m_h[tuple(bad[0])]:#x} oracle {s_o[tuple(bad[0])]}/{m_o[tuple(bad[0])]:#x}"


@pytest.mark.parametrize("kind", ["synth", "flat", "coarse", "random"])
@pytest.mark.parametrize("search", [(64, 64), (16, 16), (23, 9), (127, 127), (7, 5), (1, 1), (100, 33)])
def test_fullpel_matches_oracle(hip_ctx, oracle, kind, search):
    w, h = 192, 136
    cur, ref = _pictures(w, h, kind)
    rng = np.random.default_rng(11)
    nx, ny = cur.sb_grid()
    centers = rng.integers(-40, 41, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, *search)
    _compare(hip_ctx, oracle, cur, ref, desc)


def test_fullpel_854x480_all_sbs(hip_ctx, oracle):
    """BASELINE config-1 picture size (856x480 internally: partial right column and bottom row)."""
    cur, ref = _pictures(856, 480, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    _compare(hip_ctx, oracle, cur, ref, desc)


def test_fullpel_unaligned_reference_offsets(hip_ctx, oracle):
    """Every byte alignment of the search-window origin (the LDS staging re-aligns with v_alignbyte)."""
    cur, ref = _pictures(256, 128, "random", seed=5)
    desc = svtav1_hip.make_fullpel_desc(cur, ref, [(dx, 0) for dx in range(-3, 5)], 33, 17)
    _compare(hip_ctx, oracle, cur, ref, desc)


def test_fullpel_1080p_properties(hip_ctx, oracle):
    """Full BASELINE size: a sample of SBs against the oracle, plus size-independent properties over
    all 510 SBs: identical pictures give SAD 0 at MV (0,0) or an earlier tie; larger PUs equal sums
    is not required (independent minima) but every PU's SAD must be <= the SAD at the 64x64's MV."""
    cur, ref = _pictures(1920, 1080, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    s_h, m_h = hip_ctx.fullpel_search(cur.full, ref.full, desc)
    sample = np.random.default_rng(3).choice(desc.shape[0], 48, replace=False)
    s_o, m_o = oracle.fullpel_search_batch(cur.full, ref.full, desc[sample])
    assert np.array_equal(s_h[sample], s_o) and np.array_equal(m_h[sample], m_o)
    # property: sum of the four 32x32 minima <= the 64x64 minimum, 16x16 likewise (minimum of a sum)
    assert (s_h[:, 1:5].sum(1) <= s_h[:, 0]).all()
    assert (s_h[:, 5:21].sum(1) <= s_h[:, 1:5].sum(1)).all()
    assert (s_h[:, 21:85].sum(1) <= s_h[:, 5:21].sum(1)).all()
    # identical pictures: zero SAD everywhere
    s_z, m_z = hip_ctx.fullpel_search(cur.full, cur.full, desc)
    assert (s_z == 0).all()


def test_device_pointer_entry(hip_ctx, oracle):
    torch = pytest.importorskip("torch")
    cur, ref = _pictures(256, 192, "synth")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(cur.full).to(dev)
    d_ref = torch.from_numpy(ref.full).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)
    n = desc.shape[0]
    d_sad = torch.zeros((n, 85), dtype=torch.int32, device=dev)
    d_mv = torch.zeros((n, 85), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    hip_ctx.fullpel_search_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), n, 64, 64,
                               d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    s_o, m_o = oracle.fullpel_search_batch(cur.full, ref.full, desc)
    assert np.array_equal(d_sad.cpu().numpy().view(np.uint32), s_o)
    assert np.array_equal(d_mv.cpu().numpy().view(np.uint32), m_o)


def test_bad_arguments_fail_loudly(hip_ctx):
    cur, ref = _pictures(128, 128, "flat")
    desc = svtav1_hip.make_fullpel_desc(cur, ref, None, 64, 64)
    bad = desc.copy()
    bad[0, 4] = 200  # search width > 127
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fullpel_search(cur.full, ref.full, bad)
    bad = desc.copy()
    bad[0, 1] = cur.full.size  # window outside the plane
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fullpel_search(cur.full, ref.full, bad)


# ---------------------------------------------------------------------------------------------------------------------
# 209-PU mode (row a9)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(192, 136, "synth", 64, 64), (192, 136, "random", 23, 9), (136, 128, "flat", 16, 16),
                                  (192, 136, "pan", 127, 40), (192, 136, "extreme", 33, 33), (328, 200, "synth", 64, 64),
                                  (192, 136, "synth", 127, 127), (192, 136, "pan", 1, 1),
                                  # few grey levels: exact SAD ties at many positions, through every reduction group of the kernel,
                                  # at areas that leave lanes / positions of the last pass empty
                                  (192, 136, "coarse", 64, 64), (192, 136, "coarse", 100, 33), (192, 136, "coarse", 7, 5),
                                  (136, 128, "flat", 127, 3), (192, 136, "coarse", 17, 127)])
def test_fullpel_209pu_matches_oracle(hip_ctx, oracle, case):
    """Squares and rectangles (incl. PU 92's stale-variable recurrence) vs the reference-pinned oracle, bit-exact."""
    import torch
    from svtav1_hip import synth
    w, h, kind, sw, sh = case
    rng = np.random.default_rng(sw * 100 + sh)
    if kind == "synth":
        f = [synth.synth_luma(w, h, 3), synth.synth_luma(w, h, 0)]
    elif kind == "pan":
        big = synth.synth_luma(w + 64, h + 64, 0); f = [big[20:20 + h, 30:30 + w], big[17:17 + h, 21:21 + w]]
    elif kind == "flat":
        f = [np.full((h, w), 77, np.uint8), np.full((h, w), 77, np.uint8)]
    elif kind == "extreme":
        f = [rng.choice([0, 255], (h, w)).astype(np.uint8), rng.choice([0, 255], (h, w)).astype(np.uint8)]
    elif kind == "coarse":
        f = [(rng.integers(0, 3, (h, w)) * 100).astype(np.uint8), (rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)]
    else:
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h, w), dtype=np.uint8)]
    cur, ref = synth.PaPicture(np.ascontiguousarray(f[0])), synth.PaPicture(np.ascontiguousarray(f[1]))
    nx, ny = cur.sb_grid()
    centers = rng.integers(-20, 21, size=(nx * ny, 2))
    desc = svtav1_hip.make_fullpel_desc(cur, ref, centers, sw, sh)
    s_o, m_o = oracle.fullpel_search209_batch(cur.full, ref.full, desc)
    dev = torch.device("cuda:0")
    d_src, d_ref, d_desc = torch.from_numpy(cur.full).to(dev), torch.from_numpy(ref.full).to(dev), torch.from_numpy(desc).to(dev)
    n = desc.shape[0]
    d_sad = torch.full((n, 209), -1, dtype=torch.int32, device=dev); d_mv = torch.full((n, 209), -1, dtype=torch.int32, device=dev)
    hip_ctx.fullpel_search209_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), n, int(desc[:, 4].max()),
                                  int(desc[:, 5].max()), d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    s_h, m_h = d_sad.cpu().numpy().view(np.uint32), d_mv.cpu().numpy().view(np.uint32)
    bad = np.argwhere((s_h != s_o) | (m_h != m_o))
    assert bad.size == 0, f"{len(bad)} mismatches, first (sb,pu)={bad[0]}: hip {s_h[tuple(bad[0])]}/{m_h[tuple(bad[0])]:#x} oracle {s_o[tuple(bad[0])]}/{m_o[tuple(bad[0])]:#x}; PUs {sorted(set(bad[:, 1]))[:20]}"


def test_me_entries_reject_bad_arguments(hip_ctx):
    """Error behaviour of the C ABI: bad parameters come back as SVTHIP_ERR_BAD_PARAMETER with a message, nothing is launched."""
    import torch
    dev = torch.device("cuda:0")
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    out = torch.zeros(1 << 12, dtype=torch.int32, device=dev)
    p = buf.data_ptr()
    with pytest.raises(svtav1_hip.SvtHipError, match="search area"):
        hip_ctx.fullpel_search_dev(p, 256, p, 256, p, 1, 128, 64, out.data_ptr(), out.data_ptr())          # area > 127
    with pytest.raises(svtav1_hip.SvtHipError, match="search area"):
        hip_ctx.fullpel_search209_dev(p, 256, p, 256, p, 1, 64, 0, out.data_ptr(), out.data_ptr())         # empty area
    with pytest.raises(svtav1_hip.SvtHipError, match="multiples of 4"):
        hip_ctx.fullpel_search_dev(p, 250, p, 256, p, 1, 64, 64, out.data_ptr(), out.data_ptr())          # stride not 4-aligned
    with pytest.raises(svtav1_hip.SvtHipError, match="null"):
        hip_ctx.fullpel_search209_dev(p, 256, p, 256, None, 1, 64, 64, out.data_ptr(), out.data_ptr())
    # n_sb == 0 is a no-op, not an error
    hip_ctx.fullpel_search_dev(p, 256, p, 256, p, 0, 64, 64, out.data_ptr(), out.data_ptr())
    hip_ctx.fullpel_search209_dev(p, 256, p, 256, p, 0, 64, 64, out.data_ptr(), out.data_ptr())
    pics = [synth.PaPicture(synth.synth_luma(128, 64, t)) for t in range(2)]
    pool, pd = svtav1_hip.build_picture_pool(pics)
    d_pool = torch.from_numpy(pool).to(dev)
    sb = torch.from_numpy(svtav1_hip.sb_origins(128, 64).view(np.int16).copy()).to(dev)
    P = svtav1_hip.default_me_params(128, 64, 3, 0)
    P.number_hme_search_region_in_width = 3
    with pytest.raises(svtav1_hip.SvtHipError, match="regions"):
        hip_ctx.hme_search_center_dev(d_pool.data_ptr(), pd[1], pd[0], P, 0, sb.data_ptr(), 2, None, out.data_ptr())
    P = svtav1_hip.default_me_params(128, 64, 3, 0)
    with pytest.raises(svtav1_hip.SvtHipError, match="list_index"):
        hip_ctx.hme_search_center_dev(d_pool.data_ptr(), pd[1], pd[0], P, 2, sb.data_ptr(), 2, None, out.data_ptr())
    # the 209-PU sub-pel / bi-pred / whole-chain entries validate like their 85-PU counterparts
    with pytest.raises(svtav1_hip.SvtHipError, match="search area"):
        hip_ctx.subpel_refine209_dev(p, 256, p, 256, p, 1, 128, 64, out.data_ptr(), out.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError, match="null"):
        hip_ctx.subpel_refine209_dev(p, 256, p, 256, p, 1, 64, 64, None, out.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError, match="n_lists"):
        hip_ctx.bipred_pack209_dev(p, 256, p, 256, p, p, 256, p, 1, 64, 64, out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), 3,
                                   out.data_ptr())
    hip_ctx.subpel_refine209_dev(p, 256, p, 256, p, 0, 64, 64, out.data_ptr(), out.data_ptr())
    other = svtav1_hip.PaPictureDesc.from_buffer_copy(pd[1])
    other.full_stride += 4  # pictures of one batch must share their strides
    with pytest.raises(svtav1_hip.SvtHipError, match="strides"):
        hip_ctx.motion_estimate209_batch_dev(d_pool.data_ptr(), [pd[1], other], [pd[0], pd[0]], None, P, sb.data_ptr(), 2, out.data_ptr())
