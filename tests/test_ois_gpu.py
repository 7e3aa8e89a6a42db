"""GPU parity: open-loop intra search (svthip_open_loop_intra_search_batch_dev) through the C ABI vs the oracle
(oracle/svt_ois_oracle.c, itself pinned against the reference's OpenLoopIntraSearchLcu in tests/test_ois_vs_ref.py).
Bit-exact on every candidate word and count, every branch of the search, partial SBs, several pictures per launch."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu

OP_ORDER = ["slice_is_intra", "temporal_layer_index", "is_used_as_reference_flag", "input_resolution_4k", "limit_ois_to_dc_mode_flag",
            "cu8x8_mode", "enc_mode"]


def params_of(**kw):
    p = svtav1_hip.OisParams()
    for k, v in kw.items():
        setattr(p, k, v)
    return p, np.array([kw.get(k, 0) for k in OP_ORDER], np.int32)


def luma_of(kind, w, h, seed):
    if kind == "synth":
        return synth.synth_luma(w, h, seed)
    if kind == "random":
        return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "extreme":
        return (np.random.default_rng(seed).integers(0, 2, (h, w)) * 255).astype(np.uint8)
    return np.full((h, w), 200, np.uint8)


def me_dist_for(oracle, plane, w, h, seed):
    tab = oracle.ois_sad_table(plane, 68, w, h)
    rng = np.random.default_rng(seed)
    scale = rng.choice([0.0, 0.1, 0.5, 0.9, 1.0, 1.5, 2.2, 3.5, 6.0], size=tab.shape[:2])
    d = (tab[:, :, 1].astype(np.float64) * scale).astype(np.uint32)
    d[rng.random(d.shape) < 0.03] = 0
    return d


def run_gpu(hip_ctx, torch, pics, params, me, me_stride=85):
    pool, descs = svtav1_hip.build_picture_pool(pics)
    w, h = pics[0].width, pics[0].height
    sb = svtav1_hip.sb_origins(w, h)
    n_sb, n = len(sb), len(pics)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    d_sb = torch.from_numpy(sb.view(np.int16)).to("cuda:0")
    d_me = None
    if me is not None:
        rows = np.zeros((n * n_sb, me_stride), svtav1_hip.ME_CU_RESULT_DTYPE)
        rows["distortion"][:, :85, 0] = me.reshape(n * n_sb, 85)
        rows["distortion"][:, :, 1] = 0xdeadbeef  # must not be read
        d_me = torch.from_numpy(rows.view(np.uint8).reshape(-1)).to("cuda:0")
    d_cand = torch.full((n * n_sb * 85 * 18,), -1, dtype=torch.int32, device="cuda:0")
    d_total = torch.full((n * n_sb * 85,), 0xEE, dtype=torch.uint8, device="cuda:0")
    hip_ctx.open_loop_intra_search_batch_dev(d_pool.data_ptr(), descs, params, d_sb.data_ptr(), n_sb,
                                             None if d_me is None else d_me.data_ptr(), me_stride, d_cand.data_ptr(), d_total.data_ptr())
    hip_ctx.synchronize()
    return (d_cand.cpu().numpy().view(np.uint32).reshape(n, n_sb, 85, 18), d_total.cpu().numpy().reshape(n, n_sb, 85))


CASES = [
    dict(slice_is_intra=1),
    dict(temporal_layer_index=0),
    dict(temporal_layer_index=0, input_resolution_4k=1, is_used_as_reference_flag=1),
    dict(temporal_layer_index=1, is_used_as_reference_flag=1),
    dict(temporal_layer_index=2, is_used_as_reference_flag=1),
    dict(temporal_layer_index=3),
    dict(temporal_layer_index=3, input_resolution_4k=1),
    dict(temporal_layer_index=3, limit_ois_to_dc_mode_flag=1),
    dict(temporal_layer_index=2, cu8x8_mode=1),
    dict(temporal_layer_index=2, enc_mode=3, input_resolution_4k=1),
    dict(temporal_layer_index=0, cu8x8_mode=1),
    dict(temporal_layer_index=5, is_used_as_reference_flag=0),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("pic", [("synth", 200, 136), ("random", 128, 64), ("extreme", 328, 200), ("flat", 72, 72)])
def test_ois_matches_oracle(hip_ctx, oracle, case, pic):
    torch = pytest.importorskip("torch")
    kind, w, h = pic
    lumas = [luma_of(kind, w, h, 11 * case + t) for t in range(3)]
    pics = [synth.PaPicture(x) for x in lumas]
    params, op = params_of(**CASES[case])
    me = np.stack([me_dist_for(oracle, p.full, w, h, case + 7 * t) for t, p in enumerate(pics)])
    got_c, got_t = run_gpu(hip_ctx, torch, pics, params, me)
    for t, p in enumerate(pics):
        want_c, want_t = oracle.ois_search_picture(p.full, 68, w, h, op, me[t])
        assert np.array_equal(got_t[t], want_t), (kind, CASES[case], t)
        bad = np.argwhere(got_c[t] != want_c)
        assert bad.size == 0, (kind, CASES[case], t, bad[:4], [hex(got_c[t][tuple(b)]) for b in bad[:4]],
                               [hex(want_c[tuple(b)]) for b in bad[:4]])


def test_ois_1080p_after_me_layout(hip_ctx, oracle):
    """Full-size pictures, ME rows in the 209-PU layout (only entries 1..84 of a row are read)."""
    torch = pytest.importorskip("torch")
    w, h = 1920, 1080
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in range(2)]
    params, op = params_of(temporal_layer_index=2, is_used_as_reference_flag=1)
    me = np.stack([me_dist_for(oracle, p.full, w, h, t) for t, p in enumerate(pics)])
    got_c, got_t = run_gpu(hip_ctx, torch, pics, params, me, me_stride=209)
    for t, p in enumerate(pics):
        want_c, want_t = oracle.ois_search_picture(p.full, 68, w, h, op, me[t])
        assert np.array_equal(got_t[t], want_t)
        assert np.array_equal(got_c[t], want_c)
    # the bottom SB row is 56 rows high: its last 8x8 row and everything larger that crosses the edge is absent
    n_sb_x = 30
    last = got_t[0].reshape(17, n_sb_x, 85)[16]
    assert (last[:, 21 + 56:] == 0).all() and (last[:, 3:5] == 0).all() and (last[:, 1:3] != 0).all()


def test_ois_more_pictures_than_one_job_table(hip_ctx, oracle):
    torch = pytest.importorskip("torch")
    w, h = 128, 72
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in range(svtav1_hip.HME_MAX_JOBS + 3)]
    params, op = params_of(slice_is_intra=1)
    got_c, got_t = run_gpu(hip_ctx, torch, pics, params, None)
    for t, p in enumerate(pics):
        want_c, want_t = oracle.ois_search_picture(p.full, 68, w, h, op, None)
        assert np.array_equal(got_t[t], want_t) and np.array_equal(got_c[t], want_c), t


def test_ois_sad_table_all_modes(hip_ctx, oracle):
    """Base-layer branch keeps the best 18 of 35: check them against an independent ranking of the oracle's full SAD table."""
    torch = pytest.importorskip("torch")
    w, h = 256, 128
    pics = [synth.PaPicture(synth.synth_luma(w, h, 5))]
    params, op = params_of(temporal_layer_index=0)
    got_c, got_t = run_gpu(hip_ctx, torch, pics, params, None)
    tab = oracle.ois_sad_table(pics[0].full, 68, w, h)
    dist = got_c[0] & 0xfffff
    assert (np.diff(dist[:, 1:, :].astype(np.int64), axis=2) >= 0).all()
    want_sorted = np.sort(tab[:, 1:, :], axis=2)[:, :, :18]
    assert np.array_equal(dist[:, 1:, :], want_sorted)
    assert (got_t[0][:, 1:] == 18).all()


def test_ois_needs_me_rows_on_general_branch(hip_ctx):
    torch = pytest.importorskip("torch")
    pics = [synth.PaPicture(synth.synth_luma(64, 64, 0))]
    params, _ = params_of(temporal_layer_index=2)
    with pytest.raises(svtav1_hip.SvtHipError):
        run_gpu(hip_ctx, torch, pics, params, None)
