"""GPU parity of the whole-picture ME entry (svthip_motion_estimate_picture_dev): HME -> full-pel -> sub-pel per list,
bi-prediction, packed me_results, vs the oracle chain.  Bit-exact."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth
from me_chain_util import compare_results, device_me_picture, oracle_me_picture

pytestmark = pytest.mark.gpu


def _pics(w, h, kind):
    if kind == "synth":
        f = [synth.synth_luma(w, h, t) for t in (3, 0, 7)]
    elif kind == "pan":
        big = synth.synth_luma(w + 128, h + 96, 0)
        f = [big[40:40 + h, 50:50 + w], big[30:30 + h, 14:14 + w], big[70:70 + h, 100:100 + w]]
    elif kind == "smooth":
        yy, xx = np.mgrid[0:h + 64, 0:w + 64]
        big = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 40 * np.sin((xx + 2 * yy) / 23.0), 0, 255).astype(np.uint8)
        f = [big[20:20 + h, 20:20 + w], big[17:17 + h, 25:25 + w], big[26:26 + h, 12:12 + w]]
    else:
        rng = np.random.default_rng(5)
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    return [synth.PaPicture(np.ascontiguousarray(x)) for x in f]


CASES = [
    # w, h, kind, hier, tl, two_lists, use_subpel, cu8x8_mode
    (320, 192, "pan", 3, 0, False, True, 0),
    (320, 192, "pan", 3, 1, True, True, 0),
    (320, 192, "smooth", 3, 1, True, True, 0),
    (320, 192, "smooth", 4, 2, True, True, 1),
    (320, 192, "random", 3, 2, True, True, 0),
    (328, 200, "pan", 3, 1, True, True, 0),       # partial right column / bottom row
    (320, 192, "pan", 3, 1, True, False, 0),      # sub-pel off (the configuration the reference binary can run)
]


@pytest.mark.parametrize("case", CASES)
def test_motion_estimate_picture_matches_oracle(hip_ctx, oracle, case):
    pytest.importorskip("torch")
    w, h, kind, hier, tl, two, subpel, cu8 = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl)
    res_o, per = oracle_me_picture(oracle, pics, P, two, subpel, cu8)
    res_d, ls, lm = device_me_picture(hip_ctx, pics, P, two, subpel, cu8)
    for l in per:
        assert np.array_equal(ls[l], per[l][1]), f"list {l} SADs"
        assert np.array_equal(lm[l], per[l][2]), f"list {l} MVs"
    compare_results(res_d, res_o)


CASES209 = [
    # w, h, kind, hier, tl, two_lists, use_subpel, cu8x8_mode
    (320, 192, "pan", 3, 0, False, True, 0),
    (320, 192, "pan", 3, 1, True, True, 0),
    (320, 192, "smooth", 3, 1, True, True, 1),    # cu8x8_mode 1: 8x8 PUs keep full-pel MVs but still get a bi-pred candidate
    (320, 192, "random", 3, 2, True, True, 0),
    (328, 200, "smooth", 4, 1, True, True, 0),    # partial right column / bottom row
    (320, 192, "pan", 3, 1, True, False, 0),      # sub-pel off: the configuration pinned against the reference binary
]


@pytest.mark.parametrize("case", CASES209)
def test_motion_estimate209_matches_oracle(hip_ctx, oracle, case):
    """The 209-PU (all-partition) mode through svthip_motion_estimate209_batch_dev: HME -> 209-PU full-pel -> sub-pel of
    squares and rectangles -> bi-prediction and packing of all 209 PUs."""
    pytest.importorskip("torch")
    w, h, kind, hier, tl, two, subpel, cu8 = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl)
    res_o, per = oracle_me_picture(oracle, pics, P, two, subpel, cu8, n_pu=209)
    res_d, ls, lm = device_me_picture(hip_ctx, pics, P, two, subpel, cu8, n_pu=209)
    for l in per:
        bad = np.argwhere((ls[l] != per[l][1]) | (lm[l] != per[l][2]))
        assert bad.size == 0, f"list {l}: {len(bad)} (sb, pu) mismatches, first {bad[0]}, PUs {sorted(set(bad[:, 1].tolist()))[:20]}"
    compare_results(res_d, res_o)


def test_motion_estimate209_squares_equal_85pu_mode(hip_ctx):
    """The first 85 ME-buffer entries of the 209-PU mode are the 85-PU mode's (same search, same refinement)."""
    pytest.importorskip("torch")
    pics = _pics(320, 192, "smooth")
    P = svtav1_hip.default_me_params(320, 192, 3, 1)
    _, ls85, lm85 = device_me_picture(hip_ctx, pics, P, True, True, 0)
    _, ls209, lm209 = device_me_picture(hip_ctx, pics, P, True, True, 0, n_pu=209)
    assert np.array_equal(ls209[:, :, :85], ls85) and np.array_equal(lm209[:, :, :85], lm85)


def test_motion_estimate_picture_1080p_b_picture(hip_ctx, oracle):
    """Full-size B picture through the whole chain; a sample of SBs against the oracle chain."""
    pytest.importorskip("torch")
    pics = _pics(1920, 1080, "synth")
    P = svtav1_hip.default_me_params(1920, 1080, 3, 1)
    res_d, ls, lm = device_me_picture(hip_ctx, pics, P, True, True, 0)
    assert (res_d["totalMeCandidateIndex"] == 3).all()
    # sorted candidates
    d = res_d["distortion"]
    assert (d[:, :, 0] <= d[:, :, 1]).all() and (d[:, :, 1] <= d[:, :, 2]).all()
    sample = np.sort(np.random.default_rng(17).choice(510, 16, replace=False))
    res_o, per = oracle_me_picture(oracle, pics, P, True, True, 0, sb_subset=sample)
    for l in per:
        assert np.array_equal(ls[l][sample], per[l][1]) and np.array_equal(lm[l][sample], per[l][2])
    compare_results(res_d[sample], res_o)


def test_motion_estimate209_1080p_b_picture(hip_ctx, oracle):
    """Full-size B picture through the 209-PU chain; size-independent properties over all 510 SBs and a sample against the oracle."""
    pytest.importorskip("torch")
    pics = _pics(1920, 1080, "synth")
    P = svtav1_hip.default_me_params(1920, 1080, 3, 1)
    res_d, ls, lm = device_me_picture(hip_ctx, pics, P, True, True, 0, n_pu=209)
    assert (res_d["totalMeCandidateIndex"] == 3).all()
    d = res_d["distortion"]
    assert (d[:, :, 0] <= d[:, :, 1]).all() and (d[:, :, 1] <= d[:, :, 2]).all()
    assert (np.sort(res_d["direction"], axis=2) == np.arange(3)).all()  # each of L0 / L1 / bi-pred exactly once
    sample = np.sort(np.random.default_rng(23).choice(510, 8, replace=False))
    res_o, per = oracle_me_picture(oracle, pics, P, True, True, 0, sb_subset=sample, n_pu=209)
    for l in per:
        assert np.array_equal(ls[l][sample], per[l][1]) and np.array_equal(lm[l][sample], per[l][2])
    compare_results(res_d[sample], res_o)


def test_batched_pictures_equal_single_picture_calls(hip_ctx):
    """svthip_motion_estimate_batch_dev over several B pictures == one svthip_motion_estimate_picture_dev per picture."""
    torch = pytest.importorskip("torch")
    w, h, n_pic = 320, 192, 3
    frames = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in range(n_pic + 2)]
    pool, pd = svtav1_hip.build_picture_pool(frames)
    dev = torch.device("cuda:0")
    d_pool = torch.from_numpy(pool).to(dev)
    sb = svtav1_hip.sb_origins(w, h)
    n = sb.shape[0]
    d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    curs = [pd[i + 1] for i in range(n_pic)]; r0 = [pd[i] for i in range(n_pic)]; r1 = [pd[i + 2] for i in range(n_pic)]
    d_one = torch.zeros((n_pic * n, 85, 24), dtype=torch.uint8, device=dev)
    for i in range(n_pic):
        hip_ctx.motion_estimate_picture_dev(d_pool.data_ptr(), curs[i], r0[i], r1[i], P, d_sb.data_ptr(), n,
                                            d_one.data_ptr() + i * n * 85 * 24, True, 0)
    d_bat = torch.full((n_pic * n, 85, 24), 0x77, dtype=torch.uint8, device=dev)
    hip_ctx.motion_estimate_batch_dev(d_pool.data_ptr(), curs, r0, r1, P, d_sb.data_ptr(), n, d_bat.data_ptr(), True, 0)
    hip_ctx.synchronize()
    a = d_one.cpu().numpy().view(svtav1_hip.ME_CU_RESULT_DTYPE); b = d_bat.cpu().numpy().view(svtav1_hip.ME_CU_RESULT_DTYPE)
    for f in ("totalMeCandidateIndex", "xMvL0", "yMvL0", "xMvL1", "yMvL1", "distortion", "direction"):
        assert np.array_equal(a[f], b[f]), f
