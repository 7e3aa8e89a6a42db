"""BASELINE configs[4] on one GPU: the ME leg at 3840 x 2160 (plane stride 3976, 60 x 34 = 2040 superblocks, the 4K parameter set of
set_me_hme_params_oq: HME level 0 = 128 x 80 total, Codec/EbMotionEstimationProcess.c:94-156) through the C ABI, bit-exact against the
oracle (pinned to the reference's MotionEstimateLcu with this parameter set by tests/golden/me_lcu_4k_crop.npz), plus the
size-independent properties over all superblocks."""
import numpy as np
import pytest

import svtav1_hip
from me_chain_util import compare_results, device_me_picture, oracle_me_picture
from svtav1_hip import synth

pytestmark = pytest.mark.gpu

W, H = 3840, 2160


@pytest.fixture(scope="module")
def pics4k():
    return [synth.PaPicture(synth.synth_luma(W, H, t)) for t in (3, 0, 7)]


def test_4k_geometry(pics4k):
    assert pics4k[0].full.shape == (H + 136, W + 136) and pics4k[0].stride == 3976
    assert svtav1_hip.sb_origins(W, H).shape[0] == 2040
    P = svtav1_hip.default_me_params(W, H, 3, 0)
    assert (P.hme_level0_total_search_area_width, P.hme_level0_total_search_area_height, P.search_area_width, P.search_area_height) == (128, 80, 64, 64)


def test_search_centres_and_fullpel_all_2040_sbs(hip_ctx, oracle, pics4k):
    """Search centres of EVERY superblock against the oracle; the 85-PU full-pel search on a sample against the oracle and the
    size-independent properties (minimum of a sum; identical pictures give zero) on all of them."""
    torch = pytest.importorskip("torch")
    from test_hme_gpu import DeviceChain
    P = svtav1_hip.default_me_params(W, H, 3, 0)
    dev = DeviceChain(hip_ctx, pics4k).run(P, False)
    pool, descs = svtav1_hip.build_picture_pool(pics4k)
    sb = svtav1_hip.sb_origins(W, H)
    d0, c0 = oracle.hme_search_center_batch(pool, descs[0], descs[1], P, 0, sb)
    assert np.array_equal(dev[0][0], d0) and np.array_equal(dev[0][1], c0)
    assert len(np.unique(c0, axis=0)) > 20   # the centres are not degenerate
    sample = np.sort(np.random.default_rng(5).choice(sb.shape[0], 64, replace=False))
    sample = np.unique(np.concatenate([sample, [0, 59, 2039, 2040 - 60, 1000]]))   # corners: clipped windows, partial bottom row (2160 = 33.75 x 64)
    s0, m0 = oracle.fullpel_search_batch(pool, pool, d0[sample], descs[0].full_stride, descs[1].full_stride)
    assert np.array_equal(dev[0][2][sample], s0) and np.array_equal(dev[0][3][sample], m0)
    s_h = dev[0][2].astype(np.int64)
    assert (s_h[:, 1:5].sum(1) <= s_h[:, 0]).all() and (s_h[:, 5:21].sum(1) <= s_h[:, 1:5].sum(1)).all() and (s_h[:, 21:85].sum(1) <= s_h[:, 5:21].sum(1)).all()
    # identical pictures: zero SAD for every PU of every superblock, zero-centred windows
    desc = svtav1_hip.make_fullpel_desc(pics4k[0], pics4k[0], None, 64, 64)
    d_pic = torch.from_numpy(pics4k[0].full).to("cuda:0")
    d_desc = torch.from_numpy(desc).to("cuda:0")
    d_sad = torch.ones((desc.shape[0], 85), dtype=torch.int32, device="cuda:0")
    d_mv = torch.zeros((desc.shape[0], 85), dtype=torch.int32, device="cuda:0")
    hip_ctx.fullpel_search_dev(d_pic.data_ptr(), pics4k[0].stride, d_pic.data_ptr(), pics4k[0].stride, d_desc.data_ptr(), desc.shape[0], 64, 64,
                               d_sad.data_ptr(), d_mv.data_ptr())
    hip_ctx.synchronize()
    assert int(d_sad.abs().max()) == 0


@pytest.mark.parametrize("n_pu,n_sample", [(85, 24), (209, 12)])
def test_whole_4k_b_picture_with_subpel(hip_ctx, oracle, pics4k, n_pu, n_sample):
    """One whole 3840 x 2160 B picture through svthip_motion_estimate[209]_batch_dev (search centres, full-pel, sub-pel, bi-prediction,
    packing): a sample of superblocks against oracle_me_picture, properties on all 2040."""
    pytest.importorskip("torch")
    P = svtav1_hip.default_me_params(W, H, 3, 1)
    res_d, ls, lm = device_me_picture(hip_ctx, pics4k, P, True, True, 0, n_pu=n_pu)
    assert res_d.shape == (2040, n_pu) and (res_d["totalMeCandidateIndex"] == 3).all()
    d = res_d["distortion"]
    assert (d[:, :, 0] <= d[:, :, 1]).all() and (d[:, :, 1] <= d[:, :, 2]).all()
    assert (np.sort(res_d["direction"], axis=2) == np.arange(3)).all()
    sample = np.unique(np.concatenate([np.random.default_rng(29 + n_pu).choice(2040, n_sample, replace=False), [0, 59, 1980, 2039]]))
    res_o, per = oracle_me_picture(oracle, pics4k, P, True, True, 0, sb_subset=sample, n_pu=n_pu)
    for l in per:
        assert np.array_equal(ls[l][sample], per[l][1]) and np.array_equal(lm[l][sample], per[l][2]), l
    compare_results(res_d[sample], res_o)


def test_4k_host_pointer_picture_entry(hip_ctx, oracle, pics4k):
    """svthip_motion_estimate_picture (host pointers, MeCuResults_t rows) at 3840 x 2160: P picture, rows of one allocation."""
    import ctypes as C
    P = svtav1_hip.default_me_params(W, H, 3, 0)
    L = svtav1_hip.lib()

    class HostPicture(C.Structure):
        _fields_ = [("buffer_y", C.c_void_p), ("stride_y", C.c_uint32), ("origin_x", C.c_uint16), ("origin_y", C.c_uint16), ("width", C.c_uint16),
                    ("height", C.c_uint16)]

    hp = [HostPicture(p.full.ctypes.data, p.stride, 68, 68, W, H) for p in pics4k[:2]]
    rows = np.zeros((2040, 85, 40), np.uint8)
    ptrs = (C.c_void_p * 2040)(*[rows.ctypes.data + i * 85 * 40 for i in range(2040)])
    L.svthip_motion_estimate_picture.restype = C.c_int32
    L.svthip_motion_estimate_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p]
    hip_ctx.reserve(W, H, 85, 1, True)
    rc = L.svthip_motion_estimate_picture(hip_ctx._h, C.byref(hp[0]), C.byref(hp[1]), None, C.byref(P), 1, 0, 85, ptrs)
    assert rc == 0, L.svthip_last_error()
    sample = np.sort(np.random.default_rng(3).choice(2040, 16, replace=False))
    res_o, _ = oracle_me_picture(oracle, pics4k, P, False, True, 0, sb_subset=sample)
    r = rows[sample]
    assert np.array_equal(r[:, :, 0:2].view(np.int16)[..., 0], res_o["xMvL0"]) and np.array_equal(r[:, :, 2:4].view(np.int16)[..., 0], res_o["yMvL0"])
    assert np.array_equal(r[:, :, 8:12].view(np.uint32)[..., 0], res_o["distortion"][:, :, 0])
    assert (rows[:, :, 32] == 1).all()   # totalMeCandidateIndex of a P picture
