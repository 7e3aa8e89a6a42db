"""Golden fixtures generated from the reference's own code (tests/golden/make_golden.py): the CPU oracle must
reproduce them (not gpu), and so must the HIP path through the C ABI (gpu)."""
import os

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_oracle_fullpel_matches_golden(oracle):
    g = _load("fullpel_85pu.npz")
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    for name in ("s64", "s23x9", "s127"):
        sad, mv = oracle.fullpel_search_batch(cur.full, ref.full, g[name + "_desc"])
        assert np.array_equal(sad, g[name + "_sad"]) and np.array_equal(mv, g[name + "_mv"])


def test_oracle_me_chain_matches_golden(oracle):
    from me_chain_util import oracle_me_picture
    g = _load("me_lcu_b_picture.npz")
    pics = [synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(pics[0].width, pics[0].height, int(g["hierarchical_levels"]), int(g["temporal_layer"]))
    res, per = oracle_me_picture(oracle, pics, P, True, use_subpel=False)
    for l in (0, 1):
        assert np.array_equal(per[l][0][:, 2:4], g["origin"][:, l])
        assert np.array_equal(per[l][1], g["sad"][:, l]) and np.array_equal(per[l][2], g["mv"][:, l])
    r = g["res"]
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    assert np.array_equal(res["distortion"][:, :, 0], r[:, :, 4].astype(np.uint32))
    assert np.array_equal(res["direction"][:, :, 0], r[:, :, 5])
    assert np.array_equal(res["distortion"][:, :, 1], r[:, :, 6].astype(np.uint32))


def test_oracle_interp_planes_match_golden(oracle):
    g = _load("interp_planes.npz")
    plane = synth.pad_plane(g["img"], synth.PAD_FULL)
    b, h, j = oracle.interp_planes(plane, int(g["off"]), 0, 0, g["b"].shape[1], g["b"].shape[0])
    assert np.array_equal(b, g["b"]) and np.array_equal(h, g["h"]) and np.array_equal(j, g["j"])


@pytest.mark.gpu
def test_hip_fullpel_matches_golden(hip_ctx):
    g = _load("fullpel_85pu.npz")
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    for name in ("s64", "s23x9", "s127"):
        sad, mv = hip_ctx.fullpel_search(cur.full, ref.full, g[name + "_desc"])
        assert np.array_equal(sad, g[name + "_sad"]) and np.array_equal(mv, g[name + "_mv"])


@pytest.mark.gpu
def test_hip_me_chain_matches_golden(hip_ctx):
    pytest.importorskip("torch")
    from me_chain_util import device_me_picture
    g = _load("me_lcu_b_picture.npz")
    pics = [synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(pics[0].width, pics[0].height, int(g["hierarchical_levels"]), int(g["temporal_layer"]))
    res, ls, lm = device_me_picture(hip_ctx, pics, P, True, use_subpel=False)
    for l in (0, 1):
        assert np.array_equal(ls[l], g["sad"][:, l]) and np.array_equal(lm[l], g["mv"][:, l])
    r = g["res"]
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    assert np.array_equal(res["xMvL0"], r[:, :, 0]) and np.array_equal(res["yMvL1"], r[:, :, 3])
    assert np.array_equal(res["distortion"][:, :, 0], r[:, :, 4].astype(np.uint32))
    assert np.array_equal(res["direction"][:, :, 1], r[:, :, 7])


def _check_chain209(res, ls, lm, g):
    for l in (0, 1):
        assert np.array_equal(ls[l], g["sad"][:, l]) and np.array_equal(lm[l], g["mv"][:, l])
    r = g["res"]  # [n,209,11] = xMvL0,yMvL0,xMvL1,yMvL1, dist0,dir0, dist1,dir1, total, dist2,dir2
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    for f, c in (("xMvL0", 0), ("yMvL0", 1), ("xMvL1", 2), ("yMvL1", 3)):
        assert np.array_equal(res[f], r[:, :, c]), f
    for k, (di, dr) in enumerate(((4, 5), (6, 7), (9, 10))):
        assert np.array_equal(res["distortion"][:, :, k], r[:, :, di].astype(np.uint32)), f"distortion[{k}]"
        assert np.array_equal(res["direction"][:, :, k], r[:, :, dr]), f"direction[{k}]"


def test_oracle_me_chain209_matches_golden(oracle):
    from me_chain_util import oracle_me_picture
    gi, g = _load("me_lcu_b_picture.npz"), _load("me_lcu_b_picture_209pu.npz")
    pics = [synth.PaPicture(gi[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(pics[0].width, pics[0].height, int(g["hierarchical_levels"]), int(g["temporal_layer"]))
    res, per = oracle_me_picture(oracle, pics, P, True, use_subpel=False, n_pu=209)
    _check_chain209(res, [per[0][1], per[1][1]], [per[0][2], per[1][2]], g)


@pytest.mark.gpu
def test_hip_me_chain209_matches_golden(hip_ctx):
    """The reference's own MotionEstimateLcu output in the 209-PU mode (sub-pel off) vs svthip_motion_estimate209_batch_dev."""
    pytest.importorskip("torch")
    from me_chain_util import device_me_picture
    gi, g = _load("me_lcu_b_picture.npz"), _load("me_lcu_b_picture_209pu.npz")
    pics = [synth.PaPicture(gi[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(pics[0].width, pics[0].height, int(g["hierarchical_levels"]), int(g["temporal_layer"]))
    res, ls, lm = device_me_picture(hip_ctx, pics, P, True, use_subpel=False, n_pu=209)
    _check_chain209(res, ls, lm, g)


def test_oracle_fullpel209_matches_golden(oracle):
    g = _load("fullpel_209pu.npz")
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    for name in ("s64", "s40x17"):
        sad, mv = oracle.fullpel_search209_batch(cur.full, ref.full, g[name + "_desc"])
        assert np.array_equal(sad, g[name + "_sad"]) and np.array_equal(mv, g[name + "_mv"])


@pytest.mark.gpu
def test_hip_fullpel209_matches_golden(hip_ctx):
    torch = pytest.importorskip("torch")
    g = _load("fullpel_209pu.npz")
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    dev = torch.device("cuda:0")
    d_src, d_ref = torch.from_numpy(cur.full).to(dev), torch.from_numpy(ref.full).to(dev)
    for name in ("s64", "s40x17"):
        desc = g[name + "_desc"]
        n = desc.shape[0]
        d_desc = torch.from_numpy(np.ascontiguousarray(desc)).to(dev)
        d_sad = torch.zeros((n, 209), dtype=torch.int32, device=dev); d_mv = torch.zeros_like(d_sad)
        hip_ctx.fullpel_search209_dev(d_src.data_ptr(), cur.stride, d_ref.data_ptr(), ref.stride, d_desc.data_ptr(), n,
                                      int(desc[:, 4].max()), int(desc[:, 5].max()), d_sad.data_ptr(), d_mv.data_ptr())
        hip_ctx.synchronize()
        assert np.array_equal(d_sad.cpu().numpy().view(np.uint32), g[name + "_sad"])
        assert np.array_equal(d_mv.cpu().numpy().view(np.uint32), g[name + "_mv"])


def _transform_cases(g):
    for key in g.files:
        if key.endswith("_res"):
            n, t = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
            yield n, t, key[:-3]


def test_oracle_transforms_match_golden(oracle):
    import ctypes as C
    g = _load("transforms.npz")
    fwd, inv = oracle.lib.orc_fwd_txfm2d, oracle.lib.orc_inv_txfm2d_add
    fwd.restype = None; fwd.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    inv.restype = None; inv.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    n_cases = 0
    for n, t, k in _transform_cases(g):
        res = np.ascontiguousarray(g[k + "res"]); out = np.zeros(n * n, np.int32)
        fwd(res.ctypes.data, n, n, n, t, out.ctypes.data)
        assert np.array_equal(out, g[k + "coeff"]), (n, t)
        rec = np.ascontiguousarray(g[k + "pred"]).copy(); dq = np.ascontiguousarray(g[k + "dq"])
        inv(dq.ctypes.data, rec.ctypes.data, n, n, n, t, 8)
        assert np.array_equal(rec, g[k + "rec"]), (n, t)
        n_cases += 1
    assert n_cases == 9


@pytest.mark.gpu
def test_hip_transforms_match_golden(hip_ctx):
    """Reference outputs (generated in the build container) against the HIP forward / inverse transform entries."""
    torch = pytest.importorskip("torch")
    g = _load("transforms.npz")
    dev = torch.device("cuda:0")
    for n, t, k in _transform_cases(g):
        d_res = torch.from_numpy(np.ascontiguousarray(g[k + "res"])).to(dev)
        fd = np.zeros(1, dtype=svtav1_hip.TXFM_DESC_DTYPE); fd["in_stride"] = n; fd["tx_type"] = t
        d_fd = torch.from_numpy(fd.view(np.uint8).copy()).to(dev)
        d_c = torch.zeros(n * n, dtype=torch.int32, device=dev)
        hip_ctx.fwd_txfm2d_batch_dev(d_res.data_ptr(), d_fd.data_ptr(), 1, n, n, 8, d_c.data_ptr())
        m = min(n, 32)
        d_dq = torch.from_numpy(np.ascontiguousarray(g[k + "dq"])).to(dev)
        d_rec = torch.from_numpy(np.ascontiguousarray(g[k + "pred"]).astype(np.uint16)).to(dev)
        idd = np.zeros(1, dtype=svtav1_hip.ITXFM_DESC_DTYPE); idd["recon_stride"] = n; idd["tx_type"] = t
        d_id = torch.from_numpy(idd.view(np.uint8).copy()).to(dev)
        hip_ctx.inv_txfm2d_add_batch_dev(d_dq.data_ptr(), d_id.data_ptr(), 1, n, n, 8, True, d_rec.data_ptr())
        hip_ctx.synchronize()
        assert np.array_equal(d_c.cpu().numpy(), g[k + "coeff"]), (n, t)
        assert np.array_equal(d_rec.cpu().numpy().view(np.uint16).reshape(n, n), g[k + "rec"]), (n, t)
