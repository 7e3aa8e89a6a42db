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


_CASES_4K = [("h5t1", 5, 1), ("h4t2", 4, 2), ("h5t0", 5, 0)]


def _check_4k(res, ls, lm, g, name, n_pu):
    for l in (0, 1):
        assert np.array_equal(ls[l], g[f"{name}_{n_pu}_sad"][:, l]) and np.array_equal(lm[l], g[f"{name}_{n_pu}_mv"][:, l]), (name, n_pu, l)
    r = g[f"{name}_{n_pu}_res"]
    assert np.array_equal(res["totalMeCandidateIndex"], r[:, :, 8])
    for f, c in (("xMvL0", 0), ("yMvL0", 1), ("xMvL1", 2), ("yMvL1", 3)):
        assert np.array_equal(res[f], r[:, :, c]), (name, n_pu, f)
    for k, (di, dr) in enumerate(((4, 5), (6, 7), (9, 10))):
        assert np.array_equal(res["distortion"][:, :, k], r[:, :, di].astype(np.uint32)) and np.array_equal(res["direction"][:, :, k], r[:, :, dr]), (name, n_pu, k)


@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("case", _CASES_4K)
def test_oracle_me_chain_4k_parameter_set_matches_golden(oracle, case, n_pu):
    """BASELINE configs[4]: the reference's MotionEstimateLcu with the 4K parameter set and resolution class on a crop of a 3840 x 2160
    sequence (tests/golden/make_golden.py::me_chain_4k_fixture) vs the oracle chain."""
    from me_chain_util import oracle_me_picture
    name, hl, tl = case
    g = _load("me_lcu_4k_crop.npz")
    pics = [synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(3840, 2160, hl, tl)
    assert (P.hme_level0_total_search_area_width, P.hme_level0_total_search_area_height) == (128, 80)
    res, per = oracle_me_picture(oracle, pics, P, True, use_subpel=False, n_pu=n_pu)
    for l in (0, 1):
        assert np.array_equal(per[l][0][:, 2:4], g[f"{name}_{n_pu}_origin"][:, l])
    _check_4k(res, [per[0][1], per[1][1]], [per[0][2], per[1][2]], g, name, n_pu)
    # the pan is outside the 64 x 64 full-pel area around the zero vector: the hierarchical levels found it
    if tl > 0:
        o = g[f"{name}_{n_pu}_origin"]
        assert np.median(o[:, 0, 0] + 32) == -22 and np.median(o[:, 0, 1] + 32) == -9 and np.median(o[:, 1, 0] + 32) == 31


@pytest.mark.gpu
@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("case", _CASES_4K)
def test_hip_me_chain_4k_parameter_set_matches_golden(hip_ctx, case, n_pu):
    pytest.importorskip("torch")
    from me_chain_util import device_me_picture
    name, hl, tl = case
    g = _load("me_lcu_4k_crop.npz")
    pics = [synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(3840, 2160, hl, tl)
    res, ls, lm = device_me_picture(hip_ctx, pics, P, True, use_subpel=False, n_pu=n_pu)
    _check_4k(res, ls, lm, g, name, n_pu)


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


# ---- round-2 components: open-loop intra search, AV1 convolutions, picture-analysis planes, SadLoopKernel ----
def test_oracle_ois_matches_golden(oracle):
    g = _load("ois.npz")
    luma = g["luma"]
    h, w = luma.shape
    plane = np.ascontiguousarray(np.pad(luma, 68, mode="edge"))
    for i in range(g["op"].shape[0]):
        cand, total = oracle.ois_search_picture(plane, 68, w, h, g["op"][i], g["me"][i])
        assert np.array_equal(cand, g["cand"][i]) and np.array_equal(total, g["total"][i]), i
    for k, (cx, cy, s) in enumerate(g["pred_cus"]):
        refs = oracle.ois_neighbours(plane, 68, w, h, int(cx), int(cy), int(s))
        for mode in range(35):
            assert np.array_equal(oracle.ois_predict(refs, int(s), mode), g["pred"][k, mode, :s, :s]), (k, mode)


@pytest.mark.gpu
def test_hip_ois_matches_golden(hip_ctx):
    torch = pytest.importorskip("torch")
    g = _load("ois.npz")
    pic = synth.PaPicture(g["luma"])
    w, h = pic.width, pic.height
    pool, descs = svtav1_hip.build_picture_pool([pic])
    sb = svtav1_hip.sb_origins(w, h)
    n_sb = len(sb)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    d_sb = torch.from_numpy(sb.view(np.int16)).to("cuda:0")
    names = ["slice_is_intra", "temporal_layer_index", "is_used_as_reference_flag", "input_resolution_4k", "limit_ois_to_dc_mode_flag", "cu8x8_mode",
             "enc_mode"]
    for i in range(g["op"].shape[0]):
        prm = svtav1_hip.OisParams()
        for k, name in enumerate(names):
            setattr(prm, name, int(g["op"][i][k]))
        rows = np.zeros((n_sb, 85), svtav1_hip.ME_CU_RESULT_DTYPE)
        rows["distortion"][:, :, 0] = g["me"][i]
        d_me = torch.from_numpy(rows.view(np.uint8).reshape(-1)).to("cuda:0")
        d_cand = torch.zeros(n_sb * 85 * 18, dtype=torch.int32, device="cuda:0")
        d_total = torch.zeros(n_sb * 85, dtype=torch.uint8, device="cuda:0")
        hip_ctx.open_loop_intra_search_batch_dev(d_pool.data_ptr(), descs, prm, d_sb.data_ptr(), n_sb, d_me.data_ptr(), 85, d_cand.data_ptr(),
                                                 d_total.data_ptr())
        hip_ctx.synchronize()
        assert np.array_equal(d_cand.cpu().numpy().view(np.uint32).reshape(n_sb, 85, 18), g["cand"][i]), i
        assert np.array_equal(d_total.cpu().numpy().reshape(n_sb, 85), g["total"][i]), i


def _convolve_cases(g):
    for key in g.files:
        if key.startswith("desc_"):
            w, h = (int(v) for v in key[5:].split("x"))
            yield w, h, g[key], g[f"out_{w}x{h}"]


def test_oracle_convolve_matches_golden(oracle):
    import ctypes as C
    g = _load("convolve.npz")
    src = np.ascontiguousarray(g["src"])
    S = src.shape[1]
    f = oracle.lib.orc_av1_convolve_sr
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    for w, h, d, want in _convolve_cases(g):
        for i in range(len(d)):
            got = np.zeros((h, w), np.uint8)
            f(src.ctypes.data + int(d[i, 1]) * S + int(d[i, 0]), S, got.ctypes.data, w, w, h, int(d[i, 4]), int(d[i, 5]), int(d[i, 2]), int(d[i, 3]))
            assert np.array_equal(got, want[i]), (w, h, i)


@pytest.mark.gpu
def test_hip_convolve_matches_golden(hip_ctx):
    torch = pytest.importorskip("torch")
    g = _load("convolve.npz")
    src = np.ascontiguousarray(g["src"])
    S = src.shape[1]
    d_src = torch.from_numpy(np.concatenate([src.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    for w, h, d, want in _convolve_cases(g):
        n = len(d)
        desc = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
        desc["src_offset"] = d[:, 1] * S + d[:, 0]
        desc["dst_offset"] = np.arange(n) * w * h
        desc["subpel_x"], desc["subpel_y"], desc["filter_x"], desc["filter_y"] = d[:, 2], d[:, 3], d[:, 4], d[:, 5]
        # every block is its own w-wide strip of the destination: offsets i * w * h with stride w
        d_dst = torch.zeros(n * w * h + 64, dtype=torch.uint8, device="cuda:0")
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        hip_ctx.av1_convolve_sr_batch_dev(d_src.data_ptr(), S, d_dst.data_ptr(), w, d_desc.data_ptr(), n, w, h)
        hip_ctx.synchronize()
        got = d_dst.cpu().numpy()[:n * w * h].reshape(n, h, w)
        assert np.array_equal(got, want), (w, h)


def test_oracle_pa_planes_and_sad_loop_match_golden(oracle):
    g = _load("pa_sadloop.npz")
    full, quarter, sixteenth = oracle.pa_derive_planes(g["luma"])
    assert np.array_equal(full, g["full"]) and np.array_equal(quarter, g["quarter"]) and np.array_equal(sixteenth, g["sixteenth"])
    cur, ref = synth.PaPicture(g["sl_cur"]).full, synth.PaPicture(g["sl_ref"]).full
    S = cur.shape[1]
    for name in "abc":
        bw, bh, sw, sh, k = (int(v) for v in g[f"sl_{name}_shape"])
        for (so, ro), want in zip(g[f"sl_{name}_desc"], g[f"sl_{name}_res"]):
            assert oracle.sad_loop(cur, int(so), S * k, ref, int(ro), S * k, bh, bw, S, sw, sh) == tuple(int(v) for v in want)


@pytest.mark.gpu
def test_hip_pa_planes_and_sad_loop_match_golden(hip_ctx):
    torch = pytest.importorskip("torch")
    g = _load("pa_sadloop.npz")
    pic = synth.PaPicture(g["luma"])
    pool, descs = svtav1_hip.build_picture_pool([pic])
    raw = np.full_like(pool, 0xEE)
    d = descs[0]
    fs = d.full_stride
    plane = raw[d.full_offset:d.full_offset + fs * (pic.height + 136)].reshape(pic.height + 136, fs)
    plane[68:68 + pic.height, 68:68 + pic.width] = g["luma"]
    d_pool = torch.from_numpy(raw).to("cuda:0")
    hip_ctx.pa_derive_planes_dev(d_pool.data_ptr(), descs)
    hip_ctx.synchronize()
    got = d_pool.cpu().numpy()
    for name in ("full", "quarter", "sixteenth"):
        off = getattr(d, name + "_offset")
        assert np.array_equal(got[off:off + g[name].size].reshape(g[name].shape), g[name]), name
    cur, ref = synth.PaPicture(g["sl_cur"]).full, synth.PaPicture(g["sl_ref"]).full
    S = cur.shape[1]
    d_cur = torch.from_numpy(np.concatenate([cur.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_ref = torch.from_numpy(np.concatenate([ref.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    for name in "abc":
        bw, bh, sw, sh, k = (int(v) for v in g[f"sl_{name}_shape"])
        desc = g[f"sl_{name}_desc"].astype(np.uint32)
        n = len(desc)
        d_desc = torch.from_numpy(desc.view(np.int32).reshape(-1).copy()).to("cuda:0")
        d_sad = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        d_xy = torch.zeros(2 * n, dtype=torch.int16, device="cuda:0")
        hip_ctx.sad_loop_batch_dev(d_cur.data_ptr(), S * k, d_ref.data_ptr(), S * k, S, d_desc.data_ptr(), n, bw, bh, sw, sh, d_sad.data_ptr(),
                                   d_xy.data_ptr())
        hip_ctx.synchronize()
        want = g[f"sl_{name}_res"]
        assert np.array_equal(d_sad.cpu().numpy().astype(np.int64), want[:, 0]), name
        assert np.array_equal(d_xy.cpu().numpy().reshape(n, 2).astype(np.int64), want[:, 1:3]), name


def _compound_cases(g):
    for key in g.files:
        if key.startswith("cdesc_"):
            w, h = (int(v) for v in key[6:].split("x"))
            yield w, h, g[key], g[f"cout_{w}x{h}"]


def test_oracle_convolve_compound_matches_golden(oracle):
    import ctypes as C
    g = _load("convolve.npz")
    s0, s1 = np.ascontiguousarray(g["src"]), np.ascontiguousarray(g["src1"])
    S = s0.shape[1]
    f = oracle.lib.orc_av1_convolve_compound
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_int] * 6
    for w, h, d, want in _compound_cases(g):
        for i in range(len(d)):
            got = np.zeros((h, w), np.uint8)
            f(s0.ctypes.data + int(d[i, 1]) * S + int(d[i, 0]), S, s1.ctypes.data + int(d[i, 3]) * S + int(d[i, 2]), S, got.ctypes.data, w, w, h,
              int(d[i, 8]), int(d[i, 9]), int(d[i, 4]), int(d[i, 5]), int(d[i, 6]), int(d[i, 7]))
            assert np.array_equal(got, want[i]), (w, h, i)


@pytest.mark.gpu
def test_hip_convolve_compound_matches_golden(hip_ctx):
    torch = pytest.importorskip("torch")
    g = _load("convolve.npz")
    s0, s1 = np.ascontiguousarray(g["src"]), np.ascontiguousarray(g["src1"])
    S = s0.shape[1]
    d0 = torch.from_numpy(np.concatenate([s0.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d1 = torch.from_numpy(np.concatenate([s1.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    for w, h, d, want in _compound_cases(g):
        n = len(d)
        desc = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
        desc["src0_offset"] = d[:, 1] * S + d[:, 0]
        desc["src1_offset"] = d[:, 3] * S + d[:, 2]
        desc["dst_offset"] = np.arange(n) * w * h
        desc["subpel0"] = d[:, 4] | (d[:, 5] << 4)
        desc["subpel1"] = d[:, 6] | (d[:, 7] << 4)
        desc["filter_x"], desc["filter_y"] = d[:, 8], d[:, 9]
        d_dst = torch.zeros(n * w * h + 64, dtype=torch.uint8, device="cuda:0")
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        hip_ctx.av1_convolve_compound_batch_dev(d0.data_ptr(), S, d1.data_ptr(), S, d_dst.data_ptr(), w, d_desc.data_ptr(), n, w, h)
        hip_ctx.synchronize()
        assert np.array_equal(d_dst.cpu().numpy()[:n * w * h].reshape(n, h, w), want), (w, h)


# ---- bi-prediction for fractional vectors (the reference's own BiPredictionSearch, row a13) ----
def _bipred_inputs(g):
    cur, r0, r1 = (synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1"))
    return cur, r0, r1, g["desc0"], g["desc1"]


@pytest.mark.parametrize("n_pu", [85, 209])
def test_oracle_bipred_fractional_matches_golden(oracle, n_pu):
    g = _load("bipred_frac.npz")
    cur, r0, r1, d0, d1 = _bipred_inputs(g)
    S = cur.full.shape[1]
    big = np.full(g[f"mv0_{n_pu}"].shape, 0x00ffffff, np.uint32)
    res = oracle.bipred_pack_batch(cur.full, S, r0.full, S, d0, big, g[f"mv0_{n_pu}"], r1.full, S, d1, big, g[f"mv1_{n_pu}"], bipred_8x8=True, n_pu=n_pu)
    assert (res["direction"][:, :, 0] == 2).all() and np.array_equal(res["distortion"][:, :, 0], g[f"bisad_{n_pu}"])


@pytest.mark.gpu
@pytest.mark.parametrize("n_pu", [85, 209])
def test_hip_bipred_fractional_matches_golden(hip_ctx, n_pu):
    torch = pytest.importorskip("torch")
    g = _load("bipred_frac.npz")
    cur, r0, r1, d0, d1 = _bipred_inputs(g)
    S = cur.full.shape[1]
    n_sb = d0.shape[0]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")  # noqa: E731
    pad = np.zeros(256, np.uint8)
    d_cur, d_r0, d_r1 = (dev(np.concatenate([p.full.reshape(-1), pad])) for p in (cur, r0, r1))
    big = np.full((n_sb, n_pu), 0x00ffffff, np.uint32)
    d_big, d_m0, d_m1, d_d0, d_d1 = dev(big), dev(g[f"mv0_{n_pu}"]), dev(g[f"mv1_{n_pu}"]), dev(d0), dev(d1)
    d_out = torch.zeros(n_sb * n_pu * 24, dtype=torch.uint8, device="cuda:0")
    args = (d_cur.data_ptr(), S, d_r0.data_ptr(), S, d_d0.data_ptr(), d_r1.data_ptr(), S, d_d1.data_ptr(), n_sb, 64, 64, d_big.data_ptr(), d_m0.data_ptr(),
            d_big.data_ptr(), d_m1.data_ptr(), 2, d_out.data_ptr())
    if n_pu == 85:
        hip_ctx.bipred_pack_dev(*args, bipred_8x8=True)
    else:
        hip_ctx.bipred_pack209_dev(*args)
    hip_ctx.synchronize()
    res = d_out.cpu().numpy().view(svtav1_hip.ME_CU_RESULT_DTYPE).reshape(n_sb, n_pu)
    assert (res["direction"][:, :, 0] == 2).all() and np.array_equal(res["distortion"][:, :, 0], g[f"bisad_{n_pu}"])


# ---- sub-pel refinement control flow: the reference's HalfPelSearch_LCU + QuarterPelSearch_LCU executed with the SAD search methods ----
@pytest.mark.parametrize("n_pu", [85, 209])
@pytest.mark.parametrize("method", [0, 1])
def test_oracle_subpel_search_matches_golden(oracle, n_pu, method):
    """tests/golden/subpel_search.npz holds what the reference itself computed (make_golden.py::subpel_search_fixture); this is the
    check of rows a11 / a12's control flow that also runs where /root/reference does not exist."""
    g = _load("subpel_search.npz")
    cur, ref = synth.PaPicture(g["cur"]), synth.PaPicture(g["ref"])
    s, m, d = oracle.subpel_refine_method(cur.full, ref.full, g["desc"], g[f"sad0_{n_pu}"], g[f"mv0_{n_pu}"], method, n_pu == 209)
    assert np.array_equal(s, g[f"sad_{n_pu}_m{method}"]) and np.array_equal(m, g[f"mv_{n_pu}_m{method}"])
    assert np.array_equal(d, g[f"dir_{n_pu}_m{method}"])
    assert (m != g[f"mv0_{n_pu}"]).mean() > 0.05  # the fixture really refines
