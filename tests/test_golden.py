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
