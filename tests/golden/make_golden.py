#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own code (oracle/_ref, built by oracle/build_ref.sh from
/root/reference).  Run in the build container only; the fixtures are data (inputs + reference outputs) and are what
the GPU box checks against, since /root/reference does not exist there.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))

import svtav1_hip  # noqa: E402
from oracle.binding import Reference, ReferenceME  # noqa: E402
from svtav1_hip import synth  # noqa: E402


def fullpel_fixture():
    """Full-pel 85-PU search through the reference's SSE4.1 8-position + SSE2 single-position kernels."""
    ref = Reference()
    rng = np.random.default_rng(20261004)
    w, h = 192, 136
    cur_img = synth.synth_luma(w, h, 1)
    ref_img = synth.synth_luma(w, h, 0)
    cur, rf = synth.PaPicture(cur_img), synth.PaPicture(ref_img)
    cases = {}
    for name, search in (("s64", (64, 64)), ("s23x9", (23, 9)), ("s127", (127, 127))):
        centers = rng.integers(-40, 41, size=(9, 2))
        desc = svtav1_hip.make_fullpel_desc(cur, rf, centers, *search)
        sad, mv = ref.fullpel_search_batch(cur.full, rf.full, desc, asm_type=0)
        cases[name + "_desc"] = desc
        cases[name + "_sad"] = sad
        cases[name + "_mv"] = mv
    np.savez_compressed(os.path.join(HERE, "fullpel_85pu.npz"), cur=cur_img, ref=ref_img, **cases)


def me_chain_fixture():
    """MotionEstimateLcu (sub-pel off) on a panning B picture: per-list origins / SAD / MV and packed me_results."""
    refme = ReferenceME()
    w, h = 320, 192
    big = synth.synth_luma(w + 128, h + 96, 0)
    imgs = [np.ascontiguousarray(big[40:40 + h, 50:50 + w]), np.ascontiguousarray(big[30:30 + h, 14:14 + w]),
            np.ascontiguousarray(big[70:70 + h, 100:100 + w])]
    pics = [synth.PaPicture(x) for x in imgs]
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    out = refme.run(pics[0], pics[1], pics[2], P, two_lists=True, hierarchical_levels=3)
    np.savez_compressed(os.path.join(HERE, "me_lcu_b_picture.npz"), cur=imgs[0], ref0=imgs[1], ref1=imgs[2],
                        hierarchical_levels=3, temporal_layer=1, sad=out["sad"], mv=out["mv"], origin=out["origin"],
                        res=out["res"])


def me_chain209_fixture():
    """MotionEstimateLcu in the 209-PU mode (pic_depth_mode = PIC_ALL_DEPTH_MODE, sub-pel off) on the pictures of
    me_lcu_b_picture.npz: per-list SAD / MV [n,2,209] in ME-buffer order and me_results [n,209,11] in raster PU order."""
    refme = ReferenceME()
    g = np.load(os.path.join(HERE, "me_lcu_b_picture.npz"))
    pics = [synth.PaPicture(g[k]) for k in ("cur", "ref0", "ref1")]
    P = svtav1_hip.default_me_params(pics[0].width, pics[0].height, 3, 1)
    out = refme.run(pics[0], pics[1], pics[2], P, two_lists=True, hierarchical_levels=3, all_pu=True)
    np.savez_compressed(os.path.join(HERE, "me_lcu_b_picture_209pu.npz"), hierarchical_levels=3, temporal_layer=1, sad=out["sad"],
                        mv=out["mv"], origin=out["origin"], res=out["res"])


def interp_fixture():
    refme = ReferenceME()
    img = np.random.default_rng(7).integers(0, 256, (160, 224), dtype=np.uint8)
    plane = synth.pad_plane(img, synth.PAD_FULL)
    off = (synth.PAD_FULL + 40) * plane.shape[1] + synth.PAD_FULL + 60
    b, hh, j = refme.interp_region(plane, off, 24, 16, 16 + 64, 24 + 64)
    np.savez_compressed(os.path.join(HERE, "interp_planes.npz"), img=img, off=off, sw=24, sh=16, b=b, h=hh, j=j)


def fullpel209_fixture():
    """209-PU full-pel search through the reference's ExtSadCalculation* functions (oracle/ref_fullpel209_driver.c)."""
    refme = ReferenceME()
    rng = np.random.default_rng(20261005)
    w, h = 192, 136
    cur_img, ref_img = synth.synth_luma(w, h, 2), synth.synth_luma(w, h, 0)
    cur, rf = synth.PaPicture(cur_img), synth.PaPicture(ref_img)
    cases = {}
    for name, search in (("s64", (64, 64)), ("s40x17", (40, 17))):
        centers = rng.integers(-30, 31, size=(9, 2))
        desc = svtav1_hip.make_fullpel_desc(cur, rf, centers, *search)
        sad, mv = refme.fullpel_search209_batch(cur.full, rf.full, desc)
        cases[name + "_desc"], cases[name + "_sad"], cases[name + "_mv"] = desc, sad, mv
    np.savez_compressed(os.path.join(HERE, "fullpel_209pu.npz"), cur=cur_img, ref=ref_img, **cases)


def transform_fixture():
    """Forward transform, quantiser and inverse transform + reconstruction through the reference's C functions
    (oracle/_ref/libsvtref_tq.so) for the five square sizes, DCT_DCT and one more type each."""
    import ctypes as C
    tq = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtref_tq.so"))
    rng = np.random.default_rng(20261006)
    out = {}
    for n in (4, 8, 16, 32, 64):
        for tx_type in ((0, 3) if n <= 16 else ((0, 9) if n == 32 else (0,))):   # DCT_DCT, ADST_ADST / IDTX
            yy, xx = np.mgrid[0:n, 0:n]
            res = np.clip(40 * np.sin(xx / 3.0) * np.cos(yy / 4.0) + rng.integers(-20, 21, (n, n)), -255, 255).astype(np.int16)
            coeff = np.zeros(n * n, np.int32)
            f = getattr(tq, f"Av1TransformTwoD_{n}x{n}_c"); f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint8]
            f(res.ctypes.data, coeff.ctypes.data, n, tx_type, 8)
            m = min(n, 32)
            packed = np.ascontiguousarray(coeff.reshape(n, n)[:m, :m]).reshape(-1)      # what Av1EstimateTransform keeps
            pred = rng.integers(0, 256, (n, n)).astype(np.uint16)
            rec = pred.copy()
            dq = (packed // 8 * 8).astype(np.int32)                                      # a coarse "dequantised" block
            g = getattr(tq, f"av1_inv_txfm2d_add_{n}x{n}_c"); g.restype = None
            g.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int32]
            g(dq.ctypes.data, rec.ctypes.data, n, tx_type, 8)
            k = f"n{n}_t{tx_type}_"
            out[k + "res"], out[k + "coeff"], out[k + "dq"], out[k + "pred"], out[k + "rec"] = res, coeff, dq, pred, rec
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)


def quant_tables_fixture():
    """The reference's real quantiser rows and scan orders (oracle/ref_quant_tables_driver.c):
    av1_build_quantizer for 8- and 10-bit video with the delta-q sets av1_set_quantizer produces (chroma -20 for inter, -10 for
    intra slices, TUNE_CHROMA_OFFSET), in the ABI's row layout [256][3 planes][10]; av1_scan_orders[tx_size][tx_type] de-duplicated."""
    import ctypes as C
    me = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so"), mode=os.RTLD_LAZY)
    out = {}
    for bd in (8, 10):
        for name, dq in (("inter", -20), ("intra", -10), ("flat", 0)):
            rows = np.zeros((256, 3, 10), np.int16)
            rc = me.ref_build_quantizer_rows(bd, 0, dq, dq, dq, dq, C.c_void_p(rows.ctypes.data))
            assert rc == 0
            out[f"rows_bd{bd}_{name}"] = rows
    tables, index = [], np.full((19, 16), -1, np.int32)
    seen = {}
    for ts, (w, h) in enumerate(svtav1_hip.TX_SIZES_WH):
        n = min(w, 32) * min(h, 32)
        for tt in range(16):
            scan = np.zeros(n, np.int16); iscan = np.zeros(n, np.int16)
            got = me.ref_scan_order(ts, tt, C.c_void_p(scan.ctypes.data), C.c_void_p(iscan.ctypes.data))
            if got == 0:
                continue
            assert got == n
            key = scan.tobytes()
            if key not in seen:
                seen[key] = len(tables)
                tables.append((scan, iscan))
            index[ts, tt] = seen[key]
    out["scan_index"] = index
    out["scan_offsets"] = np.cumsum([0] + [len(t[0]) for t in tables]).astype(np.int32)
    out["scan_pool"] = np.concatenate([t[0] for t in tables])
    out["iscan_pool"] = np.concatenate([t[1] for t in tables])
    np.savez_compressed(os.path.join(HERE, "quant_tables.npz"), **out)


if __name__ == "__main__":
    quant_tables_fixture()
    fullpel209_fixture()
    transform_fixture()
    fullpel_fixture()
    me_chain_fixture()
    me_chain209_fixture()
    interp_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
