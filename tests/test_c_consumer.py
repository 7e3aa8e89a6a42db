"""The C99 consumer of include/svtav1_hip.h (tests/c_consumer/consumer.c): compiles against the header with gcc -std=c99 (struct
layouts are _Static_assert'ed there, including svthip_me_cu_result_ref == the reference's MeCuResults_t as the host compiler lays it
out), and -- on the GPU box -- drives create -> host-pointer full-pel -> whole-picture ME into MeCuResults_t rows -> host-pointer fused
TU chain -> RTCD same-signature shims -> two threads x two contexts -> destroy.  Its outputs are compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CDIR = os.path.join(ROOT, "tests", "c_consumer")
EXE = os.path.join(CDIR, "consumer")

T = dict(DIMS=1, CUR=2, REF0=3, REF1=4, PARAMS=5, FP_DESC=6, TU_SRC=7, TU_PRED=8, TU_DESC=9, TU_QP=10, TU_ISCAN=11, TU_DIMS=12, TX_RES=13,
         TX_COEFFQ=14, TX_QROW=15, TX_SCAN=16, TX_ISCAN=17, TX_PRED=18, FP_DESC_SMALL=19, ITX_COEFF=20, ITX_PRED=21, ITX_TYPE=22, SAD_BLOCK=23)
O = dict(FP_SAD=100, FP_MV=101, ME=102, TU_RECON=103, TU_Q=104, TU_EOB=105, TX_FWD=106, TX_INV=107, TX_Q=108, TX_DQ=109, TX_EOB=110, THREADS=111,
         TU_DIST=112, ME209=113, OIS_GEN_CAND=114, OIS_GEN_TOTAL=115, OIS_I_CAND=116, OIS_I_TOTAL=117, ITX_RECON=118, SAD=119)


def test_header_compiles_as_c99_and_layouts_hold():
    """CPU: the header is plain C99 and every _Static_assert in the consumer holds (no GPU, no link)."""
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only",
                        os.path.join(CDIR, "consumer.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_mirrors_match_the_header_sizes():
    import ctypes as C
    assert C.sizeof(svtav1_hip.MeParams) == 52 and C.sizeof(svtav1_hip.PaPictureDesc) == 40 and C.sizeof(svtav1_hip.FullpelDesc) == 24
    assert svtav1_hip.ME_CU_RESULT_DTYPE.itemsize == 24 and svtav1_hip.TU_DESC_DTYPE.itemsize == 32


def _write(path, sections):
    with open(path, "wb") as f:
        for tag, arr in sections:
            b = arr if isinstance(arr, (bytes, bytearray)) else np.ascontiguousarray(arr).tobytes()
            f.write(struct.pack("<IIQ", tag, 0, len(b)))
            f.write(b)


def _read(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            h = f.read(16)
            if len(h) < 16:
                break
            tag, _, n = struct.unpack("<IIQ", h)
            out[tag] = f.read(n)
    return out


@pytest.mark.gpu
def test_c_consumer_end_to_end(tmp_path, oracle):
    import ctypes as C
    from me_chain_util import oracle_me_picture
    from tq_util import RealTables, frame_encode_batch, oracle_encode_batch
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", CDIR])
    w, h = 320, 192
    pics = [synth.PaPicture(synth.synth_luma(w, h, t)) for t in (3, 0, 7)]
    P = svtav1_hip.default_me_params(w, h, 3, 1)
    desc = svtav1_hip.make_fullpel_desc(pics[0], pics[1], None, 64, 64)
    rng = np.random.default_rng(4)
    desc_small = svtav1_hip.make_fullpel_desc(pics[0], pics[2], rng.integers(-20, 21, (desc.shape[0], 2)), 16, 9)
    tables = RealTables()
    tb = frame_encode_batch(rng, 16, 16, 256, 128, tables)
    # RTCD inputs
    res = rng.integers(-255, 256, (16, 40)).astype(np.int16)
    qrow = np.ascontiguousarray(tables.rows(8, "inter")[120, 0])
    so = tables.scan_offset(2, 0)
    scan = np.ascontiguousarray(tables.scan_pool[so:so + 256]); iscan = np.ascontiguousarray(tables.iscan_pool[so:so + 256])
    coeffq = rng.laplace(0, 300, 256).astype(np.int32)
    txpred = rng.integers(0, 256, (16, 24)).astype(np.uint16)
    # av1_inv_txfm_add inputs: per transform size a packed coefficient block, a 64 x 80 prediction tile and a defined tx_type
    itx_coeff = np.zeros((19, 1024), np.int32)
    itx_pred = rng.integers(0, 256, (19, 64, 80)).astype(np.uint8)
    itx_type = np.zeros(19, np.uint8)
    for ts, (tw, th) in enumerate(svtav1_hip.TX_SIZES_WH):
        n = min(tw, 32) * min(th, 32)
        itx_coeff[ts, :n] = (rng.laplace(0, 60, n) * (rng.random(n) < 0.3)).astype(np.int32)
        itx_coeff[ts, 0] = 900 - 100 * ts
        types = svtav1_hip.valid_tx_types(tw, th)
        itx_type[ts] = types[(3 * ts) % len(types)]
    sad_plane = rng.integers(0, 256, (128, 128)).astype(np.uint8)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write(inp, [(T["DIMS"], np.array([w, h], np.uint32)), (T["CUR"], pics[0].full), (T["REF0"], pics[1].full), (T["REF1"], pics[2].full),
                 (T["PARAMS"], bytes(P)), (T["FP_DESC"], desc), (T["FP_DESC_SMALL"], desc_small),
                 (T["TU_SRC"], tb["src"]), (T["TU_PRED"], tb["pred"]), (T["TU_DESC"], tb["desc"]), (T["TU_QP"], tb["qparams"]),
                 (T["TU_ISCAN"], tb["iscan"]), (T["TU_DIMS"], np.array([16, 16, tb["src"].size, tb["qparams"].shape[0], len(tb["desc"]) * 256], np.uint32)),
                 (T["TX_RES"], res), (T["TX_COEFFQ"], coeffq), (T["TX_QROW"], qrow), (T["TX_SCAN"], scan), (T["TX_ISCAN"], iscan), (T["TX_PRED"], txpred),
                 (T["ITX_COEFF"], itx_coeff), (T["ITX_PRED"], itx_pred), (T["ITX_TYPE"], itx_type), (T["SAD_BLOCK"], sad_plane)])
    r = subprocess.run([EXE, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    got = _read(outp)
    # 1. full-pel
    s_o, m_o = oracle.fullpel_search_batch(pics[0].full, pics[1].full, desc)
    assert np.array_equal(np.frombuffer(got[O["FP_SAD"]], np.uint32).reshape(-1, 85), s_o)
    assert np.array_equal(np.frombuffer(got[O["FP_MV"]], np.uint32).reshape(-1, 85), m_o)
    # 2. whole-picture ME read back through the host's own MeCuResults_t bit-field struct
    for tag, n_pu in ((O["ME"], 85), (O["ME209"], 209)):
        want, _ = oracle_me_picture(oracle, pics, P, True, True, 0, n_pu=n_pu)
        flat = np.frombuffer(got[tag], np.int32).reshape(-1, n_pu, 11)
        assert np.array_equal(flat[:, :, 0], want["xMvL0"]) and np.array_equal(flat[:, :, 1], want["yMvL0"])
        assert np.array_equal(flat[:, :, 2], want["xMvL1"]) and np.array_equal(flat[:, :, 3], want["yMvL1"])
        assert np.array_equal(flat[:, :, 4:7].astype(np.uint32), want["distortion"]) and np.array_equal(flat[:, :, 7:10], want["direction"])
        assert np.array_equal(flat[:, :, 10], want["totalMeCandidateIndex"])
        if n_pu == 85:   # 2b. open-loop intra search fed from the host's MeCuResults_t rows (general branch) and as an intra picture
            n_sb = want["distortion"].shape[0]
            op = np.zeros(7, np.int32); op[1] = 2; op[2] = 1
            wc, wt = oracle.ois_search_picture(pics[0].full, 68, w, h, op, want["distortion"][:, :85, 0])
            assert np.array_equal(np.frombuffer(got[O["OIS_GEN_CAND"]], np.uint32).reshape(n_sb, 85, 18), wc)
            assert np.array_equal(np.frombuffer(got[O["OIS_GEN_TOTAL"]], np.uint8).reshape(n_sb, 85), wt)
            assert len(np.unique(wt)) >= 2
            op = np.zeros(7, np.int32); op[0] = 1
            wc, wt = oracle.ois_search_picture(pics[0].full, 68, w, h, op, None)
            assert np.array_equal(np.frombuffer(got[O["OIS_I_CAND"]], np.uint32).reshape(n_sb, 85, 18), wc)
            assert np.array_equal(np.frombuffer(got[O["OIS_I_TOTAL"]], np.uint8).reshape(n_sb, 85), wt)
    # 3. fused TU chain, host pointers
    want = oracle_encode_batch(oracle, tb)
    assert np.array_equal(np.frombuffer(got[O["TU_RECON"]], np.uint8), want["recon"])
    assert np.array_equal(np.frombuffer(got[O["TU_Q"]], np.int32), want["qcoeff"])
    assert np.array_equal(np.frombuffer(got[O["TU_EOB"]], np.uint16), want["eob"])
    assert np.array_equal(np.frombuffer(got[O["TU_DIST"]], np.uint64).reshape(-1, 2), want["dist"])
    # 4. RTCD shims vs the oracle stages
    f = oracle.lib.orc_fwd_txfm2d; f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    wf = np.zeros(256, np.int32)
    f(res.ctypes.data, 40, 16, 16, 3, wf.ctypes.data)
    assert np.array_equal(np.frombuffer(got[O["TX_FWD"]], np.int32), wf)
    q = oracle.lib.orc_quantize_b; q.restype = None
    q.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    oq = np.zeros(256, np.int32); odq = np.zeros(256, np.int32); oe = C.c_uint16(0)
    q(coeffq.ctypes.data, 256, qrow.ctypes.data, scan.ctypes.data, 0, 0, oq.ctypes.data, odq.ctypes.data, C.addressof(oe))
    assert np.array_equal(np.frombuffer(got[O["TX_Q"]], np.int32), oq) and np.array_equal(np.frombuffer(got[O["TX_DQ"]], np.int32), odq)
    assert np.frombuffer(got[O["TX_EOB"]], np.uint16)[0] == oe.value and oe.value > 0
    g = oracle.lib.orc_inv_txfm2d_add; g.restype = None
    g.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    wr = txpred.copy()
    g(odq.ctypes.data, wr.ctypes.data, 24, 16, 16, 0, 8)
    assert np.array_equal(np.frombuffer(got[O["TX_INV"]], np.uint16).reshape(16, 24), wr)
    # 4b. av1_inv_txfm_add (8-bit plane, TxfmParam) for all 19 sizes: the oracle reconstructs the widened tile like av1_inv_txfm_add_c does
    got_itx = np.frombuffer(got[O["ITX_RECON"]], np.uint8).reshape(19, 64, 80)
    for ts, (tw, th) in enumerate(svtav1_hip.TX_SIZES_WH):
        wide = itx_pred[ts].astype(np.uint16)
        g(itx_coeff[ts].ctypes.data, wide.ctypes.data, 80, tw, th, int(itx_type[ts]), 8)
        assert np.array_equal(got_itx[ts], wide.astype(np.uint8)), (tw, th)
        assert not np.array_equal(got_itx[ts][:th, :tw], itx_pred[ts][:th, :tw]), (tw, th)
        assert np.array_equal(got_itx[ts][th:], itx_pred[ts][th:]) and np.array_equal(got_itx[ts][:, tw:], itx_pred[ts][:, tw:])
    # 4c. leaf SAD pointers vs the oracle's restatement of the reference's C kernels
    sres = np.frombuffer(got[O["SAD"]], np.uint32)
    u8p = C.POINTER(C.c_uint8)
    P8 = lambda a, off: C.cast(a.ctypes.data + off, u8p)   # noqa: E731
    assert sres[0] == oracle.lib.orc_nxm_sad(P8(sad_plane, 8 * 128 + 8), 128, P8(sad_plane, 24 * 128 + 16), 128, 16, 16)
    assert sres[1] == oracle.lib.orc_nxm_sad(P8(sad_plane, 8 * 128 + 8), 256, P8(sad_plane, 24 * 128 + 16), 256, 16, 32)
    for k, (ss, rs, hh, ww, raw, sw_, sh_) in enumerate(((128, 128, 16, 16, 128, 33, 33), (256, 256, 8, 16, 128, 24, 12))):
        bs, bx, by = C.c_uint64(0), C.c_int16(0), C.c_int16(0)
        oracle.lib.orc_sad_loop_kernel(P8(sad_plane, 8 * 128 + 8), ss, P8(sad_plane, 24 * 128 + 16), rs, hh, ww, C.byref(bs), C.byref(bx), C.byref(by), raw, sw_, sh_)
        assert sres[2 + 2 * k] == bs.value and sres[3 + 2 * k] == (bx.value & 0xffff) | ((by.value & 0xffff) << 16), k
    # 5. two threads x two contexts with different search areas: every iteration reproduced the single-threaded result
    assert np.frombuffer(got[O["THREADS"]], np.int32).tolist() == [0, 0]
