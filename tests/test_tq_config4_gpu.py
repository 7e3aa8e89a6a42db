"""BASELINE configs[3] at its full shapes: TU batches covering one 1080p luma frame per size -- 130 560 4x4,
32 640 8x8, 8 160 16x16, 2 040 32x32, 510 64x64 (1920 x 1088: the frame rounded up to whole superblocks) -- residual Laplacian(b = 6)
and uniform [-255, 255], transform types DCT_DCT / ADST_ADST / IDTX where defined, the reference's REAL quantiser rows at qindex
20 / 120 / 200 and REAL scan orders (tests/golden/quant_tables.npz).  Fused encode chain on the GPU vs the oracle on a sample of
TUs, plus size-independent properties over every TU.  Also the 10-bit chain at 3840x2176 (configs[4])."""
import numpy as np
import pytest

import svtav1_hip
from tq_util import RealTables, frame_encode_batch, oracle_encode_batch

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def _run(hip_ctx, b, want_coeff=True):
    import torch
    n_tu, n, w, h = len(b["desc"]), b["n"], b["w"], b["h"]
    wide = b["bit_depth"] != 8
    d_src, d_pred, d_desc, d_qp, d_iscan = _dev(b["src"]), _dev(b["pred"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_q = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_dq = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_coeff = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_eob = torch.full((n_tu,), -1, dtype=torch.int16, device="cuda:0")
    d_en = torch.full((n_tu,), -1, dtype=torch.int64, device="cuda:0")
    d_dist = torch.full((n_tu, 2), -1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.encode_tu_batch_dev(d_src.data_ptr(), d_pred.data_ptr(), d_pred.data_ptr(), d_desc.data_ptr(), n_tu, w, h, d_qp.data_ptr(),
                                d_iscan.data_ptr(), d_coeff.data_ptr(), d_q.data_ptr(), d_dq.data_ptr(), d_eob.data_ptr(), d_en.data_ptr(),
                                d_dist.data_ptr(), planes_16bit=wide)
    hip_ctx.synchronize()
    rec = d_pred.cpu().numpy().view(np.uint16 if wide else np.uint8)
    return {"recon": rec, "coeff": d_coeff.cpu().numpy(), "qcoeff": d_q.cpu().numpy(), "dqcoeff": d_dq.cpu().numpy(),
            "eob": d_eob.cpu().numpy().view(np.uint16), "energy": d_en.cpu().numpy().view(np.uint64),
            "dist": d_dist.cpu().numpy().view(np.uint64)}


def _check(hip_ctx, oracle, b, pic_w, n_sample, rng):
    got = _run(hip_ctx, b)
    n_tu, n, w, h = len(b["desc"]), b["n"], b["w"], b["h"]
    # --- a sample of TUs against the oracle chain ---
    pick = np.sort(rng.choice(n_tu, min(n_sample, n_tu), replace=False))
    sub = dict(b)
    sub["desc"] = b["desc"][pick].copy()
    sub["desc"]["coeff_offset"] = np.arange(len(pick)) * n
    want = oracle_encode_batch(oracle, sub)
    for k, i in enumerate(pick):
        for f in ("coeff", "qcoeff", "dqcoeff"):
            assert np.array_equal(got[f][i * n:(i + 1) * n], want[f][k * n:(k + 1) * n]), (f, w, h, int(i))
        assert got["eob"][i] == want["eob"][k] and got["energy"][i] == want["energy"][k] and np.array_equal(got["dist"][i], want["dist"][k])
        o = int(b["desc"][i]["recon_offset"])
        for r in range(h):
            assert np.array_equal(got["recon"][o + r * pic_w:o + r * pic_w + w], want["recon"][o + r * pic_w:o + r * pic_w + w]), (w, h, int(i), r)
    # --- properties over every TU ---
    q = got["qcoeff"].reshape(n_tu, n)
    nz = (q != 0).any(axis=1)
    assert np.array_equal(nz, got["eob"] > 0)                     # eob = 0 exactly for all-zero blocks
    assert ((q != 0).sum(axis=1) <= got["eob"]).all()             # at most eob non-zero levels
    dq = got["dqcoeff"].reshape(n_tu, n)
    assert np.array_equal(dq != 0, q != 0) and np.array_equal(np.sign(dq), np.sign(q))   # dequant >= 4, log_scale <= 2
    pred = b["pred"]
    rec = got["recon"]
    empty = np.flatnonzero(~nz)[:200]
    for i in empty:                                               # nothing to add: the prediction is the reconstruction
        o = int(b["desc"][i]["recon_offset"])
        for r in range(h):
            assert np.array_equal(rec[o + r * pic_w:o + r * pic_w + w], pred[o + r * pic_w:o + r * pic_w + w])
    assert int(rec.max()) <= (255 if b["bit_depth"] == 8 else 1023)
    return got


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("residual", ["laplace", "uniform"])
def test_config4_frame_batches_real_tables(hip_ctx, oracle, n, residual):
    pytest.importorskip("torch")
    tables = RealTables()
    rng = np.random.default_rng(n * 3 + len(residual))
    b = frame_encode_batch(rng, n, n, 1920, 1088, tables, residual=residual)
    assert len(b["desc"]) == {4: 130560, 8: 32640, 16: 8160, 32: 2040, 64: 510}[n]
    got = _check(hip_ctx, oracle, b, 1920, 160 if n <= 16 else 40, rng)
    if residual == "laplace":   # qindex 20 keeps coefficients, qindex 200 kills most of a b = 6 residual
        qi = b["desc"]["qparam_index"]
        assert (got["eob"][qi == 0] > 0).mean() > (got["eob"][qi == 2] > 0).mean()


@pytest.mark.parametrize("size", [(16, 16), (64, 64), (32, 8)])
def test_config5_10bit_4k_frame_batches(hip_ctx, oracle, size):
    """configs[4]: 3840x2160 10-bit (rounded up to 2176 rows), svthip_encode_tu16_batch_dev with bd-10 rows of av1_build_quantizer."""
    pytest.importorskip("torch")
    tables = RealTables()
    w, h = size
    rng = np.random.default_rng(w + h)
    b = frame_encode_batch(rng, w, h, 3840, 2176, tables, bit_depth=10)
    assert len(b["desc"]) == (3840 // w) * (2176 // h)
    _check(hip_ctx, oracle, b, 3840, 30, rng)
