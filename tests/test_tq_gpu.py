"""GPU parity of the batched transform / quantisation entries vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import svtav1_hip
from tq_util import (oracle_encode_batch, random_encode_batch, oracle_itxfm_batch, oracle_quant_batch, oracle_txfm_batch, random_itxfm_batch, random_quant_batch,
                     random_txfm_batch)

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


@pytest.mark.parametrize("n_tu", [1, 7, 500, 20000])
def test_quantize_batch_matches_oracle(hip_ctx, oracle, n_tu):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(n_tu)
    b = random_quant_batch(rng, n_tu)
    q_o, dq_o, eob_o = oracle_quant_batch(oracle, b)
    d_coeff, d_desc, d_qp, d_iscan = _dev(b["coeff"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_q = torch.full((b["coeff"].size,), 77, dtype=torch.int32, device="cuda:0")
    d_dq = torch.full((b["coeff"].size,), 77, dtype=torch.int32, device="cuda:0")
    d_eob = torch.zeros(n_tu, dtype=torch.int16, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.quantize_b_batch_dev(d_coeff.data_ptr(), d_desc.data_ptr(), n_tu, d_qp.data_ptr(), d_iscan.data_ptr(), d_q.data_ptr(),
                                 d_dq.data_ptr(), d_eob.data_ptr())
    hip_ctx.synchronize()
    assert np.array_equal(d_q.cpu().numpy(), q_o)
    assert np.array_equal(d_dq.cpu().numpy(), dq_o)
    assert np.array_equal(d_eob.cpu().numpy().view(np.uint16), eob_o)


@pytest.mark.parametrize("size", svtav1_hip.TX_SIZES_WH)
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_fwd_txfm2d_batch_matches_oracle(hip_ctx, oracle, size, bit_depth):
    """Every AV1 transform size, every transform type the reference defines for it, 8- and 10-bit residual ranges."""
    torch = pytest.importorskip("torch")
    w, h = size
    n_tu = 333 if w * h <= 1024 else 97       # not a multiple of the TUs-per-wave group: exercises the ragged tail
    rng = np.random.default_rng(w * 1000 + h * 10 + bit_depth)
    b = random_txfm_batch(rng, n_tu, w, h, bit_depth)
    ref = oracle_txfm_batch(oracle, b)
    d_res, d_desc = _dev(b["residual"]), _dev(b["desc"])
    d_out = torch.full((n_tu * w * h,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.fwd_txfm2d_batch_dev(d_res.data_ptr(), d_desc.data_ptr(), n_tu, w, h, bit_depth, d_out.data_ptr())
    hip_ctx.synchronize()
    got = d_out.cpu().numpy()
    bad = np.flatnonzero(got != ref)
    assert bad.size == 0, (size, bit_depth, bad[:8], got[bad[:8]], ref[bad[:8]], b["desc"][bad[0] // (w * h)])


def test_fwd_txfm2d_rejects_bad_arguments(hip_ctx):
    torch = pytest.importorskip("torch")
    buf = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fwd_txfm2d_batch_dev(buf.data_ptr(), buf.data_ptr(), 1, 4, 64, 8, buf.data_ptr())    # 16:1 is not an AV1 size
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fwd_txfm2d_batch_dev(buf.data_ptr(), buf.data_ptr(), 1, 12, 12, 8, buf.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fwd_txfm2d_batch_dev(buf.data_ptr(), buf.data_ptr(), 1, 8, 8, 12, buf.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.fwd_txfm2d_batch_dev(buf.data_ptr(), buf.data_ptr(), 1, 8, 8, 8, buf.data_ptr() + 4)  # misaligned output


@pytest.mark.parametrize("size", svtav1_hip.TX_SIZES_WH)
@pytest.mark.parametrize("mode", [(8, False), (8, True), (10, True)])
def test_inv_txfm2d_add_batch_matches_oracle(hip_ctx, oracle, size, mode):
    """Every size and defined type; 8-bit plane, widened 8-bit plane and 10-bit plane; inputs that hit every clamp."""
    torch = pytest.importorskip("torch")
    w, h = size
    bit_depth, recon_16bit = mode
    n_tu = 61 if w * h <= 1024 else 23
    rng = np.random.default_rng(w * 1000 + h * 10 + bit_depth + int(recon_16bit))
    b = random_itxfm_batch(rng, n_tu, w, h, bit_depth, recon_16bit)
    ref = oracle_itxfm_batch(oracle, b)
    d_coeff, d_desc, d_rec = _dev(b["coeff"]), _dev(b["desc"]), _dev(b["pred"])
    torch.cuda.synchronize()
    hip_ctx.inv_txfm2d_add_batch_dev(d_coeff.data_ptr(), d_desc.data_ptr(), n_tu, w, h, bit_depth, recon_16bit, d_rec.data_ptr())
    hip_ctx.synchronize()
    got = d_rec.cpu().numpy().view(b["pred"].dtype)
    bad = np.flatnonzero(got != ref)
    assert bad.size == 0, (size, mode, bad[:8], got[bad[:8]], ref[bad[:8]])


def test_fwd_inv_round_trip_is_near_identity(hip_ctx):
    """Size-independent property at frame scale: inverse(forward(residual)) added to a zero prediction returns the residual
    within the transform pair's rounding error (|err| <= 2 for DCT_DCT 8-bit), for every square size."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    pic_w, pic_h = 1920, 1024
    res = rng.integers(0, 128, (pic_h, pic_w)).astype(np.int16)          # non-negative so the zero prediction does not clip it
    d_res = _dev(res)
    for n in (4, 8, 16, 32):
        per_row = pic_w // n
        n_tu = per_row * (pic_h // n)
        idx = np.arange(n_tu)
        off = (idx // per_row) * n * pic_w + (idx % per_row) * n
        fd = np.zeros(n_tu, dtype=svtav1_hip.TXFM_DESC_DTYPE)
        fd["in_offset"], fd["out_offset"], fd["in_stride"] = off, idx * n * n, pic_w
        idd = np.zeros(n_tu, dtype=svtav1_hip.ITXFM_DESC_DTYPE)
        idd["coeff_offset"], idd["recon_offset"], idd["recon_stride"] = idx * n * n, off, pic_w
        d_fd, d_id = _dev(fd), _dev(idd)
        d_coeff = torch.empty(n_tu * n * n, dtype=torch.int32, device="cuda:0")
        d_rec = torch.zeros(pic_w * pic_h, dtype=torch.uint8, device="cuda:0")
        hip_ctx.fwd_txfm2d_batch_dev(d_res.data_ptr(), d_fd.data_ptr(), n_tu, n, n, 8, d_coeff.data_ptr())
        hip_ctx.inv_txfm2d_add_batch_dev(d_coeff.data_ptr(), d_id.data_ptr(), n_tu, n, n, 8, False, d_rec.data_ptr())
        hip_ctx.synchronize()
        err = np.abs(d_rec.cpu().numpy().reshape(pic_h, pic_w).astype(np.int32) - res.astype(np.int32))
        assert err.max() <= 2, (n, int(err.max()))


@pytest.mark.parametrize("size", svtav1_hip.TX_SIZES_WH)
@pytest.mark.parametrize("in_place", [False, True])
def test_encode_tu_batch_matches_oracle(hip_ctx, oracle, size, in_place):
    """Fused residual -> transform -> quantise -> dequantise -> inverse -> reconstruct vs the chain of pinned oracle stages."""
    torch = pytest.importorskip("torch")
    w, h = size
    n_tu = 53 if w * h <= 1024 else 21
    rng = np.random.default_rng(w * 100 + h + int(in_place))
    b = random_encode_batch(rng, n_tu, w, h)
    ref = oracle_encode_batch(oracle, b)
    n = b["n"]
    d_src, d_pred, d_desc, d_qp, d_iscan = _dev(b["src"]), _dev(b["pred"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_recon = d_pred if in_place else torch.full_like(d_pred, 0x33)
    d_coeff = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_q = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_dq = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_eob = torch.full((n_tu,), -1, dtype=torch.int16, device="cuda:0")
    d_en = torch.full((n_tu,), -1, dtype=torch.int64, device="cuda:0")
    d_dist = torch.full((n_tu, 2), -1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.encode_tu_batch_dev(d_src.data_ptr(), d_pred.data_ptr(), d_recon.data_ptr(), d_desc.data_ptr(), n_tu, w, h, d_qp.data_ptr(),
                                d_iscan.data_ptr(), d_coeff.data_ptr(), d_q.data_ptr(), d_dq.data_ptr(), d_eob.data_ptr(),
                                d_en.data_ptr(), d_dist.data_ptr())
    hip_ctx.synchronize()
    assert np.array_equal(d_coeff.cpu().numpy(), ref["coeff"])
    assert np.array_equal(d_q.cpu().numpy(), ref["qcoeff"])
    assert np.array_equal(d_dq.cpu().numpy(), ref["dqcoeff"])
    assert np.array_equal(d_eob.cpu().numpy().view(np.uint16), ref["eob"])
    assert np.array_equal(d_en.cpu().numpy().view(np.uint64), ref["energy"])
    assert np.array_equal(d_dist.cpu().numpy().view(np.uint64), ref["dist"])
    got = d_recon.cpu().numpy()
    if in_place:
        assert np.array_equal(got, ref["recon"])
    else:   # only the TU areas are written
        mask = np.zeros(got.size, bool)
        for d in b["desc"]:
            for r in range(h):
                o = int(d["recon_offset"]) + r * int(d["recon_stride"]); mask[o:o + w] = True
        assert np.array_equal(got[mask], ref["recon"][mask]) and (got[~mask] == 0x33).all()
    assert (ref["eob"] > 0).any() and (ref["eob"] == 0).any()      # both branches of the reference's "has coefficients" test


def test_encode_tu_optional_outputs_may_be_null(hip_ctx, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    b = random_encode_batch(rng, 40, 16, 16)
    ref = oracle_encode_batch(oracle, b)
    d_src, d_pred, d_desc, d_qp, d_iscan = _dev(b["src"]), _dev(b["pred"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_q = torch.zeros(40 * 256, dtype=torch.int32, device="cuda:0")
    d_eob = torch.zeros(40, dtype=torch.int16, device="cuda:0")
    hip_ctx.encode_tu_batch_dev(d_src.data_ptr(), d_pred.data_ptr(), d_pred.data_ptr(), d_desc.data_ptr(), 40, 16, 16, d_qp.data_ptr(),
                                d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None)
    hip_ctx.synchronize()
    assert np.array_equal(d_q.cpu().numpy(), ref["qcoeff"])
    assert np.array_equal(d_pred.cpu().numpy(), ref["recon"])


@pytest.mark.parametrize("size", [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (64, 32)])
def test_encode_tu16_batch_matches_oracle(hip_ctx, oracle, size):
    """Fused chain on 10-bit samples in 16-bit planes (high-bit-depth quantiser, bd = 10 inverse transform), in place."""
    torch = pytest.importorskip("torch")
    w, h = size
    n_tu = 53 if w * h <= 1024 else 21
    rng = np.random.default_rng(w * 100 + h + 7)
    b = random_encode_batch(rng, n_tu, w, h, bit_depth=10)
    ref = oracle_encode_batch(oracle, b)
    n = b["n"]
    d_src, d_pred, d_desc, d_qp, d_iscan = _dev(b["src"]), _dev(b["pred"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_coeff = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_q = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_dq = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_eob = torch.full((n_tu,), -1, dtype=torch.int16, device="cuda:0")
    d_en = torch.full((n_tu,), -1, dtype=torch.int64, device="cuda:0")
    d_dist = torch.full((n_tu, 2), -1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.encode_tu_batch_dev(d_src.data_ptr(), d_pred.data_ptr(), d_pred.data_ptr(), d_desc.data_ptr(), n_tu, w, h, d_qp.data_ptr(),
                                d_iscan.data_ptr(), d_coeff.data_ptr(), d_q.data_ptr(), d_dq.data_ptr(), d_eob.data_ptr(),
                                d_en.data_ptr(), d_dist.data_ptr(), planes_16bit=True)
    hip_ctx.synchronize()
    assert np.array_equal(d_coeff.cpu().numpy(), ref["coeff"])
    assert np.array_equal(d_q.cpu().numpy(), ref["qcoeff"])
    assert np.array_equal(d_dq.cpu().numpy(), ref["dqcoeff"])
    assert np.array_equal(d_eob.cpu().numpy().view(np.uint16), ref["eob"])
    assert np.array_equal(d_en.cpu().numpy().view(np.uint64), ref["energy"])
    assert np.array_equal(d_dist.cpu().numpy().view(np.uint64), ref["dist"])
    assert np.array_equal(d_pred.cpu().numpy().view(np.uint16), ref["recon"])


@pytest.mark.parametrize("case", [(4, 4, 8, 3001), (8, 8, 8, 1501), (16, 16, 8, 1203), (32, 32, 8, 331), (64, 64, 8, 37), (16, 8, 8, 403), (32, 64, 8, 45),
                                  (4, 4, 10, 1999), (16, 16, 10, 801), (32, 32, 10, 203), (64, 64, 10, 21)])
@pytest.mark.parametrize("max_wg", [1, 3])
def test_encode_tu_many_groups_per_wave(hip_ctx, oracle, case, max_wg):
    """The path a large batch takes -- few workgroups, every wave walking MANY groups of TUs with the next group's descriptor and rows
    prefetched and the reconstruction store deferred by one group -- on a small batch (SVTHIP_OPT_TQ_MAX_WORKGROUPS), in place, with a TU
    count that is not a multiple of the TUs per wave: quantised coefficients, end of block and reconstruction against the oracle."""
    torch = pytest.importorskip("torch")
    w, h, bd, n_tu = case
    rng = np.random.default_rng(w * 1000 + h * 10 + bd + max_wg)
    pic_w = 1024
    pic_h = ((n_tu + pic_w // w - 1) // (pic_w // w) + 2) * h
    b = random_encode_batch(rng, n_tu, w, h, pic_w=pic_w, pic_h=pic_h, bit_depth=bd)
    ref = oracle_encode_batch(oracle, b)
    n = b["n"]
    d_src, d_pred, d_desc, d_qp, d_iscan = _dev(b["src"]), _dev(b["pred"]), _dev(b["desc"]), _dev(b["qparams"]), _dev(b["iscan"])
    d_q = torch.full((n_tu * n,), 5, dtype=torch.int32, device="cuda:0")
    d_eob = torch.full((n_tu,), -1, dtype=torch.int16, device="cuda:0")
    hip_ctx.set_option(svtav1_hip.OPT_TQ_MAX_WORKGROUPS, max_wg)
    try:
        hip_ctx.encode_tu_batch_dev(d_src.data_ptr(), d_pred.data_ptr(), d_pred.data_ptr(), d_desc.data_ptr(), n_tu, w, h, d_qp.data_ptr(),
                                    d_iscan.data_ptr(), None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None, planes_16bit=(bd == 10))
        hip_ctx.synchronize()
    finally:
        hip_ctx.set_option(svtav1_hip.OPT_TQ_MAX_WORKGROUPS, 0)
    assert np.array_equal(d_q.cpu().numpy(), ref["qcoeff"])
    assert np.array_equal(d_eob.cpu().numpy().view(np.uint16), ref["eob"])
    got = d_pred.cpu().numpy()
    assert np.array_equal(got.view(np.uint16) if bd == 10 else got, ref["recon"])
