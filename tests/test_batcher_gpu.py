"""the host-side batching layer (svthip_tu_batcher) used the way ProductFullLoopTxSearch would use it
(Codec/EbFullLoop.c:1138-1352): every transform-type candidate of every TU is ADDED (private scratch reconstruction), one FLUSH, and
the caller picks per TU the candidate with the smallest distortion -- the same winner, eob and levels as evaluating the candidates one
by one with the oracle chain.  Then the winners are encoded in place (encode-pass shape, Codec/EbCodingLoop.c:552-760) through the
same batcher with mixed sizes in arbitrary order."""
import numpy as np
import pytest

import svtav1_hip
from tq_util import RealTables, oracle_encode_batch

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def test_tx_type_search_through_the_batcher(hip_ctx, oracle):
    torch = pytest.importorskip("torch")
    tables = RealTables()
    rng = np.random.default_rng(12)
    pic_w, pic_h = 256, 128
    yy, xx = np.mgrid[0:pic_h, 0:pic_w]
    src = np.clip(128 + 70 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 6, (pic_h, pic_w)), 0, 255).astype(np.uint8)
    pred = np.clip(src.astype(np.int32) + rng.laplace(0, 7, (pic_h, pic_w)).astype(np.int32), 0, 255).astype(np.uint8)
    qrows = np.ascontiguousarray(tables.rows(8, "inter")[[60, 140], 0, :])
    # a partition of the picture into TUs of mixed sizes (32x32, 16x16, 8x8, 16x8 / 8x16 rectangles), visited in shuffled order
    tus = []
    for by in range(0, pic_h, 32):
        for bx in range(0, pic_w, 32):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                tus.append((3, bx, by))
            elif kind == 1:
                tus += [(2, bx + dx, by + dy) for dy in (0, 16) for dx in (0, 16)]
            elif kind == 2:
                tus += [(1, bx + dx, by + dy) for dy in range(0, 32, 8) for dx in range(0, 32, 8)]
            else:
                tus += [(8, bx + dx, by + dy) for dy in range(0, 32, 8) for dx in (0, 16)][:4] + [(7, bx + dx, by + 16 * (dy // 16)) for dy in (16,) for dx in range(0, 32, 8)]
    order = rng.permutation(len(tus))
    d_src, d_pred = _dev(src), _dev(pred)
    d_qp, d_iscan = _dev(qrows), _dev(tables.iscan_pool)
    bat = svtav1_hip.TuBatcher(hip_ctx, 8192, 1 << 20)
    bat.begin(d_src.data_ptr(), d_pred.data_ptr(), None, False, d_qp.data_ptr(), d_iscan.data_ptr())
    handles = {}
    for ti in order:
        ts, bx, by = tus[ti]
        w, h = svtav1_hip.TX_SIZES_WH[ts]
        off = by * pic_w + bx
        qi = (bx // 32 + by // 32) & 1
        for tt in svtav1_hip.valid_tx_types(w, h):
            handles[(ti, tt)] = bat.add(ts, tt, off, pic_w, off, pic_w, svtav1_hip.TU_RECON_SCRATCH, 0, qi, tables.scan_offset(ts, tt))
    bat.flush()
    # the serial loop of the reference, through the oracle: one candidate at a time
    n_checked = 0
    winners = {}
    for ti in range(len(tus)):
        ts, bx, by = tus[ti]
        w, h = svtav1_hip.TX_SIZES_WH[ts]
        n = min(w, 32) * min(h, 32)
        off = by * pic_w + bx
        qi = (bx // 32 + by // 32) & 1
        best = None
        for tt in svtav1_hip.valid_tx_types(w, h):
            d = np.zeros(1, dtype=svtav1_hip.TU_DESC_DTYPE)
            d["src_offset"] = d["pred_offset"] = d["recon_offset"] = off
            d["src_stride"] = d["pred_stride"] = d["recon_stride"] = pic_w
            d["iscan_offset"] = tables.scan_offset(ts, tt); d["qparam_index"] = qi; d["tx_type"] = tt
            want = oracle_encode_batch(oracle, {"src": src.reshape(-1), "pred": pred.reshape(-1), "desc": d, "qparams": qrows, "scan": tables.scan_pool,
                                                "w": w, "h": h, "n": n, "bit_depth": 8})
            r = bat.result(handles[(ti, tt)])
            assert (r.eob, r.three_quad_energy, r.distortion[0], r.distortion[1]) == (int(want["eob"][0]), int(want["energy"][0]),
                                                                                      int(want["dist"][0, 0]), int(want["dist"][0, 1])), (ti, tt)
            assert (r.tx_size, r.tx_type) == (ts, tt)
            if ti % 7 == 0:
                q, dq = bat.read_coeffs(handles[(ti, tt)], n)
                assert np.array_equal(q, want["qcoeff"]) and np.array_equal(dq, want["dqcoeff"])
                n_checked += 1
            cost = int(want["dist"][0, 0])
            if best is None or cost < best[0]:
                best = (cost, tt)
        winners[ti] = best[1]
    assert n_checked > 20
    # encode pass: the winners, reconstructed in place into the prediction plane, mixed sizes in one flush
    bat.begin(d_src.data_ptr(), d_pred.data_ptr(), d_pred.data_ptr(), False, d_qp.data_ptr(), d_iscan.data_ptr())
    hs = {}
    for ti in order[::-1]:
        ts, bx, by = tus[ti]
        off = by * pic_w + bx
        hs[ti] = bat.add(ts, winners[ti], off, pic_w, off, pic_w, off, pic_w, (bx // 32 + by // 32) & 1, tables.scan_offset(ts, winners[ti]))
    bat.flush()
    rec = d_pred.cpu().numpy().reshape(pic_h, pic_w)
    want_rec = pred.copy()
    for ti in range(len(tus)):
        ts, bx, by = tus[ti]
        w, h = svtav1_hip.TX_SIZES_WH[ts]
        d = np.zeros(1, dtype=svtav1_hip.TU_DESC_DTYPE)
        off = by * pic_w + bx
        d["src_offset"] = d["pred_offset"] = d["recon_offset"] = off
        d["src_stride"] = d["pred_stride"] = d["recon_stride"] = pic_w
        d["iscan_offset"] = tables.scan_offset(ts, winners[ti]); d["qparam_index"] = (bx // 32 + by // 32) & 1; d["tx_type"] = winners[ti]
        o = oracle_encode_batch(oracle, {"src": src.reshape(-1), "pred": pred.reshape(-1), "desc": d, "qparams": qrows, "scan": tables.scan_pool,
                                         "w": w, "h": h, "n": min(w, 32) * min(h, 32), "bit_depth": 8})
        want_rec[by:by + h, bx:bx + w] = o["recon"].reshape(pic_h, pic_w)[by:by + h, bx:bx + w]
        assert bat.result(hs[ti]).eob == int(o["eob"][0])
    assert np.array_equal(rec, want_rec)
    assert len(set(winners.values())) > 2    # the search really chose between transform types
    bat.close()


def test_batcher_refuses_overflow_and_bad_handles(hip_ctx):
    torch = pytest.importorskip("torch")
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
    bat = svtav1_hip.TuBatcher(hip_ctx, 4, 4096)
    with pytest.raises(svtav1_hip.SvtHipError):
        bat.add(0, 0, 0, 64, 0, 64, svtav1_hip.TU_RECON_SCRATCH, 0, 0, 0)        # not begun
    bat.begin(buf.data_ptr(), buf.data_ptr(), None, False, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError):
        bat.add(0, 0, 0, 64, 0, 64, 0, 64, 0, 0)                                   # in-place reconstruction without a plane
    with pytest.raises(svtav1_hip.SvtHipError):
        bat.add(19, 0, 0, 64, 0, 64, svtav1_hip.TU_RECON_SCRATCH, 0, 0, 0)       # no such TxSize
    for _ in range(4):
        bat.add(3, 0, 0, 64, 0, 64, svtav1_hip.TU_RECON_SCRATCH, 0, 0, 0)
    with pytest.raises(svtav1_hip.SvtHipError):
        bat.add(0, 0, 0, 64, 0, 64, svtav1_hip.TU_RECON_SCRATCH, 0, 0, 0)        # candidate capacity
    with pytest.raises(svtav1_hip.SvtHipError):
        bat.result(0)                                                             # not flushed yet
    bat.close()
