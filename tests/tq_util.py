"""Shared helpers for the transform / quantisation tests."""
import ctypes as C

import numpy as np

import svtav1_hip


def make_qparams(q_dc, q_ac):
    """One Quants/Dequants row the way av1_build_quantizer builds it (Codec/EbModeDecisionConfigurationProcess.c:417-506):
    invert_quant -> quant/quant_shift, zbin = ROUND(84*q, 7), round = 64*q >> 7, dequant = q."""
    out = np.zeros(10, np.int16)
    for i, q in enumerate((q_dc, q_ac)):
        l = int(q).bit_length() - 1
        m = 1 + (1 << (16 + l)) // q
        out[4 + i] = np.int16(np.uint16((m - (1 << 16)) & 0xffff))
        out[6 + i] = 1 << (16 - l)
        out[0 + i] = (84 * q + 64) >> 7
        out[2 + i] = (64 * q) >> 7
        out[8 + i] = q
    return out


def random_quant_batch(rng, n_tu, sizes=(16, 64, 256, 1024)):
    """Random TU batch: returns dict with coeff pool, desc, qparams table, scan/iscan pools."""
    n_rows = 12
    qparams = np.stack([make_qparams(int(rng.integers(4, 1337)), int(rng.integers(4, 1829))) for _ in range(n_rows)])
    scans, iscans, scan_off = [], [], {}
    off = 0
    for n in sizes:
        for v in range(3):
            sc = rng.permutation(n).astype(np.int16)
            isc = np.zeros(n, np.int16); isc[sc] = np.arange(n, dtype=np.int16)
            scan_off[(n, v)] = off
            scans.append(sc); iscans.append(isc)
            off += n
    scan_pool = np.concatenate(scans); iscan_pool = np.concatenate(iscans)
    desc = np.zeros(n_tu, dtype=svtav1_hip.QUANT_DESC_DTYPE)
    chunks = []
    coff = 0
    for i in range(n_tu):
        n = int(rng.choice(sizes))
        row = int(rng.integers(0, n_rows))
        kind = i % 4
        q_ac = int(qparams[row, 9])
        if kind == 0:
            c = rng.integers(-40, 41, n)
        elif kind == 1:
            c = rng.laplace(0, 4 * q_ac, n).astype(np.int64)
        elif kind == 2:
            c = rng.integers(-(1 << 17), 1 << 17, n)
        else:
            c = np.zeros(n, np.int64); c[rng.integers(0, n, 3)] = rng.integers(-3000, 3000, 3)
        chunks.append(c.astype(np.int32))
        desc[i] = (coff, scan_off[(n, int(rng.integers(0, 3)))], row, n, int(rng.integers(0, 3)), int(rng.integers(0, 2)))
        coff += n
    return {"coeff": np.concatenate(chunks), "desc": desc, "qparams": qparams, "scan": scan_pool, "iscan": iscan_pool}


def oracle_quant_batch(oracle, b):
    orc = oracle.lib.orc_quantize_b
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    coeff = b["coeff"]
    q = np.zeros_like(coeff); dq = np.zeros_like(coeff); eob = np.zeros(len(b["desc"]), np.uint16)
    for i, d in enumerate(b["desc"]):
        co, so, n = int(d["coeff_offset"]), int(d["iscan_offset"]), int(d["n_coeffs"])
        e = C.c_uint16(0)
        qp = np.ascontiguousarray(b["qparams"][int(d["qparam_index"])])
        c = np.ascontiguousarray(coeff[co:co + n]); sc = np.ascontiguousarray(b["scan"][so:so + n])
        qq = np.zeros(n, np.int32); dd = np.zeros(n, np.int32)
        orc(c.ctypes.data, n, qp.ctypes.data, sc.ctypes.data, int(d["log_scale"]), int(d["highbd"]), qq.ctypes.data, dd.ctypes.data,
            C.addressof(e))
        q[co:co + n] = qq; dq[co:co + n] = dd; eob[i] = e.value
    return q, dq, eob


def random_txfm_batch(rng, n_tu, w, h, bit_depth=8, pic_w=256, pic_h=128):
    """TUs of one size scattered over a residual picture (stride = picture width, like the encoder's residual buffers)."""
    lim = (1 << bit_depth) - 1
    yy, xx = np.mgrid[0:pic_h, 0:pic_w]
    smooth = 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
    res = np.clip(smooth + rng.normal(0, 12, (pic_h, pic_w)), -lim, lim).astype(np.int16)
    res[: pic_h // 4] = rng.integers(-lim, lim + 1, (pic_h // 4, pic_w))     # full-range noise band
    res[pic_h // 4: pic_h // 4 + 8] = rng.choice([-lim, lim], (8, pic_w))    # worst-case magnitudes
    res[-8:] = 0                                                               # all-zero TUs
    types = svtav1_hip.valid_tx_types(w, h)
    desc = np.zeros(n_tu, dtype=svtav1_hip.TXFM_DESC_DTYPE)
    for i in range(n_tu):
        x0 = int(rng.integers(0, (pic_w - w) // 4 + 1)) * 4
        y0 = int(rng.integers(0, pic_h - h + 1))
        desc[i] = (y0 * pic_w + x0, i * w * h, pic_w, types[int(rng.integers(0, len(types)))], 0)
    return {"residual": res.reshape(-1), "desc": desc, "w": w, "h": h, "bit_depth": bit_depth}


def oracle_txfm_batch(oracle, b):
    orc = oracle.lib.orc_fwd_txfm2d
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    w, h = b["w"], b["h"]
    res = b["residual"]
    out = np.zeros(len(b["desc"]) * w * h, np.int32)
    for d in b["desc"]:
        o = int(d["out_offset"])
        orc(res.ctypes.data + 2 * int(d["in_offset"]), int(d["in_stride"]), w, h, int(d["tx_type"]), out.ctypes.data + 4 * o)
    return out


def random_itxfm_batch(rng, n_tu, w, h, bit_depth=8, recon_16bit=False, pic_w=512, pic_h=256):
    """Non-overlapping TUs on a prediction plane; dequantised-coefficient-like inputs incl. out-of-range ones (clamps)."""
    win, hin = min(w, 32), min(h, 32)
    per_row = pic_w // w
    assert n_tu <= per_row * (pic_h // h)
    slots = rng.permutation(per_row * (pic_h // h))[:n_tu]
    types = svtav1_hip.valid_tx_types(w, h)
    desc = np.zeros(n_tu, dtype=svtav1_hip.ITXFM_DESC_DTYPE)
    coeff = np.zeros(n_tu * win * hin, np.int32)
    for i in range(n_tu):
        sx, sy = int(slots[i]) % per_row, int(slots[i]) // per_row
        desc[i] = (i * win * hin, sy * h * pic_w + sx * w, pic_w, types[int(rng.integers(0, len(types)))], 0)
        kind = i % 6
        n = win * hin
        if kind == 0:
            c = rng.laplace(0, 40 << (bit_depth - 8), n).astype(np.int64)
        elif kind == 1:
            c = np.zeros(n, np.int64); k = rng.integers(0, n, 4); c[k] = rng.integers(-(1 << (bit_depth + 6)), 1 << (bit_depth + 6), 4)
        elif kind == 2:
            c = rng.integers(-(1 << (bit_depth + 7)), 1 << (bit_depth + 7), n)
        elif kind == 3:
            c = rng.integers(-(1 << 20), 1 << 20, n)
        elif kind == 4:
            c = rng.choice([-(1 << (bit_depth + 7)), (1 << (bit_depth + 7)) - 1], n)
        else:
            c = np.zeros(n, np.int64)
        coeff[i * n:(i + 1) * n] = c
    pred = rng.integers(0, 1 << bit_depth, pic_w * pic_h).astype(np.uint16 if recon_16bit else np.uint8)
    return {"coeff": coeff, "desc": desc, "pred": pred, "w": w, "h": h, "bit_depth": bit_depth, "recon_16bit": recon_16bit}


def oracle_itxfm_batch(oracle, b):
    orc = oracle.lib.orc_inv_txfm2d_add
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    rec = b["pred"].astype(np.uint16)         # av1_inv_txfm_add_c widens the 8-bit plane the same way (EbTransforms.c:8321-8340)
    for d in b["desc"]:
        orc(b["coeff"].ctypes.data + 4 * int(d["coeff_offset"]), rec.ctypes.data + 2 * int(d["recon_offset"]), int(d["recon_stride"]),
            b["w"], b["h"], int(d["tx_type"]), b["bit_depth"])
    return rec.astype(b["pred"].dtype)


def random_encode_batch(rng, n_tu, w, h, pic_w=512, pic_h=256, bit_depth=8):
    """Source / prediction planes and TU descriptors for the fused encode chain; TUs do not overlap.  bit_depth 10 gives
    uint16 planes with 10-bit samples."""
    win, hin = min(w, 32), min(h, 32)
    n = win * hin
    yy, xx = np.mgrid[0:pic_h, 0:pic_w]
    base = 128 + 70 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    sc = 1 << (bit_depth - 8)
    dt = np.uint8 if bit_depth == 8 else np.uint16
    src = np.clip(sc * (base + rng.normal(0, 6, (pic_h, pic_w))), 0, 256 * sc - 1).astype(dt)
    pred = np.clip(sc * (base + rng.normal(0, 10, (pic_h, pic_w)) + 8 * np.sin(xx / 3.0)), 0, 256 * sc - 1).astype(dt)
    pred[: pic_h // 8] = rng.integers(0, 256 * sc, (pic_h // 8, pic_w))       # large residuals
    pred[-pic_h // 4:] = src[-pic_h // 4:]                                     # zero residuals -> eob 0
    per_row = pic_w // w
    slots = rng.permutation(per_row * (pic_h // h))[:n_tu]
    n_rows = 8
    qparams = np.stack([make_qparams(int(rng.integers(4, 400)) * sc, int(rng.integers(4, 500)) * sc) for _ in range(n_rows)])
    scans, iscans = [], []
    for v in range(3):
        sc = rng.permutation(n).astype(np.int16)
        isc = np.zeros(n, np.int16); isc[sc] = np.arange(n, dtype=np.int16)
        scans.append(sc); iscans.append(isc)
    types = svtav1_hip.valid_tx_types(w, h)
    desc = np.zeros(n_tu, dtype=svtav1_hip.TU_DESC_DTYPE)
    for i in range(n_tu):
        sx, sy = int(slots[i]) % per_row, int(slots[i]) // per_row
        off = sy * h * pic_w + sx * w
        desc[i]["src_offset"] = off; desc[i]["pred_offset"] = off; desc[i]["recon_offset"] = off
        desc[i]["coeff_offset"] = i * n; desc[i]["iscan_offset"] = int(rng.integers(0, 3)) * n
        desc[i]["src_stride"] = pic_w; desc[i]["pred_stride"] = pic_w; desc[i]["recon_stride"] = pic_w
        desc[i]["qparam_index"] = int(rng.integers(0, n_rows)); desc[i]["tx_type"] = types[int(rng.integers(0, len(types)))]
    return {"src": src.reshape(-1), "pred": pred.reshape(-1), "desc": desc, "qparams": qparams, "scan": np.concatenate(scans),
            "iscan": np.concatenate(iscans), "w": w, "h": h, "n": n, "bit_depth": bit_depth}


def oracle_encode_batch(oracle, b):
    wide = b.get("bit_depth", 8) != 8
    es = 2 if wide else 1          # bytes per sample
    orc = oracle.lib.orc_encode_tu16 if wide else oracle.lib.orc_encode_tu
    orc.restype = None
    orc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8
    n_tu, n = len(b["desc"]), b["n"]
    out = {"recon": b["pred"].copy(), "coeff": np.zeros(n_tu * n, np.int32), "qcoeff": np.zeros(n_tu * n, np.int32),
           "dqcoeff": np.zeros(n_tu * n, np.int32), "eob": np.zeros(n_tu, np.uint16), "energy": np.zeros(n_tu, np.uint64),
           "dist": np.zeros((n_tu, 2), np.uint64)}
    for i, d in enumerate(b["desc"]):
        co = int(d["coeff_offset"])
        qp = np.ascontiguousarray(b["qparams"][int(d["qparam_index"])])
        sc = np.ascontiguousarray(b["scan"][int(d["iscan_offset"]):int(d["iscan_offset"]) + n])
        orc(b["src"].ctypes.data + es * int(d["src_offset"]), int(d["src_stride"]), b["pred"].ctypes.data + es * int(d["pred_offset"]),
            int(d["pred_stride"]), out["recon"].ctypes.data + es * int(d["recon_offset"]), int(d["recon_stride"]), b["w"], b["h"],
            int(d["tx_type"]), qp.ctypes.data, sc.ctypes.data, out["coeff"].ctypes.data + 4 * co, out["qcoeff"].ctypes.data + 4 * co,
            out["dqcoeff"].ctypes.data + 4 * co, out["eob"].ctypes.data + 2 * i, out["energy"].ctypes.data + 8 * i,
            out["dist"].ctypes.data + 16 * i)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the reference's REAL quantiser rows and scan orders (tests/golden/quant_tables.npz, generated by tests/golden/make_golden.py from
# av1_build_quantizer / av1_scan_orders)
# ---------------------------------------------------------------------------------------------------------------------
import os  # noqa: E402

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quant_tables.npz")


class RealTables:
    def __init__(self):
        z = np.load(_GOLDEN)
        self.z = z
        self.scan_index, self.scan_offsets = z["scan_index"], z["scan_offsets"]
        self.scan_pool, self.iscan_pool = z["scan_pool"], z["iscan_pool"]

    def rows(self, bit_depth=8, kind="inter"):
        """[256][3][10] int16: zbin[2], round[2], quant[2], quant_shift[2], dequant[2] per (qindex, plane)."""
        return self.z[f"rows_bd{bit_depth}_{kind}"]

    def scan_offset(self, tx_size_index, tx_type):
        """element offset of the (scan, iscan) pair of av1_scan_orders[tx_size][tx_type] in scan_pool / iscan_pool"""
        k = int(self.scan_index[tx_size_index, tx_type])
        assert k >= 0
        return int(self.scan_offsets[k])


def frame_encode_batch(rng, w, h, pic_w, pic_h, tables, qindices=(20, 120, 200), tx_types=None, residual="laplace", bit_depth=8,
                       kind="inter"):
    """One picture tiled completely by w x h TUs (1920x1080 -> 129 600 4x4 ... 510 64x64), prediction =
    source + residual, real quantiser rows (luma plane of `qindices`) and the real scan order of every TU's transform type."""
    ts = svtav1_hip.TX_SIZES_WH.index((w, h))
    win, hin = min(w, 32), min(h, 32)
    n = win * hin
    cols, rows_ = pic_w // w, pic_h // h
    n_tu = cols * rows_
    sc = 1 << (bit_depth - 8)
    dt = np.uint8 if bit_depth == 8 else np.uint16
    yy, xx = np.mgrid[0:pic_h, 0:pic_w]
    src = np.clip(sc * (128 + 70 * np.sin(xx / 9.0) * np.cos(yy / 7.0)), 0, 256 * sc - 1)
    if residual == "laplace":
        r = rng.laplace(0, 6 * sc, (pic_h, pic_w))
    else:
        r = rng.integers(-255 * sc, 255 * sc + 1, (pic_h, pic_w))
    pred = np.clip(src - r, 0, 256 * sc - 1).astype(dt)
    src = src.astype(dt)
    if tx_types is None:
        tx_types = [t for t in (0, 3, 9) if t in svtav1_hip.valid_tx_types(w, h)]  # DCT_DCT, ADST_ADST, IDTX
    qrows = np.ascontiguousarray(tables.rows(bit_depth, kind)[list(qindices), 0, :])   # luma rows
    idx = np.arange(n_tu)
    off = (idx // cols) * h * pic_w + (idx % cols) * w
    desc = np.zeros(n_tu, dtype=svtav1_hip.TU_DESC_DTYPE)
    desc["src_offset"] = desc["pred_offset"] = desc["recon_offset"] = off
    desc["coeff_offset"] = idx * n
    tt = np.asarray(tx_types)[rng.integers(0, len(tx_types), n_tu)]
    desc["tx_type"] = tt
    desc["iscan_offset"] = np.asarray([tables.scan_offset(ts, int(t)) for t in tx_types])[np.searchsorted(np.asarray(tx_types), tt)] \
        if sorted(tx_types) == list(tx_types) else [tables.scan_offset(ts, int(t)) for t in tt]
    desc["src_stride"] = desc["pred_stride"] = desc["recon_stride"] = pic_w
    desc["qparam_index"] = rng.integers(0, len(qindices), n_tu)
    return {"src": src.reshape(-1), "pred": pred.reshape(-1), "desc": desc, "qparams": qrows, "scan": tables.scan_pool,
            "iscan": tables.iscan_pool, "w": w, "h": h, "n": n, "bit_depth": bit_depth}
