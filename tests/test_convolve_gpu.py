"""GPU parity: svthip_av1_convolve_sr_batch_dev (AV1 8-tap / 4-tap single-reference inter prediction, 8-bit) vs the oracle, which
tests/test_convolve_vs_ref.py pins against the reference's av1_convolve_*_sr_c.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import svtav1_hip

pytestmark = pytest.mark.gpu


def _oracle_batch(oracle, src, S, dst, D, desc, w, h):
    f = oracle.lib.orc_av1_convolve_sr_batch
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
    d = np.zeros((len(desc), 4), np.uint32)
    d[:, 0], d[:, 1] = desc["src_offset"], desc["dst_offset"]
    d[:, 2] = desc["subpel_x"].astype(np.uint32) | (desc["subpel_y"].astype(np.uint32) << 8) | (desc["filter_x"].astype(np.uint32) << 16) | \
        (desc["filter_y"].astype(np.uint32) << 24)
    f(src.ctypes.data, S, dst.ctypes.data, D, d.ctypes.data, len(desc), w, h)


def _run(hip_ctx, src, S, dst, D, desc, w, h):
    import torch
    d_src = torch.from_numpy(np.concatenate([src.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_dst = torch.from_numpy(dst.reshape(-1).copy()).to("cuda:0")
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    hip_ctx.av1_convolve_sr_batch_dev(d_src.data_ptr(), S, d_dst.data_ptr(), D, d_desc.data_ptr(), len(desc), w, h)
    hip_ctx.synchronize()
    return d_dst.cpu().numpy().reshape(dst.shape)


@pytest.mark.parametrize("size", svtav1_hip.AV1_BLOCK_SIZES_WH)
def test_convolve_every_block_size(hip_ctx, oracle, size):
    pytest.importorskip("torch")
    w, h = size
    rng = np.random.default_rng(w * 257 + h)
    S, R = 640, 400
    src = rng.integers(0, 256, (R, S), dtype=np.uint8)
    src[:40] = (((np.arange(S)[None, :] // 2 + np.arange(40)[:, None] // 3) & 1) * 255).astype(np.uint8)   # both clips
    cols, rows = 512 // w, 256 // h
    n = min(cols * rows, 300) - 1     # not a multiple of the blocks-per-workgroup group: ragged last workgroup
    slots = rng.permutation(cols * rows)[:n]
    D = 512 + 3                        # odd destination stride: byte-granular stores
    dst = np.full((256, D), 0x55, np.uint8)
    desc = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    for i, sl in enumerate(slots):
        bx, by = (int(sl) % cols) * w, (int(sl) // cols) * h
        mvx, mvy = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
        sx = min(max(40 + bx + mvx, 8), S - w - 8); sy = min(max(40 + by + mvy, 8), R - h - 8)
        desc[i] = (sy * S + sx, by * D + bx, int(rng.integers(0, 16)), int(rng.integers(0, 16)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), 0)
    desc["subpel_x"][:6] = [0, 0, 5, 0, 15, 8]; desc["subpel_y"][:6] = [0, 7, 0, 0, 15, 8]   # copy / y-only / x-only / 2-D all present
    want = dst.copy()
    _oracle_batch(oracle, src, S, want, D, desc, w, h)
    got = _run(hip_ctx, src, S, dst, D, desc, w, h)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (size, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
    if w % 32 == 0 and h % 32 == 0:   # these sizes run on the matrix cores; the VALU kernel must agree on them as well
        hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 1)
        try:
            got2 = _run(hip_ctx, src, S, dst, D, desc, w, h)
        finally:
            hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 0)
        assert np.array_equal(got2, want), size


def test_convolve_64x64_all_phases_1080p(hip_ctx, oracle):
    """64x64 blocks at all 15 x 15 fractional phases -- here all 16 x 16 (the zero phases take the x-only / y-only /
    copy functions) over one 1080p frame: 510 blocks x 256 phase pairs, one launch.  Oracle on a sample; properties over everything:
    phase (0,0) is a copy, bilinear at phase 8 in x only is the rounded average of horizontal neighbours."""
    torch = pytest.importorskip("torch")
    from svtav1_hip import synth
    pic = synth.PaPicture(synth.synth_luma(1920, 1088, 1))
    S = pic.stride
    src = pic.full
    nbx, nby = 30, 17
    n_ph = 256
    desc = np.zeros(nbx * nby * n_ph, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    i = np.arange(desc.size)
    blk, ph = i // n_ph, i % n_ph
    bx, by = (blk % nbx) * 64, (blk // nbx) * 64
    desc["src_offset"] = (68 + by) * S + 68 + bx
    D = 64 * n_ph                                   # destination: one row of 64 x 64 tiles per block, one tile per phase
    desc["dst_offset"] = blk * (64 * D) + ph * 64
    desc["subpel_x"], desc["subpel_y"] = ph % 16, ph // 16
    desc["filter_x"] = (blk + ph) % 4
    desc["filter_y"] = (blk // 3 + ph // 5) % 4
    d_src = torch.from_numpy(np.concatenate([src.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_dst = torch.zeros(nbx * nby * 64 * D, dtype=torch.uint8, device="cuda:0")
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    hip_ctx.av1_convolve_sr_batch_dev(d_src.data_ptr(), S, d_dst.data_ptr(), D, d_desc.data_ptr(), desc.size, 64, 64)
    hip_ctx.synchronize()
    got = d_dst.cpu().numpy().reshape(nbx * nby * 64, D)
    rng = np.random.default_rng(1)
    pick = rng.choice(desc.size, 400, replace=False)
    f = oracle.lib.orc_av1_convolve_sr
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]
    for k in pick:
        d = desc[k]
        tile = np.zeros((64, 64), np.uint8)
        f(src.ctypes.data + int(d["src_offset"]), S, tile.ctypes.data, 64, 64, 64, int(d["filter_x"]), int(d["filter_y"]), int(d["subpel_x"]),
          int(d["subpel_y"]))
        b, p = int(k) // n_ph, int(k) % n_ph
        assert np.array_equal(got[b * 64:(b + 1) * 64, p * 64:(p + 1) * 64], tile), (b, p)
    for b in range(0, nbx * nby, 7):
        x0, y0 = (b % nbx) * 64, (b // nbx) * 64
        blk_src = src[68 + y0:68 + y0 + 64, 68 + x0:68 + x0 + 64 + 1].astype(np.int32)
        assert np.array_equal(got[b * 64:(b + 1) * 64, 0:64], blk_src[:, :64])                 # phase (0,0): copy
        d8 = desc[b * n_ph + 8]
        if d8["filter_x"] == 3:                                                                  # bilinear, x phase 8, y phase 0
            avg = (((64 * blk_src[:, :64] + 64 * blk_src[:, 1:65] + 4) >> 3) + 8) >> 4
            assert np.array_equal(got[b * 64:(b + 1) * 64, 8 * 64:9 * 64], avg)


def test_convolve_rejects_bad_arguments(hip_ctx):
    torch = pytest.importorskip("torch")
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.av1_convolve_sr_batch_dev(buf.data_ptr(), 256, buf.data_ptr(), 256, buf.data_ptr(), 1, 12, 12)    # not a block size
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.av1_convolve_sr_batch_dev(buf.data_ptr(), 256, buf.data_ptr(), 256, buf.data_ptr(), 1, 4, 32)     # 8:1
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.av1_convolve_sr_batch_dev(buf.data_ptr(), 256, buf.data_ptr(), 256, buf.data_ptr() + 4, 1, 8, 8)  # misaligned descriptors


def _oracle_compound(oracle, s0, S0, s1, S1, dst, D, desc, w, h):
    f = oracle.lib.orc_av1_convolve_compound_batch
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
    d = np.zeros((len(desc), 4), np.uint32)
    d[:, 0], d[:, 1], d[:, 2] = desc["src0_offset"], desc["src1_offset"], desc["dst_offset"]
    d[:, 3] = desc["subpel0"].astype(np.uint32) | (desc["subpel1"].astype(np.uint32) << 8) | (desc["filter_x"].astype(np.uint32) << 16) | \
        (desc["filter_y"].astype(np.uint32) << 24)
    f(s0.ctypes.data, S0, s1.ctypes.data, S1, dst.ctypes.data, D, d.ctypes.data, len(desc), w, h)


def _run_compound(hip_ctx, s0, S0, s1, S1, dst, D, desc, w, h):
    import torch
    d0 = torch.from_numpy(np.concatenate([s0.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d1 = torch.from_numpy(np.concatenate([s1.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_dst = torch.from_numpy(dst.reshape(-1).copy()).to("cuda:0")
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    hip_ctx.av1_convolve_compound_batch_dev(d0.data_ptr(), S0, d1.data_ptr(), S1, d_dst.data_ptr(), D, d_desc.data_ptr(), len(desc), w, h)
    hip_ctx.synchronize()
    return d_dst.cpu().numpy().reshape(dst.shape)


@pytest.mark.parametrize("size", svtav1_hip.AV1_BLOCK_SIZES_WH)
def test_convolve_compound_every_block_size(hip_ctx, oracle, size):
    """BI_PRED prediction (av1_jnt_convolve_* pair) of every AV1 block size: random phases incl. all copy / x-only / y-only / 2-D
    combinations of the two lists, two planes with different strides, odd destination stride; both kernels where both exist."""
    pytest.importorskip("torch")
    w, h = size
    rng = np.random.default_rng(w * 263 + h)
    S0, S1, R = 640, 704, 400
    s0 = rng.integers(0, 256, (R, S0), dtype=np.uint8)
    s1 = rng.integers(0, 256, (R, S1), dtype=np.uint8)
    s0[:40] = (((np.arange(S0)[None, :] // 2 + np.arange(40)[:, None] // 3) & 1) * 255).astype(np.uint8)
    s1[:60, ::2] = 255; s1[:60, 1::2] = 0
    cols, rows = 512 // w, 256 // h
    n = min(cols * rows, 200) - 1
    slots = rng.permutation(cols * rows)[:n]
    D = 512 + 1
    dst = np.full((256, D), 0x55, np.uint8)
    desc = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
    for i, sl in enumerate(slots):
        bx, by = (int(sl) % cols) * w, (int(sl) // cols) * h
        off = []
        for S in (S0, S1):
            sx = min(max(40 + bx + int(rng.integers(-20, 21)), 8), S - w - 8); sy = min(max(40 + by + int(rng.integers(-20, 21)), 8), R - h - 8)
            off.append(sy * S + sx)
        ph = [int(v) for v in rng.integers(0, 16, 4)]
        if i < 16:
            a, b = i & 3, i >> 2
            ph = [(ph[0] or 5) if a & 1 else 0, (ph[1] or 9) if a & 2 else 0, (ph[2] or 3) if b & 1 else 0, (ph[3] or 12) if b & 2 else 0]
        desc[i] = (off[0], off[1], by * D + bx, ph[0] | (ph[1] << 4), ph[2] | (ph[3] << 4), int(rng.integers(0, 4)), int(rng.integers(0, 4)))
    want = dst.copy()
    _oracle_compound(oracle, s0, S0, s1, S1, want, D, desc, w, h)
    got = _run_compound(hip_ctx, s0, S0, s1, S1, dst, D, desc, w, h)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (size, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
    if w % 32 == 0 and h % 32 == 0:
        hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 1)
        try:
            got2 = _run_compound(hip_ctx, s0, S0, s1, S1, dst, D, desc, w, h)
        finally:
            hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 0)
        assert np.array_equal(got2, want), size


def test_convolve_1080p_all_blocks_two_kernels_and_oracle_sample(hip_ctx, oracle):
    """BASELINE configs[2] shape at full size: every 64x64 block of a 1080p frame at 24 phase / filter combinations (12 240 blocks), uni- and
    bi-predicted.  Size-independent checks: the matrix-core kernel and the VALU kernel (independent implementations) agree on every byte,
    phase (0, 0) returns the source block, and a 300-block sample equals the oracle."""
    torch = pytest.importorskip("torch")
    import os
    from svtav1_hip import synth
    w = h = 64
    pics = [synth.PaPicture(synth.synth_luma(1920, 1080, t)) for t in (1, 4)]
    S = pics[0].full.shape[1]
    nbx, nby, n_ph = 30, 16, 24          # the 17th SB row is 56 rows high: whole 64x64 blocks only
    n = nbx * nby * n_ph
    rng = np.random.default_rng(12)
    i = np.arange(n)
    blk, ph = i // n_ph, i % n_ph
    base = (68 + (blk // nbx) * 64) * S + 68 + (blk % nbx) * 64
    mvx, mvy = rng.integers(-30, 31, n), rng.integers(-30, 31, n)
    d = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    d["src_offset"] = base + mvy * S + mvx
    d["dst_offset"] = i * 4096
    d["subpel_x"], d["subpel_y"] = rng.integers(0, 16, n), rng.integers(0, 16, n)
    d["subpel_x"][ph == 0] = 0; d["subpel_y"][ph == 0] = 0
    d["filter_x"], d["filter_y"] = rng.integers(0, 4, n), rng.integers(0, 4, n)
    c = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
    c["src0_offset"], c["src1_offset"], c["dst_offset"] = d["src_offset"], base - mvy * S - mvx, d["dst_offset"]
    c["subpel0"] = d["subpel_x"] | (d["subpel_y"] << 4)
    c["subpel1"] = rng.integers(0, 256, n)
    c["filter_x"], c["filter_y"] = d["filter_x"], d["filter_y"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")  # noqa: E731
    pad = np.zeros(256, np.uint8)
    d_s0, d_s1 = dev(np.concatenate([pics[0].full.reshape(-1), pad])), dev(np.concatenate([pics[1].full.reshape(-1), pad]))
    d_d, d_c = dev(d), dev(c)

    def run(compound):
        out = torch.zeros(n * 4096, dtype=torch.uint8, device="cuda:0")
        if compound:
            hip_ctx.av1_convolve_compound_batch_dev(d_s0.data_ptr(), S, d_s1.data_ptr(), S, out.data_ptr(), 64, d_c.data_ptr(), n, w, h)
        else:
            hip_ctx.av1_convolve_sr_batch_dev(d_s0.data_ptr(), S, out.data_ptr(), 64, d_d.data_ptr(), n, w, h)
        hip_ctx.synchronize()
        return out.cpu().numpy().reshape(n, 64, 64)

    got = {k: run(k) for k in (False, True)}
    hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 1)
    try:
        for k in (False, True):
            assert np.array_equal(run(k), got[k]), "matrix-core and VALU kernels differ (compound=%s)" % k
    finally:
        hip_ctx.set_option(svtav1_hip.OPT_CONVOLVE_VALU, 0)
    flat = pics[0].full.reshape(-1)
    for j in np.nonzero(ph == 0)[0][:200]:
        o = int(d["src_offset"][j])
        assert np.array_equal(got[False][j], flat[o:o + 64 * S].reshape(64, S)[:, :64])
    sample = rng.choice(n, 300, replace=False)
    want = np.zeros((300, 64, 64), np.uint8)
    ds = d[sample].copy(); ds["dst_offset"] = np.arange(300) * 4096
    _oracle_batch(oracle, pics[0].full, S, want, 64, ds, w, h)
    assert np.array_equal(got[False][sample], want)
    cs = c[sample].copy(); cs["dst_offset"] = np.arange(300) * 4096
    _oracle_compound(oracle, pics[0].full, S, pics[1].full, S, want, 64, cs, w, h)
    assert np.array_equal(got[True][sample], want)


@pytest.mark.parametrize("size", svtav1_hip.AV1_BLOCK_SIZES_WH)
def test_highbd_convolve_every_block_size(hip_ctx, oracle, size):
    """10-bit inter prediction in 16-bit planes (av1_highbd_convolve_*_sr / av1_highbd_jnt_convolve_* pair), uni- and bi-predicted, every AV1
    block size, all case combinations, random and extreme (0 / 1023) samples, odd sample offsets and strides."""
    torch = pytest.importorskip("torch")
    w, h = size
    rng = np.random.default_rng(w * 269 + h)
    S0, S1, R = 641, 705, 400
    s0 = rng.integers(0, 1024, (R, S0), dtype=np.uint16)
    s1 = rng.integers(0, 1024, (R, S1), dtype=np.uint16)
    s0[:40] = (((np.arange(S0)[None, :] // 2 + np.arange(40)[:, None] // 3) & 1) * 1023).astype(np.uint16)
    s1[:60, ::2] = 1023; s1[:60, 1::2] = 0
    cols, rows = 512 // w, 256 // h
    n = min(cols * rows, 120) - 1
    slots = rng.permutation(cols * rows)[:n]
    D = 512 + 1
    cd = np.zeros(n, dtype=svtav1_hip.CONVOLVE_COMPOUND_DESC_DTYPE)
    ud = np.zeros(n, dtype=svtav1_hip.CONVOLVE_DESC_DTYPE)
    for i, sl in enumerate(slots):
        bx, by = (int(sl) % cols) * w, (int(sl) // cols) * h
        off = []
        for S in (S0, S1):
            sx = min(max(40 + bx + int(rng.integers(-20, 21)), 8), S - w - 8); sy = min(max(40 + by + int(rng.integers(-20, 21)), 8), R - h - 8)
            off.append(sy * S + sx)
        ph = [int(v) for v in rng.integers(0, 16, 4)]
        if i < 16:
            a, b = i & 3, i >> 2
            ph = [(ph[0] or 5) if a & 1 else 0, (ph[1] or 9) if a & 2 else 0, (ph[2] or 3) if b & 1 else 0, (ph[3] or 12) if b & 2 else 0]
        fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        cd[i] = (off[0], off[1], by * D + bx, ph[0] | (ph[1] << 4), ph[2] | (ph[3] << 4), fx, fy)
        ud[i] = (off[0], by * D + bx, ph[0], ph[1], fx, fy, 0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")  # noqa: E731
    pad = np.zeros(64, np.uint16)
    d0, d1 = dev(np.concatenate([s0.reshape(-1), pad])), dev(np.concatenate([s1.reshape(-1), pad]))
    A = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int]
    B = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int]
    fu, fc = oracle.lib.orc_av1_highbd_convolve_sr_batch, oracle.lib.orc_av1_highbd_convolve_compound_batch
    fu.restype = None; fu.argtypes = A
    fc.restype = None; fc.argtypes = B
    for compound in (False, True):
        want = np.full((256, D), 0x155, np.uint16)
        d_dst = dev(want)
        if compound:
            dd = np.zeros((n, 4), np.uint32)
            dd[:, 0], dd[:, 1], dd[:, 2] = cd["src0_offset"], cd["src1_offset"], cd["dst_offset"]
            dd[:, 3] = cd["subpel0"].astype(np.uint32) | (cd["subpel1"].astype(np.uint32) << 8) | (cd["filter_x"].astype(np.uint32) << 16) | \
                (cd["filter_y"].astype(np.uint32) << 24)
            fc(s0.ctypes.data, S0, s1.ctypes.data, S1, want.ctypes.data, D, dd.ctypes.data, n, w, h, 10)
            d_desc = dev(cd)
        else:
            dd = np.zeros((n, 4), np.uint32)
            dd[:, 0], dd[:, 1] = ud["src_offset"], ud["dst_offset"]
            dd[:, 2] = ud["subpel_x"].astype(np.uint32) | (ud["subpel_y"].astype(np.uint32) << 8) | (ud["filter_x"].astype(np.uint32) << 16) | \
                (ud["filter_y"].astype(np.uint32) << 24)
            fu(s0.ctypes.data, S0, want.ctypes.data, D, dd.ctypes.data, n, w, h, 10)
            d_desc = dev(ud)
        hip_ctx.av1_highbd_convolve_batch_dev(d0.data_ptr(), S0, d1.data_ptr(), S1, d_dst.data_ptr(), D, d_desc.data_ptr(), compound, n, w, h, 10)
        hip_ctx.synchronize()
        got = d_dst.cpu().numpy().view(np.uint16).reshape(256, D)
        bad = np.argwhere(got != want)
        assert bad.size == 0, (size, compound, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])
