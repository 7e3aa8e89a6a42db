"""GPU parity: picture-analysis producers on the device (svthip_pa_derive_planes_dev, svthip_pad_plane_dev) through the C ABI vs
the oracle (itself pinned against the reference's generate_padding / Decimation2D).  Bit-exact."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu


def _pool_with_interiors_only(pics):
    """A pool laid out like build_picture_pool, but holding ONLY the picture interiors of the full-resolution planes (what a
    host uploads once); borders and decimated planes are garbage the kernel must overwrite."""
    pool, descs = svtav1_hip.build_picture_pool(pics)
    raw = np.full_like(pool, 0xEE)
    for p, d in zip(pics, descs):
        fs = d.full_stride
        plane = raw[d.full_offset:d.full_offset + fs * (p.height + 136)].reshape(p.height + 136, fs)
        plane[68:68 + p.height, 68:68 + p.width] = p.full[68:68 + p.height, 68:68 + p.width]
    return raw, pool, descs


@pytest.mark.parametrize("size", [(64, 64), (328, 200), (856, 480), (1920, 1080), (3840, 2160)])
def test_pa_derive_planes_matches_oracle(hip_ctx, oracle, size):
    torch = pytest.importorskip("torch")
    w, h = size
    lumas = [synth.synth_luma(w, h, t) for t in (0, 3)]
    pics = [synth.PaPicture(x) for x in lumas]
    raw, want_pool, descs = _pool_with_interiors_only(pics)
    for x, p in zip(lumas, pics):  # the numpy planes ARE the oracle's planes (tests/test_pa_vs_ref.py pins both to the reference)
        f, q, s = oracle.pa_derive_planes(x)
        assert np.array_equal(f, p.full) and np.array_equal(q, p.quarter) and np.array_equal(s, p.sixteenth)
    d_pool = torch.from_numpy(raw).to("cuda:0")
    hip_ctx.pa_derive_planes_dev(d_pool.data_ptr(), descs)
    hip_ctx.synchronize()
    got = d_pool.cpu().numpy()
    for p, d in zip(pics, descs):
        for name, arr in (("full", p.full), ("quarter", p.quarter), ("sixteenth", p.sixteenth)):
            off = getattr(d, name + "_offset")
            g = got[off:off + arr.size].reshape(arr.shape)
            assert np.array_equal(g, arr), f"{name} plane of a {w}x{h} picture"
    # nothing outside the planes was touched (inter-plane alignment gaps and the tail keep their fill)
    mask = np.ones(got.size, bool)
    for p, d in zip(pics, descs):
        for name, arr in (("full", p.full), ("quarter", p.quarter), ("sixteenth", p.sixteenth)):
            off = getattr(d, name + "_offset")
            mask[off:off + arr.size] = False
    assert (got[mask] == 0xEE).all()


def test_pa_derive_planes_respects_level_flags(hip_ctx):
    torch = pytest.importorskip("torch")
    pics = [synth.PaPicture(synth.synth_luma(128, 64, 1))]
    raw, want, descs = _pool_with_interiors_only(pics)
    d_pool = torch.from_numpy(raw).to("cuda:0")
    hip_ctx.pa_derive_planes_dev(d_pool.data_ptr(), descs, want_quarter=False, want_sixteenth=True)
    hip_ctx.synchronize()
    got = d_pool.cpu().numpy()
    d = descs[0]
    q = got[d.quarter_offset:d.quarter_offset + pics[0].quarter.size]
    assert (q == 0xEE).all()
    s = got[d.sixteenth_offset:d.sixteenth_offset + pics[0].sixteenth.size].reshape(pics[0].sixteenth.shape)
    assert np.array_equal(s, pics[0].sixteenth)


@pytest.mark.parametrize("case", [(1920, 1080, 160, 160, 1), (960, 540, 80, 80, 1), (3840, 2160, 160, 160, 2), (1920, 1080, 160, 160, 2),
                                  (72, 40, 9, 5, 1), (8, 8, 96, 96, 2)])
def test_pad_plane_matches_oracle(hip_ctx, oracle, case):
    """generate_padding / generate_padding16_bit as PadRefAndSetFlags applies them to a reconstructed picture (luma origin 160,
    chroma 80)."""
    torch = pytest.importorskip("torch")
    w, h, pw, ph, sb = case
    rng = np.random.default_rng(w + h + sb)
    stride = w + 2 * pw + (4 if w < 100 else 0)
    dt = np.uint8 if sb == 1 else np.uint16
    a = rng.integers(0, 256 if sb == 1 else 1024, (h + 2 * ph, stride), dtype=dt)
    want = a.copy()
    oracle.generate_padding(want, w, h, pw, ph)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    hip_ctx.pad_plane_dev(t.data_ptr(), stride, w, h, pw, ph, sb)
    hip_ctx.synchronize()
    got = t.cpu().numpy().view(dt).reshape(a.shape)
    assert np.array_equal(got, want)


def test_pad_plane_rejects_bad_geometry(hip_ctx):
    torch = pytest.importorskip("torch")
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.pad_plane_dev(t.data_ptr(), 16, 32, 8, 4, 4, 1)   # stride smaller than the padded width
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.pad_plane_dev(t.data_ptr(), 64, 32, 8, 4, 4, 3)   # sample size
