"""GPU parity: svthip_sad_loop_batch_dev (SadLoopKernel over a batch of blocks) vs the oracle's orc_sad_loop_kernel, which
tests/test_oracle_vs_ref.py pins against the reference's C and SSE4.1 SadLoopKernel.  Includes BASELINE configs[0] at full size:
every 16x16 block of an 856x480 picture, +-16 search (33x33 positions)."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu


def _run(hip_ctx, cur, ref, desc, w, h, sw, sh, k=1):
    import torch
    S = cur.shape[1]
    d_src = torch.from_numpy(np.concatenate([cur.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_ref = torch.from_numpy(np.concatenate([ref.reshape(-1), np.zeros(64, np.uint8)])).to("cuda:0")
    d_desc = torch.from_numpy(desc.astype(np.uint32).view(np.int32).reshape(-1).copy()).to("cuda:0")
    n = desc.shape[0]
    d_sad = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    d_xy = torch.full((n, 2), -7, dtype=torch.int16, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.sad_loop_batch_dev(d_src.data_ptr(), S * k, d_ref.data_ptr(), ref.shape[1] * k, ref.shape[1], d_desc.data_ptr(), n, w, h, sw, sh,
                               d_sad.data_ptr(), d_xy.data_ptr())
    hip_ctx.synchronize()
    return d_sad.cpu().numpy().view(np.uint32), d_xy.cpu().numpy()


def _oracle(oracle, cur, ref, desc, w, h, sw, sh, k=1):
    S, R = cur.shape[1], ref.shape[1]
    sad = np.zeros(desc.shape[0], np.uint32); xy = np.zeros((desc.shape[0], 2), np.int16)
    for i, (so, ro) in enumerate(desc):
        b, x, y = oracle.sad_loop(cur, int(so), S * k, ref, int(ro), R * k, h, w, R, sw, sh)
        sad[i], xy[i] = b, (x, y)
    return sad, xy


def test_sad_loop_config0_856x480_16x16_pm16(hip_ctx, oracle):
    pytest.importorskip("torch")
    cur = synth.PaPicture(synth.synth_luma(856, 480, 1)).full
    ref = synth.PaPicture(synth.synth_luma(856, 480, 0)).full
    S = cur.shape[1]
    desc = np.array([((68 + by) * S + 68 + bx, (68 + by - 16) * S + 68 + bx - 16) for by in range(0, 480, 16) for bx in range(0, 848, 16)], np.int64)
    assert desc.shape[0] == 53 * 30
    got = _run(hip_ctx, cur, ref, desc, 16, 16, 33, 33)
    want = _oracle(oracle, cur, ref, desc, 16, 16, 33, 33)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert len(set(map(tuple, want[1].tolist()))) > 3     # several distinct minima (global motion (3,2) plus noise), none at the origin


@pytest.mark.parametrize("case", [(16, 8, 48, 24, 2, "synth"), (32, 16, 16, 16, 2, "synth"), (64, 32, 8, 8, 2, "random"), (8, 8, 64, 64, 1, "flat"),
                                  (64, 64, 5, 3, 1, "random"), (4, 1, 33, 7, 1, "random"), (24, 13, 19, 11, 1, "synth"), (48, 64, 9, 9, 1, "extreme"),
                                  (64, 64, 9, 9, 1, "extreme"), (32, 32, 17, 5, 1, "extreme"), (16, 16, 33, 33, 1, "extreme"), (8, 64, 7, 30, 1, "extreme"),
                                  (4, 64, 64, 2, 1, "extreme"), (16, 16, 1, 1, 1, "random"), (64, 17, 8, 8, 2, "extreme"), (64, 64, 64, 64, 1, "synth"),
                                  (16, 16, 36, 33, 1, "random"), (8, 8, 12, 5, 1, "random"), (32, 32, 24, 24, 1, "synth"), (16, 4, 13, 1, 1, "random")])
def test_sad_loop_shapes(hip_ctx, oracle, case):
    """HME block shapes with row skipping (ref_stride = 2 x raw), ties on flat pictures (first position wins), odd shapes."""
    pytest.importorskip("torch")
    w, h, sw, sh, k, kind = case
    W, H = 320, 256
    rng = np.random.default_rng(w * 64 + h)
    if kind == "synth":
        cur, ref = synth.synth_luma(W, H, 2), synth.synth_luma(W, H, 0)
    elif kind == "flat":
        cur = np.full((H, W), 90, np.uint8); ref = cur.copy()
    elif kind == "extreme":
        yy, xx = np.mgrid[0:H, 0:W]
        cur = (((xx // 3 + yy // 5) & 1) * 255).astype(np.uint8); ref = (((xx // 4 + yy // 3) & 1) * 255).astype(np.uint8)
    else:
        cur, ref = rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)
    n = 41
    desc = np.zeros((n, 2), np.int64)
    for i in range(n):
        bx, by = int(rng.integers(0, W - w)), int(rng.integers(0, H - h * k))
        rx, ry = int(rng.integers(0, W - w - sw)), int(rng.integers(0, H - h * k - sh))
        desc[i] = (by * W + bx, ry * W + rx)
    got = _run(hip_ctx, cur, ref, desc, w, h, sw, sh, k)
    want = _oracle(oracle, cur, ref, desc, w, h, sw, sh, k)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    if kind == "flat":
        assert (got[1] == 0).all() and (got[0] == 0).all()
    if w in (4, 8, 16, 32, 64) and w * h + (w + sw) * (h * k + sh) < 15000:  # these widths take the packed-SAD kernel; the generic kernel (four
        # blocks and windows per workgroup in LDS) must agree on them as well where it can hold them
        hip_ctx.set_option(svtav1_hip.OPT_SADLOOP_GENERIC, 1)
        try:
            got2 = _run(hip_ctx, cur, ref, desc, w, h, sw, sh, k)
        finally:
            hip_ctx.set_option(svtav1_hip.OPT_SADLOOP_GENERIC, 0)
        assert np.array_equal(got2[0], want[0]) and np.array_equal(got2[1], want[1])


def test_sad_loop_rejects_bad_arguments(hip_ctx):
    torch = pytest.importorskip("torch")
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
    a = (buf.data_ptr(), 256, buf.data_ptr(), 256, 256, buf.data_ptr(), 1)
    for bad in ((6, 8, 8, 8), (16, 16, 65, 64), (16, 0, 8, 8), (64, 64, 4096, 1)):   # width not x4; > 4096 positions; zero height; LDS window
        with pytest.raises(svtav1_hip.SvtHipError):
            hip_ctx.sad_loop_batch_dev(*a, *bad, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(svtav1_hip.SvtHipError):
        hip_ctx.sad_loop_batch_dev(buf.data_ptr(), 256, buf.data_ptr(), 768, 256, buf.data_ptr(), 1, 16, 16, 8, 8, buf.data_ptr(), buf.data_ptr())
