"""GPU parity: the HIP hierarchical-ME chain (centre check -> HME L0/L1/L2 -> region pick -> zero-centre check ->
window clipping -> full-pel 85-PU search) through the C ABI vs the CPU oracle (which tests/test_hme_vs_ref.py pins
against the reference's own MotionEstimateLcu).  Bit-exact."""
import numpy as np
import pytest

import svtav1_hip
from svtav1_hip import synth

pytestmark = pytest.mark.gpu


def _pics(w, h, kind):
    if kind == "synth":
        f = [synth.synth_luma(w, h, t) for t in (3, 0, 7)]
    elif kind == "pan":
        big = synth.synth_luma(w + 128, h + 96, 0)
        f = [big[40:40 + h, 50:50 + w], big[30:30 + h, 14:14 + w], big[70:70 + h, 100:100 + w]]
    elif kind == "flat":
        f = [np.full((h, w), 90, np.uint8)] * 3
    else:
        rng = np.random.default_rng(99)
        f = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    return [synth.PaPicture(np.ascontiguousarray(x)) for x in f]


class DeviceChain:
    """Runs hme -> fullpel for one or two lists entirely on device buffers."""

    def __init__(self, ctx, pics):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = torch.device("cuda:0")
        self.pool, self.descs = svtav1_hip.build_picture_pool(pics)
        self.sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
        self.n = self.sb.shape[0]
        self.d_pool = torch.from_numpy(np.concatenate([self.pool, np.zeros(64, np.uint8)])).to(self.dev)
        self.d_sb = torch.from_numpy(self.sb.view(np.int16).copy()).to(self.dev)
        self.d_state = torch.zeros((self.n, 25), dtype=torch.int16, device=self.dev)

    def run(self, P, two_lists, use_state=True):
        torch = self.torch
        out = {}
        d_mv_prev = None
        for l in range(2 if two_lists else 1):
            d_desc = torch.zeros((self.n, 6), dtype=torch.int32, device=self.dev)
            d_center = torch.zeros((self.n, 2), dtype=torch.int16, device=self.dev)
            d_sad = torch.zeros((self.n, 85), dtype=torch.int32, device=self.dev)
            d_mv = torch.zeros((self.n, 85), dtype=torch.int32, device=self.dev)
            d_l0 = None
            if l == 1:
                d_l0 = d_mv_prev[:, 0].contiguous()
            torch.cuda.synchronize()
            self.ctx.hme_search_center_dev(self.d_pool.data_ptr(), self.descs[0], self.descs[1 + l], P, l,
                                           self.d_sb.data_ptr(), self.n, d_l0.data_ptr() if d_l0 is not None else None,
                                           d_desc.data_ptr(), d_center.data_ptr(),
                                           self.d_state.data_ptr() if use_state else None)
            self.ctx.fullpel_search_dev(self.d_pool.data_ptr(), self.descs[0].full_stride, self.d_pool.data_ptr(),
                                        self.descs[1 + l].full_stride, d_desc.data_ptr(), self.n,
                                        min(P.search_area_width, 127), min(P.search_area_height, 127), d_sad.data_ptr(),
                                        d_mv.data_ptr())
            self.ctx.synchronize()
            d_mv_prev = d_mv
            out[l] = (d_desc.cpu().numpy(), d_center.cpu().numpy(), d_sad.cpu().numpy().view(np.uint32),
                      d_mv.cpu().numpy().view(np.uint32))
        return out


def oracle_chain(oracle, pics, P, two_lists):
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
    state = np.zeros((sb.shape[0], 25), np.int16)
    out = {}
    d0, c0 = oracle.hme_search_center_batch(pool, descs[0], descs[1], P, 0, sb, None, state)
    s0, m0 = oracle.fullpel_search_batch(pool, pool, d0, descs[0].full_stride, descs[1].full_stride)
    out[0] = (d0, c0, s0, m0)
    if two_lists:
        d1, c1 = oracle.hme_search_center_batch(pool, descs[0], descs[2], P, 1, sb, m0[:, 0], state)
        s1, m1 = oracle.fullpel_search_batch(pool, pool, d1, descs[0].full_stride, descs[2].full_stride)
        out[1] = (d1, c1, s1, m1)
    return out


def _check(dev, ora):
    for l in ora:
        for name, a, b in zip(("desc", "center", "sad", "mv"), dev[l], ora[l]):
            bad = np.argwhere(a != b)
            assert bad.size == 0, f"list {l} {name}: {len(bad)} mismatches, first at {bad[0]}: hip {a[tuple(bad[0])]} oracle {b[tuple(bad[0])]}"


CASES = [
    (448, 320, "synth", 3, 0, False, True, False),
    (448, 320, "pan", 3, 1, True, True, False),
    (448, 320, "pan", 4, 0, True, True, False),
    (448, 320, "pan", 5, 0, True, False, False),
    (448, 320, "pan", 3, 3, True, True, True),
    (448, 320, "random", 3, 2, True, True, False),
    (448, 320, "flat", 3, 1, True, True, False),
    (456, 328, "pan", 3, 1, True, True, False),
    (856, 480, "synth", 3, 0, False, True, False),
]


@pytest.mark.parametrize("case", CASES)
def test_hme_chain_matches_oracle(hip_ctx, oracle, case):
    pytest.importorskip("torch")
    w, h, kind, hier, tl, two, is_ref, poc_eq = case
    pics = _pics(w, h, kind)
    P = svtav1_hip.default_me_params(w, h, hier, tl, is_ref, poc_eq)
    _check(DeviceChain(hip_ctx, pics).run(P, two), oracle_chain(oracle, pics, P, two))


def test_hme_disabled_levels_carry_state(hip_ctx, oracle):
    pytest.importorskip("torch")
    pics = _pics(448, 320, "pan")
    chain = DeviceChain(hip_ctx, pics)
    for flags in [(1, 0, 0), (1, 1, 0), (0, 1, 1), (0, 0, 1)]:
        P = svtav1_hip.default_me_params(448, 320, 3, 1)
        P.enable_hme_level0_flag, P.enable_hme_level1_flag, P.enable_hme_level2_flag = flags
        _check(chain.run(P, True), oracle_chain(oracle, pics, P, True))


def test_hme_1080p_default(hip_ctx, oracle):
    """BASELINE config 2 at full size: 510 SBs, default M0-M3 parameters, against the oracle chain."""
    pytest.importorskip("torch")
    pics = _pics(1920, 1080, "synth")
    P = svtav1_hip.default_me_params(1920, 1080, 3, 0)
    dev = DeviceChain(hip_ctx, pics).run(P, False)
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(1920, 1080)
    d0, c0 = oracle.hme_search_center_batch(pool, descs[0], descs[1], P, 0, sb)
    assert np.array_equal(dev[0][0], d0) and np.array_equal(dev[0][1], c0)
    sample = np.random.default_rng(5).choice(sb.shape[0], 40, replace=False)
    s0, m0 = oracle.fullpel_search_batch(pool, pool, d0[sample], descs[0].full_stride, descs[1].full_stride)
    assert np.array_equal(dev[0][2][sample], s0) and np.array_equal(dev[0][3][sample], m0)


def test_batched_launch_equals_per_picture_launches(hip_ctx):
    """svthip_me_hme_search_center_batch_dev over several picture pairs == one svthip_me_hme_search_center_dev per pair."""
    import torch
    from svtav1_hip import synth

    w, h, n_pic = 320, 192, 5
    pics = [synth.PaPicture(synth.synth_luma(w, h, 2 * i)) for i in range(n_pic + 1)]
    pool, pd = svtav1_hip.build_picture_pool(pics)
    d_pool = torch.from_numpy(pool).to("cuda:0")
    sbs = svtav1_hip.sb_origins(w, h)
    n_sb = sbs.shape[0]
    d_sb = torch.from_numpy(sbs.view(np.int16).copy()).to("cuda:0")
    params = svtav1_hip.default_me_params(w, h, 3, 0)
    d_one = torch.zeros((n_pic * n_sb, 6), dtype=torch.int32, device="cuda:0")
    d_cen1 = torch.zeros((n_pic * n_sb, 2), dtype=torch.int16, device="cuda:0")
    for i in range(n_pic):
        hip_ctx.hme_search_center_dev(d_pool.data_ptr(), pd[i + 1], pd[i], params, 0, d_sb.data_ptr(), n_sb, None,
                                      d_one.data_ptr() + i * n_sb * 24, d_cen1.data_ptr() + i * n_sb * 4)
    d_bat = torch.zeros((n_pic * n_sb, 6), dtype=torch.int32, device="cuda:0")
    d_cen2 = torch.zeros((n_pic * n_sb, 2), dtype=torch.int16, device="cuda:0")
    hip_ctx.hme_search_center_batch_dev(d_pool.data_ptr(), [pd[i + 1] for i in range(n_pic)], [pd[i] for i in range(n_pic)], params, 0,
                                        d_sb.data_ptr(), n_sb, None, d_bat.data_ptr(), d_cen2.data_ptr())
    hip_ctx.synchronize()
    assert torch.equal(d_one, d_bat) and torch.equal(d_cen1, d_cen2)
    assert (d_cen2 != 0).any()
