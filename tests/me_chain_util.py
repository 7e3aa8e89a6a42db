"""Shared test helper: the oracle's whole-picture ME chain (what MotionEstimateLcu does for every SB)."""
import numpy as np

import svtav1_hip


def oracle_me_picture(oracle, pics, P, two_lists, use_subpel=True, cu8x8_mode=0, sb_subset=None, n_pu=85):
    """pics = [cur, ref0, ref1] PaPicture.  Returns (results [n,n_pu] structured, per-list dict l -> (desc, sad, mv)).
    sb_subset: optional indices into the picture's SB list (every SB is independent).  n_pu = 85 or 209 (all-partition mode)."""
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
    if sb_subset is not None:
        sb = np.ascontiguousarray(sb[sb_subset])
    fs = descs[0].full_stride
    state = np.zeros((sb.shape[0], 25), np.int16)
    per = {}
    prev_mv = None
    for l in range(2 if two_lists else 1):
        d, c = oracle.hme_search_center_batch(pool, descs[0], descs[1 + l], P, l, sb, prev_mv, state)
        if n_pu == 209:
            s, m = oracle.fullpel_search209_batch(pool, pool, d, fs, descs[1 + l].full_stride)
            if use_subpel:
                s, m = oracle.subpel_refine209_batch(pool, pool, d, s, m, disable_8x8=(cu8x8_mode == 1), src_stride=fs,
                                                     ref_stride=descs[1 + l].full_stride)
        else:
            s, m = oracle.fullpel_search_batch(pool, pool, d, fs, descs[1 + l].full_stride)
            if use_subpel:
                s, m, _, _ = oracle.subpel_refine_batch(pool, pool, d, s, m, disable_8x8=(cu8x8_mode == 1), src_stride=fs,
                                                        ref_stride=descs[1 + l].full_stride)
        per[l] = (d, s, m)
        prev_mv = np.ascontiguousarray(m[:, 0])
    if two_lists:
        res = oracle.bipred_pack_batch(pool, fs, pool, descs[1].full_stride, per[0][0], per[0][1], per[0][2], pool,
                                       descs[2].full_stride, per[1][0], per[1][1], per[1][2], bipred_8x8=(cu8x8_mode == 0), n_pu=n_pu)
    else:
        res = oracle.bipred_pack_batch(pool, fs, pool, descs[1].full_stride, per[0][0], per[0][1], per[0][2], n_pu=n_pu)
    return res, per


def device_me_picture(ctx, pics, P, two_lists, use_subpel=True, cu8x8_mode=0, n_pu=85):
    import torch
    dev = torch.device("cuda:0")
    pool, descs = svtav1_hip.build_picture_pool(pics)
    sb = svtav1_hip.sb_origins(pics[0].width, pics[0].height)
    n = sb.shape[0]
    d_pool = torch.from_numpy(np.concatenate([pool, np.zeros(64, np.uint8)])).to(dev)
    d_sb = torch.from_numpy(sb.view(np.int16).copy()).to(dev)
    d_out = torch.zeros((n, n_pu, 24), dtype=torch.uint8, device=dev)
    d_ls = torch.zeros((2, n, n_pu), dtype=torch.int32, device=dev)
    d_lm = torch.zeros((2, n, n_pu), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if n_pu == 209:
        ctx.motion_estimate209_batch_dev(d_pool.data_ptr(), [descs[0]], [descs[1]], [descs[2]] if two_lists else None, P, d_sb.data_ptr(), n,
                                         d_out.data_ptr(), use_subpel, cu8x8_mode, d_ls.data_ptr(), d_lm.data_ptr())
    else:
        ctx.motion_estimate_picture_dev(d_pool.data_ptr(), descs[0], descs[1], descs[2] if two_lists else None, P, d_sb.data_ptr(), n,
                                        d_out.data_ptr(), use_subpel, cu8x8_mode, d_ls.data_ptr(), d_lm.data_ptr())
    ctx.synchronize()
    res = d_out.cpu().numpy().view(svtav1_hip.ME_CU_RESULT_DTYPE).reshape(n, n_pu)
    return res, d_ls.cpu().numpy().view(np.uint32), d_lm.cpu().numpy().view(np.uint32)


def compare_results(res_dev, res_ora):
    for f in ("totalMeCandidateIndex", "xMvL0", "yMvL0", "xMvL1", "yMvL1", "distortion", "direction"):
        a, b = res_dev[f], res_ora[f]
        bad = np.argwhere(a != b)
        assert bad.size == 0, f"{f}: {len(bad)} mismatches, first at {bad[0]}: hip {a[tuple(bad[0])]} oracle {b[tuple(bad[0])]}"
