#!/usr/bin/env python3
"""Non-parity experiment: 32x32 forward DCT as two matrix products on the matrix cores
(tools/mfma_dct_experiment.hip) against the bit-exact butterfly kernel of the product (svthip_fwd_txfm2d_batch_dev).
Reports the mismatch (how many coefficients differ from the reference's integers, and by how much), both kernels' distance from the
infinitely precise transform, their time, algorithmic bytes rate and the matrix-core occupancy of the MFMA version.
Usage: python tools/mfma_dct_probe.py [--tus N] [--iters K]      (builds tools/_build/libmfma_dct.so with hipcc when missing)"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))
import torch  # noqa: E402
import svtav1_hip  # noqa: E402

SO = os.path.join(ROOT, "tools", "_build", "libmfma_dct.so")


def dct_matrix():
    k = np.arange(32)[:, None]
    n = np.arange(32)[None, :]
    m = np.cos((2 * n + 1) * k * np.pi / 64)
    m[0, :] = 1 / np.sqrt(2)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tus", type=int, default=1 << 16)   # 64 M coefficients: 384 MB working set, beyond the 256 MB infinity cache
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    if not os.path.exists(SO):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "mfma_dct_experiment.hip"), "-o", SO])
    lib = C.CDLL(SO)
    lib.mfma_dct32_run.restype = C.c_int
    lib.mfma_dct32_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_void_p]
    ctx = svtav1_hip.Context(0)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    n = args.tus
    M = dct_matrix()
    d_M = torch.from_numpy(M.astype(np.float32)).to("cuda:0")
    out = {"n_tu": n, "transform": "32x32 DCT_DCT forward, 8-bit residual"}
    for kind in ("uniform", "laplacian_b6"):
        rng = np.random.default_rng(5)
        if kind == "uniform":
            x = rng.integers(-255, 256, (n, 32, 32), dtype=np.int16)
        else:
            x = np.clip(np.rint(rng.laplace(0, 6, (n, 32, 32))), -255, 255).astype(np.int16)
        d_x = torch.from_numpy(x.reshape(-1)).to("cuda:0")
        desc = np.zeros(n, svtav1_hip.TXFM_DESC_DTYPE)
        desc["in_offset"] = np.arange(n, dtype=np.uint32) * 1024
        desc["out_offset"] = np.arange(n, dtype=np.uint32) * 1024
        desc["in_stride"] = 32
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to("cuda:0")
        d_ref = torch.zeros(n * 1024, dtype=torch.int32, device="cuda:0")
        d_mm = torch.zeros(n * 1024, dtype=torch.int32, device="cuda:0")

        def exact():
            ctx.fwd_txfm2d_batch_dev(d_x.data_ptr(), d_desc.data_ptr(), n, 32, 32, 8, d_ref.data_ptr(), stream)

        def mfma():
            rc = lib.mfma_dct32_run(d_x.data_ptr(), d_mm.data_ptr(), n, d_M.data_ptr(), 0.25, stream)
            assert rc == 0

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters

        t_exact, t_mfma = timed(exact), timed(mfma)
        ref = d_ref.cpu().numpy().reshape(n, 32, 32)
        mm = d_mm.cpu().numpy().reshape(n, 32, 32)
        diff = (mm.astype(np.int64) - ref).reshape(-1)
        ns = min(n, 2048)   # infinitely precise transform of a sample: (4 X) through M twice, / 16  ->  0.25 M X M^T
        ideal = 0.25 * np.einsum("ky,nyx,vx->nkv", M, x[:ns].astype(np.float64), M)
        e_ref = ref[:ns] - ideal
        e_mm = mm[:ns] - ideal
        coeffs = n * 1024
        res = {
            "butterfly_bit_exact": {"ms": round(t_exact, 4), "algorithmic_tbps": round(coeffs * 6 / t_exact * 1e-9, 3),
                                    "max_abs_error_vs_ideal": round(float(np.abs(e_ref).max()), 3),
                                    "rms_error_vs_ideal": round(float(np.sqrt((e_ref ** 2).mean())), 4)},
            "mfma_matmul": {"ms": round(t_mfma, 4), "algorithmic_tbps": round(coeffs * 6 / t_mfma * 1e-9, 3),
                            "max_abs_error_vs_ideal": round(float(np.abs(e_mm).max()), 3),
                            "rms_error_vs_ideal": round(float(np.sqrt((e_mm ** 2).mean())), 4),
                            "mfma_per_tu": 10,
                            "matrix_core_occupancy": round(n * 10 * 32 / (1024 * 2.4e9 * t_mfma * 1e-3), 4)},
            "mismatch_vs_reference_integers": {"coefficients_differing": round(float((diff != 0).mean()), 5),
                                               "tus_with_any_difference": round(float((diff.reshape(n, -1) != 0).any(axis=1).mean()), 5),
                                               "max_abs_difference": int(np.abs(diff).max()),
                                               "histogram_-2..2": [int((diff == v).sum()) for v in (-2, -1, 0, 1, 2)]},
        }
        out[kind] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
