"""GPU parity of the batched transform / quantisation entries vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import svtav1_hip
from tq_util import oracle_quant_batch, random_quant_batch

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
