/*
 * oracle/svt_tq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 * Plain-C restatement of the reference's quantisation (Source/Lib/Codec/EbFullLoop.c) and, below, its forward
 * transforms (Source/Lib/Codec/EbTransforms.c).  Pinned against the reference's own C functions in
 * oracle/_ref/libsvtref_tq.so by tests/test_tq_vs_ref.py.
 */
#include "svt_me_oracle.h"

#include <string.h>

static inline int32_t rpot(int32_t v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); } /* ROUND_POWER_OF_TWO */

/* quantize_b_helper_c_II (:46-108, 8-bit: |coeff|+round is clamped to int16) and highbd_quantize_b_helper_c (:242-299, no
 * clamp) with the flat quantisation matrix the encoder uses (qm_ptr == NULL -> wt = iwt = 1 << AOM_QM_BITS).
 * The pre-scan passes of both only skip coefficients inside the dead zone, which the per-coefficient test repeats, so the
 * result is a pure per-coefficient function plus eob = 1 + last scan position with a non-zero level.
 * qp = {zbin[2], round[2], quant[2], quant_shift[2], dequant[2]} (index 0 = DC, 1 = AC). */
void orc_quantize_b(const int32_t *coeff, int32_t n_coeffs, const int16_t *qp, const int16_t *scan, int log_scale, int highbd,
                    int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob_ptr)
{
    const int16_t *zbin = qp, *round = qp + 2, *quant = qp + 4, *quant_shift = qp + 6, *dequant = qp + 8;
    const int32_t zbins[2] = {rpot(zbin[0], log_scale), rpot(zbin[1], log_scale)};
    int32_t eob = -1;
    memset(qcoeff, 0, sizeof(int32_t) * (size_t)n_coeffs);
    memset(dqcoeff, 0, sizeof(int32_t) * (size_t)n_coeffs);
    for (int32_t i = 0; i < n_coeffs; i++) {
        const int32_t rc = scan[i];
        const int ac = rc != 0;
        const int32_t c = coeff[rc];
        const int32_t sign = c >> 31;
        const int32_t abs_c = (c ^ sign) - sign;
        if (abs_c < zbins[ac]) continue;
        int64_t tmp = (int64_t)abs_c + rpot(round[ac], log_scale);
        if (!highbd) tmp = tmp < -32768 ? -32768 : (tmp > 32767 ? 32767 : tmp); /* clamp(.., INT16_MIN, INT16_MAX), :88-90 */
        tmp *= 32;                                                              /* wt = 1 << AOM_QM_BITS */
        const int32_t level = (int32_t)(((((tmp * quant[ac]) >> 16) + tmp) * quant_shift[ac]) >> (16 - log_scale + 5));
        qcoeff[rc] = (level ^ sign) - sign;
        const int32_t dq = (dequant[ac] * 32 + 16) >> 5;
        const int32_t abs_dq = (int32_t)((uint32_t)level * (uint32_t)dq) >> log_scale; /* int32 product as in the reference */
        dqcoeff[rc] = (abs_dq ^ sign) - sign;
        if (level) eob = i;
    }
    *eob_ptr = (uint16_t)(eob + 1);
}
