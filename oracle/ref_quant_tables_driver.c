/*
 * oracle/ref_quant_tables_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Exposes the REFERENCE's own quantiser tables and scan orders (compiled from /root/reference into
 * oracle/_ref/libsvtref_me.so by oracle/build_ref.sh) in the row layout of include/svtav1_hip.h:
 *   ref_build_quantizer_rows : av1_build_quantizer (Codec/EbModeDecisionConfigurationProcess.c:417-506) ->
 *                              int16 [256 qindex][3 planes Y,U,V][10] = zbin[2], round[2], quant[2], quant_shift[2], dequant_QTX[2]
 *                              (the fields Av1QuantizeInvQuantize_II hands to the quantiser, Codec/EbFullLoop.c:793-823)
 *   ref_scan_order           : av1_scan_orders[tx_size][tx_type] (Codec/EbTransforms.h:3336, used at Codec/EbFullLoop.c:826)
 * Contains no reference code, only calls and table reads.  tests/golden/make_golden.py turns the outputs into fixtures.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbTransforms.h"

void av1_build_quantizer(aom_bit_depth_t bit_depth, int32_t y_dc_delta_q, int32_t u_dc_delta_q, int32_t u_ac_delta_q, int32_t v_dc_delta_q,
                         int32_t v_ac_delta_q, Quants *const quants, Dequants *const deq);

int ref_build_quantizer_rows(int bit_depth, int y_dc_delta_q, int u_dc_delta_q, int u_ac_delta_q, int v_dc_delta_q, int v_ac_delta_q,
                             int16_t *rows /* [256][3][10] */)
{
    Quants *q = (Quants *)aligned_alloc(64, (sizeof(Quants) + 63) & ~(size_t)63);
    Dequants *d = (Dequants *)aligned_alloc(64, (sizeof(Dequants) + 63) & ~(size_t)63);
    if (!q || !d) return -1;
    memset(q, 0, sizeof(*q));
    memset(d, 0, sizeof(*d));
    av1_build_quantizer((aom_bit_depth_t)bit_depth, y_dc_delta_q, u_dc_delta_q, u_ac_delta_q, v_dc_delta_q, v_ac_delta_q, q, d);
    for (int i = 0; i < QINDEX_RANGE; i++) {
        const int16_t *zb[3] = {q->y_zbin[i], q->u_zbin[i], q->v_zbin[i]};
        const int16_t *rd[3] = {q->y_round[i], q->u_round[i], q->v_round[i]};
        const int16_t *qu[3] = {q->y_quant[i], q->u_quant[i], q->v_quant[i]};
        const int16_t *qs[3] = {q->y_quant_shift[i], q->u_quant_shift[i], q->v_quant_shift[i]};
        const int16_t *dq[3] = {d->y_dequant_QTX[i], d->u_dequant_QTX[i], d->v_dequant_QTX[i]};
        for (int p = 0; p < 3; p++) {
            int16_t *r = rows + ((size_t)i * 3 + p) * 10;
            for (int k = 0; k < 2; k++) {
                r[0 + k] = zb[p][k];
                r[2 + k] = rd[p][k];
                r[4 + k] = qu[p][k];
                r[6 + k] = qs[p][k];
                r[8 + k] = dq[p][k];
            }
        }
    }
    free(q);
    free(d);
    return 0;
}

/* copies n = min(w,32) * min(h,32) entries of scan and iscan; returns n (0 if the table is absent) */
int ref_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan)
{
    static const int wh[TX_SIZES_ALL][2] = {{4, 4}, {8, 8}, {16, 16}, {32, 32}, {64, 64}, {4, 8}, {8, 4}, {8, 16}, {16, 8}, {16, 32},
                                            {32, 16}, {32, 64}, {64, 32}, {4, 16}, {16, 4}, {8, 32}, {32, 8}, {16, 64}, {64, 16}};
    const SCAN_ORDER *so = &av1_scan_orders[tx_size][tx_type];
    if (!so->scan || !so->iscan) return 0;
    const int w = wh[tx_size][0] > 32 ? 32 : wh[tx_size][0], h = wh[tx_size][1] > 32 ? 32 : wh[tx_size][1];
    const int n = w * h;
    memcpy(scan, so->scan, sizeof(int16_t) * n);
    memcpy(iscan, so->iscan, sizeof(int16_t) * n);
    return n;
}
