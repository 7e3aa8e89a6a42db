/*
 * oracle/ref_bench_driver.c -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Batch loops over the REFERENCE's own C functions (compiled from /root/reference into oracle/_ref/libsvtref_bench.so by
 * oracle/build_ref.sh) for bench.py's per-leg `cpu_baseline`: a Python loop calling one function per TU through ctypes would time the
 * interpreter, not the reference.  Contains no reference code, only calls into it; every entry is re-entrant (called from several
 * host threads over disjoint ranges).
 *
 *   ref_bench_tq_chain   : per TU what Av1EncodeLoop does for an 8-bit luma TU (Codec/EbCodingLoop.c:552-760):
 *                          ResidualKernel_c -> Av1TransformTwoD_NxN_c (+ HandleTransform64x64_c) -> aom_quantize_b*_c_II ->
 *                          av1_inv_txfm2d_add_NxN_c on the widened block, as av1_inv_txfm_add_c -- what Av1InvTransformRecon8bit
 *                          reaches -- does (Codec/EbTransforms.c:8321-8340; its inner calls go through RTCD pointers that only
 *                          setup_rtcd_internal fills, so the C bodies are called directly), square sizes 4..64.
 *   ref_bench_convolve   : av1_convolve_{2d,x,y,2d_copy}_sr_c driven like av1_inter_prediction (oracle/ref_convolve_driver.c), per block.
 */
#include <stdint.h>
#include <string.h>

#include "EbDefinitions.h"

void ResidualKernel_c(uint8_t *input, uint32_t inputStride, uint8_t *pred, uint32_t predStride, int16_t *residual, uint32_t residualStride,
                      uint32_t areaWidth, uint32_t areaHeight);
#define DECL_FWD(N) void Av1TransformTwoD_##N##x##N##_c(int16_t *input, int32_t *output, uint32_t inputStride, TxType transform_type, uint8_t bit_depth);
DECL_FWD(4) DECL_FWD(8) DECL_FWD(16) DECL_FWD(32) DECL_FWD(64)
uint64_t HandleTransform64x64_c(int32_t *output, uint32_t outputStride);
#define DECL_Q(n)                                                                                                                      \
    void n(const int32_t *coeff_ptr, intptr_t n_coeffs, int32_t skip_block, const int16_t *zbin_ptr, const int16_t *round_ptr,         \
           const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, \
           uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan);
DECL_Q(aom_quantize_b_c_II) DECL_Q(aom_quantize_b_32x32_c_II) DECL_Q(aom_quantize_b_64x64_c_II)
#define DECL_INV(N) void av1_inv_txfm2d_add_##N##x##N##_c(const int32_t *input, uint16_t *output, int32_t stride, TxType tx_type, int32_t bd);
DECL_INV(4) DECL_INV(8) DECL_INV(16) DECL_INV(32) DECL_INV(64)

/* TU i: source / prediction / reconstruction at offs[i] in planes of `stride`; tx_type[i]; quantiser row qrows + 10 * qidx[i]
 * (zbin[2], round[2], quant[2], quant_shift[2], dequant[2]); scan / iscan tables per tx_type given by scan_of_type[tx_type] (int16 arrays of
 * min(n,32)^2 entries).  Returns the sum of the eobs (so the work cannot be optimised away). */
uint64_t ref_bench_tq_chain(uint8_t *src, uint8_t *pred, uint8_t *recon, uint32_t stride, const uint32_t *offs, const uint8_t *tx_type,
                            const uint8_t *qidx, uint32_t n_tu, int n, const int16_t *qrows, const int16_t *const *scan_of_type,
                            const int16_t *const *iscan_of_type)
{
    int16_t residual[64 * 64];
    int32_t coeff[64 * 64], qcoeff[32 * 32], dqcoeff[32 * 32];
    uint64_t acc = 0;
    const int nc = (n > 32 ? 32 : n) * (n > 32 ? 32 : n);
    for (uint32_t i = 0; i < n_tu; i++) {
        const uint32_t o = offs[i];
        ResidualKernel_c(src + o, stride, pred + o, stride, residual, (uint32_t)n, (uint32_t)n, (uint32_t)n);
        switch (n) {
        case 4: Av1TransformTwoD_4x4_c(residual, coeff, 4, (TxType)tx_type[i], 8); break;
        case 8: Av1TransformTwoD_8x8_c(residual, coeff, 8, (TxType)tx_type[i], 8); break;
        case 16: Av1TransformTwoD_16x16_c(residual, coeff, 16, (TxType)tx_type[i], 8); break;
        case 32: Av1TransformTwoD_32x32_c(residual, coeff, 32, (TxType)tx_type[i], 8); break;
        default:
            Av1TransformTwoD_64x64_c(residual, coeff, 64, (TxType)tx_type[i], 8);
            acc += HandleTransform64x64_c(coeff, 64) & 1u; /* zeroes the dropped quadrants; rows stay 64 apart */
            for (int r = 1; r < 32; r++) memmove(coeff + 32 * r, coeff + 64 * r, 32 * sizeof(int32_t)); /* pack like Av1EstimateTransform */
            break;
        }
        const int16_t *q = qrows + 10 * qidx[i];
        uint16_t eob = 0;
        const int16_t *sc = scan_of_type[tx_type[i]], *isc = iscan_of_type[tx_type[i]];
        if (n <= 16) aom_quantize_b_c_II(coeff, nc, 0, q, q + 2, q + 4, q + 6, qcoeff, dqcoeff, q + 8, &eob, sc, isc);
        else if (n == 32) aom_quantize_b_32x32_c_II(coeff, nc, 0, q, q + 2, q + 4, q + 6, qcoeff, dqcoeff, q + 8, &eob, sc, isc);
        else aom_quantize_b_64x64_c_II(coeff, nc, 0, q, q + 2, q + 4, q + 6, qcoeff, dqcoeff, q + 8, &eob, sc, isc);
        acc += eob;
        /* reconstruction on top of the prediction, like the encode pass (recon starts as a copy of pred) */
        for (int r = 0; r < n; r++) memcpy(recon + o + (size_t)r * stride, pred + o + (size_t)r * stride, (size_t)n);
        if (eob) {
            uint16_t wide[64 * 64];
            for (int r = 0; r < n; r++)
                for (int c = 0; c < n; c++) wide[r * n + c] = recon[o + (size_t)r * stride + c];
            switch (n) {
            case 4: av1_inv_txfm2d_add_4x4_c(dqcoeff, wide, 4, (TxType)tx_type[i], 8); break;
            case 8: av1_inv_txfm2d_add_8x8_c(dqcoeff, wide, 8, (TxType)tx_type[i], 8); break;
            case 16: av1_inv_txfm2d_add_16x16_c(dqcoeff, wide, 16, (TxType)tx_type[i], 8); break;
            case 32: av1_inv_txfm2d_add_32x32_c(dqcoeff, wide, 32, (TxType)tx_type[i], 8); break;
            default: av1_inv_txfm2d_add_64x64_c(dqcoeff, wide, 64, (TxType)tx_type[i], 8); break;
            }
            for (int r = 0; r < n; r++)
                for (int c = 0; c < n; c++) recon[o + (size_t)r * stride + c] = (uint8_t)wide[r * n + c];
        }
    }
    return acc;
}

void ref_av1_convolve_sr(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, int32_t w, int32_t h, int filter_x,
                         int filter_y, int subpel_x, int subpel_y);

/* block i: source at src + src_off[i], destination tile dst + dst_off[i] (dst_stride), phases / filters packed as
 * subpel_x | subpel_y << 4 | filter_x << 8 | filter_y << 12 */
void ref_bench_convolve(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, const uint32_t *src_off, const uint32_t *dst_off,
                        const uint16_t *mode, uint32_t n_blocks, int32_t w, int32_t h)
{
    for (uint32_t i = 0; i < n_blocks; i++)
        ref_av1_convolve_sr(src + src_off[i], src_stride, dst + dst_off[i], dst_stride, w, h, (mode[i] >> 8) & 15, (mode[i] >> 12) & 15, mode[i] & 15,
                            (mode[i] >> 4) & 15);
}
