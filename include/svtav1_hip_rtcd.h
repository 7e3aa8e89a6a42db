/*
 * include/svtav1_hip_rtcd.h -- same-signature, single-TU, synchronous drop-ins for the RTCD pointers of the reference's
 * transform / quantisation surface (Source/Lib/Codec/aom_dsp_rtcd.h).  A maintainer overrides the pointers after
 * setup_rtcd_internal() (Codec/aom_dsp_rtcd.h:1972-1983), e.g.
 *
 *     av1_fwd_txfm2d_16x16   = svthip_av1_fwd_txfm2d_16x16;     (Codec/aom_dsp_rtcd.h:224-238, assigned :2676-2726)
 *     av1_inv_txfm2d_add_8x8 = svthip_av1_inv_txfm2d_add_8x8;   (:336-404, assigned :2098-2139)
 *     aom_quantize_b         = svthip_aom_quantize_b;           (:310-331, assigned :2086-2096)
 *     av1_inv_txfm_add       = svthip_av1_inv_txfm_add;         (:410-412, assigned :2138-2139)
 *
 * Every call uploads its one TU, runs the batch kernel of include/svtav1_hip.h with n_tu = 1 on a context owned by the calling
 * thread (created on first use on device $SVTHIP_DEVICE, default 0) and copies the result back: bit-identical to the reference's C
 * functions, and a PCIe round trip per TU -- this surface exists so that the library is a drop-in at the reference's own granularity;
 * throughput comes from the batch entries.  There is no CPU path: without a device the first call aborts with a
 * message (the reference's signatures return void).
 *
 * TxType / TxSize are ATTRIBUTE_PACKED enums in the reference (one byte, Codec/EbDefinitions.h); they are declared uint8_t here.
 */
#ifndef SVTAV1_HIP_RTCD_H
#define SVTAV1_HIP_RTCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* void (*av1_fwd_txfm2d_WxH)(int16_t *input, int32_t *output, uint32_t inputStride, TxType transform_type, uint8_t bit_depth) */
#define SVTHIP_DECL_FWD(W, H) \
    void svthip_av1_fwd_txfm2d_##W##x##H(int16_t *input, int32_t *output, uint32_t input_stride, uint8_t transform_type, uint8_t bit_depth);
SVTHIP_DECL_FWD(4, 4) SVTHIP_DECL_FWD(8, 8) SVTHIP_DECL_FWD(16, 16) SVTHIP_DECL_FWD(32, 32) SVTHIP_DECL_FWD(64, 64)
SVTHIP_DECL_FWD(4, 8) SVTHIP_DECL_FWD(8, 4) SVTHIP_DECL_FWD(8, 16) SVTHIP_DECL_FWD(16, 8) SVTHIP_DECL_FWD(16, 32) SVTHIP_DECL_FWD(32, 16)
SVTHIP_DECL_FWD(32, 64) SVTHIP_DECL_FWD(64, 32) SVTHIP_DECL_FWD(4, 16) SVTHIP_DECL_FWD(16, 4) SVTHIP_DECL_FWD(8, 32) SVTHIP_DECL_FWD(32, 8)
SVTHIP_DECL_FWD(16, 64) SVTHIP_DECL_FWD(64, 16)
#undef SVTHIP_DECL_FWD

/* squares: void (*av1_inv_txfm2d_add_NxN)(const int32_t *input, uint16_t *output, int32_t stride, TxType tx_type, int32_t bd) */
#define SVTHIP_DECL_INV_SQ(N) \
    void svthip_av1_inv_txfm2d_add_##N##x##N(const int32_t *input, uint16_t *output, int32_t stride, uint8_t tx_type, int32_t bd);
SVTHIP_DECL_INV_SQ(4) SVTHIP_DECL_INV_SQ(8) SVTHIP_DECL_INV_SQ(16) SVTHIP_DECL_INV_SQ(32) SVTHIP_DECL_INV_SQ(64)
#undef SVTHIP_DECL_INV_SQ
/* 2:1 / 4:1 rectangles with at least 8 on the short side: (..., TxType tx_type, TxSize tx_size, int32_t eob, int32_t bd) */
#define SVTHIP_DECL_INV_RE(W, H) \
    void svthip_av1_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int32_t stride, uint8_t tx_type, uint8_t tx_size, int32_t eob, int32_t bd);
SVTHIP_DECL_INV_RE(8, 16) SVTHIP_DECL_INV_RE(16, 8) SVTHIP_DECL_INV_RE(16, 32) SVTHIP_DECL_INV_RE(32, 16) SVTHIP_DECL_INV_RE(32, 8)
SVTHIP_DECL_INV_RE(8, 32) SVTHIP_DECL_INV_RE(32, 64) SVTHIP_DECL_INV_RE(64, 32) SVTHIP_DECL_INV_RE(16, 64) SVTHIP_DECL_INV_RE(64, 16)
#undef SVTHIP_DECL_INV_RE
/* 4-wide / 4-high rectangles: (..., TxType tx_type, TxSize tx_size, int32_t bd) */
#define SVTHIP_DECL_INV_R4(W, H) \
    void svthip_av1_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int32_t stride, uint8_t tx_type, uint8_t tx_size, int32_t bd);
SVTHIP_DECL_INV_R4(4, 8) SVTHIP_DECL_INV_R4(8, 4) SVTHIP_DECL_INV_R4(4, 16) SVTHIP_DECL_INV_R4(16, 4)
#undef SVTHIP_DECL_INV_R4

/* the 13-argument libaom quantiser signature (Codec/aom_dsp_rtcd.h:309-331); tran_low_t = int32_t */
#define SVTHIP_DECL_QUANT(NAME)                                                                                                          \
    void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, int32_t skip_block, const int16_t *zbin_ptr, const int16_t *round_ptr,       \
              const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr,                      \
              const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan);
SVTHIP_DECL_QUANT(svthip_aom_quantize_b) SVTHIP_DECL_QUANT(svthip_aom_quantize_b_32x32) SVTHIP_DECL_QUANT(svthip_aom_quantize_b_64x64)
SVTHIP_DECL_QUANT(svthip_aom_highbd_quantize_b) SVTHIP_DECL_QUANT(svthip_aom_highbd_quantize_b_32x32)
SVTHIP_DECL_QUANT(svthip_aom_highbd_quantize_b_64x64)
#undef SVTHIP_DECL_QUANT

/* av1_inv_txfm_add (Codec/aom_dsp_rtcd.h:410-412, assigned :2138-2139): the pointer the reference's 8-bit reconstruction calls,
 *     Av1InvTransformRecon8bit -> av1_inv_txfm_add(coeffBuffer, reconBuffer, reconStride, &txfm_param)   (Codec/EbTransforms.c:8374-8399)
 * C body av1_inv_txfm_add_c (:8321-8340): widen the 8-bit block, highbd_inv_txfm_add by tx_size (:8252-8319), narrow again --
 * equivalent to reconstructing the 8-bit plane directly, which is what the batch kernel's 8-bit plane mode does.
 * svthip_txfm_param is layout-identical to TxfmParam (Codec/EbDefinitions.h:725-737; TxType / TxSize / TxSetType are one-byte
 * packed enums): 24 bytes.  tx_type, tx_size and bd (8) are read; lossless must be 0 (the reference never sets it:
 * Codec/EbTransforms.c:8362, :8391; the Walsh-Hadamard path is not provided and the call aborts with a message); eob is not needed
 * (an all-zero block inverse-transforms to zero). */
typedef struct svthip_txfm_param {
    uint8_t tx_type; /* TxType */
    uint8_t tx_size; /* TxSize */
    int32_t lossless;
    int32_t bd;
    int32_t is_hbd;
    uint8_t tx_set_type; /* TxSetType */
    int32_t eob;
} svthip_txfm_param;
void svthip_av1_inv_txfm_add(const int32_t *dqcoeff, uint8_t *dst, int32_t stride, const svthip_txfm_param *txfm_param);

/* Leaf SAD surface, for completeness ("too fine-grained for a GPU" -- the engine's own granularity is the whole-picture /
 * batch entries; these exist so that every pointer of the path has a same-signature stand-in):
 *   EB_SADKERNELNxM_TYPE      (Codec/EbComputeSAD.h:27-33)  -- NxMSadKernel_funcPtrArray[asm][width / 8] (:127-180): SAD of one block
 *   EB_SADLOOPKERNELNxM_TYPE  (Codec/EbComputeSAD.h:37-49)  -- NxMSadLoopKernel_funcPtrArray[asm] (:183-189) = SadLoopKernel
 *                             (C_DEFAULT/EbComputeSAD_C.c:73-119): best SAD and position over a search area; refStride may be
 *                             2 * srcStrideRaw (the HME callers skip every other row).  *bestSad starts from 0xffffff like the C body.
 * One synchronous call of svthip_sad_loop_batch_dev with n_blocks = 1 each (width: multiple of 4, 4..64; height 1..64). */
uint32_t svthip_nxm_sad_kernel(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
void svthip_sad_loop_kernel(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width,
                            uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw,
                            int16_t search_area_width, int16_t search_area_height);

#ifdef __cplusplus
}
#endif
#endif /* SVTAV1_HIP_RTCD_H */
