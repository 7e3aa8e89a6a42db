/*
 * oracle/svt_me_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's (ateme-developers/SVT-AV1-1) motion-estimation hot path,
 * asm_type = ASM_NON_AVX2 semantics.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call into this; the product path (svt-av1-1_amd/csrc) never does.
 *
 * Pinning: the reference ships no tests / golden vectors for this path, so every
 * function here is pinned against the reference's own kernels compiled from /root/reference into
 * oracle/_ref/ (oracle/build_ref.sh) by tests/test_oracle_vs_ref.py, and against tests/golden/
 * fixtures generated from those kernels (tests/golden/make_golden.py).
 *
 * All paths cited are under /root/reference/Source/Lib/.
 */
#ifndef SVT_ME_ORACLE_H
#define SVT_ME_ORACLE_H
#include <stdint.h>
#include "../include/svtav1_hip.h" /* public ABI types only (descriptors, parameter blocks) */
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SAD_VALUE (128u * 128u * 255u) /* Codec/EbMotionEstimation.h:78 */
#define ORC_NUM_SQ_PU 85

/* C_DEFAULT/EbComputeSAD_C.c:49-71 (FastLoop_NxMSadKernel) */
uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                     uint32_t height, uint32_t width);

/* C_DEFAULT/EbComputeSAD_C.c:73-119 (SadLoopKernel): exhaustive search, strict '<' in raster order,
 * bestSad starts at 0xffffff; x/y untouched when nothing beats it. */
void orc_sad_loop_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                         uint32_t height, uint32_t width, uint64_t *best_sad, int16_t *x_center,
                         int16_t *y_center, uint32_t ref_stride_raw, int16_t search_area_width,
                         int16_t search_area_height);

/* FullPelSearch_LCU (Codec/EbMotionEstimation.c:1504-1551) over one SB and one reference list.
 *   src        : top-left of the 64x64 source SB inside the padded source plane (stride src_stride)
 *   ref        : reference sample at search position (0,0), i.e. integer_buffer_ptr + 2 + 2*stride
 *   best_sad/mv: 85 entries each, ME-buffer order (0: 64x64, 1-4: 32x32, 5-20: 16x16 z-order,
 *                21-84: 8x8 = 21 + 4*z16 + raster-in-16x16); updated in place with strict '<'.
 * SADs are the vertically 2:1 subsampled ones, doubled.
 * MV word = (uint16)(4*(y_origin+ys)) << 16 | (uint16)(4*(x_origin+xs))  (:1389-1391). */
void orc_fullpel_search_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                             int16_t x_search_area_origin, int16_t y_search_area_origin,
                             uint32_t search_area_width, uint32_t search_area_height, uint32_t *best_sad,
                             uint32_t *best_mv);

void orc_init_best(uint32_t *best_sad, uint32_t *best_mv, uint32_t n);

/* n_sb independent searches; desc[i] = {src_offset, ref_offset, x_origin, y_origin, sw, sh} with the
 * offsets in bytes into the two planes; outputs [n_sb][85], initialised to MAX_SAD_VALUE / 0 first. */
void orc_fullpel_search_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane,
                              uint32_t ref_stride, const int32_t *desc, uint32_t n_sb, uint32_t *best_sad,
                              uint32_t *best_mv);

/* Search-centre half of MotionEstimateLcu for one SB and one list (Codec/EbMotionEstimation.c:6300-6738):
 * hme_mv_center_check, HmeLevel0/1/2 per region, region pick, CheckZeroZeroCenter, window clipping.
 * Writes the full-pel descriptor (offsets relative to `pool`) and optionally the final centre. */
void orc_hme_search_center(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                           const svthip_me_params *p, uint32_t list_index, uint32_t sb_origin_x, uint32_t sb_origin_y,
                           uint32_t l0_best_mv64, svthip_fullpel_desc *desc, int16_t *center_xy, int16_t *hme_state);
void orc_hme_search_center_batch(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                                 const svthip_me_params *p, uint32_t list_index, const svthip_sb_origin *sb, uint32_t n_sb,
                                 const uint32_t *l0_best_mv64, svthip_fullpel_desc *desc, int16_t *center_xy,
                                 int16_t *hme_state /* [n_sb][25], carried from list 0 to list 1; may be NULL */);

/* ---- sub-pel (oracle/svt_subpel_oracle.c) ---- */
/* b / h / j half-pel planes of InterpolateSearchRegionAVC as w x h tiles whose (0,0) is search-region coordinate (x0,y0):
 * b[x,y] = half-pel (x-1/2, y), h[x,y] = (x, y-1/2), j[x,y] = (x-1/2, y-1/2) (vertical filter of the rounded b). */
void orc_interp_planes(const uint8_t *ref00, uint32_t ref_stride, int x0, int y0, int w, int h, uint8_t *b, uint8_t *hh,
                       uint8_t *j);
/* SpatialFullDistortionKernel*_SSSE3_INTRIN semantics: sum of squared 8-bit WRAPPED differences */
uint32_t orc_ssd_wrapped(const uint8_t *src, uint32_t src_stride, const uint8_t *rec, uint32_t rec_stride, uint32_t w, uint32_t h);
/* half-pel + quarter-pel refinement of the 85 square PUs (SSD_SEARCH mode, M0/M1 flags) */
void orc_subpel_refine_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                            int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, uint32_t *best_sad,
                            uint32_t *best_mv, uint32_t *out_ssd, uint8_t *out_dir);
void orc_subpel_refine_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane, uint32_t ref_stride,
                             const int32_t *desc, uint32_t n_sb, int disable_8x8, uint32_t *best_sad, uint32_t *best_mv,
                             uint32_t *out_ssd, uint8_t *out_dir);

/* bi-prediction SAD + Sort3Elements packing of the 85 PUs into raster-ordered results */
void orc_bipred_pack_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref0_00, uint32_t ref0_stride, int16_t xo0,
                          int16_t yo0, const uint8_t *ref1_00, uint32_t ref1_stride, int16_t xo1, int16_t yo1,
                          const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1, int n_lists,
                          int bipred_8x8, svthip_me_cu_result *out);
void orc_bipred_pack_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref0_plane, uint32_t ref0_stride,
                           const int32_t *desc0, const uint8_t *ref1_plane, uint32_t ref1_stride, const int32_t *desc1,
                           uint32_t n_sb, const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1,
                           int n_lists, int bipred_8x8, svthip_me_cu_result *out);

/* ---- the 209-PU (all-partition) mode ---- */
/* [209][5] = w, h, px, py, ME-buffer index, by raster PU index */
void orc_pu_geometry209(uint8_t *out);
void orc_subpel_refine_209pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                             int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, uint32_t *best_sad,
                             uint32_t *best_mv);
/* the refinement under any fractional search method (0 SUB_SAD_SEARCH, 1 FULL_SAD_SEARCH, 2 SSD_SEARCH): 85 or 209 PUs, ME-buffer order */
void orc_subpel_refine_method(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride, int16_t x_search_area_origin,
                              int16_t y_search_area_origin, int disable_8x8, int all_pu, int method, uint32_t *best_sad, uint32_t *best_mv,
                              uint8_t *out_dir);
void orc_subpel_refine209_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane, uint32_t ref_stride,
                                const int32_t *desc, uint32_t n_sb, int disable_8x8, uint32_t *best_sad, uint32_t *best_mv);
/* orc_bipred_pack_batch over [n_sb][n_pu] arrays, n_pu = 85 or 209 */
void orc_bipred_pack_batch_npu(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref0_plane, uint32_t ref0_stride,
                               const int32_t *desc0, const uint8_t *ref1_plane, uint32_t ref1_stride, const int32_t *desc1,
                               uint32_t n_sb, const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1,
                               int n_lists, int bipred_8x8, int n_pu, svthip_me_cu_result *out);

/* ---- transform / quantisation (oracle/svt_tq_oracle.c) ---- */
/* aom_quantize_b{,_32x32,_64x64}_c_II (highbd = 0) / aom_highbd_quantize_b*_c (highbd = 1), flat qmatrix.
 * qp = {zbin[2], round[2], quant[2], quant_shift[2], dequant[2]} */
void orc_quantize_b(const int32_t *coeff, int32_t n_coeffs, const int16_t *qp, const int16_t *scan, int log_scale, int highbd,
                    int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob_ptr);

#ifdef __cplusplus
}
#endif
#endif
