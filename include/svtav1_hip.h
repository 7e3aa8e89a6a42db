/*
 * include/svtav1_hip.h -- C ABI of the MI355X (gfx950) motion-estimation / transform / quantisation
 * engine for the SVT-AV1 encoder pipeline.
 *
 * Plain C, no C++/torch types: this is what the reference's C host code binds.
 * Every entry point cites the reference interface it replaces; paths are under
 * Source/Lib/ of ateme-developers/SVT-AV1-1.
 *
 * Conventions
 *   - return value: 0 = EB_ErrorNone; a negative value carries an EbErrorType-compatible code
 *     (Source/API/EbApi.h:85-100), text via svthip_last_error().  Entry points never fall back to a CPU
 *     implementation: if the device or the code object is unavailable they fail.
 *   - `_dev` entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t passed as void*;
 *     NULL = the context's own stream) without synchronising; the host-pointer forms copy in, run,
 *     copy out and synchronise, so the caller's buffers are authoritative on return (SURVEY 8b ownership).
 *   - all planes are 8-bit luma, laid out exactly like EbPictureBufferDesc_t::bufferY: `stride` bytes per
 *     row, picture origin at (origin_x, origin_y) = the padding (68 / 32 / 16 px).
 */
#ifndef SVTAV1_HIP_H
#define SVTAV1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVTHIP_OK 0
#define SVTHIP_ERR_INSUFFICIENT_RESOURCES ((int32_t)0x80001000) /* EB_ErrorInsufficientResources */
#define SVTHIP_ERR_BAD_PARAMETER ((int32_t)0x80001005)          /* EB_ErrorBadParameter */
#define SVTHIP_ERR_DEVICE ((int32_t)0x80002000)                 /* no GPU / HIP failure (no CPU fallback) */

#define SVTHIP_NUM_SQ_PU 85          /* MAX_ME_PU_COUNT square part: 64x64, 4x32x32, 16x16x16, 64x8x8 */
#define SVTHIP_MAX_SAD_VALUE (128u * 128u * 255u) /* Codec/EbMotionEstimation.h:78 */

typedef struct svthip_ctx svthip_ctx;

/* One context per ME/EncDec thread context (MeContext_t / EncDecContext_t): owns a HIP stream and
 * device scratch; re-entrant across contexts (SURVEY 8b threading): any number of threads may call concurrently, each on
 * its own context, with no lock on the hot path.  ONE context is used by one thread at a time.  Every entry makes the
 * context's device current for the calling thread.  Entries that use context-owned scratch (the whole-picture ME, the
 * 209-PU bi-prediction, the host-pointer forms) order themselves on the stream they are given: if a call passes a different
 * stream than the previous scratch-using call of the same context, the new stream waits for the old one's work.
 * Created where the reference builds MeContext_t (Codec/EbMotionEstimationContext.c:28-122). */
int32_t svthip_create(int32_t device, svthip_ctx **out_ctx);
void svthip_destroy(svthip_ctx *ctx);
const char *svthip_last_error(void);
/* HIP stream of the context (hipStream_t as void*), for callers that order their own copies. */
void *svthip_stream(svthip_ctx *ctx);
int32_t svthip_synchronize(svthip_ctx *ctx);

/* Pre-size the context-owned device scratch for whole-picture ME calls of up to n_jobs pictures of width x height with n_pu (85 or
 * 209) PUs per SB, so that no later call allocates (where the reference sizes MeContext_t's buffers once in MeContextCtor,
 * Codec/EbMotionEstimationContext.c:28-122).  host_forms != 0 also sizes what svthip_motion_estimate_picture (host pointers) needs.
 * Optional: every entry grows its scratch on demand, stream-ordered (hipMallocAsync / hipFreeAsync on the stream that last used the
 * context's scratch -- never a device-wide synchronisation under other contexts' work). */
int32_t svthip_reserve(svthip_ctx *ctx, uint32_t width, uint32_t height, uint32_t n_pu, uint32_t n_jobs, int32_t host_forms);

/* Kernel-selection overrides of ONE context, default 0.  Two entries have a specialised kernel for the common shapes and a general
 * one for the rest; setting the option routes the common shapes through the general kernel too, which is how the tests cross-check the
 * two against each other.  Nothing is read from the environment. */
#define SVTHIP_OPT_SADLOOP_GENERIC 0 /* svthip_sad_loop_batch_dev: one-position-per-lane kernel for every width */
#define SVTHIP_OPT_CONVOLVE_VALU 1   /* svthip_av1_convolve_*_batch_dev: vector-unit kernel also for sides that are multiples of 32 */
#define SVTHIP_OPT_TQ_MAX_WORKGROUPS 2 /* svthip_encode_tu[16]_batch_dev: upper bound on the workgroups of a launch (0 = none).  The kernels walk
                                        * their transform units with whatever grid they get (a large batch already runs on as many workgroups
                                        * as the chip holds, each wave walking several groups with its operands prefetched one group ahead);
                                        * a small bound makes a small batch take that path, which is how the tests cover it */
#define SVTHIP_OPT_COUNT 3
int32_t svthip_set_option(svthip_ctx *ctx, int32_t option, int32_t value);

/* ---------------------------------------------------------------------------------------------
 * Full-pel 85-PU search of a batch of superblocks against one reference list.
 * Replaces FullPelSearch_LCU + GetEightHorizontalSearchPointResultsAll85PUs + GetSearchPointResults
 * (Codec/EbMotionEstimation.c:1504-1551, :1369-1499, :1237-1364) and the leaf kernels behind
 * GetEightHorizontalSearchPointResults_8x8_16x16_funcPtrArray / _32x32_64x64_funcPtrArray
 * (Codec/EbComputeSAD.h:183-212) and SadCalculation_*_funcPtrArray (Codec/EbMeSadCalculation.h:97-120),
 * asm_type = ASM_NON_AVX2 semantics, for all SBs of an ME segment in one launch.
 *
 * desc[i] (six int32 per SB):
 *   [0] src_offset : byte offset of the SB's top-left source sample in the source plane
 *                    (== MeContext_t::sb_src_ptr - bufferY, Codec/EbMotionEstimationProcess.c:514);
 *                    must be a multiple of 4 (always true for the reference's planes)
 *   [1] ref_offset : byte offset of search position (0,0) in the reference plane
 *                    (== integer_buffer_ptr + 2 + 2*stride - bufferY, Codec/EbMotionEstimation.c:1379)
 *   [2] x_search_area_origin, [3] y_search_area_origin : full-pel, relative to the SB origin (:6725-6726)
 *   [4] search_area_width, [5] search_area_height      : 1..127 (:6644-6645, after edge clipping :6689-6723)
 *
 * best_sad / best_mv : [n_sb][85] in ME-buffer order (p_sb_best_sad/p_sb_best_mv, PU 0 = 64x64, 1-4 = 32x32,
 *   5-20 = 16x16 z-order, 21-84 = 8x8 as 21 + 4*z16 + raster-in-16x16).  Both arrays are fully
 *   overwritten: the search starts from MAX_SAD_VALUE like MotionEstimateLcu does (:6820).
 *   SADs are the reference's vertically 2:1 sub-sampled, doubled SADs; MV word =
 *   (uint16)(4*y) << 16 | (uint16)(4*x) with strict-'<' raster-order tie breaking (first minimum wins).
 */
typedef struct svthip_fullpel_desc {
    int32_t src_offset;
    int32_t ref_offset;
    int32_t x_search_area_origin;
    int32_t y_search_area_origin;
    int32_t search_area_width;
    int32_t search_area_height;
} svthip_fullpel_desc;

int32_t svthip_me_fullpel_search_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                     const uint8_t *d_ref_plane, uint32_t ref_stride,
                                     const svthip_fullpel_desc *d_desc, uint32_t n_sb, uint32_t max_search_area_width,
                                     uint32_t max_search_area_height, uint32_t *d_best_sad, uint32_t *d_best_mv,
                                     void *stream);

/* Host-pointer form: planes are caller-owned host buffers of `src_plane_bytes` / `ref_plane_bytes`. */
int32_t svthip_me_fullpel_search(svthip_ctx *ctx, const uint8_t *src_plane, size_t src_plane_bytes, uint32_t src_stride,
                                 const uint8_t *ref_plane, size_t ref_plane_bytes, uint32_t ref_stride,
                                 const svthip_fullpel_desc *desc, uint32_t n_sb, uint32_t *best_sad, uint32_t *best_mv);


/* ---------------------------------------------------------------------------------------------
 * Hierarchical ME (HME): search-centre derivation for a batch of superblocks against one list.
 * Replaces the first half of MotionEstimateLcu (Codec/EbMotionEstimation.c:6300-6738):
 * hme_mv_center_check (:5882-6145), HmeLevel0/1/2 (:4306-4758) over the 2x2 search regions, the
 * best-region pick (:6573-6631), CheckZeroZeroCenter (:5466-5552) and the search-window clipping
 * (:6667-6723).  Output is the svthip_fullpel_desc array consumed by svthip_me_fullpel_search*.
 *
 * The three planes of a picture are what EbPaReferenceObject_t holds (Codec/EbReferenceObject.c:220-258):
 * padded full-resolution luma (origin 68,68), "quarter" (every 2nd pixel/row, origin 32,32) and
 * "sixteenth" (every 4th, origin 16,16).  Offsets are in bytes from `pool`, so one device buffer can
 * hold many pictures (the reference's picture pools).
 */
typedef struct svthip_pa_picture {
    int64_t full_offset;      /* byte offset of the padded full-res plane's first byte in the pool */
    int64_t quarter_offset;
    int64_t sixteenth_offset;
    uint32_t full_stride, quarter_stride, sixteenth_stride;
    uint16_t width, height;   /* luma_width / luma_height (multiples of 8) */
} svthip_pa_picture;

/* MeContext_t search parameters (Codec/EbMotionEstimationProcess.c:94-156) + the picture-level
 * signals MotionEstimateLcu reads from PictureParentControlSet_t. */
typedef struct svthip_me_params {
    uint16_t search_area_width, search_area_height;                 /* full-pel search area (<=127 used) */
    uint16_t number_hme_search_region_in_width, number_hme_search_region_in_height; /* 1..2 each */
    uint16_t hme_level0_total_search_area_width, hme_level0_total_search_area_height;
    uint16_t hme_level0_search_area_in_width_array[2], hme_level0_search_area_in_height_array[2];
    uint16_t hme_level1_search_area_in_width_array[2], hme_level1_search_area_in_height_array[2];
    uint16_t hme_level2_search_area_in_width_array[2], hme_level2_search_area_in_height_array[2];
    uint32_t hme_level0_multiplier_x, hme_level0_multiplier_y;      /* HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_X/Y[hier][tl] */
    uint8_t enable_hme_flag, enable_hme_level0_flag, enable_hme_level1_flag, enable_hme_level2_flag;
    uint8_t temporal_layer_index;
    uint8_t is_used_as_reference_flag;
    uint8_t ref_poc_equal;   /* ref0Poc == ref1Poc: list 1 takes the second-best HME L2 region (:6606-6631) */
    uint8_t reserved;
} svthip_me_params;

/* One SB of the batch: origin in luma samples (multiples of 64). */
typedef struct svthip_sb_origin {
    uint16_t x, y;
} svthip_sb_origin;

/* d_l0_best_mv64: for list_index 1, the final list-0 MV word of the 64x64 PU of every SB
 * (p_sb_best_mv[0][0][0], used by hme_mv_center_check :6076-6077); may be NULL for list 0.  SB i's word is
 * d_l0_best_mv64[i * l0_mv_stride] (stride 1 for a packed array, 85 when pointing at a [n_sb][85] MV array).
 * d_hme_state  : [n_sb][SVTHIP_HME_STATE_INT16] int16 scratch carried from the list-0 call to the list-1 call
 *                of the same SBs.  The reference keeps the per-region centre arrays (and the loop counters that
 *                guard their initialisation, :6325-6345) alive across its list loop, so list 1 starts from list 0's
 *                values whenever an HME level is disabled.  May be NULL when every enabled level is on (default).
 * Outputs: d_desc[n_sb] (ready for svthip_me_fullpel_search_dev with the SAME pool as both planes),
 *          d_center[n_sb] = (int16 x, int16 y) final search centre, for inspection (may be NULL).
 * `cur`, `ref`, `params` are HOST structs (passed by value to the kernel); d_* are device pointers.
 * The pool must stay readable 64 bytes past the end of its last plane (windows are staged with aligned 16-byte groups). */
#define SVTHIP_HME_STATE_INT16 25
#define SVTHIP_HME_MAX_JOBS 32
int32_t svthip_me_hme_search_center_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                        const svthip_pa_picture *ref, const svthip_me_params *params,
                                        uint32_t list_index, const svthip_sb_origin *d_sb, uint32_t n_sb,
                                        const uint32_t *d_l0_best_mv64, uint32_t l0_mv_stride, svthip_fullpel_desc *d_desc,
                                        int16_t *d_center, int16_t *d_hme_state, void *stream);

/* The same for n_jobs (current, reference) picture pairs of equal size in ONE launch (a segment of pictures handed to the ME
 * stage together): `cur` / `ref` are HOST arrays of n_jobs descriptors; every per-SB device array holds the jobs back to back,
 * job j's SB i at index j * n_sb + i (d_desc, d_center, d_hme_state, d_l0_best_mv64 * l0_mv_stride).  One launch keeps all
 * 256 CUs busy where a single 1080p picture (510 workgroups) cannot. */
int32_t svthip_me_hme_search_center_batch_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                              const svthip_pa_picture *ref, uint32_t n_jobs, const svthip_me_params *params,
                                              uint32_t list_index, const svthip_sb_origin *d_sb, uint32_t n_sb,
                                              const uint32_t *d_l0_best_mv64, uint32_t l0_mv_stride, svthip_fullpel_desc *d_desc,
                                              int16_t *d_center, int16_t *d_hme_state, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Sub-pel refinement (half-pel then quarter-pel) of the 85 square PUs of a batch of superblocks, one list.
 * Replaces InterpolateSearchRegionAVC + HalfPelSearch_LCU + QuarterPelSearch_LCU and the kernels behind them
 * (Codec/EbMotionEstimation.c:1707-1835, :2246-2786, :3337-4114; AvcStyleLumaInterpolationFilter* in
 * ASM_SSSE3/EbAvcStyleMcp_Intrinsic_SSSE3.c, SpatialFullDistortionKernel* in
 * ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c, NxMSadAveragingKernel / CombinedAveragingSSD), in the
 * configuration MotionEstimateLcu uses when use_subpel_flag = 1 at enc modes M0/M1 (:6857-6964): SSD_SEARCH
 * metric, half-pel on every PU size incl. 64x64, quarter-pel on.
 *
 * d_desc is the SAME descriptor array the full-pel search consumed; d_best_sad / d_best_mv ([n_sb][85], ME-buffer
 * order) hold the full-pel results on entry and the refined results on return (in place, like p_sb_best_sad/mv).
 * disable_8x8_refinement = (cu8x8_mode == CU_8x8_MODE_1): 8x8 PUs keep their full-pel result. */
int32_t svthip_me_subpel_refine_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                    const uint8_t *d_ref_plane, uint32_t ref_stride, const svthip_fullpel_desc *d_desc,
                                    uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                    int32_t disable_8x8_refinement, uint32_t *d_best_sad, uint32_t *d_best_mv, void *stream);

/* The same refinement with the distortion selectable like MeContext_t::fractionalSearchMethod (Codec/EbMotionEstimationContext.h:392,
 * values Codec/EbDefinitions.h:1846-1848), for the 85 squares (all_pu = 0) or all 209 PUs (all_pu != 0, [n_sb][209] arrays):
 *   SUB_SAD_SEARCH  : every candidate costs 2 x its SAD over every second row (NxMSadKernel / NxMSadAveragingKernel with doubled strides
 *                     and half the height, :1930-1931, :2915-2916), compared with and stored as the best SAD;
 *   FULL_SAD_SEARCH : the SAD over every row (:1932, :2917);
 *   SSD_SEARCH      : what MotionEstimateLcu hard-wires (:6254) and the two entries above / below compute.
 * The statements around the distortion -- candidate order, strict '<', direction choice, valid quarter-pel positions, buffer selection --
 * are shared by the three methods; with the SAD methods the reference's own HalfPelSearch_LCU + QuarterPelSearch_LCU can be executed in the
 * build container (no NASM-only symbol is reached) and this entry is checked against them. */
#define SVTHIP_FRACTIONAL_SUB_SAD_SEARCH 0
#define SVTHIP_FRACTIONAL_FULL_SAD_SEARCH 1
#define SVTHIP_FRACTIONAL_SSD_SEARCH 2
int32_t svthip_me_subpel_search_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                    const uint8_t *d_ref_plane, uint32_t ref_stride, const svthip_fullpel_desc *d_desc,
                                    uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                    int32_t disable_8x8_refinement, int32_t all_pu, int32_t fractional_search_method,
                                    uint32_t *d_best_sad, uint32_t *d_best_mv, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Bi-prediction SAD + result packing for a batch of superblocks.
 * Replaces the tail of MotionEstimateLcu (Codec/EbMotionEstimation.c:6973-7146): BiPredictionSearch /
 * BiPredictionCompensation / BiPredAverging / SelectBuffer / QuarterPelCompensation (:5261-5342, :5090-5254,
 * :4933-5081, :4762-4920) and the Sort3Elements-ordered fill of me_results[sb][pu] (:7047-7143).
 *
 * svthip_me_cu_result mirrors MeCuResults_t (Codec/EbMotionEstimationLcuResults.h:56-76) with the bit-fields
 * widened: out[sb][pu], pu = 0..84 in RASTER order (0: 64x64, 1-4: 32x32, 5-20: 16x16 raster, 21-84: 8x8 raster),
 * candidates sorted by distortion with '<=' ties favouring L0 then L1.  direction: 0 = UNI_PRED_LIST_0,
 * 1 = UNI_PRED_LIST_1, 2 = BI_PRED.  For P pictures (n_lists = 1) the list-1 MV fields are written as 0 (the
 * reference leaves whatever an earlier SB stored there).
 * d_desc0 / d_desc1: the descriptor arrays of the two lists (search-area origins, plane offsets);
 * d_sad*, d_mv*: final per-list results [n_sb][85] in ME-buffer order (after sub-pel refinement).
 * bipred_8x8: cu8x8_mode == CU_8x8_MODE_0 (8x8 PUs get a bi-pred candidate as well, :7028). */
typedef struct svthip_me_cu_result {
    int16_t xMvL0, yMvL0, xMvL1, yMvL1;
    uint32_t distortion[3];
    uint8_t direction[3];
    uint8_t totalMeCandidateIndex;
} svthip_me_cu_result;

int32_t svthip_me_bipred_pack_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                  const uint8_t *d_ref0_plane, uint32_t ref0_stride, const svthip_fullpel_desc *d_desc0,
                                  const uint8_t *d_ref1_plane, uint32_t ref1_stride, const svthip_fullpel_desc *d_desc1,
                                  uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                  const uint32_t *d_sad0, const uint32_t *d_mv0, const uint32_t *d_sad1, const uint32_t *d_mv1,
                                  uint32_t n_lists, int32_t bipred_8x8, svthip_me_cu_result *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-picture motion estimation: the batched equivalent of the SB loop of MotionEstimationKernel calling
 * MotionEstimateLcu (Codec/EbMotionEstimationProcess.c:478-556 -> Codec/EbMotionEstimation.c:6152-7162) for every SB
 * in d_sb: per list { search-centre (HME) -> full-pel 85-PU search -> sub-pel refinement }, then bi-prediction and
 * result packing.  This is the entry the ME process binds (one call per picture or per segment).
 *   ref1 = NULL          : P picture (one list, numOfListToSearch = 0, :6271)
 *   use_subpel_flag      : PictureParentControlSet_t::use_subpel_flag (:6857)
 *   cu8x8_mode           : 0 = CU_8x8_MODE_0 (8x8 PUs are sub-pel refined and bi-predicted), 1 = CU_8x8_MODE_1
 *   d_out                : [n_sb][85] svthip_me_cu_result, raster PU order (me_results[sb][pu])
 *   d_list_sad/d_list_mv : optional [2][n_sb][85] copies of p_sb_best_sad / p_sb_best_mv (ME-buffer order); may be NULL
 * All launches go to `stream` (NULL = context stream); the call does not synchronise.  Device scratch is owned by the
 * context and grows on demand (allocation happens only when a larger batch than ever before is submitted). */
int32_t svthip_motion_estimate_picture_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                           const svthip_pa_picture *ref0, const svthip_pa_picture *ref1,
                                           const svthip_me_params *params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                           const svthip_sb_origin *d_sb, uint32_t n_sb, svthip_me_cu_result *d_out,
                                           uint32_t *d_list_sad, uint32_t *d_list_mv, void *stream);

/* The same for n_jobs pictures of equal geometry and strides in one call (cur / ref0 / ref1: HOST arrays of n_jobs
 * descriptors, ref1 == NULL for P pictures): seven kernel launches whatever n_jobs is.  Per-SB device arrays hold the jobs
 * back to back: d_out[(j * n_sb + i) * 85 + pu]; d_list_sad / d_list_mv (optional) are [2][n_jobs * n_sb][85]. */
int32_t svthip_motion_estimate_batch_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                         const svthip_pa_picture *ref0, const svthip_pa_picture *ref1, uint32_t n_jobs,
                                         const svthip_me_params *params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                         const svthip_sb_origin *d_sb, uint32_t n_sb, svthip_me_cu_result *d_out,
                                         uint32_t *d_list_sad, uint32_t *d_list_mv, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batched quantisation + dequantisation + eob of transform units.
 * Replaces aom_quantize_b / aom_quantize_b_32x32 / aom_quantize_b_64x64 (RTCD, Codec/aom_dsp_rtcd.h:310-331; C bodies
 * aom_quantize_b*_c_II, Codec/EbFullLoop.c:109-143) and aom_highbd_quantize_b* (:301-336) as called by
 * av1_quantize_b_facade_II / av1_highbd_quantize_b_facade (:596-664) with the flat quantisation matrix.
 *
 * d_coeff / d_qcoeff / d_dqcoeff : int32 pools; TU i occupies [coeff_offset, coeff_offset + n_coeffs) in all three
 *                                  (coeff_offset multiple of 4; n_coeffs = 16, 64, 256 or 1024: transform sizes up to 32x32,
 *                                  64-point transforms are quantised on their 32x32 top-left part like the reference).
 * d_qparams  : int16 [n_rows][10] = zbin[2], round[2], quant[2], quant_shift[2], dequant[2] (index 0 DC, 1 AC): one row of
 *              Quants/Dequants per (qindex, plane) as built by av1_build_quantizer
 *              (Codec/EbModeDecisionConfigurationProcess.c:417-506); the host builds and uploads it once per picture.
 * d_iscan    : int16 pool of INVERSE scan tables (SCAN_ORDER::iscan of av1_scan_orders[tx_size][tx_type],
 *              Codec/EbFullLoop.c:826); iscan_offset (multiple of 4) selects the TU's table.
 * d_eob[i]   : uint16, 1 + last scan position with a non-zero level (0 = empty block).
 * Outputs are fully written (the reference memsets q/dq first, :60-61). */
typedef struct svthip_quant_desc {
    uint32_t coeff_offset;
    uint32_t iscan_offset;
    uint32_t qparam_index;
    uint16_t n_coeffs;
    uint8_t log_scale;  /* 0: up to 16x16-class, 1: 32x32-class, 2: 64x64-class (av1_get_tx_scale) */
    uint8_t highbd;     /* 0: 8-bit path (int16 clamp, quantize_b_helper_c_II), 1: high bit-depth path */
} svthip_quant_desc;

int32_t svthip_quantize_b_batch_dev(svthip_ctx *ctx, const int32_t *d_coeff, const svthip_quant_desc *d_desc, uint32_t n_tu,
                                    const int16_t *d_qparams, const int16_t *d_iscan, int32_t *d_qcoeff, int32_t *d_dqcoeff,
                                    uint16_t *d_eob, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched forward 2-D transform.  Replaces, per TU, Av1TransformTwoD_{4x4,8x8,16x16,32x32,64x64}_c and
 * av1_fwd_txfm2d_{WxH}_c (Source/Lib/Codec/EbTransforms.c:3928-4400; signature (int16_t *input, int32_t *output,
 * uint32_t inputStride, TxType transform_type, uint8_t bit_depth)), i.e. Av1TranformTwoDCore_c (:3701-3780) as configured
 * by Av1TransformConfig (:3847-3867), which Av1EstimateTransform (:4410-4728) dispatches to.
 *
 * One call transforms n_tu units of ONE size tx_width x tx_height (any of the 19 AV1 sizes: 4..64, aspect <= 4:1); the host
 * groups TUs by size, as the reference's per-size function pointers already do.
 * d_residual : int16 pool; TU i reads rows at in_offset + r * in_stride (elements), like `input` / `inputStride`.
 * d_coeff    : int32 pool; TU i writes tx_width * tx_height coefficients, row-major with row stride tx_width, at out_offset
 *              (multiple of 4) -- the reference's `output` layout before Av1EstimateTransform repacks 64-wide outputs.
 * tx_type    : TxType 0..15 (Source/Lib/Codec/EbDefinitions.h: DCT_DCT .. H_FLIPADST).  Only combinations for which the
 *              reference has a 1-D network are defined: ADST/FLIPADST up to 16 points, identity up to 32, 64-point DCT only.
 * bit_depth  : 8 or 10, as in the reference signature (it only feeds the reference's debug range checks; results are
 *              bit-identical to the reference for residuals of that depth, |residual| <= 2^bit_depth - 1). */
typedef struct svthip_txfm_desc {
    uint32_t in_offset;
    uint32_t out_offset;
    uint16_t in_stride;
    uint8_t tx_type;
    uint8_t reserved;
} svthip_txfm_desc;

int32_t svthip_fwd_txfm2d_batch_dev(svthip_ctx *ctx, const int16_t *d_residual, const svthip_txfm_desc *d_desc, uint32_t n_tu,
                                    uint32_t tx_width, uint32_t tx_height, uint32_t bit_depth, int32_t *d_coeff, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched inverse 2-D transform + reconstruction.  Replaces, per TU, av1_inv_txfm2d_add_{WxH}_c
 * (Source/Lib/Codec/EbTransforms.c:7714-7900; (const int32_t *input, uint16_t *output, int32_t stride, TxType tx_type,
 * [TxSize, [eob,]] int32_t bd)), which Av1InvTransformRecon / Av1InvTransformRecon8bit (:8344-8399) reach through
 * highbd_inv_txfm_add (:8252-8320) / av1_inv_txfm_add_c (:8321-8340, the 8-bit plane is widened, reconstructed and
 * narrowed again -- equivalent to reconstructing the 8-bit plane directly, which is what recon_16bit = 0 does).
 *
 * One call handles n_tu units of ONE size.  d_coeff: int32 pool of dequantised coefficients; TU i reads
 * min(W,32) x min(H,32) values, row stride min(W,32), at coeff_offset (multiple of 4) -- 64-point dimensions are stored
 * packed like the reference's input (:7736-7760).  d_recon: the prediction plane (uint8 when recon_16bit == 0, else
 * uint16), updated in place at recon_offset + r * recon_stride (elements) with clip(pred + residual) as the reference does.
 * TUs of one call must not overlap in d_recon. */
typedef struct svthip_itxfm_desc {
    uint32_t coeff_offset;
    uint32_t recon_offset;
    uint16_t recon_stride;
    uint8_t tx_type;
    uint8_t reserved;
} svthip_itxfm_desc;

int32_t svthip_inv_txfm2d_add_batch_dev(svthip_ctx *ctx, const int32_t *d_coeff, const svthip_itxfm_desc *d_desc, uint32_t n_tu,
                                        uint32_t tx_width, uint32_t tx_height, uint32_t bit_depth, uint32_t recon_16bit,
                                        void *d_recon, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused per-TU encode chain (8-bit planes).  One call does, for n_tu units of ONE size, what Av1EncodeLoop
 * (Source/Lib/Codec/EbCodingLoop.c:552-760) does per TU with ResidualKernel (Codec/EbPictureOperators.c:257-285),
 * Av1EstimateTransform (Codec/EbTransforms.c:4410-4728), Av1QuantizeInvQuantize (Codec/EbFullLoop.c:877-941),
 * FullDistortionKernel32Bits (Codec/EbPictureOperators.c:374-404) and Av1InvTransformRecon8bit
 * (Codec/EbTransforms.c:8374-8399), without the intermediate buffers going through memory.
 *
 * d_src / d_pred / d_recon : uint8 planes; TU i reads source and prediction at src_offset / pred_offset (+ r * stride) and
 *                writes clip(pred + inverse(dequantised)) at recon_offset.  d_recon may be d_pred (same offsets and stride):
 *                in-place reconstruction as the reference does.  TUs of one call must not overlap in d_recon.
 * coefficient pools (int32, 16-byte aligned; TU i at coeff_offset, multiple of 4, min(W,32) x min(H,32) values with row
 *                stride min(W,32) -- Av1EstimateTransform's packed layout):
 *                d_coeff (transform output, may be NULL), d_qcoeff (required), d_dqcoeff (may be NULL).
 * d_qparams / d_iscan : as svthip_quantize_b_batch_dev; log_scale is av1_get_tx_scale of the size.
 * d_eob[i]     : uint16 end of block (= y_count_non_zero_coeffs of the reference).
 * d_three_quad_energy[i] (may be NULL) : energy of the coefficients a 64-point dimension drops (HandleTransform64x64_c etc.,
 *                Codec/EbTransforms.c:3894-3926), 0 for the other sizes.
 * d_distortion (may be NULL) : uint64 [n_tu][2] = {sum (coeff - dqcoeff)^2, sum coeff^2} over the packed block
 *                (DIST_CALC_RESIDUAL, DIST_CALC_PREDICTION; the caller adds three_quad_energy and shifts as in
 *                Codec/EbFullLoop.c:1040-1045). */
typedef struct svthip_tu_desc {
    uint32_t src_offset;
    uint32_t pred_offset;
    uint32_t recon_offset;
    uint32_t coeff_offset;
    uint32_t iscan_offset;
    uint16_t src_stride;
    uint16_t pred_stride;
    uint16_t recon_stride;
    uint16_t qparam_index;
    uint8_t tx_type;
    uint8_t reserved[3];
} svthip_tu_desc;

int32_t svthip_encode_tu_batch_dev(svthip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                   const svthip_tu_desc *d_desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height,
                                   const int16_t *d_qparams, const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                                   int32_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_three_quad_energy, uint64_t *d_distortion,
                                   void *stream);

/* The same chain for 10-bit video held in 16-bit planes (offsets and strides of the descriptors in SAMPLES): high-bit-depth
 * quantiser (highbd_quantize_b_helper_c, no int16 clamp), inverse transform with bd = 10 (Av1InvTransformRecon,
 * Codec/EbTransforms.c:8344-8372), reconstruction clipped to 0..1023. */
int32_t svthip_encode_tu16_batch_dev(svthip_ctx *ctx, const uint16_t *d_src, const uint16_t *d_pred, uint16_t *d_recon,
                                     const svthip_tu_desc *d_desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height,
                                     const int16_t *d_qparams, const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                                     int32_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_three_quad_energy, uint64_t *d_distortion,
                                     void *stream);

/* 209-PU mode of the same search (pic_depth_mode <= PIC_ALL_C_DEPTH_MODE): open_loop_me_fullpel_search_sblock +
 * ExtSadCalculation_8x8_16x16 / ExtSadCalculation_32x32_64x64 / ExtSadCalculation
 * (Source/Lib/Codec/EbMotionEstimation.c:1556-1595, :1065-1231, :159-1052): the 85 square PUs plus the 124 rectangular ones
 * (64x32, 32x16, 16x8, 32x64, 16x32, 8x16, 32x8, 8x32, 64x16, 16x64), d_best_sad / d_best_mv = [n_sb][209] in the reference's
 * ME-buffer order (Codec/EbMotionEstimationContext.h:42-265), including the reference's stale-variable update of PU 92
 * (32x16[5], :343-347).  Same descriptors and window rules as svthip_me_fullpel_search_dev. */
int32_t svthip_me_fullpel_search209_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                        const uint8_t *d_ref_plane, uint32_t ref_stride, const svthip_fullpel_desc *d_desc,
                                        uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                        uint32_t *d_best_sad, uint32_t *d_best_mv, void *stream);

/* Sub-pel refinement of all 209 PUs: svthip_me_subpel_refine_dev's squares plus the rectangular half of HalfPelSearch_LCU
 * (Source/Lib/Codec/EbMotionEstimation.c:2418-2786) and of QuarterPelSearch_LCU (:3580-4114).  d_best_sad / d_best_mv =
 * [n_sb][209] in ME-buffer order, refined in place; every other argument as in svthip_me_subpel_refine_dev. */
int32_t svthip_me_subpel_refine209_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                       const uint8_t *d_ref_plane, uint32_t ref_stride, const svthip_fullpel_desc *d_desc,
                                       uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                       int32_t disable_8x8_refinement, uint32_t *d_best_sad, uint32_t *d_best_mv, void *stream);

/* Bi-prediction search and result packing over all 209 PUs (the loop of MotionEstimateLcu :6973-7146 with
 * max_number_of_pus_per_sb = 209; BiPredictionSearch :5261-5342 uses partitionWidth / partitionHeight / puSearchIndexMap,
 * Codec/EbMotionEstimation.h:177-322).  In this mode every PU gets a bi-prediction candidate whatever cu8x8_mode is
 * (:7028).  d_sad* / d_mv* = [n_sb][209] in ME-buffer order; d_out = [n_sb][209] in raster PU order (me_results), the
 * translation between the two being tab16x16 .. tab8x32 (EbMotionEstimation.h:89-171). */
int32_t svthip_me_bipred_pack209_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                     const uint8_t *d_ref0_plane, uint32_t ref0_stride, const svthip_fullpel_desc *d_desc0,
                                     const uint8_t *d_ref1_plane, uint32_t ref1_stride, const svthip_fullpel_desc *d_desc1,
                                     uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                     const uint32_t *d_sad0, const uint32_t *d_mv0, const uint32_t *d_sad1, const uint32_t *d_mv1,
                                     uint32_t n_lists, svthip_me_cu_result *d_out, void *stream);

/* MotionEstimateLcu (:6152) in the 209-PU mode for a batch of pictures: svthip_motion_estimate_batch_dev with the 209-PU
 * full-pel search, sub-pel refinement and bi-prediction.  d_out = [n_jobs][n_sb][209]; the optional d_list_sad /
 * d_list_mv = [2][n_jobs * n_sb][209]. */
int32_t svthip_motion_estimate209_batch_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                            const svthip_pa_picture *ref0, const svthip_pa_picture *ref1, uint32_t n_jobs,
                                            const svthip_me_params *params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                            const svthip_sb_origin *d_sb, uint32_t n_sb, svthip_me_cu_result *d_out,
                                            uint32_t *d_list_sad, uint32_t *d_list_mv, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Picture-analysis producers of the ME inputs, on the device.  The host uploads the padded full-resolution
 * luma ONCE (only its width x height interior has to be valid); this call then
 *   - replicates the picture edges into the 68-sample border of the full-resolution plane
 *     (PadPictureToMultipleOfLcuDimensions -> generate_padding, Codec/EbPictureAnalysisProcess.c:4866-4880, Codec/EbMcp.c:173-215),
 *   - writes the complete "quarter" plane (every 2nd sample / row, 32-sample border) when want_quarter != 0 and the complete
 *     "sixteenth" plane (every 4th, 16-sample border) when want_sixteenth != 0
 *     (DecimateInputPicture = Decimation2D + generate_padding, Codec/EbPictureAnalysisProcess.c:100-125, :4885-4936; the
 *     reference gates them on enable_hme_level1_flag / enable_hme_level0_flag),
 * for n_pics pictures of the pool in one launch.  pics: HOST array of descriptors (plane offsets / strides in d_pool); the
 * decimated planes' strides must be at least width/2 + 64 resp. width/4 + 32.  Bit-identical to the reference's two-pass
 * padding and its decimate-then-pad order (every output sample is the input sample at clamped coordinates). */
int32_t svthip_pa_derive_planes_dev(svthip_ctx *ctx, uint8_t *d_pool, const svthip_pa_picture *pics, uint32_t n_pics,
                                    int32_t want_quarter, int32_t want_sixteenth, void *stream);

/* generate_padding (sample_bytes = 1, Codec/EbMcp.c:173-215) / generate_padding16_bit (sample_bytes = 2, :220-262) of one plane
 * in place, as PadRefAndSetFlags applies them to a reconstructed reference picture (Codec/EbEncDecProcess.c:1135-1204).
 * d_plane points at the first sample of the PADDED plane; stride, width, height, pad_width, pad_height are in SAMPLES
 * (the reference's 16-bit variant takes bytes; this entry takes samples for both depths).  The width x height interior at
 * (pad_width, pad_height) is read, the border is written. */
int32_t svthip_pad_plane_dev(svthip_ctx *ctx, void *d_plane, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_width,
                             uint32_t pad_height, uint32_t sample_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SadLoopKernel for a batch of blocks: NxMSadLoopKernel_funcPtrArray[asm_type] (Codec/EbComputeSAD.h:183-189) = SadLoopKernel
 * (C_DEFAULT/EbComputeSAD_C.c:73-119; signature EB_SADLOOPKERNELNxM_TYPE, Codec/EbComputeSAD.h:38-51) with the reference's full
 * argument set, for n_blocks blocks per launch.  Block i: source block at d_src + src_offset (rows src_stride apart), search grid
 * origin at d_ref + ref_offset; candidate (x, y), 0 <= x < search_area_width, 0 <= y < search_area_height, compares block row r with
 * reference row y * ref_stride_raw + r * ref_stride at column x.  ref_stride must be ref_stride_raw or 2 * ref_stride_raw (the HME
 * callers skip every other row, Codec/EbMotionEstimation.c:4453-4470).  Result = the first minimum in raster order (strict '<' from
 * 0xffffff): d_best_sad[i] (the reference's *bestSad), d_best_xy[2 i] = xSearchCenter index, [2 i + 1] = ySearchCenter index.
 * width: multiple of 4, 4..64; height 1..64; search_area_width * search_area_height <= 4096; the per-block window must fit the LDS
 * slice (SVTHIP_ERR_BAD_PARAMETER otherwise).  BASELINE configs[0] = 16x16 blocks, 33x33 positions at 856x480. */
typedef struct svthip_sad_loop_desc {
    uint32_t src_offset;
    uint32_t ref_offset;
} svthip_sad_loop_desc;

int32_t svthip_sad_loop_batch_dev(svthip_ctx *ctx, const uint8_t *d_src, uint32_t src_stride, const uint8_t *d_ref, uint32_t ref_stride,
                                  uint32_t ref_stride_raw, const svthip_sad_loop_desc *d_desc, uint32_t n_blocks, uint32_t width,
                                  uint32_t height, uint32_t search_area_width, uint32_t search_area_height, uint32_t *d_best_sad,
                                  int16_t *d_best_xy, void *stream);

/* ---------------------------------------------------------------------------------------------
 * AV1 inter prediction: 8-bit single-reference convolutions of a batch of blocks of ONE size.
 * Per block, what av1_inter_prediction does for one plane of a uni-predicted block (Codec/EbInterPrediction.c:1255-1287):
 * convolve[subpel_x != 0][subpel_y != 0][0] = av1_convolve_2d_sr / av1_convolve_x_sr / av1_convolve_y_sr / av1_convolve_2d_copy_sr
 * (C bodies :145-286, RTCD Codec/aom_dsp_rtcd.h:2067-2076) with filter kernels chosen by
 * av1_get_interp_filter_params_with_block_size (:985-995: blocks <= 4 samples wide / high use the 4-tap tables) and
 * get_conv_params_no_round(.., is_compound = 0, EB_8BIT) rounding (round_0 = 3, round_1 = 11).
 *
 * d_src / d_dst : uint8 planes.  Block i reads around d_src + src_offset (the block's top-left sample AFTER the integer part of the
 *                 motion vector has been applied: srcPtr + (mv_q4.row >> 4) * stride + (mv_q4.col >> 4), :1265) -- 3 samples
 *                 left / above and 4 right / below when the respective phase is non-zero -- and writes width x height samples at
 *                 d_dst + dst_offset.  The source plane must stay readable 16 bytes past the last sample a block needs.
 * subpel_x / subpel_y : mv_q4 & SUBPEL_MASK, 0..15 (1/16 sample).
 * filter_x / filter_y : InterpFilter of each direction: 0 EIGHTTAP_REGULAR, 1 EIGHTTAP_SMOOTH, 2 MULTITAP_SHARP, 3 BILINEAR
 *                 (av1_extract_interp_filter(interp_filters, 1 / 0)).
 * width x height : one of the 22 AV1 block sizes (4..128, aspect <= 4:1; the host groups blocks by size). */
typedef struct svthip_convolve_desc {
    uint32_t src_offset;
    uint32_t dst_offset;
    uint8_t subpel_x, subpel_y, filter_x, filter_y;
    uint32_t reserved;
} svthip_convolve_desc;

int32_t svthip_av1_convolve_sr_batch_dev(svthip_ctx *ctx, const uint8_t *d_src, uint32_t src_stride, uint8_t *d_dst, uint32_t dst_stride,
                                         const svthip_convolve_desc *d_desc, uint32_t n_blocks, uint32_t width, uint32_t height, void *stream);

/* The same for bi-predicted (BI_PRED) blocks: what av1_inter_prediction does for the luma plane when mv_unit->predDirection == BI_PRED
 * (Codec/EbInterPrediction.c:1254-1290 and :1346-1385): list 0 through convolve[..][..][1] = av1_jnt_convolve_2d / _x / _y / _2d_copy
 * (C bodies :290-528) into the 16-bit buffer with get_conv_params_no_round(.., do_average = 0, is_compound = 1) (round_0 = 3,
 * round_1 = COMPOUND_ROUND1_BITS = 7), then list 1 with do_average = 1 averaged into the 8-bit destination (use_jnt_comp_avg = 0,
 * round_bits = 4).  One interp_filters pair serves both lists, as in the reference's call.  Block i reads around
 * d_src0 + src0_offset with (subpel_x0, subpel_y0) and around d_src1 + src1_offset with (subpel_x1, subpel_y1) under the window
 * rules of the single-reference entry, and writes width x height samples at d_dst + dst_offset. */
typedef struct svthip_convolve_compound_desc {
    uint32_t src0_offset;
    uint32_t src1_offset;
    uint32_t dst_offset;
    uint8_t subpel0; /* subpel_x0 | subpel_y0 << 4 */
    uint8_t subpel1; /* subpel_x1 | subpel_y1 << 4 */
    uint8_t filter_x, filter_y;
} svthip_convolve_compound_desc;

int32_t svthip_av1_convolve_compound_batch_dev(svthip_ctx *ctx, const uint8_t *d_src0, uint32_t src0_stride, const uint8_t *d_src1,
                                               uint32_t src1_stride, uint8_t *d_dst, uint32_t dst_stride,
                                               const svthip_convolve_compound_desc *d_desc, uint32_t n_blocks, uint32_t width,
                                               uint32_t height, void *stream);

/* The same two entries for 10-bit video held in 16-bit planes: convolveHbd[..][..][is_compound] = av1_highbd_convolve_{2d,x,y,2d_copy}_sr /
 * av1_highbd_jnt_convolve_* (Codec/EbInterPrediction.c:530-895) with get_conv_params_no_round(.., bd).  Offsets and strides are in
 * SAMPLES; d_desc is a svthip_convolve_desc array (compound = 0; d_src1 unused) or a svthip_convolve_compound_desc array (compound != 0).
 * bit_depth must be 10 (round_0 changes at 12 bits). */
int32_t svthip_av1_highbd_convolve_batch_dev(svthip_ctx *ctx, const uint16_t *d_src0, uint32_t src0_stride, const uint16_t *d_src1,
                                             uint32_t src1_stride, uint16_t *d_dst, uint32_t dst_stride, const void *d_desc,
                                             int32_t compound, uint32_t n_blocks, uint32_t width, uint32_t height, uint32_t bit_depth,
                                             void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batching layer for the transform / quantisation callers.  The reference calls its T/Q kernels one TU and one
 * transform type at a time from ProductFullLoopTxSearch (Codec/EbFullLoop.c:1138-1352: for every tx_type candidate of a TU:
 * Av1EstimateTransform -> Av1QuantizeInvQuantize -> distortion -> cost), encode_pass_tx_search (:1354-1550) and Av1EncodeLoop
 * (Codec/EbCodingLoop.c:552-913).  A batcher is the host-side gather / scatter that lets those callers keep their control flow:
 * they ADD every (TU, tx_type) candidate they would have evaluated, FLUSH once (descriptors are grouped by transform size, one
 * fused-chain launch per size present -- svthip_encode_tu[16]_batch_dev), and READ each candidate's eob / energy / distortion (and,
 * when wanted, its quantised levels) back by handle to make the same decision the serial loop makes.
 *
 * One batcher belongs to one context (one thread).  Planes and tables are DEVICE buffers bound with _begin (the picture's source
 * plane, the prediction plane, the reconstruction plane or NULL; qparams rows and the inverse-scan pool uploaded once per picture).
 * recon_offset = SVTHIP_TU_RECON_SCRATCH gives the candidate a private tile in batcher-owned scratch (tx-type search: several
 * candidates of the same TU must not overwrite each other, and none of them is the final reconstruction). */
typedef struct svthip_tu_batcher svthip_tu_batcher;
#define SVTHIP_TU_RECON_SCRATCH 0xffffffffu

typedef struct svthip_tu_result {
    uint64_t distortion[2];     /* sum (coeff - dqcoeff)^2, sum coeff^2 (DIST_CALC_RESIDUAL, DIST_CALC_PREDICTION) */
    uint64_t three_quad_energy; /* dropped 64-point quadrants (HandleTransform64x64_c) */
    uint32_t coeff_offset;      /* of this candidate's block in the batcher's coefficient pools */
    uint16_t eob;
    uint8_t tx_size, tx_type;
} svthip_tu_result;

int32_t svthip_tu_batcher_create(svthip_ctx *ctx, uint32_t max_candidates, uint32_t max_coeff_samples, svthip_tu_batcher **out);
void svthip_tu_batcher_destroy(svthip_tu_batcher *b);
int32_t svthip_tu_batcher_begin(svthip_tu_batcher *b, const void *d_src, const void *d_pred, void *d_recon, int32_t planes_16bit,
                                const int16_t *d_qparams, const int16_t *d_iscan);
/* tx_size: TxSize 0..18 (TX_4X4 .. TX_64X16, Codec/EbDefinitions.h); offsets / strides in samples, as in svthip_tu_desc */
int32_t svthip_tu_batcher_add(svthip_tu_batcher *b, uint32_t tx_size, uint32_t tx_type, uint32_t src_offset, uint32_t src_stride,
                              uint32_t pred_offset, uint32_t pred_stride, uint32_t recon_offset, uint32_t recon_stride, uint32_t qparam_index,
                              uint32_t iscan_offset, uint32_t *out_handle);
/* launches everything added since _begin (or the last flush) and waits; results stay readable until the next _begin */
int32_t svthip_tu_batcher_flush(svthip_tu_batcher *b);
int32_t svthip_tu_batcher_result(const svthip_tu_batcher *b, uint32_t handle, svthip_tu_result *out);
/* copies the candidate's min(W,32) x min(H,32) quantised levels / dequantised coefficients to host buffers (either may be NULL) */
int32_t svthip_tu_batcher_read_coeffs(svthip_tu_batcher *b, uint32_t handle, int32_t *qcoeff, int32_t *dqcoeff);
/* device addresses of the pools, for a caller that keeps the winner's levels on the device (entropy coding input) */
int32_t svthip_tu_batcher_pools(const svthip_tu_batcher *b, const int32_t **d_qcoeff, const int32_t **d_dqcoeff, const void **d_recon_scratch);

/* ---------------------------------------------------------------------------------------------
 * Reference-layout results and host-pointer forms (what a C host that owns host memory binds).
 *
 * svthip_me_cu_result_ref has the memory layout of the reference's MeCuResults_t (Codec/EbMotionEstimationLcuResults.h:56-76) as
 * GCC lays it out on x86-64: the four MV components, then three DistDir_t { unsigned distortion : 32; unsigned direction : 2; }
 * (8 bytes each: the distortion word, then a word whose 2 low bits are the direction), then totalMeCandidateIndex -- 40 bytes.
 * A host may hand its own MeCuResults_t arrays to the calls below; unused bits / padding bytes are written as 0. */
typedef struct svthip_me_cu_result_ref {
    int16_t xMvL0, yMvL0, xMvL1, yMvL1;
    struct {
        uint32_t distortion;
        uint32_t direction; /* 0 = UNI_PRED_LIST_0, 1 = UNI_PRED_LIST_1, 2 = BI_PRED (only the 2 low bits are defined in the reference) */
    } distortionDirection[3];
    uint8_t totalMeCandidateIndex;
    uint8_t pad_[7];
} svthip_me_cu_result_ref;

/* d_in [n] svthip_me_cu_result -> d_out [n] svthip_me_cu_result_ref */
int32_t svthip_me_results_to_ref_layout_dev(svthip_ctx *ctx, const svthip_me_cu_result *d_in, uint32_t n, svthip_me_cu_result_ref *d_out,
                                            void *stream);

/* The fields of an EbPictureBufferDesc_t (Codec/EbPictureBufferDesc.h) the host-pointer ME entry reads: the padded luma plane of
 * an EbPaReferenceObject_t::inputPaddedPicturePtr.  origin_x / origin_y must be 68 (Codec/EbEncHandle.c:1006-1009). */
typedef struct svthip_host_picture {
    const uint8_t *buffer_y;
    uint32_t stride_y;
    uint16_t origin_x, origin_y;
    uint16_t width, height;
} svthip_host_picture;

/* MotionEstimationKernel's SB loop (Codec/EbMotionEstimationProcess.c:478-556) for one whole picture with HOST buffers: uploads the
 * three luma planes (picture rows only), derives borders and the 1/4 and 1/16 planes on the device (svthip_pa_derive_planes_dev),
 * runs svthip_motion_estimate[209]_batch_dev over every SB in raster order, and writes me_results[sb][pu] in the reference's own
 * MeCuResults_t layout: me_results is the picture's `MeCuResults_t **me_results` (PictureParentControlSet_t), n_sb row pointers of
 * n_pu (85 or 209) records each.  ref1 = NULL for P pictures.  Synchronous; the host buffers are authoritative on return. */
int32_t svthip_motion_estimate_picture(svthip_ctx *ctx, const svthip_host_picture *cur, const svthip_host_picture *ref0,
                                       const svthip_host_picture *ref1, const svthip_me_params *params, int32_t use_subpel_flag,
                                       int32_t cu8x8_mode, uint32_t n_pu, void *const *me_results);

/* svthip_encode_tu_batch_dev with HOST buffers: planes of `plane_samples` samples each (8-bit when planes_16bit == 0, else 16-bit),
 * recon may alias pred (in-place reconstruction); qparams has n_qparam_rows rows of 10 int16; iscan n_iscan int16 entries;
 * coefficient outputs hold coeff_samples int32 each (qcoeff required; coeff / dqcoeff / three_quad_energy / distortion may be NULL).
 * Copies in, runs the fused chain, copies out and synchronises. */
int32_t svthip_encode_tu_batch(svthip_ctx *ctx, const void *src, const void *pred, void *recon, size_t plane_samples, int32_t planes_16bit,
                               const svthip_tu_desc *desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height, const int16_t *qparams,
                               uint32_t n_qparam_rows, const int16_t *iscan, uint32_t n_iscan, size_t coeff_samples, int32_t *coeff,
                               int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, uint64_t *three_quad_energy, uint64_t *distortion);

/* ---------------------------------------------------------------------------------------------
 * Open-loop intra search: OpenLoopIntraSearchLcu (Codec/EbMotionEstimation.c:8047-8355), which the ME process
 * runs for every SB right after MotionEstimateLcu (Codec/EbMotionEstimationProcess.c:558-579), for n_jobs pictures of equal
 * geometry in one launch: neighbour samples from the SOURCE picture (UpdateNeighborSamplesArrayOpenLoop,
 * Codec/EbIntraPrediction.c:5233), the 35 HEVC-style luma predictors (IntraPredictionOpenLoop, :5353), plain SAD, and the
 * candidate selection of the picture's branch:
 *   slice_is_intra                                    : 7 modes, best-mode based list (:8076-8153)
 *   temporal_layer_index == 0 && !input_resolution_4k : all 35 modes, best 18 sorted by SAD (:8219-8270)
 *   limit_ois_to_dc_mode_flag                         : DC only (OpenLoopIntraDC, :7951)
 *   otherwise                                         : DC SAD vs the CU's ME distortion -> OIS point -> 1..9 stage-1 modes ->
 *                                                       injected list (GetInterIntraSadDistance / GetOisPoint /
 *                                                       InjectIntraCandidatesBasedOnBestMode, :7525-7852); needs d_me
 * cur: HOST array of n_jobs picture descriptors (only the full-resolution plane is read; origin (68,68)).
 * d_me: the ME results of the same (job, SB) items, [n_jobs * n_sb][me_pu_stride] (85 or 209) in raster PU order, of which
 *       entry [cu].distortion[0] is read for cu = 1..84 (me_results[sb][rasterScanCuIndex].distortionDirection[0], :8291).
 * Outputs, per (job, SB) item and raster CU index 0..84 (RASTER_SCAN_CU_INDEX, 0 = the unused 64x64 slot):
 *   d_cand  [n_jobs * n_sb][85][18] : OisCandidate_t words (distortion : 20, valid_distortion : 1, - : 3, intra_mode : 8;
 *                                     Codec/EbCodingUnit.h:303-313), i.e. sorted_ois_candidate[cu][0..17]
 *   d_total [n_jobs * n_sb][85]     : total_intra_luma_mode[cu]
 * Every word is written; fields the reference leaves untouched (it keeps what an earlier picture stored there: e.g.
 * distortion of candidates 1..8 on the general branch, total_intra_luma_mode of 32x32 CUs of I pictures, everything of CUs
 * that are not wholly inside the picture) are written as 0, so a host copy-out should copy only what the branch defines
 * (INTEGRATION.md 1.8).  ASM_NON_AVX2 semantics (the AVX2 DC-only shortcut UpdateNeighborDcIntraPred_AVX2_INTRIN is not
 * reproduced). */
typedef struct svthip_ois_params {
    uint8_t slice_is_intra;             /* slice_type == I_SLICE */
    uint8_t temporal_layer_index;       /* 0..5 */
    uint8_t is_used_as_reference_flag;
    uint8_t input_resolution_4k;        /* sequence input_resolution == INPUT_SIZE_4K_RANGE */
    uint8_t limit_ois_to_dc_mode_flag;
    uint8_t cu8x8_mode;                 /* 1 = CU_8x8_MODE_1: 8x8 CUs are skipped on non-intra pictures (:8203) */
    uint8_t enc_mode;                   /* only read for the vertical winner's valid flag (:7609) */
    uint8_t reserved;
} svthip_ois_params;

int32_t svthip_open_loop_intra_search_batch_dev(svthip_ctx *ctx, const uint8_t *d_pool, const svthip_pa_picture *cur,
                                                uint32_t n_jobs, const svthip_ois_params *params,
                                                const svthip_sb_origin *d_sb, uint32_t n_sb, const svthip_me_cu_result *d_me,
                                                uint32_t me_pu_stride, uint32_t *d_cand, uint8_t *d_total, void *stream);

/* OpenLoopIntraSearchLcu for every SB of one picture with HOST buffers (the second SB loop of MotionEstimationKernel,
 * Codec/EbMotionEstimationProcess.c:558-579): uploads the luma interior (any origin: the search never reads outside the picture),
 * takes distortionDirection[0].distortion of entries 0..84 of every SB's me_results row (me_results[sb] -> MeCuResults_t[n_pu],
 * host memory, 40-byte layout above; may be NULL on the branches that do not read it), runs
 * svthip_open_loop_intra_search_batch_dev and returns cand [n_sb][85][18] / total [n_sb][85] in host memory.  Synchronous. */
int32_t svthip_open_loop_intra_search_picture(svthip_ctx *ctx, const svthip_host_picture *cur, const svthip_ois_params *params,
                                              const void *const *me_results, uint32_t n_pu, uint32_t *cand, uint8_t *total);


/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: superblocks of a picture sharded over the GPUs of one node, RCCL over xGMI for the one real exchange.
 * One rank per GPU -- a thread of the reference's single process (its ME / EncDec threads already own one context each) or one
 * process per GPU.  Every rank holds the read-only source planes, so ME needs NO data-path collective; the encode pass's T/Q shards by
 * SB-row slab, and the reconstructed slabs are exchanged so that every rank holds the whole padded reference picture for the next
 * picture's inter prediction -- what PadRefAndSetFlags (Codec/EbEncDecProcess.c:1135-1204, called :1852) finishes on one host.
 *
 * Partition (pure host arithmetic, callable without a device):
 *   svthip_shard_range     : contiguous share of n_units (superblocks for ME, balanced to ONE unit: 1080p = 510 SBs -> 64/64/64/64/64/64/63/63)
 *   svthip_recon_slab_rows : luma rows [first_row, first_row + n_rows) of rank's SB-row slab (1080p = 17 SB rows -> 3/2/2/2/2/2/2/2 rows of 64)
 * Plans: the byte ranges a rank sends / receives, as the device entries below will issue them; a test (or another transport) can
 * execute the same list.  offset / bytes are relative to the plane (`plane` 0 = Y, 1 = Cb, 2 = Cr) resp. to d_full / d_local (`plane` = job). */
typedef struct svthip_comm svthip_comm;
#define SVTHIP_COMM_ID_BYTES 128 /* ncclUniqueId */

void svthip_shard_range(uint32_t n_units, int32_t world, int32_t rank, uint32_t *first, uint32_t *count);
void svthip_recon_slab_rows(uint32_t height, int32_t world, int32_t rank, uint32_t *first_row, uint32_t *n_rows);

/* The planes of an EbPictureBufferDesc_t reference picture (EbReferenceObject_t::referencePicture / referencePicture16bit) on the
 * device: pointers to the first sample of the PADDED planes; luma origin (origin_x, origin_y), chroma (4:2:0) origin, size and
 * slab rows are the luma values >> 1 exactly as PadRefAndSetFlags passes them.  cb == cr == NULL: luma only. */
typedef struct svthip_recon_picture {
    void *y, *cb, *cr;
    uint32_t stride_y, stride_cb, stride_cr; /* samples */
    uint16_t width, height;                  /* luma, even */
    uint16_t origin_x, origin_y;             /* luma padding (the reference allocates 160 + for reference pictures, Codec/EbEncHandle.c) */
    uint8_t sample_bytes;                    /* 1: 8-bit planes, 2: 16-bit planes */
    uint8_t reserved[3];
} svthip_recon_picture;

typedef struct svthip_xfer {
    int32_t peer;
    uint32_t plane;
    uint32_t send; /* 1: this rank -> peer, 0: peer -> this rank */
    uint64_t offset, bytes;
} svthip_xfer;

/* n_xfers receives the number of transfers; out may be NULL to query it. */
int32_t svthip_recon_exchange_plan(const svthip_recon_picture *pic, int32_t world, int32_t rank, svthip_xfer *out, uint32_t max_xfers,
                                   uint32_t *n_xfers);
int32_t svthip_me_gather_plan(uint32_t n_sb_total, uint32_t n_jobs, uint32_t record_bytes, int32_t world, int32_t rank, svthip_xfer *out,
                              uint32_t max_xfers, uint32_t *n_xfers);

/* Communicator of one rank, bound to the context whose device it uses.  Rank 0 calls svthip_comm_get_unique_id and hands the 128 bytes
 * to the other ranks by whatever the host has (shared memory between the reference's threads; a file, socket or launcher store between
 * processes); every rank then calls svthip_comm_create (collective: ncclCommInitRank).  world == 1 needs no id and never touches RCCL. */
int32_t svthip_comm_get_unique_id(uint8_t *id /* [SVTHIP_COMM_ID_BYTES] */);
int32_t svthip_comm_create(svthip_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, svthip_comm **out);
void svthip_comm_destroy(svthip_comm *comm);
int32_t svthip_comm_rank(const svthip_comm *comm);
int32_t svthip_comm_world(const svthip_comm *comm);
const char *svthip_comm_last_error(void);

/* Reconstructed-picture exchange + PadRefAndSetFlags.  On entry every rank's planes hold ITS slab (svthip_recon_slab_rows; chroma rows
 * >> 1) reconstructed in place; on return (stream-ordered, no host synchronisation) every rank's planes hold the whole picture with
 * its borders replicated (generate_padding / generate_padding16_bit of Y, Cb, Cr, Codec/EbMcp.c:173-262).  xGMI is a point-to-point
 * mesh, so the exchange is ONE RCCL group of direct sends / receives: each slab goes over its own link straight into its place in the
 * peer's plane (whole rows, side borders included -- they are rewritten by the padding), no staging buffer. */
int32_t svthip_recon_exchange_dev(svthip_comm *comm, const svthip_recon_picture *pic, void *stream);

/* ME results of a frame-sharded search on every rank (for a consumer that needs all of me_results everywhere; the ME itself needs no
 * exchange): d_local = this rank's [n_jobs][count][record_bytes] (count from svthip_shard_range(n_sb_total, ..)), d_full =
 * [n_jobs][n_sb_total][record_bytes] on every rank.  record_bytes = n_pu * sizeof(svthip_me_cu_result). */
int32_t svthip_me_gather_results_dev(svthip_comm *comm, const void *d_local, void *d_full, uint32_t n_jobs, uint32_t n_sb_total,
                                     uint32_t record_bytes, void *stream);

/* Launch-duration probe for bench.py: runs the same launch `iters` times on the context stream between
 * two HIP events and returns the average kernel time in milliseconds (inputs are device pointers). */
int32_t svthip_me_fullpel_search_time_dev(svthip_ctx *ctx, const uint8_t *d_src_plane, uint32_t src_stride,
                                          const uint8_t *d_ref_plane, uint32_t ref_stride,
                                          const svthip_fullpel_desc *d_desc, uint32_t n_sb,
                                          uint32_t max_search_area_width, uint32_t max_search_area_height,
                                          uint32_t *d_best_sad, uint32_t *d_best_mv, uint32_t iters, float *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SVTAV1_HIP_H */
