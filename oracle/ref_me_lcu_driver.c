/*
 * oracle/ref_me_lcu_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the REFERENCE's own MotionEstimateLcu (Source/Lib/Codec/EbMotionEstimation.c:6152) standalone,
 * compiled from /root/reference into oracle/_ref/libsvtref_me.so by oracle/build_ref.sh.  This file only
 * builds the structures the function reads (the field list is SURVEY.md section 8c) and stages the SB
 * buffers the way MotionEstimationKernel does (Codec/EbMotionEstimationProcess.c:481-552); it compiles
 * against the reference's headers where they lie and contains no reference code.
 *
 * Limits (see DESIGN.md "oracle"): the reference's 9 NASM files cannot be assembled in this image and no
 * stand-ins are written, so the library keeps `Log2f_SSE2` and `PictureCopyKernel_SSE2` unresolved
 * (lazy binding).  `Log2f_SSE2` is called by PU_HalfPelRefinement (:1912), therefore this driver can only
 * be used with use_subpel_flag = 0: it pins the centre checks, HME L0/L1/L2, region pick, window clipping,
 * the full-pel 85-PU search and the result packing -- not the sub-pel refinement.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbEncodeContext.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationContext.h"
#include "EbPictureBufferDesc.h"
#include "EbPictureControlSet.h"
#include "EbReferenceObject.h"
#include "EbSequenceControlSet.h"
#include "EbSystemResourceManager.h"

extern EbMemoryMapEntry *memoryMap;
extern uint32_t *memoryMapIndex;
extern uint64_t *totalLibMemory;

static EbPictureBufferDesc_t *make_desc(uint8_t *buf, int stride, int origin, int width, int height)
{
    EbPictureBufferDesc_t *d = (EbPictureBufferDesc_t *)calloc(1, sizeof(*d));
    d->bufferY = buf;
    d->strideY = (uint16_t)stride;
    d->origin_x = d->origin_y = (uint16_t)origin;
    d->width = (uint16_t)width;
    d->height = (uint16_t)height;
    d->maxWidth = (uint16_t)width;
    d->maxHeight = (uint16_t)height;
    return d;
}

static EbObjectWrapper_t *make_pa_ref(uint8_t *full, uint8_t *quarter, uint8_t *sixteenth, int w, int h)
{
    EbPaReferenceObject_t *o = (EbPaReferenceObject_t *)calloc(1, sizeof(*o));
    o->inputPaddedPicturePtr = make_desc(full, w + 136, 68, w, h);
    o->quarterDecimatedPicturePtr = make_desc(quarter, (w >> 1) + 64, 32, w >> 1, h >> 1);
    o->sixteenthDecimatedPicturePtr = make_desc(sixteenth, (w >> 2) + 32, 16, w >> 2, h >> 2);
    EbObjectWrapper_t *wr = (EbObjectWrapper_t *)calloc(1, sizeof(*wr));
    wr->objectPtr = o;
    return wr;
}

enum {
    IP_SEARCH_W, IP_SEARCH_H, IP_NHME_W, IP_NHME_H, IP_L0_TOTAL_W, IP_L0_TOTAL_H,
    IP_L0_W0, IP_L0_W1, IP_L0_H0, IP_L0_H1, IP_L1_W0, IP_L1_W1, IP_L1_H0, IP_L1_H1, IP_L2_W0, IP_L2_W1, IP_L2_H0, IP_L2_H1,
    IP_EN_HME, IP_EN_L0, IP_EN_L1, IP_EN_L2, IP_TWO_LISTS, IP_TEMPORAL_LAYER, IP_HIER_LEVELS, IP_IS_REF, IP_USE_SUBPEL,
    IP_REF0_POC, IP_REF1_POC, IP_ASM_TYPE, IP_ALL_PU, IP_RES_4K, IP_COUNT
};

/* planes: [0]=current, [1]=list-0 reference, [2]=list-1 reference; each {full, quarter, sixteenth}.
 * ip[IP_ALL_PU] selects the 209-PU mode (pic_depth_mode PIC_ALL_DEPTH_MODE, max_number_of_pus_per_sb 209: the
 * open_loop_me_fullpel_search_sblock branch of :6745-6813 and bi-prediction over all 209 PUs, :7028); NPU = 209 or 85.
 * out_sad / out_mv : [n_sb][2][NPU] ME-buffer order; out_origin: [n_sb][2][4] = x_origin, y_origin (as left in
 * MeContext_t) and two spare; out_res: [n_sb][NPU][11] = xMvL0,yMvL0,xMvL1,yMvL1, dist0,dir0, dist1,dir1, total,
 * dist2,dir2.  Returns 0, or a negative code. */
int ref_me_lcu_run(uint8_t **planes, int width, int height, const int32_t *ip, uint32_t *out_sad, uint32_t *out_mv,
                   int32_t *out_origin, int32_t *out_res)
{
    static EbMemoryMapEntry *mm = NULL;
    static uint32_t mm_index;
    static uint64_t mm_total;
    if (!mm) mm = (EbMemoryMapEntry *)calloc(1 << 16, sizeof(EbMemoryMapEntry));
    memoryMap = mm;
    mm_index = 0;
    memoryMapIndex = &mm_index;
    totalLibMemory = &mm_total;

    if (ip[IP_USE_SUBPEL]) return -2; /* would call Log2f_SSE2 (NASM), unavailable */

    SequenceControlSet_t *scs = (SequenceControlSet_t *)calloc(1, sizeof(*scs));
    EncodeContext_t *ec = (EncodeContext_t *)calloc(1, sizeof(*ec));
    scs->encode_context_ptr = ec;
    ec->asm_type = (EbAsm)ip[IP_ASM_TYPE];
    scs->luma_width = (uint16_t)width;
    scs->luma_height = (uint16_t)height;
    scs->sb_sz = 64;
    scs->input_resolution = ip[IP_RES_4K] ? INPUT_SIZE_4K_RANGE  /* a crop of a 3840x2160 sequence keeps its sequence's resolution class */
                           : (width * height < INPUT_SIZE_1080i_TH) ? INPUT_SIZE_576p_RANGE_OR_LOWER : INPUT_SIZE_1080p_RANGE;
    scs->static_config.rate_control_mode = 0;
    EbObjectWrapper_t *scs_wr = (EbObjectWrapper_t *)calloc(1, sizeof(*scs_wr));
    scs_wr->objectPtr = scs;

    PictureParentControlSet_t *pcs = (PictureParentControlSet_t *)calloc(1, sizeof(*pcs));
    pcs->sequence_control_set_wrapper_ptr = scs_wr;
    const int npu = ip[IP_ALL_PU] ? 209 : 85;
    pcs->max_number_of_pus_per_sb = (uint8_t)npu;
    pcs->pic_depth_mode = ip[IP_ALL_PU] ? PIC_ALL_DEPTH_MODE : PIC_SQ_DEPTH_MODE;
    pcs->cu8x8_mode = CU_8x8_MODE_0;
    pcs->enable_hme_flag = (EbBool)ip[IP_EN_HME];
    pcs->enable_hme_level0_flag = (EbBool)ip[IP_EN_L0];
    pcs->enable_hme_level1_flag = (EbBool)ip[IP_EN_L1];
    pcs->enable_hme_level2_flag = (EbBool)ip[IP_EN_L2];
    pcs->slice_type = ip[IP_TWO_LISTS] ? B_SLICE : P_SLICE;
    pcs->temporal_layer_index = (uint8_t)ip[IP_TEMPORAL_LAYER];
    pcs->hierarchical_levels = (uint8_t)ip[IP_HIER_LEVELS];
    pcs->is_used_as_reference_flag = (EbBool)ip[IP_IS_REF];
    pcs->use_subpel_flag = 0;
    pcs->ref_pa_pic_ptr_array[0] = make_pa_ref(planes[3], planes[4], planes[5], width, height);
    pcs->ref_pa_pic_ptr_array[1] = make_pa_ref(planes[6], planes[7], planes[8], width, height);
    pcs->ref_pic_poc_array[0] = (uint64_t)ip[IP_REF0_POC];
    pcs->ref_pic_poc_array[1] = (uint64_t)ip[IP_REF1_POC];

    const int nx = (width + 63) / 64, ny = (height + 63) / 64, nsb = nx * ny;
    pcs->me_results = (MeCuResults_t **)calloc(nsb, sizeof(MeCuResults_t *));
    for (int i = 0; i < nsb; i++) pcs->me_results[i] = (MeCuResults_t *)calloc(MAX_ME_PU_COUNT, sizeof(MeCuResults_t));
    pcs->rc_me_distortion = (uint32_t *)calloc(nsb, sizeof(uint32_t));

    MeContext_t *ctx = NULL;
    if (MeContextCtor(&ctx) != EB_ErrorNone) return -3;
    ctx->search_area_width = (uint8_t)ip[IP_SEARCH_W];
    ctx->search_area_height = (uint8_t)ip[IP_SEARCH_H];
    ctx->number_hme_search_region_in_width = (uint16_t)ip[IP_NHME_W];
    ctx->number_hme_search_region_in_height = (uint16_t)ip[IP_NHME_H];
    ctx->hme_level0_total_search_area_width = (uint16_t)ip[IP_L0_TOTAL_W];
    ctx->hme_level0_total_search_area_height = (uint16_t)ip[IP_L0_TOTAL_H];
    for (int k = 0; k < 2; k++) {
        ctx->hme_level0_search_area_in_width_array[k] = (uint16_t)ip[IP_L0_W0 + k];
        ctx->hme_level0_search_area_in_height_array[k] = (uint16_t)ip[IP_L0_H0 + k];
        ctx->hme_level1_search_area_in_width_array[k] = (uint16_t)ip[IP_L1_W0 + k];
        ctx->hme_level1_search_area_in_height_array[k] = (uint16_t)ip[IP_L1_H0 + k];
        ctx->hme_level2_search_area_in_width_array[k] = (uint16_t)ip[IP_L2_W0 + k];
        ctx->hme_level2_search_area_in_height_array[k] = (uint16_t)ip[IP_L2_H0 + k];
    }
    ctx->lambda = 0;

    /* the current picture: padded planes double as the "enhanced picture" (same samples, same stride) */
    EbPictureBufferDesc_t *cur_full = make_desc(planes[0], width + 136, 68, width, height);
    EbPictureBufferDesc_t *cur_q = make_desc(planes[1], (width >> 1) + 64, 32, width >> 1, height >> 1);
    EbPictureBufferDesc_t *cur_s = make_desc(planes[2], (width >> 2) + 32, 16, width >> 2, height >> 2);

    for (int sy = 0; sy < ny; sy++)
        for (int sx = 0; sx < nx; sx++) {
            const uint32_t sb_index = (uint32_t)(sx + sy * nx);
            const uint32_t sb_origin_x = sx * 64, sb_origin_y = sy * 64;
            const uint32_t sb_width = (width - sb_origin_x) < 64 ? width - sb_origin_x : 64;
            const uint32_t sb_height = (height - sb_origin_y) < 64 ? height - sb_origin_y : 64;
            /* Codec/EbMotionEstimationProcess.c:491-544 */
            uint32_t bufferIndex = (cur_full->origin_y + sb_origin_y) * cur_full->strideY + cur_full->origin_x + sb_origin_x;
            ctx->hme_search_type = HME_RECTANGULAR;
            for (uint32_t r = 0; r < 64; r++)
                memcpy(&ctx->sb_buffer[r * 64], &cur_full->bufferY[bufferIndex + r * cur_full->strideY], 64);
            ctx->sb_src_ptr = &cur_full->bufferY[bufferIndex];
            ctx->sb_src_stride = cur_full->strideY;
            if (pcs->enable_hme_level1_flag) {
                bufferIndex = (cur_q->origin_y + (sb_origin_y >> 1)) * cur_q->strideY + cur_q->origin_x + (sb_origin_x >> 1);
                for (uint32_t r = 0; r < (sb_height >> 1); r++)
                    memcpy(&ctx->quarter_sb_buffer[r * ctx->quarter_sb_buffer_stride], &cur_q->bufferY[bufferIndex + r * cur_q->strideY],
                           sb_width >> 1);
            }
            if (pcs->enable_hme_level0_flag) {
                bufferIndex = (cur_s->origin_y + (sb_origin_y >> 2)) * cur_s->strideY + cur_s->origin_x + (sb_origin_x >> 2);
                uint8_t *framePtr = &cur_s->bufferY[bufferIndex];
                uint8_t *localPtr = ctx->sixteenth_sb_buffer;
                for (uint32_t r = 0; r < (sb_height >> 2); r += 2) {
                    memcpy(localPtr, framePtr, sb_width >> 2);
                    localPtr += 16;
                    framePtr += cur_s->strideY << 1;
                }
            }
            MotionEstimateLcu(pcs, sb_index, sb_origin_x, sb_origin_y, ctx, cur_full);

            for (int l = 0; l < 2; l++) {
                memcpy(out_sad + ((size_t)sb_index * 2 + l) * npu, ctx->p_sb_best_sad[l][0], (size_t)npu * 4);
                memcpy(out_mv + ((size_t)sb_index * 2 + l) * npu, ctx->p_sb_best_mv[l][0], (size_t)npu * 4);
                out_origin[((size_t)sb_index * 2 + l) * 4 + 0] = ctx->x_search_area_origin[l][0];
                out_origin[((size_t)sb_index * 2 + l) * 4 + 1] = ctx->y_search_area_origin[l][0];
            }
            for (int pu = 0; pu < npu; pu++) {
                const MeCuResults_t *r = &pcs->me_results[sb_index][pu];
                int32_t *o = out_res + ((size_t)sb_index * npu + pu) * 11;
                o[0] = r->xMvL0; o[1] = r->yMvL0; o[2] = r->xMvL1; o[3] = r->yMvL1;
                o[4] = (int32_t)r->distortionDirection[0].distortion; o[5] = r->distortionDirection[0].direction;
                o[6] = (int32_t)r->distortionDirection[1].distortion; o[7] = r->distortionDirection[1].direction;
                o[8] = r->totalMeCandidateIndex;
                o[9] = (int32_t)r->distortionDirection[2].distortion; o[10] = r->distortionDirection[2].direction;
            }
        }
    return 0;
}

/* Runs the reference's InterpolateSearchRegionAVC (Codec/EbMotionEstimation.c:1707) on a search region whose
 * position (0,0) is `ref00` and copies out the three planes as MotionEstimateLcu hands them to the refinement
 * (:6923-6926): b from row 2, h from column 1, j from (0,0); out arrays are [rows][cols] with the context's stride
 * cropped to `cols`.  Used to pin the plane geometry of oracle/svt_subpel_oracle.c. */
int ref_interp_region(uint8_t *ref00, int stride, int sw, int sh, int rows, int cols, uint8_t *out_b, uint8_t *out_h,
                      uint8_t *out_j)
{
    static EbMemoryMapEntry *mm = NULL;
    static uint32_t mm_index;
    static uint64_t mm_total;
    if (!mm) mm = (EbMemoryMapEntry *)calloc(1 << 16, sizeof(EbMemoryMapEntry));
    memoryMap = mm;
    mm_index = 0;
    memoryMapIndex = &mm_index;
    totalLibMemory = &mm_total;
    MeContext_t *ctx = NULL;
    if (MeContextCtor(&ctx) != EB_ErrorNone) return -3;
    InterpolateSearchRegionAVC(ctx, 0, ref00, (uint32_t)stride, (uint32_t)sw + 63, (uint32_t)sh + 63, 8, ASM_NON_AVX2);
    const uint32_t is = ctx->interpolated_stride;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            out_b[y * cols + x] = ctx->pos_b_buffer[0][0][(2 + y) * is + x];
            out_h[y * cols + x] = ctx->pos_h_buffer[0][0][y * is + 1 + x];
            out_j[y * cols + x] = ctx->pos_j_buffer[0][0][y * is + x];
        }
    return 0;
}
