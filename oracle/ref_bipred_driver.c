/*
 * oracle/ref_bipred_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the REFERENCE's own bi-prediction search of one superblock (BiPredictionSearch, Source/Lib/Codec/EbMotionEstimation.c:5261,
 * -> BiPredictionCompensation :5090 -> BiPredAverging :4933 -> SelectBuffer :4762 / QuarterPelCompensation :4818 ->
 * picture_average_array / NxMSadAveragingKernel_funcPtrArray) for caller-chosen QUARTER-PEL vectors, on half-pel planes produced by the
 * reference's own InterpolateSearchRegionAVC (:1707).  None of these reaches a NASM-only symbol (PictureAverageKernel_SSE2_INTRIN and
 * CombinedAveragingSAD are C / intrinsics), so the fractional half of the bi-prediction path is pinned here although the sub-pel
 * refinement that normally produces the vectors cannot run (Log2f_SSE2).  Compiles against the reference's headers where they lie and
 * contains no reference code.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationContext.h"
#include "EbPictureControlSet.h"
#include "EbSystemResourceManager.h"

extern EbMemoryMapEntry *memoryMap;
extern uint32_t *memoryMapIndex;
extern uint64_t *totalLibMemory;

EbErrorType BiPredictionSearch(MeContext_t *context_ptr, uint32_t pu_index, uint8_t candidateIndex, uint32_t activeRefPicFirstLisNum,
                               uint32_t activeRefPicSecondLisNum, uint8_t *totalMeCandidateIndex, EbAsm asm_type,
                               PictureParentControlSet_t *picture_control_set_ptr);

/* geo = { xo0, yo0, sw0, sh0, xo1, yo1, sw1, sh1 }: search-area origins (relative to the SB) and sizes of the two lists.
 * src00 / ref0_00 / ref1_00: the SB's first sample in the source plane / the co-located sample in each padded reference plane.
 * mv0 / mv1: [n_pu] quarter-pel vector words in ME-buffer order (p_sb_best_mv[list][0][nIndex]); out_sad[pu] (raster PU index) =
 * me_candidate[0].pu[pu].distortion = the return value of BiPredAverging. */
int ref_bipred_search(uint8_t *src00, int src_stride, uint8_t *ref0_00, int ref0_stride, uint8_t *ref1_00, int ref1_stride, const int32_t *geo,
                      const uint32_t *mv0, const uint32_t *mv1, int n_pu, int asm_type, uint32_t *out_sad)
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
    ctx->fractionalSearchMethod = SSD_SEARCH; /* MotionEstimateLcu :6247-6258: the full-SAD branch of BiPredAverging */
    ctx->sb_src_ptr = src00;
    ctx->sb_src_stride = (uint32_t)src_stride;
    uint8_t *refs[2] = {ref0_00, ref1_00};
    const int strides[2] = {ref0_stride, ref1_stride};
    for (int l = 0; l < 2; l++) {
        const int xo = geo[4 * l], yo = geo[4 * l + 1], sw = geo[4 * l + 2], sh = geo[4 * l + 3];
        ctx->x_search_area_origin[l][0] = (int16_t)xo;
        ctx->y_search_area_origin[l][0] = (int16_t)yo;
        /* :6880-6897: the integer buffer starts ME_FILTER_TAP / 2 samples left of / above the search region */
        ctx->integer_buffer_ptr[l][0] = refs[l] + (xo - (ME_FILTER_TAP >> 1)) + (ptrdiff_t)(yo - (ME_FILTER_TAP >> 1)) * strides[l];
        ctx->interpolated_full_stride[l][0] = (uint32_t)strides[l];
        InterpolateSearchRegionAVC(ctx, (uint32_t)l, ctx->integer_buffer_ptr[l][0] + (ME_FILTER_TAP >> 1) + (ME_FILTER_TAP >> 1) * strides[l],
                                   (uint32_t)strides[l], (uint32_t)sw + (BLOCK_SIZE_64 - 1), (uint32_t)sh + (BLOCK_SIZE_64 - 1), 8, (EbAsm)asm_type);
        memcpy(ctx->p_sb_best_mv[l][0], l ? mv1 : mv0, (size_t)n_pu * 4);
    }
    for (int pu = 0; pu < n_pu; pu++) {
        uint8_t total = 0;
        BiPredictionSearch(ctx, (uint32_t)pu, 0, 1, 1, &total, (EbAsm)asm_type, NULL);
        out_sad[pu] = ctx->me_candidate[0].pu[pu].distortion;
    }
    return 0;
}
