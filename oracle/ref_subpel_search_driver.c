/*
 * oracle/ref_subpel_search_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the REFERENCE's own sub-pel refinement of one superblock and list -- InterpolateSearchRegionAVC (Source/Lib/Codec/
 * EbMotionEstimation.c:1707), HalfPelSearch_LCU (:2246) with PU_HalfPelRefinement (:1844) and QuarterPelSearch_LCU (:3337) with
 * PU_QuarterPelRefinementOnTheFly (:2824) -- on caller-chosen full-pel results, called the way MotionEstimateLcu calls them (:6857-6964).
 *
 * Why this can run although MotionEstimateLcu with use_subpel_flag = 1 cannot: the refinement's distortion is selected at run time by
 * MeContext_t::fractionalSearchMethod (Codec/EbDefinitions.h:1846-1848).  MotionEstimateLcu hard-wires SSD_SEARCH (:6254), whose leaf is
 * looked up with Log2f(pu_width) -- Log2f_SSE2, a NASM-only symbol that cannot be assembled in this image and for which no stand-in is
 * written.  With SUB_SAD_SEARCH or FULL_SAD_SEARCH the same statements run (candidate order, strict '<', direction choice, the valid
 * quarter-pel positions, buffer selection, the 64x64 PU's 32x32 quarter-pel block, the tab* index maps of the 209-PU mode) with the SAD
 * leaves NxMSadKernel / NxMSadAveragingKernel instead, and the conditional operator never evaluates the Log2f branch.  This driver
 * therefore REFUSES method 2 (SSD_SEARCH); the SSD leaves and their dispatch per PU shape are pinned separately
 * (oracle/ref_subpel_leaf_driver.c).
 *
 * QuarterPelSearch_LCU is `static`, so this translation unit is the reference's EbMotionEstimation.c itself, included where it lies
 * (nothing is copied into the repository), followed by one entry point that only fills the structures those functions read.
 * oracle/build_ref.sh links it INSTEAD of the separately compiled EbMotionEstimation.o into oracle/_ref/libsvtref_subpel.so.
 */
#include "EbMotionEstimation.c"

#include <stdlib.h>
#include <string.h>

extern EbMemoryMapEntry *memoryMap;
extern uint32_t *memoryMapIndex;
extern uint64_t *totalLibMemory;

/* src00 / ref00: the SB's first sample in the padded source plane / the co-located sample in the padded reference plane.
 * geo = { x_search_area_origin, y_search_area_origin, search_area_width, search_area_height } (relative to the SB).
 * method: 0 SUB_SAD_SEARCH, 1 FULL_SAD_SEARCH.  all_pu: 209-PU mode (pic_depth_mode PIC_ALL_DEPTH_MODE) or the 85 squares.
 * sad / mv: [209 or 85] in ME-buffer order (p_sb_best_sad / p_sb_best_mv), full-pel results in, refined results out.
 * out_dir (optional): the half-pel direction of every PU in the same order (psub_pel_direction*). */
int ref_subpel_search(uint8_t *src00, int src_stride, uint8_t *ref00, int ref_stride, const int32_t *geo, int method, int all_pu,
                      int disable_8x8, int asm_type, uint32_t *sad, uint32_t *mv, uint8_t *out_dir)
{
    static EbMemoryMapEntry *mm = NULL;
    static uint32_t mm_index;
    static uint64_t mm_total;
    if (method != SUB_SAD_SEARCH && method != FULL_SAD_SEARCH) return -2; /* SSD_SEARCH would call Log2f_SSE2 (NASM), unavailable */
    if (!mm) mm = (EbMemoryMapEntry *)calloc(1 << 16, sizeof(EbMemoryMapEntry));
    memoryMap = mm;
    mm_index = 0;
    memoryMapIndex = &mm_index;
    totalLibMemory = &mm_total;

    MeContext_t *context_ptr = NULL;
    if (MeContextCtor(&context_ptr) != EB_ErrorNone) return -3;
    const int npu = all_pu ? 209 : 85;
    const uint32_t listIndex = 0, refPicIndex = 0;
    const int16_t x_search_area_origin = (int16_t)geo[0], y_search_area_origin = (int16_t)geo[1];
    const int search_area_width = geo[2], search_area_height = geo[3];

    SequenceControlSet_t *sequence_control_set_ptr = (SequenceControlSet_t *)calloc(1, sizeof(SequenceControlSet_t));
    PictureParentControlSet_t *picture_control_set_ptr = (PictureParentControlSet_t *)calloc(1, sizeof(PictureParentControlSet_t));
    picture_control_set_ptr->pic_depth_mode = all_pu ? PIC_ALL_DEPTH_MODE : PIC_SQ_DEPTH_MODE;
    picture_control_set_ptr->cu8x8_mode = disable_8x8 ? CU_8x8_MODE_1 : CU_8x8_MODE_0;

    context_ptr->fractionalSearchMethod = (uint8_t)method;
    context_ptr->fractional_search64x64 = EB_TRUE; /* :6261 */
    /* Codec/EbMotionEstimationProcess.c:491-505: the SB's source samples, in the picture and as a 64-wide copy */
    context_ptr->sb_src_ptr = src00;
    context_ptr->sb_src_stride = (uint32_t)src_stride;
    for (int r = 0; r < 64; r++) memcpy(&context_ptr->sb_buffer[r * 64], src00 + (ptrdiff_t)r * src_stride, 64);
    context_ptr->x_search_area_origin[listIndex][refPicIndex] = x_search_area_origin;
    context_ptr->y_search_area_origin[listIndex][refPicIndex] = y_search_area_origin;
    /* :6711-6722: the integer buffer starts ME_FILTER_TAP / 2 samples left of / above the search region */
    context_ptr->integer_buffer_ptr[listIndex][refPicIndex] =
        ref00 + (x_search_area_origin - (ME_FILTER_TAP >> 1)) + (ptrdiff_t)(y_search_area_origin - (ME_FILTER_TAP >> 1)) * ref_stride;
    context_ptr->interpolated_full_stride[listIndex][refPicIndex] = (uint32_t)ref_stride;

    /* the per-shape result pointers MotionEstimateLcu sets up (:6745-6800): for each of the 14 PU shapes, the SAD / MV / SSD pointers are
     * the three per-(list, reference) result arrays offset by the index of the shape's first PU */
    {
        uint32_t *const base_sad = context_ptr->p_sb_best_sad[listIndex][refPicIndex];
        uint32_t *const base_mv = context_ptr->p_sb_best_mv[listIndex][refPicIndex];
        uint32_t *const base_ssd = context_ptr->p_sb_best_ssd[listIndex][refPicIndex];
#define SHAPES(X) \
    X(64x64, ME_TIER_ZERO_PU_64x64) X(32x32, ME_TIER_ZERO_PU_32x32_0) X(16x16, ME_TIER_ZERO_PU_16x16_0) X(8x8, ME_TIER_ZERO_PU_8x8_0)       \
    X(64x32, ME_TIER_ZERO_PU_64x32_0) X(32x16, ME_TIER_ZERO_PU_32x16_0) X(16x8, ME_TIER_ZERO_PU_16x8_0) X(32x64, ME_TIER_ZERO_PU_32x64_0)   \
    X(16x32, ME_TIER_ZERO_PU_16x32_0) X(8x16, ME_TIER_ZERO_PU_8x16_0) X(32x8, ME_TIER_ZERO_PU_32x8_0) X(8x32, ME_TIER_ZERO_PU_8x32_0)       \
    X(64x16, ME_TIER_ZERO_PU_64x16_0) X(16x64, ME_TIER_ZERO_PU_16x64_0)
#define BIND(shape, first)                               \
    context_ptr->p_best_sad##shape = base_sad + (first); \
    context_ptr->p_best_mv##shape = base_mv + (first);   \
    context_ptr->p_best_ssd##shape = base_ssd + (first);
        SHAPES(BIND)
#undef BIND
#undef SHAPES
    }

    memcpy(context_ptr->p_sb_best_sad[listIndex][refPicIndex], sad, (size_t)npu * 4);
    memcpy(context_ptr->p_sb_best_mv[listIndex][refPicIndex], mv, (size_t)npu * 4);

    /* :6894-6960, argument for argument (M0_HIGH_PRECISION_INTERPOLATION is not defined in this reference) */
    InterpolateSearchRegionAVC(
        context_ptr,
        listIndex,
        context_ptr->integer_buffer_ptr[listIndex][0] + (ME_FILTER_TAP >> 1) + ((ME_FILTER_TAP >> 1) * context_ptr->interpolated_full_stride[listIndex][0]),
        context_ptr->interpolated_full_stride[listIndex][0],
        (uint32_t)search_area_width + (BLOCK_SIZE_64 - 1),
        (uint32_t)search_area_height + (BLOCK_SIZE_64 - 1),
        8,
        (EbAsm)asm_type);

    HalfPelSearch_LCU(
        sequence_control_set_ptr,
        picture_control_set_ptr,
        context_ptr,
        context_ptr->integer_buffer_ptr[listIndex][0] + (ME_FILTER_TAP >> 1) + ((ME_FILTER_TAP >> 1) * context_ptr->interpolated_full_stride[listIndex][0]),
        context_ptr->interpolated_full_stride[listIndex][0],
        &(context_ptr->pos_b_buffer[listIndex][0][(ME_FILTER_TAP >> 1) * context_ptr->interpolated_stride]),
        &(context_ptr->pos_h_buffer[listIndex][0][1]),
        &(context_ptr->pos_j_buffer[listIndex][0][0]),
        x_search_area_origin,
        y_search_area_origin,
        (EbAsm)asm_type,
        picture_control_set_ptr->cu8x8_mode == CU_8x8_MODE_1,
        EB_TRUE,
        EB_TRUE,
        EB_TRUE);

    QuarterPelSearch_LCU(
        context_ptr,
        context_ptr->integer_buffer_ptr[listIndex][0] + (ME_FILTER_TAP >> 1) + ((ME_FILTER_TAP >> 1) * context_ptr->interpolated_full_stride[listIndex][0]),
        context_ptr->interpolated_full_stride[listIndex][0],
        &(context_ptr->pos_b_buffer[listIndex][0][(ME_FILTER_TAP >> 1) * context_ptr->interpolated_stride]),
        &(context_ptr->pos_h_buffer[listIndex][0][1]),
        &(context_ptr->pos_j_buffer[listIndex][0][0]),
        x_search_area_origin,
        y_search_area_origin,
        (EbAsm)asm_type,
        picture_control_set_ptr->cu8x8_mode == CU_8x8_MODE_1,
        EB_TRUE,
        picture_control_set_ptr->pic_depth_mode <= PIC_ALL_C_DEPTH_MODE);

    memcpy(sad, context_ptr->p_sb_best_sad[listIndex][refPicIndex], (size_t)npu * 4);
    memcpy(mv, context_ptr->p_sb_best_mv[listIndex][refPicIndex], (size_t)npu * 4);
    if (out_dir) {
        /* ME-buffer order of the direction arrays = the order of the result arrays they travel with */
        out_dir[0] = context_ptr->psub_pel_direction64x64;
        memcpy(out_dir + 1, context_ptr->psub_pel_direction32x32, 4);
        memcpy(out_dir + 5, context_ptr->psub_pel_direction16x16, 16);
        memcpy(out_dir + 21, context_ptr->psub_pel_direction8x8, 64);
        if (all_pu) {
            memcpy(out_dir + 85, context_ptr->psub_pel_direction64x32, 2);
            memcpy(out_dir + 87, context_ptr->psub_pel_direction32x16, 8);
            memcpy(out_dir + 95, context_ptr->psub_pel_direction16x8, 32);
            memcpy(out_dir + 127, context_ptr->psub_pel_direction32x64, 2);
            memcpy(out_dir + 129, context_ptr->psub_pel_direction16x32, 8);
            memcpy(out_dir + 137, context_ptr->psub_pel_direction8x16, 32);
            memcpy(out_dir + 169, context_ptr->psub_pel_direction32x8, 16);
            memcpy(out_dir + 185, context_ptr->psub_pel_direction8x32, 16);
            memcpy(out_dir + 201, context_ptr->psub_pel_direction64x16, 4);
            memcpy(out_dir + 205, context_ptr->psub_pel_direction16x64, 4);
        }
    }
    free(sequence_control_set_ptr);
    free(picture_control_set_ptr);
    return 0;
}
