/*
 * oracle/ref_ois_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the REFERENCE's own open-loop intra search (OpenLoopIntraSearchLcu, Source/Lib/Codec/EbMotionEstimation.c:8047)
 * and its two building blocks (UpdateNeighborSamplesArrayOpenLoop / IntraPredictionOpenLoop,
 * Codec/EbIntraPrediction.c:5233 / :5353) standalone, compiled from /root/reference into oracle/_ref/libsvtref_me.so by
 * oracle/build_ref.sh.  This file only builds the structures those functions read and copies their outputs; it compiles
 * against the reference's headers where they lie and contains no reference code.  asm_type is ASM_NON_AVX2 (the
 * oracle definition of SURVEY 8c); every leaf reached is a C / intrinsics function, nothing NASM-only.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "EbDefinitions.h"
#include "EbEncodeContext.h"
#include "EbIntraPrediction.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationContext.h"
#include "EbMotionEstimationProcess.h"
#include "EbPictureBufferDesc.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbSystemResourceManager.h"

extern EbMemoryMapEntry *memoryMap;
extern uint32_t *memoryMapIndex;
extern uint64_t *totalLibMemory;

static void mm_reset(void)
{
    static EbMemoryMapEntry *mm = NULL;
    static uint32_t mm_index;
    static uint64_t mm_total;
    if (!mm) mm = (EbMemoryMapEntry *)calloc(1 << 16, sizeof(EbMemoryMapEntry));
    memoryMap = mm;
    mm_index = 0;
    memoryMapIndex = &mm_index;
    totalLibMemory = &mm_total;
}

static MotionEstimationContext_t *make_ctx(void)
{
    MotionEstimationContext_t *c = (MotionEstimationContext_t *)calloc(1, sizeof(*c));
    if (IntraOpenLoopReferenceSamplesCtor(&c->intra_ref_ptr) != EB_ErrorNone) return NULL;
    if (MeContextCtor(&c->me_context_ptr) != EB_ErrorNone) return NULL;
    return c;
}

static void fill_desc(EbPictureBufferDesc_t *d, uint8_t *buf, int stride, int origin, int width, int height)
{
    memset(d, 0, sizeof(*d));
    d->bufferY = buf;
    d->strideY = (uint16_t)stride;
    d->origin_x = d->origin_y = (uint16_t)origin;
    d->width = (uint16_t)width;
    d->height = (uint16_t)height;
    d->maxWidth = (uint16_t)width;
    d->maxHeight = (uint16_t)height;
}

/* One CU: neighbour gathering + prediction of `mode` (0..34) as the open-loop search does them.  plane = padded luma
 * (first byte of the buffer), out_pred = size x size, out_refs = the 4*size+1 "reverse" neighbour array
 * (left top->bottom, top-left, top). */
int ref_ois_predict(uint8_t *plane, int stride, int origin, int width, int height, int cu_x, int cu_y, int size, int mode,
                    uint8_t *out_pred, uint8_t *out_refs)
{
    mm_reset();
    MotionEstimationContext_t *c = make_ctx();
    if (!c) return -3;
    EbPictureBufferDesc_t d;
    fill_desc(&d, plane, stride, origin, width, height);
    UpdateNeighborSamplesArrayOpenLoop(c->intra_ref_ptr, &d, d.strideY, (uint32_t)cu_x, (uint32_t)cu_y, (uint32_t)size);
    memcpy(out_refs, c->intra_ref_ptr->y_intra_reference_array_reverse, (size_t)4 * size + 1);
    IntraPredictionOpenLoop((uint32_t)size, c, (uint32_t)mode, ASM_NON_AVX2);
    const uint32_t ps = c->me_context_ptr->sb_buffer_stride;
    for (int y = 0; y < size; y++) memcpy(out_pred + y * size, c->me_context_ptr->sb_buffer + y * ps, (size_t)size);
    return 0;
}

enum { OP_SLICE_I, OP_TEMPORAL_LAYER, OP_IS_REF, OP_RES_4K, OP_LIMIT_DC, OP_CU8X8_MODE, OP_ENC_MODE, OP_COUNT };

/* Whole picture.  me_dist: [n_sb][85] = me_results[sb][raster cu].distortionDirection[0].distortion (may be NULL when the
 * path does not read it).  out_cand: [n_sb][85][18] OisCandidate_t words (index 0 = the unused 64x64 slot, zero),
 * out_total: [n_sb][85] total_intra_luma_mode.  The result buffers are zero-initialised before the search, so every field
 * the reference does not write reads as zero. */
int ref_ois_search_picture(uint8_t *plane, int stride, int origin, int width, int height, const int32_t *op,
                           const uint32_t *me_dist, uint32_t *out_cand, uint8_t *out_total)
{
    mm_reset();
    MotionEstimationContext_t *c = make_ctx();
    if (!c) return -3;
    const int nx = (width + 63) / 64, ny = (height + 63) / 64, nsb = nx * ny;

    SequenceControlSet_t *scs = (SequenceControlSet_t *)calloc(1, sizeof(*scs));
    scs->luma_width = (uint16_t)width;
    scs->luma_height = (uint16_t)height;
    scs->sb_sz = 64;
    scs->input_resolution = op[OP_RES_4K] ? INPUT_SIZE_4K_RANGE : INPUT_SIZE_1080p_RANGE;
    scs->sb_params_array = (SbParams_t *)calloc(nsb, sizeof(SbParams_t));
    for (int i = 0; i < nsb; i++) {
        SbParams_t *p = &scs->sb_params_array[i];
        p->origin_x = (uint16_t)((i % nx) * 64);
        p->origin_y = (uint16_t)((i / nx) * 64);
        /* Codec/EbSequenceControlSet.c:483-488 */
        for (int cu = 0; cu < CU_MAX_COUNT; cu++)
            p->raster_scan_cu_validity[cu] =
                (p->origin_x + RASTER_SCAN_CU_X[cu] + RASTER_SCAN_CU_SIZE[cu] > (uint32_t)width ||
                 p->origin_y + RASTER_SCAN_CU_Y[cu] + RASTER_SCAN_CU_SIZE[cu] > (uint32_t)height)
                    ? EB_FALSE
                    : EB_TRUE;
    }
    EbObjectWrapper_t *scs_wr = (EbObjectWrapper_t *)calloc(1, sizeof(*scs_wr));
    scs_wr->objectPtr = scs;

    PictureParentControlSet_t *pcs = (PictureParentControlSet_t *)calloc(1, sizeof(*pcs));
    pcs->sequence_control_set_wrapper_ptr = scs_wr;
    pcs->slice_type = op[OP_SLICE_I] ? I_SLICE : B_SLICE;
    pcs->temporal_layer_index = (uint8_t)op[OP_TEMPORAL_LAYER];
    pcs->is_used_as_reference_flag = (EbBool)op[OP_IS_REF];
    pcs->limit_ois_to_dc_mode_flag = (EbBool)op[OP_LIMIT_DC];
    pcs->cu8x8_mode = op[OP_CU8X8_MODE] ? CU_8x8_MODE_1 : CU_8x8_MODE_0;
    pcs->enc_mode = (EbEncMode)op[OP_ENC_MODE];
    pcs->ois_cu32_cu16_results = (OisCu32Cu16Results_t **)calloc(nsb, sizeof(void *));
    pcs->ois_cu8_results = (OisCu8Results_t **)calloc(nsb, sizeof(void *));
    pcs->me_results = (MeCuResults_t **)calloc(nsb, sizeof(MeCuResults_t *));
    for (int i = 0; i < nsb; i++) {
        pcs->ois_cu32_cu16_results[i] = (OisCu32Cu16Results_t *)calloc(1, sizeof(OisCu32Cu16Results_t));
        pcs->ois_cu8_results[i] = (OisCu8Results_t *)calloc(1, sizeof(OisCu8Results_t));
        OisCandidate_t *a = (OisCandidate_t *)calloc(21 * MAX_OPEN_LOOP_INTRA_CANDIDATES, sizeof(OisCandidate_t));
        OisCandidate_t *b = (OisCandidate_t *)calloc(64 * MAX_OPEN_LOOP_INTRA_CANDIDATES, sizeof(OisCandidate_t));
        for (int k = 0; k < 21; k++) pcs->ois_cu32_cu16_results[i]->sorted_ois_candidate[k] = a + k * MAX_OPEN_LOOP_INTRA_CANDIDATES;
        for (int k = 0; k < 64; k++) pcs->ois_cu8_results[i]->sorted_ois_candidate[k] = b + k * MAX_OPEN_LOOP_INTRA_CANDIDATES;
        pcs->me_results[i] = (MeCuResults_t *)calloc(MAX_ME_PU_COUNT, sizeof(MeCuResults_t));
        if (me_dist)
            for (int cu = 0; cu < 85; cu++) pcs->me_results[i][cu].distortionDirection[0].distortion = me_dist[i * 85 + cu];
    }

    EbPictureBufferDesc_t d;
    fill_desc(&d, plane, stride, origin, width, height);
    for (int i = 0; i < nsb; i++) {
        OpenLoopIntraSearchLcu(pcs, (uint32_t)i, c, &d, ASM_NON_AVX2);
        for (int cu = 0; cu < 85; cu++) {
            uint32_t *o = out_cand + ((size_t)i * 85 + cu) * MAX_OPEN_LOOP_INTRA_CANDIDATES;
            if (cu == 0) {
                memset(o, 0, MAX_OPEN_LOOP_INTRA_CANDIDATES * 4);
                out_total[i * 85] = 0;
                continue;
            }
            const OisCandidate_t *s = cu < 21 ? pcs->ois_cu32_cu16_results[i]->sorted_ois_candidate[cu]
                                              : pcs->ois_cu8_results[i]->sorted_ois_candidate[cu - 21];
            for (int k = 0; k < MAX_OPEN_LOOP_INTRA_CANDIDATES; k++) o[k] = s[k].ois_results;
            out_total[i * 85 + cu] = cu < 21 ? pcs->ois_cu32_cu16_results[i]->total_intra_luma_mode[cu]
                                              : pcs->ois_cu8_results[i]->total_intra_luma_mode[cu - 21];
        }
    }
    return 0;
}
