/*
 * oracle/svt_me_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 * Plain-C restatement of the reference ME hot path, ASM_NON_AVX2 semantics.
 */
#include "svt_me_oracle.h"
#include <string.h>

static inline uint32_t absdiff(uint8_t a, uint8_t b) { return a > b ? (uint32_t)(a - b) : (uint32_t)(b - a); }

uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                     uint32_t height, uint32_t width)
{
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++)
            sad += absdiff(src[y * src_stride + x], ref[y * ref_stride + x]);
    return sad;
}

void orc_sad_loop_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                         uint32_t height, uint32_t width, uint64_t *best_sad, int16_t *x_center,
                         int16_t *y_center, uint32_t ref_stride_raw, int16_t search_area_width,
                         int16_t search_area_height)
{
    *best_sad = 0xffffff;
    for (int16_t ys = 0; ys < search_area_height; ys++) {
        for (int16_t xs = 0; xs < search_area_width; xs++) {
            uint32_t sad = orc_nxm_sad(src, src_stride, ref + xs, ref_stride, height, width);
            if (sad < *best_sad) {
                *best_sad = sad;
                *x_center = xs;
                *y_center = ys;
            }
        }
        ref += ref_stride_raw; /* C_DEFAULT/EbComputeSAD_C.c:115 */
    }
}

void orc_init_best(uint32_t *best_sad, uint32_t *best_mv, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        best_sad[i] = ORC_MAX_SAD_VALUE; /* Codec/EbMotionEstimation.c:6820 */
        if (best_mv)
            best_mv[i] = 0;
    }
}

/* 8x8 SAD as the ME kernels compute it: rows 0,2,4,6 only (ASM_SSE2/EbMeSadCalculation_Intrinsic_SSE2.c:25-43,
 * ASM_SSE4_1/EbComputeSAD_Intrinsic_SSE4_1.c:4487-4509); the caller doubles it. */
static inline uint32_t sad8x4_even_rows(const uint8_t *s, uint32_t ss, const uint8_t *r, uint32_t rs)
{
    uint32_t sad = 0;
    for (int y = 0; y < 8; y += 2)
        for (int x = 0; x < 8; x++)
            sad += absdiff(s[y * ss + x], r[y * rs + x]);
    return sad;
}

void orc_fullpel_search_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                             int16_t x_search_area_origin, int16_t y_search_area_origin,
                             uint32_t search_area_width, uint32_t search_area_height, uint32_t *best_sad,
                             uint32_t *best_mv)
{
    for (uint32_t ys = 0; ys < search_area_height; ys++) {
        for (uint32_t xs = 0; xs < search_area_width; xs++) {
            const uint8_t *r0 = ref + ys * ref_stride + xs;
            /* Codec/EbMotionEstimation.c:1389-1391 / :1261-1263 */
            int32_t ymv = (int32_t)ys + y_search_area_origin;
            int32_t xmv = (int32_t)xs + x_search_area_origin;
            uint32_t mv = ((uint32_t)(uint16_t)(ymv * 4) << 16) | (uint32_t)(uint16_t)(xmv * 4);

            uint32_t s16[16], s32[4], s64 = 0;
            for (int z16 = 0; z16 < 16; z16++) {
                /* z-order of 16x16 blocks, Codec/EbMotionEstimation.c:1405-1415 */
                int col16 = ((z16 >> 2) & 1) * 2 + (z16 & 1);
                int row16 = ((z16 >> 3) & 1) * 2 + ((z16 >> 1) & 1);
                uint32_t sum16 = 0;
                for (int k = 0; k < 4; k++) {
                    int bx = col16 * 16 + (k & 1) * 8;
                    int by = row16 * 16 + (k >> 1) * 8;
                    uint32_t s8 = 2 * sad8x4_even_rows(src + by * src_stride + bx, src_stride,
                                                       r0 + by * ref_stride + bx, ref_stride);
                    sum16 += s8;
                    int pu = 21 + 4 * z16 + k;
                    if (s8 < best_sad[pu]) {
                        best_sad[pu] = s8;
                        best_mv[pu] = mv;
                    }
                }
                s16[z16] = sum16;
                if (sum16 < best_sad[5 + z16]) {
                    best_sad[5 + z16] = sum16;
                    best_mv[5 + z16] = mv;
                }
            }
            for (int q = 0; q < 4; q++) {
                /* ASM_SSE2/EbMeSadCalculation_Intrinsic_SSE2.c:91-116: 32x32_q = sum of p_sad16x16[4q..4q+3] */
                s32[q] = s16[4 * q] + s16[4 * q + 1] + s16[4 * q + 2] + s16[4 * q + 3];
                s64 += s32[q];
                if (s32[q] < best_sad[1 + q]) {
                    best_sad[1 + q] = s32[q];
                    best_mv[1 + q] = mv;
                }
            }
            if (s64 < best_sad[0]) {
                best_sad[0] = s64;
                best_mv[0] = mv;
            }
        }
    }
}

/* Batch form (same descriptor layout as oracle/ref_fullpel_driver.c): n_sb independent searches,
 * desc[i] = {src_offset, ref_offset, x_origin, y_origin, sw, sh}. */
void orc_fullpel_search_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane,
                              uint32_t ref_stride, const int32_t *desc, uint32_t n_sb, uint32_t *best_sad,
                              uint32_t *best_mv)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d = desc + 6 * i;
        orc_init_best(best_sad + 85 * i, best_mv + 85 * i, 85);
        orc_fullpel_search_85pu(src_plane + d[0], src_stride, ref_plane + d[1], ref_stride, (int16_t)d[2],
                                (int16_t)d[3], (uint32_t)d[4], (uint32_t)d[5], best_sad + 85 * i, best_mv + 85 * i);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * 209-PU full-pel search (row a9): open_loop_me_fullpel_search_sblock / open_loop_me_get_search_point_results_block
 * (Codec/EbMotionEstimation.c:1556-1595, :1065-1231) with ExtSadCalculation_8x8_16x16 (:159-212),
 * ExtSadCalculation_32x32_64x64 (:218-260) and ExtSadCalculation (:266-1052), ASM_NON_AVX2 row.  Every position is
 * visited in raster order; the 85 square PUs are searched exactly like the 85-PU mode; the 124 rectangular PUs are sums of
 * the stored square SADs.  PU order: 0..84 squares, 85 64x32[2], 87 32x16[8], 95 16x8[32], 127 32x64[2], 129 16x32[8],
 * 137 8x16[32], 169 32x8[16], 185 8x32[16], 201 64x16[4], 205 16x64[4] (Codec/EbMotionEstimationContext.h:132-261).
 * Quirk (:343-347): the update of 32x16[5] is guarded by the STALE variable `sad`, which at that point holds the SAD of
 * 64x32[1]; when that is below the stored best of 32x16[5], the best is overwritten with the current 32x16[5] SAD whether or
 * not it is smaller.  Reproduced literally.
 * ------------------------------------------------------------------------------------------------------------------ */
#define UPD(pu, v) do { if ((v) < best_sad[pu]) { best_sad[pu] = (v); best_mv[pu] = mv; } } while (0)
void orc_fullpel_search_209pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                              int16_t x_search_area_origin, int16_t y_search_area_origin, uint32_t search_area_width,
                              uint32_t search_area_height, uint32_t *best_sad, uint32_t *best_mv)
{
    for (uint32_t ys = 0; ys < search_area_height; ys++) {
        for (uint32_t xs = 0; xs < search_area_width; xs++) {
            const uint8_t *r0 = ref + ys * ref_stride + xs;
            const int32_t ymv = (int32_t)ys + y_search_area_origin, xmv = (int32_t)xs + x_search_area_origin;
            const uint32_t mv = ((uint32_t)(uint16_t)(ymv * 4) << 16) | (uint32_t)(uint16_t)(xmv * 4);
            uint32_t s8[64], s16[16], s32[4], s64 = 0; /* s8 index = 4 * z16 + raster-in-16x16, s16 index = z16 */
            for (int z16 = 0; z16 < 16; z16++) {
                const int col16 = ((z16 >> 2) & 1) * 2 + (z16 & 1), row16 = ((z16 >> 3) & 1) * 2 + ((z16 >> 1) & 1);
                uint32_t sum16 = 0;
                for (int k = 0; k < 4; k++) {
                    const int bx = col16 * 16 + (k & 1) * 8, by = row16 * 16 + (k >> 1) * 8;
                    const uint32_t v = 2 * sad8x4_even_rows(src + by * src_stride + bx, src_stride, r0 + by * ref_stride + bx, ref_stride);
                    s8[4 * z16 + k] = v;
                    sum16 += v;
                    UPD(21 + 4 * z16 + k, v);
                }
                s16[z16] = sum16;
                UPD(5 + z16, sum16);
            }
            for (int q = 0; q < 4; q++) {
                s32[q] = s16[4 * q] + s16[4 * q + 1] + s16[4 * q + 2] + s16[4 * q + 3];
                s64 += s32[q];
                UPD(1 + q, s32[q]);
            }
            UPD(0, s64);
            /* ExtSadCalculation, in its statement order */
            uint32_t sad, s32x16[8], s16x32[8], s16x8[32], s8x16[32];
            sad = s32[0] + s32[1]; UPD(85, sad);
            sad = s32[2] + s32[3]; UPD(86, sad);
            for (int i = 0; i < 8; i++) {
                s32x16[i] = s16[2 * i] + s16[2 * i + 1];
                if (i == 5) { if (sad < best_sad[87 + 5]) { best_sad[87 + 5] = s32x16[5]; best_mv[87 + 5] = mv; } } /* stale `sad` */
                else UPD(87 + i, s32x16[i]);
            }
            sad = s32x16[0] + s32x16[2]; UPD(201, sad);
            sad = s32x16[1] + s32x16[3]; UPD(202, sad);
            sad = s32x16[4] + s32x16[6]; UPD(203, sad);
            sad = s32x16[5] + s32x16[7]; UPD(204, sad);
            for (int i = 0; i < 32; i++) { s16x8[i] = s8[2 * i] + s8[2 * i + 1]; UPD(95 + i, s16x8[i]); }
            sad = s32[0] + s32[2]; UPD(127, sad);
            sad = s32[1] + s32[3]; UPD(128, sad);
            for (int i = 0; i < 8; i++) { s16x32[i] = s16[4 * (i >> 1) + (i & 1)] + s16[4 * (i >> 1) + (i & 1) + 2]; UPD(129 + i, s16x32[i]); }
            sad = s16x32[0] + s16x32[4]; UPD(205, sad);
            sad = s16x32[1] + s16x32[5]; UPD(206, sad);
            sad = s16x32[2] + s16x32[6]; UPD(207, sad);
            sad = s16x32[3] + s16x32[7]; UPD(208, sad);
            for (int i = 0; i < 32; i++) { s8x16[i] = s8[4 * (i >> 1) + (i & 1)] + s8[4 * (i >> 1) + (i & 1) + 2]; UPD(137 + i, s8x16[i]); }
            for (int i = 0; i < 16; i++) { sad = s16x8[4 * (i >> 1) + (i & 1)] + s16x8[4 * (i >> 1) + (i & 1) + 2]; UPD(169 + i, sad); }
            for (int i = 0; i < 16; i++) { sad = s8x16[8 * (i >> 2) + (i & 3)] + s8x16[8 * (i >> 2) + (i & 3) + 4]; UPD(185 + i, sad); }
        }
    }
}
#undef UPD

void orc_fullpel_search209_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane, uint32_t ref_stride,
                                 const int32_t *desc, uint32_t n_sb, uint32_t *best_sad, uint32_t *best_mv)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d = desc + 6 * i;
        for (int p = 0; p < 209; p++) { best_sad[209 * i + p] = 128 * 128 * 255; best_mv[209 * i + p] = 0; } /* MAX_SAD_VALUE */
        orc_fullpel_search_209pu(src_plane + d[0], src_stride, ref_plane + d[1], ref_stride, (int16_t)d[2], (int16_t)d[3],
                                 (uint32_t)d[4], (uint32_t)d[5], best_sad + 209 * i, best_mv + 209 * i);
    }
}
