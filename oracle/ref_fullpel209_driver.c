/*
 * oracle/ref_fullpel209_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Drives the REFERENCE's own ExtSadCalculation_8x8_16x16 / ExtSadCalculation_32x32_64x64 / ExtSadCalculation (compiled
 * from /root/reference into oracle/_ref/libsvtref_me.so by oracle/build_ref.sh) in exactly the call sequence of the
 * reference's static open_loop_me_fullpel_search_sblock / open_loop_me_get_search_point_results_block
 * (Codec/EbMotionEstimation.c:1556-1595, :1065-1231), which cannot be linked directly.  Contains no reference code, only
 * calls into it.  Pins oracle/svt_me_oracle.c::orc_fullpel_search_209pu.
 */
#include <stdint.h>

/* Codec/EbMotionEstimation.c:159, :218, :266 (ASM_NON_AVX2 row of the function tables, :129-141, :1054-1060) */
void ExtSadCalculation_8x8_16x16(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t refStride, uint32_t *p_best_sad8x8,
                                 uint32_t *p_best_sad16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv,
                                 uint32_t *p_sad16x16, uint32_t *p_sad8x8);
void ExtSadCalculation_32x32_64x64(uint32_t *p_sad16x16, uint32_t *p_best_sad32x32, uint32_t *p_best_sad64x64, uint32_t *p_best_mv32x32,
                                   uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);
void ExtSadCalculation(uint32_t *p_sad8x8, uint32_t *p_sad16x16, uint32_t *p_sad32x32, uint32_t *p_best_sad64x32, uint32_t *p_best_mv64x32,
                       uint32_t *p_best_sad32x16, uint32_t *p_best_mv32x16, uint32_t *p_best_sad16x8, uint32_t *p_best_mv16x8,
                       uint32_t *p_best_sad32x64, uint32_t *p_best_mv32x64, uint32_t *p_best_sad16x32, uint32_t *p_best_mv16x32,
                       uint32_t *p_best_sad8x16, uint32_t *p_best_mv8x16, uint32_t *p_best_sad32x8, uint32_t *p_best_mv32x8,
                       uint32_t *p_best_sad8x32, uint32_t *p_best_mv8x32, uint32_t *p_best_sad64x16, uint32_t *p_best_mv64x16,
                       uint32_t *p_best_sad16x64, uint32_t *p_best_mv16x64, uint32_t mv);

/* order in which the reference visits the sixteen 16x16 blocks: (z-index, col16, row16), :1127-1206 */
static const uint8_t kVisit[16][3] = {{0, 0, 0}, {1, 1, 0}, {4, 2, 0}, {5, 3, 0}, {2, 0, 1}, {3, 1, 1}, {6, 2, 1}, {7, 3, 1},
                                      {8, 0, 2}, {9, 1, 2}, {12, 2, 2}, {13, 3, 2}, {10, 0, 3}, {11, 1, 3}, {14, 2, 3}, {15, 3, 3}};

/* best_sad / best_mv: 209 entries in ME-buffer order, initialised by the caller (MAX_SAD_VALUE / 0) */
void ref_fullpel_search_209pu(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, int16_t x_origin, int16_t y_origin,
                              uint32_t sw, uint32_t sh, uint32_t *bs, uint32_t *bm)
{
    uint32_t sad8[64], sad16[16], sad32[4];
    for (uint32_t ys = 0; ys < sh; ys++)
        for (uint32_t xs = 0; xs < sw; xs++) {
            const int32_t xi = (int32_t)xs + x_origin, yi = (int32_t)ys + y_origin;
            const uint32_t mv1 = ((uint32_t)(uint16_t)yi) << 18; /* :1085 */
            const uint16_t mv2 = (uint16_t)((uint16_t)xi << 2);   /* :1086 */
            const uint32_t mv = mv1 | mv2;
            uint8_t *r0 = ref + ys * ref_stride + xs;
            for (int i = 0; i < 16; i++) {
                const int z = kVisit[i][0];
                ExtSadCalculation_8x8_16x16(src + kVisit[i][2] * 16 * src_stride + kVisit[i][1] * 16, src_stride,
                                            r0 + kVisit[i][2] * 16 * ref_stride + kVisit[i][1] * 16, ref_stride, &bs[21 + 4 * z], &bs[5 + z],
                                            &bm[21 + 4 * z], &bm[5 + z], mv, &sad16[z], &sad8[4 * z]);
            }
            ExtSadCalculation_32x32_64x64(sad16, &bs[1], &bs[0], &bm[1], &bm[0], mv, sad32);
            ExtSadCalculation(sad8, sad16, sad32, &bs[85], &bm[85], &bs[87], &bm[87], &bs[95], &bm[95], &bs[127], &bm[127], &bs[129], &bm[129],
                              &bs[137], &bm[137], &bs[169], &bm[169], &bs[185], &bm[185], &bs[201], &bm[201], &bs[205], &bm[205], mv);
        }
}
