/*
 * oracle/ref_fullpel_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Drives the REFERENCE's own leaf kernels (compiled from /root/reference into
 * oracle/_ref/libsvtref_kernels.so by oracle/build_ref.sh) in exactly the call sequence of the
 * reference's static FullPelSearch_LCU / GetEightHorizontalSearchPointResultsAll85PUs /
 * GetSearchPointResults (Codec/EbMotionEstimation.c:1504-1551, :1369-1499, :1237-1364), which are
 * `static` and cannot be linked directly.  Used (a) to pin oracle/svt_me_oracle.c, (b) to generate
 * tests/golden fixtures, (c) as bench.py's cpu_baseline (kind "reference").
 *
 * asm_type 0 = ASM_NON_AVX2 row (SSE4.1 8-position kernels + SSE2 single-position kernels) -- the
 *              parity oracle; asm_type 1 = ASM_AVX2 row (timing only; its 32x32 MVs are wrong when
 *              built with GCC).
 */
#include <stdint.h>
#include <string.h>

/* reference kernel prototypes (Codec/EbComputeSAD.h:27-91, Codec/EbMeSadCalculation.h:22-95) */
void GetEightHorizontalSearchPointResults_8x8_16x16_PU_SSE41_INTRIN(uint8_t *src, uint32_t src_stride, uint8_t *ref,
    uint32_t refStride, uint32_t *p_best_sad8x8, uint32_t *p_best_mv8x8, uint32_t *p_best_sad16x16,
    uint32_t *p_best_mv16x16, uint32_t mv, uint16_t *p_sad16x16);
void GetEightHorizontalSearchPointResults_32x32_64x64_PU_SSE41_INTRIN(uint16_t *p_sad16x16, uint32_t *p_best_sad32x32,
    uint32_t *p_best_sad64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv);
void GetEightHorizontalSearchPointResults_8x8_16x16_PU_AVX2_INTRIN(uint8_t *src, uint32_t src_stride, uint8_t *ref,
    uint32_t refStride, uint32_t *p_best_sad8x8, uint32_t *p_best_mv8x8, uint32_t *p_best_sad16x16,
    uint32_t *p_best_mv16x16, uint32_t mv, uint16_t *p_sad16x16);
void GetEightHorizontalSearchPointResults_32x32_64x64_PU_AVX2_INTRIN(uint16_t *p_sad16x16, uint32_t *p_best_sad32x32,
    uint32_t *p_best_sad64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv);
void SadCalculation_8x8_16x16_SSE2_INTRIN(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t refStride,
    uint32_t *p_best_sad8x8, uint32_t *p_best_sad16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16,
    uint32_t mv, uint32_t *p_sad16x16);
void SadCalculation_32x32_64x64_SSE2_INTRIN(uint32_t *p_sad16x16, uint32_t *p_best_sad32x32, uint32_t *p_best_sad64x64,
    uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv);

/* order in which the reference visits the sixteen 16x16 blocks: (z-index, col16, row16) */
static const uint8_t kVisit[16][3] = {
    {0, 0, 0}, {1, 1, 0}, {4, 2, 0}, {5, 3, 0}, {2, 0, 1}, {3, 1, 1}, {6, 2, 1}, {7, 3, 1},
    {8, 0, 2}, {9, 1, 2}, {12, 2, 2}, {13, 3, 2}, {10, 0, 3}, {11, 1, 3}, {14, 2, 3}, {15, 3, 3}};

void ref_fullpel_search_85pu(int asm_type, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                             int16_t x_origin, int16_t y_origin, uint32_t sw, uint32_t sh, uint32_t *best_sad,
                             uint32_t *best_mv)
{
    uint32_t *bs64 = best_sad + 0, *bs32 = best_sad + 1, *bs16 = best_sad + 5, *bs8 = best_sad + 21;
    uint32_t *bm64 = best_mv + 0, *bm32 = best_mv + 1, *bm16 = best_mv + 5, *bm8 = best_mv + 21;
    uint16_t eight_pos_sad16[128] __attribute__((aligned(32)));
    uint32_t sad16[16];
    uint32_t sw8 = sw & ~7u;

    for (uint32_t ys = 0; ys < sh; ys++) {
        for (uint32_t xs = 0; xs < sw8; xs += 8) {
            int32_t xi = (int32_t)xs + x_origin, yi = (int32_t)ys + y_origin;
            uint32_t mvy = ((uint32_t)(uint16_t)yi) << 18; /* :1389 */
            uint16_t mvx = (uint16_t)((uint16_t)xi << 2);   /* :1390 */
            uint32_t mv = mvy | mvx;
            uint8_t *r0 = ref + ys * ref_stride + xs;
            for (int i = 0; i < 16; i++) {
                int z = kVisit[i][0];
                uint8_t *s = src + kVisit[i][2] * 16 * src_stride + kVisit[i][1] * 16;
                uint8_t *r = r0 + kVisit[i][2] * 16 * ref_stride + kVisit[i][1] * 16;
                if (asm_type)
                    GetEightHorizontalSearchPointResults_8x8_16x16_PU_AVX2_INTRIN(s, src_stride, r, ref_stride,
                        &bs8[4 * z], &bm8[4 * z], &bs16[z], &bm16[z], mv, &eight_pos_sad16[z * 8]);
                else
                    GetEightHorizontalSearchPointResults_8x8_16x16_PU_SSE41_INTRIN(s, src_stride, r, ref_stride,
                        &bs8[4 * z], &bm8[4 * z], &bs16[z], &bm16[z], mv, &eight_pos_sad16[z * 8]);
            }
            if (asm_type)
                GetEightHorizontalSearchPointResults_32x32_64x64_PU_AVX2_INTRIN(eight_pos_sad16, bs32, bs64, bm32, bm64, mv);
            else
                GetEightHorizontalSearchPointResults_32x32_64x64_PU_SSE41_INTRIN(eight_pos_sad16, bs32, bs64, bm32, bm64, mv);
        }
        for (uint32_t xs = sw8; xs < sw; xs++) {
            int32_t xi = (int32_t)xs + x_origin, yi = (int32_t)ys + y_origin;
            uint32_t mv1 = ((uint32_t)(uint16_t)yi) << 18; /* :1261 */
            uint16_t mv2 = (uint16_t)((uint16_t)xi << 2);   /* :1262 */
            uint32_t mv = mv1 | mv2;
            uint8_t *r0 = ref + ys * ref_stride + xs;
            for (int i = 0; i < 16; i++) {
                int z = kVisit[i][0];
                uint8_t *s = src + kVisit[i][2] * 16 * src_stride + kVisit[i][1] * 16;
                uint8_t *r = r0 + kVisit[i][2] * 16 * ref_stride + kVisit[i][1] * 16;
                /* both asm rows use the SSE2 single-position kernel (Codec/EbMeSadCalculation.h) */
                SadCalculation_8x8_16x16_SSE2_INTRIN(s, src_stride, r, ref_stride, &bs8[4 * z], &bs16[z], &bm8[4 * z],
                                                     &bm16[z], mv, &sad16[z]);
            }
            SadCalculation_32x32_64x64_SSE2_INTRIN(sad16, bs32, bs64, bm32, bm64, mv);
        }
    }
}

/* Batch form used for fixtures and for the CPU timing baseline: n_sb independent searches.
 * desc[i] = {src_offset, ref_offset, x_origin, y_origin, sw, sh} (offsets into the two planes). */
void ref_fullpel_search_batch(int asm_type, uint8_t *src_plane, uint32_t src_stride, uint8_t *ref_plane,
                              uint32_t ref_stride, const int32_t *desc, uint32_t n_sb, uint32_t *best_sad,
                              uint32_t *best_mv)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d = desc + 6 * i;
        uint32_t *bs = best_sad + 85 * i, *bm = best_mv + 85 * i;
        for (int k = 0; k < 85; k++) {
            bs[k] = 128u * 128u * 255u;
            bm[k] = 0;
        }
        ref_fullpel_search_85pu(asm_type, src_plane + d[0], src_stride, ref_plane + d[1], ref_stride, (int16_t)d[2],
                                (int16_t)d[3], (uint32_t)d[4], (uint32_t)d[5], bs, bm);
    }
}
