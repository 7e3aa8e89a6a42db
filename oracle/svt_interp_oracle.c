/*
 * oracle/svt_interp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 *
 * Plain-C restatement of the 8-bit single-reference AV1 inter-prediction convolutions, paths under
 * Source/Lib/Codec of the reference:
 *   av1_convolve_2d_sr_c       EbInterPrediction.c:145-198   horizontal 8-tap -> int16 (round_0 = 3, offset 1 << 14), vertical 8-tap
 *                                                            (round_1 = 11, offset 1 << 19 removed again), clip
 *   av1_convolve_y_sr_c        :200-232                      vertical only, (sum + 64) >> 7
 *   av1_convolve_x_sr_c        :234-267                      horizontal only, ((sum + 4) >> 3 + 8) >> 4  (two roundings)
 *   av1_convolve_2d_copy_sr_c  :269-286
 *   dispatch convolve[subpel_x != 0][subpel_y != 0][0]       :898-911, call site :1276-1287
 *   filter choice av1_get_interp_filter_params_with_block_size :985-995 (4-tap kernels for blocks <= 4 wide / high)
 *   conv params get_conv_params_no_round(.., is_compound = 0, bd = 8): round_0 = 3, round_1 = 11 (convolve.h:115-143)
 *   av1_jnt_convolve_{2d,x,y,2d_copy}_c  :290-528           the compound (BI_PRED) forms: 16-bit intermediate per list, plain average
 * PINNED against the reference's own functions (oracle/ref_convolve_driver.c) in tests/test_convolve_vs_ref.py for every filter,
 * every phase pair and every AV1 block size.
 */
#include <stdint.h>

static const int16_t kFilters[6][16][8] =
#include "svt_interp_filters.inc"
    ;

static inline uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* index into kFilters for InterpFilter `f` (0 regular, 1 smooth, 2 sharp, 3 bilinear) on a dimension of `size` samples */
static int filter_index(int f, int size)
{
    if (size <= 4 && (f == 2 || f == 0)) return 4;
    if (size <= 4 && f == 1) return 5;
    return f;
}

void orc_av1_convolve_sr(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, int32_t w, int32_t h, int filter_x,
                         int filter_y, int subpel_x, int subpel_y)
{
    const int16_t *fx = kFilters[filter_index(filter_x, w)][subpel_x & 15], *fy = kFilters[filter_index(filter_y, h)][subpel_y & 15];
    if (subpel_x && subpel_y) {
        static int16_t im[(128 + 7) * 128];
        const int im_h = h + 7;
        const uint8_t *s = src - 3 * src_stride;
        for (int y = 0; y < im_h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << 14; /* 1 << (bd + FILTER_BITS - 1) */
                for (int k = 0; k < 8; k++) sum += fx[k] * s[y * src_stride + x - 3 + k];
                im[y * w + x] = (int16_t)((sum + 4) >> 3);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << 19; /* offset_bits = bd + 2 * FILTER_BITS - round_0 */
                for (int k = 0; k < 8; k++) sum += fy[k] * im[(y + k) * w + x];
                const int16_t res = (int16_t)(uint16_t)(((sum + (1 << 10)) >> 11) - ((1 << 8) + (1 << 7)));
                dst[y * dst_stride + x] = clip8(res); /* bits = 0 */
            }
    } else if (subpel_y) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t res = 0;
                for (int k = 0; k < 8; k++) res += fy[k] * src[(y - 3 + k) * src_stride + x];
                dst[y * dst_stride + x] = clip8((res + 64) >> 7);
            }
    } else if (subpel_x) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t res = 0;
                for (int k = 0; k < 8; k++) res += fx[k] * src[y * src_stride + x - 3 + k];
                res = (res + 4) >> 3;
                dst[y * dst_stride + x] = clip8((res + 8) >> 4);
            }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) dst[y * dst_stride + x] = src[y * src_stride + x];
    }
}

/* One list of a compound (BI_PRED) block: the 16-bit value av1_jnt_convolve_{2d,x,y,2d_copy}_c store into conv_params->dst
 * (EbInterPrediction.c:290-528; round_0 = 3, round_1 = COMPOUND_ROUND1_BITS = 7: offset_bits = 19, round_offset = 6144). */
static void compound_list(const uint8_t *src, int32_t src_stride, uint16_t *res, int32_t w, int32_t h, const int16_t *fx, const int16_t *fy,
                          int subpel_x, int subpel_y)
{
    if (subpel_x && subpel_y) {
        static int16_t im[(128 + 7) * 128];
        const uint8_t *s = src - 3 * src_stride;
        for (int y = 0; y < h + 7; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << 14;
                for (int k = 0; k < 8; k++) sum += fx[k] * s[y * src_stride + x - 3 + k];
                im[y * w + x] = (int16_t)((sum + 4) >> 3);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << 19;
                for (int k = 0; k < 8; k++) sum += fy[k] * im[(y + k) * w + x];
                res[y * w + x] = (uint16_t)((sum + 64) >> 7);
            }
    } else if (subpel_y) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t r = 0;
                for (int k = 0; k < 8; k++) r += fy[k] * src[(y - 3 + k) * src_stride + x];
                r *= 16; /* 1 << (FILTER_BITS - round_0) */
                res[y * w + x] = (uint16_t)(((r + 64) >> 7) + 6144);
            }
    } else if (subpel_x) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t r = 0;
                for (int k = 0; k < 8; k++) r += fx[k] * src[y * src_stride + x - 3 + k];
                res[y * w + x] = (uint16_t)(((r + 4) >> 3) + 6144); /* bits = FILTER_BITS - round_1 = 0 */
            }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) res[y * w + x] = (uint16_t)((src[y * src_stride + x] << 4) + 6144);
    }
}

/* BI_PRED luma of av1_inter_prediction (:1254-1290 + :1346-1385): list 0 into the 16-bit buffer, list 1 averaged with it
 * (use_jnt_comp_avg = 0), round_bits = 4. */
void orc_av1_convolve_compound(const uint8_t *src0, int32_t src0_stride, const uint8_t *src1, int32_t src1_stride, uint8_t *dst,
                               int32_t dst_stride, int32_t w, int32_t h, int filter_x, int filter_y, int subpel_x0, int subpel_y0,
                               int subpel_x1, int subpel_y1)
{
    static uint16_t r0[128 * 128], r1[128 * 128];
    const int fxi = filter_index(filter_x, w), fyi = filter_index(filter_y, h);
    compound_list(src0, src0_stride, r0, w, h, kFilters[fxi][subpel_x0 & 15], kFilters[fyi][subpel_y0 & 15], subpel_x0, subpel_y0);
    compound_list(src1, src1_stride, r1, w, h, kFilters[fxi][subpel_x1 & 15], kFilters[fyi][subpel_y1 & 15], subpel_x1, subpel_y1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t tmp = ((int32_t)r0[y * w + x] + (int32_t)r1[y * w + x]) >> 1;
            tmp -= 6144;
            dst[y * dst_stride + x] = clip8((tmp + 8) >> 4);
        }
}

/* batch form of the device entry: desc = { src0_offset, src1_offset, dst_offset,
 * subpel_x0 | subpel_y0 << 4 | subpel_x1 << 8 | subpel_y1 << 12 | filter_x << 16 | filter_y << 24 } */
void orc_av1_convolve_compound_batch(const uint8_t *src0, int32_t src0_stride, const uint8_t *src1, int32_t src1_stride, uint8_t *dst,
                                     int32_t dst_stride, const uint32_t *desc, uint32_t n, int32_t w, int32_t h)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *d = desc + 4 * i, p = d[3];
        orc_av1_convolve_compound(src0 + d[0], src0_stride, src1 + d[1], src1_stride, dst + d[2], dst_stride, w, h, (p >> 16) & 255, (p >> 24) & 255,
                                  p & 15, (p >> 4) & 15, (p >> 8) & 15, (p >> 12) & 15);
    }
}

/* ---- 10-bit video in 16-bit planes: av1_highbd_convolve_*_sr_c (:530-660) and av1_highbd_jnt_convolve_*_c (:661-880).  The same arithmetic
 * with bd in the offsets: pass-1 offset 1 << (bd + 6), offset_bits = bd + 11, round_offset = (1 << (bd + 4)) + (1 << (bd + 3)) in the
 * compound forms; round_0 = 3 holds for bd <= 10 (get_conv_params_no_round raises it only from bd = 12). ---- */
static uint16_t clip_bd(int v, int bd) { const int m = (1 << bd) - 1; return (uint16_t)(v < 0 ? 0 : (v > m ? m : v)); }

static void hbd_list(const uint16_t *src, int32_t src_stride, int32_t *res, int32_t w, int32_t h, const int16_t *fx, const int16_t *fy, int subpel_x,
                     int subpel_y, int bd, int compound)
{
    const int ro = (1 << (bd + 4)) + (1 << (bd + 3));
    if (subpel_x && subpel_y) {
        static int16_t im[(128 + 7) * 128];
        const uint16_t *s = src - 3 * src_stride;
        for (int y = 0; y < h + 7; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << (bd + 6);
                for (int k = 0; k < 8; k++) sum += fx[k] * s[y * src_stride + x - 3 + k];
                im[y * w + x] = (int16_t)((sum + 4) >> 3);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << (bd + 11);
                for (int k = 0; k < 8; k++) sum += fy[k] * im[(y + k) * w + x];
                res[y * w + x] = compound ? (int32_t)(uint16_t)((sum + 64) >> 7) : ((sum + 1024) >> 11) - ((1 << bd) + (1 << (bd - 1)));
            }
    } else if (subpel_y) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t r = 0;
                for (int k = 0; k < 8; k++) r += fy[k] * src[(y - 3 + k) * src_stride + x];
                res[y * w + x] = compound ? (int32_t)(uint16_t)((((r * 16) + 64) >> 7) + ro) : ((r + 64) >> 7);
            }
    } else if (subpel_x) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t r = 0;
                for (int k = 0; k < 8; k++) r += fx[k] * src[y * src_stride + x - 3 + k];
                r = (r + 4) >> 3;
                res[y * w + x] = compound ? (int32_t)(uint16_t)(r + ro) : ((r + 8) >> 4);
            }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) res[y * w + x] = compound ? (int32_t)(uint16_t)((src[y * src_stride + x] << 4) + ro) : src[y * src_stride + x];
    }
}

void orc_av1_highbd_convolve_sr(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, int32_t w, int32_t h, int filter_x,
                                int filter_y, int subpel_x, int subpel_y, int bd)
{
    static int32_t r[128 * 128];
    hbd_list(src, src_stride, r, w, h, kFilters[filter_index(filter_x, w)][subpel_x & 15], kFilters[filter_index(filter_y, h)][subpel_y & 15], subpel_x,
             subpel_y, bd, 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * dst_stride + x] = clip_bd(r[y * w + x], bd);
}

void orc_av1_highbd_convolve_compound(const uint16_t *src0, int32_t src0_stride, const uint16_t *src1, int32_t src1_stride, uint16_t *dst,
                                      int32_t dst_stride, int32_t w, int32_t h, int filter_x, int filter_y, int subpel_x0, int subpel_y0,
                                      int subpel_x1, int subpel_y1, int bd)
{
    static int32_t r0[128 * 128], r1[128 * 128];
    const int fxi = filter_index(filter_x, w), fyi = filter_index(filter_y, h), ro = (1 << (bd + 4)) + (1 << (bd + 3));
    hbd_list(src0, src0_stride, r0, w, h, kFilters[fxi][subpel_x0 & 15], kFilters[fyi][subpel_y0 & 15], subpel_x0, subpel_y0, bd, 1);
    hbd_list(src1, src1_stride, r1, w, h, kFilters[fxi][subpel_x1 & 15], kFilters[fyi][subpel_y1 & 15], subpel_x1, subpel_y1, bd, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * dst_stride + x] = clip_bd((((r0[y * w + x] + r1[y * w + x]) >> 1) - ro + 8) >> 4, bd);
}

/* batch forms (offsets and strides in SAMPLES; descriptors as in the 8-bit batches) */
void orc_av1_highbd_convolve_sr_batch(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, const uint32_t *desc, uint32_t n,
                                      int32_t w, int32_t h, int bd)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *d = desc + 4 * i;
        orc_av1_highbd_convolve_sr(src + d[0], src_stride, dst + d[1], dst_stride, w, h, (d[2] >> 16) & 255, (d[2] >> 24) & 255, d[2] & 15, (d[2] >> 8) & 15,
                                   bd);
    }
}

void orc_av1_highbd_convolve_compound_batch(const uint16_t *src0, int32_t src0_stride, const uint16_t *src1, int32_t src1_stride, uint16_t *dst,
                                            int32_t dst_stride, const uint32_t *desc, uint32_t n, int32_t w, int32_t h, int bd)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *d = desc + 4 * i, p = d[3];
        orc_av1_highbd_convolve_compound(src0 + d[0], src0_stride, src1 + d[1], src1_stride, dst + d[2], dst_stride, w, h, (p >> 16) & 255,
                                         (p >> 24) & 255, p & 15, (p >> 4) & 15, (p >> 8) & 15, (p >> 12) & 15, bd);
    }
}

/* the batch the device entry takes: desc = { src_offset, dst_offset, subpel_x | subpel_y << 8 | filter_x << 16 | filter_y << 24, 0 } */
void orc_av1_convolve_sr_batch(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, const uint32_t *desc, uint32_t n,
                               int32_t w, int32_t h)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *d = desc + 4 * i;
        orc_av1_convolve_sr(src + d[0], src_stride, dst + d[1], dst_stride, w, h, (d[2] >> 16) & 255, (d[2] >> 24) & 255, d[2] & 255, (d[2] >> 8) & 255);
    }
}
