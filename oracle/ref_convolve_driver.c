/*
 * oracle/ref_convolve_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Calls the REFERENCE's own 8-bit single-reference convolutions (compiled from /root/reference into oracle/_ref/libsvtref_me.so)
 * the way av1_inter_prediction does for the luma plane of a uni-predicted block (Codec/EbInterPrediction.c:1255-1287):
 * filter parameters from av1_get_interp_filter_params_with_block_size, conv params from get_conv_params_no_round(.., 0, EB_8BIT),
 * dispatch on (subpel_x != 0, subpel_y != 0).  Contains no reference code, only calls into it.
 */
#include <stdint.h>

#include "EbDefinitions.h"
#include "convolve.h"

InterpFilterParams av1_get_interp_filter_params_with_block_size(const InterpFilter interp_filter, const int32_t w);
#define DECL(n)                                                                                                                          \
    void n(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, int32_t w, int32_t h, InterpFilterParams *filter_params_x, \
           InterpFilterParams *filter_params_y, const int32_t subpel_x_q4, const int32_t subpel_y_q4, ConvolveParams *conv_params);
DECL(av1_convolve_2d_sr_c) DECL(av1_convolve_x_sr_c) DECL(av1_convolve_y_sr_c) DECL(av1_convolve_2d_copy_sr_c)

void ref_av1_convolve_sr(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, int32_t w, int32_t h, int filter_x,
                         int filter_y, int subpel_x, int subpel_y)
{
    uint16_t tmp[128 * 128];
    ConvolveParams cp = get_conv_params_no_round(0, 0, 0, tmp, 128, 0, EB_8BIT);
    InterpFilterParams px = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_x, w);
    InterpFilterParams py = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_y, h);
    if (subpel_x && subpel_y) av1_convolve_2d_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp);
    else if (subpel_y) av1_convolve_y_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp);
    else if (subpel_x) av1_convolve_x_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp);
    else av1_convolve_2d_copy_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp);
}

/* The luma plane of a BI_PRED block (Codec/EbInterPrediction.c:1254-1290 list 0 with do_average = 0, :1346-1385 list 1 with
 * do_average = 1): both lists through convolve[subpel_x != 0][subpel_y != 0][1] = av1_jnt_convolve_* with
 * get_conv_params_no_round(.., is_compound = 1, EB_8BIT) (round_1 = COMPOUND_ROUND1_BITS, use_jnt_comp_avg = 0) and the shared
 * 16-bit tmp_dstY buffer of stride 128; one interp_filters pair for both lists. */
DECL(av1_jnt_convolve_2d_c) DECL(av1_jnt_convolve_x_c) DECL(av1_jnt_convolve_y_c) DECL(av1_jnt_convolve_2d_copy_c)

void ref_av1_convolve_compound(const uint8_t *src0, int32_t src0_stride, const uint8_t *src1, int32_t src1_stride, uint8_t *dst,
                               int32_t dst_stride, int32_t w, int32_t h, int filter_x, int filter_y, int subpel_x0, int subpel_y0,
                               int subpel_x1, int subpel_y1)
{
    static uint16_t tmp[128 * 128];
    InterpFilterParams px = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_x, w);
    InterpFilterParams py = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_y, h);
    for (int list = 0; list < 2; list++) {
        ConvolveParams cp = get_conv_params_no_round(0, list, 0, tmp, 128, 1, EB_8BIT);
        const uint8_t *src = list ? src1 : src0;
        const int32_t st = list ? src1_stride : src0_stride, sx = list ? subpel_x1 : subpel_x0, sy = list ? subpel_y1 : subpel_y0;
        if (sx && sy) av1_jnt_convolve_2d_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp);
        else if (sy) av1_jnt_convolve_y_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp);
        else if (sx) av1_jnt_convolve_x_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp);
        else av1_jnt_convolve_2d_copy_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp);
    }
}

/* 10-bit video in 16-bit planes: the highbd forms av1_inter_prediction's 16-bit twin dispatches (convolveHbd[..][..][is_compound],
 * Codec/EbInterPrediction.c:882-895, C bodies :530-880), conv params from get_conv_params_no_round(.., bd). */
#define DECLH(n)                                                                                                                       \
    void n(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, int32_t w, int32_t h,                          \
           const InterpFilterParams *filter_params_x, const InterpFilterParams *filter_params_y, const int32_t subpel_x_q4,           \
           const int32_t subpel_y_q4, ConvolveParams *conv_params, int32_t bd);
DECLH(av1_highbd_convolve_2d_sr_c) DECLH(av1_highbd_convolve_x_sr_c) DECLH(av1_highbd_convolve_y_sr_c) DECLH(av1_highbd_convolve_2d_copy_sr_c)
DECLH(av1_highbd_jnt_convolve_2d_c) DECLH(av1_highbd_jnt_convolve_x_c) DECLH(av1_highbd_jnt_convolve_y_c) DECLH(av1_highbd_jnt_convolve_2d_copy_c)

void ref_av1_highbd_convolve_sr(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, int32_t w, int32_t h, int filter_x,
                                int filter_y, int subpel_x, int subpel_y, int bd)
{
    static uint16_t tmp[128 * 128];
    ConvolveParams cp = get_conv_params_no_round(0, 0, 0, tmp, 128, 0, bd);
    InterpFilterParams px = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_x, w);
    InterpFilterParams py = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_y, h);
    if (subpel_x && subpel_y) av1_highbd_convolve_2d_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp, bd);
    else if (subpel_y) av1_highbd_convolve_y_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp, bd);
    else if (subpel_x) av1_highbd_convolve_x_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp, bd);
    else av1_highbd_convolve_2d_copy_sr_c(src, src_stride, dst, dst_stride, w, h, &px, &py, subpel_x, subpel_y, &cp, bd);
}

void ref_av1_highbd_convolve_compound(const uint16_t *src0, int32_t src0_stride, const uint16_t *src1, int32_t src1_stride, uint16_t *dst,
                                      int32_t dst_stride, int32_t w, int32_t h, int filter_x, int filter_y, int subpel_x0, int subpel_y0,
                                      int subpel_x1, int subpel_y1, int bd)
{
    static uint16_t tmp[128 * 128];
    InterpFilterParams px = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_x, w);
    InterpFilterParams py = av1_get_interp_filter_params_with_block_size((InterpFilter)filter_y, h);
    for (int list = 0; list < 2; list++) {
        ConvolveParams cp = get_conv_params_no_round(0, list, 0, tmp, 128, 1, bd);
        const uint16_t *src = list ? src1 : src0;
        const int32_t st = list ? src1_stride : src0_stride, sx = list ? subpel_x1 : subpel_x0, sy = list ? subpel_y1 : subpel_y0;
        if (sx && sy) av1_highbd_jnt_convolve_2d_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp, bd);
        else if (sy) av1_highbd_jnt_convolve_y_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp, bd);
        else if (sx) av1_highbd_jnt_convolve_x_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp, bd);
        else av1_highbd_jnt_convolve_2d_copy_c(src, st, dst, dst_stride, w, h, &px, &py, sx, sy, &cp, bd);
    }
}
