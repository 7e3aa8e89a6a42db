/*
 * oracle/svt_hme_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 *
 * Plain-C restatement of the search-centre half of MotionEstimateLcu
 * (Source/Lib/Codec/EbMotionEstimation.c:6300-6738): hme_mv_center_check, HmeLevel0/1/2, region pick,
 * CheckZeroZeroCenter and the search-window clipping.  Statement order, int16 truncations and the
 * reference's quirks (stale index for candidate A, un-shrunk width at the left/top edge, unclamped centre
 * returned by the centre check) are reproduced literally.
 *
 * Pinned by tests/test_hme_vs_ref.py against the reference's own MotionEstimateLcu where that can run
 * (oracle/_ref, sub-pel disabled), see DESIGN.md "oracle".
 */
#include "svt_me_oracle.h"

#define MAXV(a, b) ((a) > (b) ? (a) : (b))

static const uint8_t *full_at(const uint8_t *pool, const svthip_pa_picture *p, int x, int y)
{
    return pool + p->full_offset + (int64_t)(68 + y) * p->full_stride + 68 + x;
}

/* NxMSadKernel on every other row, doubled: the "<< subsampleSad" pattern of the centre checks */
static uint32_t sad_sb_subsampled(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                  uint32_t sb_width, uint32_t sb_height)
{
    return orc_nxm_sad(src, src_stride << 1, ref, ref_stride << 1, sb_height >> 1, sb_width) << 1;
}

static void clamp_center(int16_t *cx, int16_t *cy, int16_t origin_x, int16_t origin_y, int16_t pic_w, int16_t pic_h)
{
    const int16_t pad_width = 63, pad_height = 63;
    int16_t x = *cx, y = *cy;
    x = ((origin_x + x) < -pad_width) ? (int16_t)(-pad_width - origin_x) : x;
    x = ((origin_x + x) > pic_w - 1) ? (int16_t)(x - ((origin_x + x) - (pic_w - 1))) : x;
    y = ((origin_y + y) < -pad_height) ? (int16_t)(-pad_height - origin_y) : y;
    y = ((origin_y + y) > pic_h - 1) ? (int16_t)(y - ((origin_y + y) - (pic_h - 1))) : y;
    *cx = x;
    *cy = y;
}

/* Codec/EbMotionEstimation.c:5882-6145 */
static void hme_mv_center_check(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                                const svthip_me_params *p, uint32_t list_index, int16_t origin_x, int16_t origin_y,
                                uint32_t sb_width, uint32_t sb_height, uint32_t l0_best_mv64, int16_t *xsc, int16_t *ysc)
{
    const uint8_t *src = full_at(pool, cur, origin_x, origin_y);
    const int16_t pw = (int16_t)ref->width, ph = (int16_t)ref->height;
    const uint64_t COSTP = 8; /* COST_PRECISION */
    int16_t cx, cy;

    uint64_t zero_mv_cost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x, origin_y),
                                                         ref->full_stride, sb_width, sb_height) << COSTP;
    /* A: clipped centre is computed but the SAD is taken at the stale (zero-MV) index (:5953-5959) */
    uint64_t mv_a_cost = zero_mv_cost;

    cx = (int16_t)p->hme_level0_total_search_area_width; cy = 0; /* B */
    clamp_center(&cx, &cy, origin_x, origin_y, pw, ph);
    uint64_t mv_b_cost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x + cx, origin_y + cy),
                                                     ref->full_stride, sb_width, sb_height) << COSTP;
    cx = 0; cy = (int16_t)(0 - p->hme_level0_total_search_area_height); /* C */
    clamp_center(&cx, &cy, origin_x, origin_y, pw, ph);
    uint64_t mv_c_cost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x + cx, origin_y + cy),
                                                     ref->full_stride, sb_width, sb_height) << COSTP;
    cx = 0; cy = (int16_t)p->hme_level0_total_search_area_height; /* D */
    clamp_center(&cx, &cy, origin_x, origin_y, pw, ph);
    uint64_t mv_d_cost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x + cx, origin_y + cy),
                                                     ref->full_stride, sb_width, sb_height) << COSTP;
    uint64_t direct_mv_cost = 0xFFFFFFFFFFFFFull;
    const int16_t dx = (int16_t)(0 - ((int16_t)(l0_best_mv64 & 0xffff) >> 2));
    const int16_t dy = (int16_t)(0 - ((int16_t)(l0_best_mv64 >> 16) >> 2));
    if (list_index == 1) {
        cx = dx; cy = dy;
        clamp_center(&cx, &cy, origin_x, origin_y, pw, ph);
        direct_mv_cost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x + cx, origin_y + cy),
                                                     ref->full_stride, sb_width, sb_height) << COSTP;
    }
    uint64_t best = zero_mv_cost;
    if (mv_a_cost < best) best = mv_a_cost;
    if (mv_b_cost < best) best = mv_b_cost;
    if (mv_c_cost < best) best = mv_c_cost;
    if (mv_d_cost < best) best = mv_d_cost;
    if (direct_mv_cost < best) best = direct_mv_cost;
    /* returned centre is the UNCLAMPED candidate (:6114-6137) */
    if (best == zero_mv_cost) { cx = 0; cy = 0; }
    else if (best == mv_a_cost) { cx = (int16_t)(0 - p->hme_level0_total_search_area_width); cy = 0; }
    else if (best == mv_b_cost) { cx = (int16_t)p->hme_level0_total_search_area_width; cy = 0; }
    else if (best == mv_c_cost) { cx = 0; cy = (int16_t)(0 - p->hme_level0_total_search_area_height); }
    else if (best == direct_mv_cost) { cx = list_index ? dx : 0; cy = list_index ? dy : 0; }
    else { cx = 0; cy = (int16_t)p->hme_level0_total_search_area_height; }
    *xsc = cx;
    *ysc = cy;
}

/* the four-statement-per-axis window clipping shared by HmeLevel0/1/2 and the full-pel window
 * (:4378-4411, :4541-4573, :4674-4707, :6690-6723): statement 2 re-reads the corrected origin and never fires */
static void clip_window(int16_t *xo_, int16_t *yo_, int16_t *sw_, int16_t *sh_, int16_t origin_x, int16_t origin_y,
                        int16_t pad_w, int16_t pad_h, int16_t pic_w, int16_t pic_h)
{
    int16_t xo = *xo_, yo = *yo_, sw = *sw_, sh = *sh_;
    xo = ((origin_x + xo) < -pad_w) ? (int16_t)(-pad_w - origin_x) : xo;
    sw = ((origin_x + xo) < -pad_w) ? (int16_t)(sw - (-pad_w - (origin_x + xo))) : sw;
    xo = ((origin_x + xo) > pic_w - 1) ? (int16_t)(xo - ((origin_x + xo) - (pic_w - 1))) : xo;
    sw = ((origin_x + xo + sw) > pic_w) ? (int16_t)MAXV(1, sw - ((origin_x + xo + sw) - pic_w)) : sw;
    yo = ((origin_y + yo) < -pad_h) ? (int16_t)(-pad_h - origin_y) : yo;
    sh = ((origin_y + yo) < -pad_h) ? (int16_t)(sh - (-pad_h - (origin_y + yo))) : sh;
    yo = ((origin_y + yo) > pic_h - 1) ? (int16_t)(yo - ((origin_y + yo) - (pic_h - 1))) : yo;
    sh = ((origin_y + yo + sh) > pic_h) ? (int16_t)MAXV(1, sh - ((origin_y + yo + sh) - pic_h)) : sh;
    *xo_ = xo; *yo_ = yo; *sw_ = sw; *sh_ = sh;
}

/* Codec/EbMotionEstimation.c:4306-4503.  origin/size/centre are already in 1/16-picture units. */
static void hme_level0(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                       const svthip_me_params *p, int16_t origin_x, int16_t origin_y, uint32_t sb_width, uint32_t sb_height,
                       int16_t xc, int16_t yc, uint32_t rw, uint32_t rh, uint64_t *best_sad, int16_t *xout, int16_t *yout)
{
    const uint32_t mx = p->hme_level0_multiplier_x, my = p->hme_level0_multiplier_y;
    int16_t sw = (int16_t)((p->hme_level0_search_area_in_width_array[rw] * mx) / 100);
    int16_t sh = (int16_t)((p->hme_level0_search_area_in_height_array[rh] * my) / 100);
    int16_t xdist = xc, ydist = yc;
    const int16_t pic_w = (int16_t)(ref->width >> 2), pic_h = (int16_t)(ref->height >> 2);
    uint32_t k = rw;
    while (k) { k--; xdist += (int16_t)((p->hme_level0_search_area_in_width_array[k] * mx) / 100); }
    k = rh;
    while (k) { k--; ydist += (int16_t)((p->hme_level0_search_area_in_height_array[k] * my) / 100); }
    int16_t xo = (int16_t)(-(int16_t)(((p->hme_level0_total_search_area_width * mx) / 100) >> 1) + xdist);
    int16_t yo = (int16_t)(-(int16_t)(((p->hme_level0_total_search_area_height * my) / 100) >> 1) + ydist);
    clip_window(&xo, &yo, &sw, &sh, origin_x, origin_y, 16 - 1, 16 - 1, pic_w, pic_h);

    /* source = even rows of the 1/16 SB, packed at stride 16 (Codec/EbMotionEstimationProcess.c:530-544):
     * read here straight from the plane with a doubled stride, which is the same bytes */
    const uint8_t *src = pool + cur->sixteenth_offset + (int64_t)(16 + origin_y) * cur->sixteenth_stride + 16 + origin_x;
    const uint8_t *r = pool + ref->sixteenth_offset + (int64_t)(16 + origin_y + yo) * ref->sixteenth_stride + 16 + origin_x + xo;
    int16_t bx = 0, by = 0;
    orc_sad_loop_kernel(src, cur->sixteenth_stride * 2, r, ref->sixteenth_stride * 2, sb_height >> 1, sb_width, best_sad, &bx,
                        &by, ref->sixteenth_stride, sw, sh);
    *best_sad *= 2;
    *xout = (int16_t)((int16_t)(bx + xo) * 4);
    *yout = (int16_t)((int16_t)(by + yo) * 4);
}

static int16_t round_hme_width(int16_t w)
{
    /* :4528 / :4658 -- not a round-up: adds the remainder */
    return (w < 8) ? 8 : (w & 7) ? (int16_t)(w + (w - ((w >> 3) << 3))) : w;
}

/* Codec/EbMotionEstimation.c:4505-4625 (quarter-picture units) */
static void hme_level1(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref, int16_t origin_x,
                       int16_t origin_y, uint32_t sb_width, uint32_t sb_height, int16_t area_w, int16_t area_h, int16_t xc,
                       int16_t yc, uint64_t *best_sad, int16_t *xout, int16_t *yout)
{
    int16_t sw = round_hme_width(area_w), sh = area_h;
    int16_t xo = (int16_t)(-(sw >> 1) + xc), yo = (int16_t)(-(sh >> 1) + yc);
    clip_window(&xo, &yo, &sw, &sh, origin_x, origin_y, 32 - 1, 32 - 1, (int16_t)(ref->width >> 1), (int16_t)(ref->height >> 1));
    const uint8_t *src = pool + cur->quarter_offset + (int64_t)(32 + origin_y) * cur->quarter_stride + 32 + origin_x;
    const uint8_t *r = pool + ref->quarter_offset + (int64_t)(32 + origin_y + yo) * ref->quarter_stride + 32 + origin_x + xo;
    int16_t bx = 0, by = 0;
    orc_sad_loop_kernel(src, cur->quarter_stride * 2, r, ref->quarter_stride * 2, sb_height >> 1, sb_width, best_sad, &bx, &by,
                        ref->quarter_stride, sw, sh);
    *best_sad *= 2;
    *xout = (int16_t)((int16_t)(bx + xo) * 2);
    *yout = (int16_t)((int16_t)(by + yo) * 2);
}

/* Codec/EbMotionEstimation.c:4627-4758 (full resolution) */
static void hme_level2(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                       const svthip_me_params *p, int16_t origin_x, int16_t origin_y, uint32_t sb_width, uint32_t sb_height,
                       uint32_t rw, uint32_t rh, int16_t xc, int16_t yc, uint64_t *best_sad, int16_t *xout, int16_t *yout)
{
    int16_t sw = round_hme_width((int16_t)p->hme_level2_search_area_in_width_array[rw]);
    int16_t sh = (int16_t)p->hme_level2_search_area_in_height_array[rh];
    int16_t xo = (int16_t)(-(sw >> 1) + xc), yo = (int16_t)(-(sh >> 1) + yc);
    clip_window(&xo, &yo, &sw, &sh, origin_x, origin_y, 63, 63, (int16_t)ref->width, (int16_t)ref->height);
    const uint8_t *src = full_at(pool, cur, origin_x, origin_y);
    const uint8_t *r = full_at(pool, ref, origin_x + xo, origin_y + yo);
    int16_t bx = 0, by = 0;
    orc_sad_loop_kernel(src, cur->full_stride * 2, r, ref->full_stride * 2, sb_height >> 1, sb_width, best_sad, &bx, &by,
                        ref->full_stride, sw, sh);
    *best_sad *= 2;
    *xout = (int16_t)(bx + xo);
    *yout = (int16_t)(by + yo);
}

/* hme_state: 25 int16 per SB carried from the list-0 call to the list-1 call of the same SB
 * ([0..23] = x0,y0,x1,y1,x2,y2 as [w][h], [24] = "centre arrays already initialised").  The reference keeps
 * these arrays, and the loop counters that guard their initialisation (:6325-6345), alive across the list
 * loop, so list 1 starts from list 0's values whenever a level is disabled.  NULL = no carry-over. */
void orc_hme_search_center(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                           const svthip_me_params *p, uint32_t list_index, uint32_t sb_origin_x, uint32_t sb_origin_y,
                           uint32_t l0_best_mv64, svthip_fullpel_desc *desc, int16_t *center_xy, int16_t *hme_state)
{
    const int16_t picture_width = (int16_t)cur->width, picture_height = (int16_t)cur->height;
    const uint32_t sb_width = (cur->width - sb_origin_x) < 64 ? cur->width - sb_origin_x : 64;
    const uint32_t sb_height = (cur->height - sb_origin_y) < 64 ? cur->height - sb_origin_y : 64;
    const int16_t origin_x = (int16_t)sb_origin_x, origin_y = (int16_t)sb_origin_y;
    int16_t xSearchCenter = 0, ySearchCenter = 0;
    const uint32_t nw = p->number_hme_search_region_in_width, nh = p->number_hme_search_region_in_height;

    if (p->temporal_layer_index > 0 || list_index == 0) { /* :6300 */
        hme_mv_center_check(pool, cur, ref, p, list_index, origin_x, origin_y, sb_width, sb_height, l0_best_mv64,
                            &xSearchCenter, &ySearchCenter);
        if (p->enable_hme_flag && sb_height == 64) { /* :6323 */
            int16_t x0[2][2] = {{0}}, y0[2][2] = {{0}}, x1[2][2] = {{0}}, y1[2][2] = {{0}}, x2[2][2] = {{0}}, y2[2][2] = {{0}};
            uint64_t s0[2][2] = {{0}}, s1[2][2] = {{0}}, s2[2][2] = {{0}};
            int16_t xHme = 0, yHme = 0;
            uint64_t hmeSad = 0;
            const int carried = hme_state && list_index == 1 && hme_state[24];
            if (carried) {
                for (int k = 0; k < 4; k++) {
                    x0[k >> 1][k & 1] = hme_state[k]; y0[k >> 1][k & 1] = hme_state[4 + k];
                    x1[k >> 1][k & 1] = hme_state[8 + k]; y1[k >> 1][k & 1] = hme_state[12 + k];
                    x2[k >> 1][k & 1] = hme_state[16 + k]; y2[k >> 1][k & 1] = hme_state[20 + k];
                }
            } else {
                for (uint32_t h = 0; h < nh; h++)
                    for (uint32_t w = 0; w < nw; w++) {
                        x0[w][h] = x1[w][h] = x2[w][h] = xSearchCenter;
                        y0[w][h] = y1[w][h] = y2[w][h] = ySearchCenter;
                    }
            }
            if (p->enable_hme_level0_flag)
                for (uint32_t h = 0; h < nh; h++)
                    for (uint32_t w = 0; w < nw; w++)
                        hme_level0(pool, cur, ref, p, origin_x >> 2, origin_y >> 2, sb_width >> 2, sb_height >> 2,
                                   xSearchCenter >> 2, ySearchCenter >> 2, w, h, &s0[w][h], &x0[w][h], &y0[w][h]);
            if (p->enable_hme_level1_flag)
                for (uint32_t h = 0; h < nh; h++)
                    for (uint32_t w = 0; w < nw; w++)
                        hme_level1(pool, cur, ref, origin_x >> 1, origin_y >> 1, sb_width >> 1, sb_height >> 1,
                                   (int16_t)p->hme_level1_search_area_in_width_array[w],
                                   (int16_t)p->hme_level1_search_area_in_height_array[h], x0[w][h] >> 1, y0[w][h] >> 1,
                                   &s1[w][h], &x1[w][h], &y1[w][h]);
            if (p->enable_hme_level2_flag)
                for (uint32_t h = 0; h < nh; h++)
                    for (uint32_t w = 0; w < nw; w++)
                        hme_level2(pool, cur, ref, p, origin_x, origin_y, sb_width, sb_height, w, h, x1[w][h], y1[w][h],
                                   &s2[w][h], &x2[w][h], &y2[w][h]);

            /* region pick: starts from [0][0], then [w][h] in w-inner order from w=1, strict '<' (:6510-6596) */
            int16_t(*xs)[2] = NULL, (*ys)[2] = NULL;
            uint64_t(*ss)[2] = NULL;
            if (p->enable_hme_level0_flag && !p->enable_hme_level1_flag && !p->enable_hme_level2_flag) { xs = x0; ys = y0; ss = s0; }
            if (p->enable_hme_level1_flag && !p->enable_hme_level2_flag) { xs = x1; ys = y1; ss = s1; }
            if (p->enable_hme_level2_flag) { xs = x2; ys = y2; ss = s2; }
            if (xs) {
                xHme = xs[0][0]; yHme = ys[0][0]; hmeSad = ss[0][0];
                uint32_t w = 1, h = 0;
                while (h < nh) {
                    while (w < nw) {
                        if (ss[w][h] < hmeSad) { xHme = xs[w][h]; yHme = ys[w][h]; hmeSad = ss[w][h]; }
                        w++;
                    }
                    w = 0;
                    h++;
                }
            }
            if (p->enable_hme_level2_flag) {
                /* same-POC list 1: bubble-sort the regions by SAD, indexing [q / nw][q % nw] (sic), then take [0][1] (:6606-6631) */
                const uint32_t total = nh * nw;
                if (p->ref_poc_equal && list_index == 1 && total > 1) {
                    for (uint32_t q = 0; q < total - 1; q++)
                        for (uint32_t n = q + 1; n < total; n++) {
                            uint32_t a0 = q / nw, a1 = q % nw, b0 = n / nw, b1 = n % nw;
                            if (s2[a0][a1] > s2[b0][b1]) {
                                int16_t tx = x2[a0][a1], ty = y2[a0][a1];
                                uint64_t ts = s2[a0][a1];
                                x2[a0][a1] = x2[b0][b1]; y2[a0][a1] = y2[b0][b1]; s2[a0][a1] = s2[b0][b1];
                                x2[b0][b1] = tx; y2[b0][b1] = ty; s2[b0][b1] = ts;
                            }
                        }
                    xHme = x2[0][1];
                    yHme = y2[0][1];
                }
            }
            xSearchCenter = xHme;
            ySearchCenter = yHme;
            if (hme_state) {
                for (int k = 0; k < 4; k++) {
                    hme_state[k] = x0[k >> 1][k & 1]; hme_state[4 + k] = y0[k >> 1][k & 1];
                    hme_state[8 + k] = x1[k >> 1][k & 1]; hme_state[12 + k] = y1[k >> 1][k & 1];
                    hme_state[16 + k] = x2[k >> 1][k & 1]; hme_state[20 + k] = y2[k >> 1][k & 1];
                }
                hme_state[24] = 1;
            }
        }
    } else {
        xSearchCenter = 0;
        ySearchCenter = 0;
    }

    int16_t search_area_width = (int16_t)(p->search_area_width < 127 ? p->search_area_width : 127);
    int16_t search_area_height = (int16_t)(p->search_area_height < 127 ? p->search_area_height : 127);

    if ((xSearchCenter != 0 || ySearchCenter != 0) && p->is_used_as_reference_flag) { /* :6651, CheckZeroZeroCenter :5466-5552 */
        const uint8_t *src = full_at(pool, cur, origin_x, origin_y);
        uint64_t zeroMvCost = (uint64_t)sad_sb_subsampled(src, cur->full_stride, full_at(pool, ref, origin_x, origin_y),
                                                          ref->full_stride, sb_width, sb_height) << 8;
        clamp_center(&xSearchCenter, &ySearchCenter, origin_x, origin_y, (int16_t)ref->width, (int16_t)ref->height);
        uint64_t hmeMvCost = (uint64_t)sad_sb_subsampled(src, cur->full_stride,
                                                         full_at(pool, ref, origin_x + xSearchCenter, origin_y + ySearchCenter),
                                                         ref->full_stride, sb_width, sb_height) << 8; /* + rate 0 */
        uint64_t c = zeroMvCost < hmeMvCost ? zeroMvCost : hmeMvCost;
        if (c == zeroMvCost) { xSearchCenter = 0; ySearchCenter = 0; }
    }
    int16_t xo = (int16_t)(xSearchCenter - (search_area_width >> 1));
    int16_t yo = (int16_t)(ySearchCenter - (search_area_height >> 1));
    clip_window(&xo, &yo, &search_area_width, &search_area_height, origin_x, origin_y, 63, 63, picture_width, picture_height);

    desc->src_offset = (int32_t)(cur->full_offset + (int64_t)(68 + sb_origin_y) * cur->full_stride + 68 + sb_origin_x);
    desc->ref_offset = (int32_t)(ref->full_offset + (int64_t)(68 + (int)sb_origin_y + yo) * ref->full_stride + 68 + (int)sb_origin_x + xo);
    desc->x_search_area_origin = xo;
    desc->y_search_area_origin = yo;
    desc->search_area_width = search_area_width;
    desc->search_area_height = search_area_height;
    if (center_xy) { center_xy[0] = xSearchCenter; center_xy[1] = ySearchCenter; }
}

void orc_hme_search_center_batch(const uint8_t *pool, const svthip_pa_picture *cur, const svthip_pa_picture *ref,
                                 const svthip_me_params *p, uint32_t list_index, const svthip_sb_origin *sb, uint32_t n_sb,
                                 const uint32_t *l0_best_mv64, svthip_fullpel_desc *desc, int16_t *center_xy,
                                 int16_t *hme_state)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        if (hme_state && list_index == 0) hme_state[25 * i + 24] = 0;
        orc_hme_search_center(pool, cur, ref, p, list_index, sb[i].x, sb[i].y, l0_best_mv64 ? l0_best_mv64[i] : 0, &desc[i],
                              center_xy ? center_xy + 2 * i : 0, hme_state ? hme_state + 25 * i : 0);
    }
}
