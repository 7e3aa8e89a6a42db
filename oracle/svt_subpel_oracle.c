/*
 * oracle/svt_subpel_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 *
 * Plain-C restatement of the sub-pel half of MotionEstimateLcu (Source/Lib/Codec/EbMotionEstimation.c):
 *   InterpolateSearchRegionAVC (:1707-1835)  -> the b / h / j half-pel planes as pure functions of the reference
 *   HalfPelSearch_LCU / PU_HalfPelRefinement (:2246-2786 / :1842-2240), SSD_SEARCH mode
 *   QuarterPelSearch_LCU / SetQuarterPelRefinementInputsOnTheFly / PU_QuarterPelRefinementOnTheFly /
 *   CombinedAveragingSSD (:3337-4114 / :3246-3331 / :2824-3239 / :2792-2817)
 *
 * PINNING STATUS.  The leaf arithmetic is pinned against the reference's own kernels in oracle/_ref
 * (AvcStyleLumaInterpolationFilter{Horizontal,Vertical}_SSSE3_INTRIN, SpatialFullDistortionKernel*_SSSE3_INTRIN,
 * CombinedAveragingSAD) and the plane geometry against the reference's InterpolateSearchRegionAVC
 * (tests/test_subpel_vs_ref.py).  The leaf DISPATCH is pinned per (width, height): orc_halfpel_ssd_dispatched /
 * orc_halfpel_sad_dispatched / orc_quarterpel_*_dispatched below restate which table entry PU_HalfPelRefinement /
 * PU_QuarterPelRefinementOnTheFly index for a PU size, and tests/test_subpel_vs_ref.py compares them with the reference's
 * own function-pointer tables (oracle/ref_subpel_leaf_driver.c) for every (w,h) pair HalfPelSearch_LCU passes.
 * The per-PU refinement CONTROL FLOW (PU_HalfPelRefinement, PU_QuarterPelRefinementOnTheFly and the two _LCU drivers with their index
 * maps) is pinned against the reference EXECUTING it: the distortion is a run-time choice (MeContext_t::fractionalSearchMethod), and with
 * SUB_SAD_SEARCH / FULL_SAD_SEARCH the reference's HalfPelSearch_LCU + QuarterPelSearch_LCU run in this image (Log2f_SSE2, NASM-only, is
 * evaluated only by the SSD_SEARCH branch of a conditional operator): oracle/ref_subpel_search_driver.c, tests/test_subpel_vs_ref.py
 * ::test_refinement_control_flow_matches_reference_execution, tests/golden/subpel_search.npz.  pu_half_pel / pu_quarter_pel below take the
 * method as an argument and differ between methods only in the distortion lines.  What stays unexecuted by the reference here is the
 * SSD_SEARCH distortion INSIDE that control flow (MotionEstimateLcu's setting): its leaves and dispatch are pinned as said above, the
 * bookkeeping of pBestSsd (:1911-1919, :1941-1946, :2927-2932) is restated from the text.
 */
#include "svt_me_oracle.h"

static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* 4-tap AVC-style half-pel filter {-2,18,18,-2}, +16 >> 5 (arithmetic), clip to u8
 * (ASM_SSSE3/EbAvcStyleMcp_Intrinsic_SSSE3.c:200-246 horizontal, :318-370 vertical). */
static inline int f4(int a, int b, int c, int d) { return clip8((-2 * a + 18 * b + 18 * c - 2 * d + 16) >> 5); }

typedef struct {
    const uint8_t *ref; /* reference sample at search position (0,0) */
    int stride;
} RefView;

static inline int A_(const RefView *r, int x, int y) { return r->ref[y * r->stride + x]; }
/* b[x,y]: half-pel between x-1 and x (pos_b consumed from row 2, :6923) */
static inline int B_(const RefView *r, int x, int y) { return f4(A_(r, x - 2, y), A_(r, x - 1, y), A_(r, x, y), A_(r, x + 1, y)); }
/* h[x,y]: half-pel between y-1 and y (pos_h consumed from column 1, :6925) */
static inline int H_(const RefView *r, int x, int y) { return f4(A_(r, x, y - 2), A_(r, x, y - 1), A_(r, x, y), A_(r, x, y + 1)); }
/* j[x,y]: vertical filter of the ROUNDED b plane (:1782-1791) */
static inline int J_(const RefView *r, int x, int y) { return f4(B_(r, x, y - 2), B_(r, x, y - 1), B_(r, x, y), B_(r, x, y + 1)); }

static inline int sample(const RefView *r, int plane, int x, int y)
{
    switch (plane) {
    case 0: return A_(r, x, y);
    case 1: return B_(r, x, y);
    case 2: return H_(r, x, y);
    default: return J_(r, x, y);
    }
}

void orc_interp_planes(const uint8_t *ref00, uint32_t ref_stride, int x0, int y0, int w, int h, uint8_t *b, uint8_t *hh,
                       uint8_t *j)
{
    RefView r = {ref00, (int)ref_stride};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            b[y * w + x] = (uint8_t)B_(&r, x0 + x, y0 + y);
            hh[y * w + x] = (uint8_t)H_(&r, x0 + x, y0 + y);
            j[y * w + x] = (uint8_t)J_(&r, x0 + x, y0 + y);
        }
}

/* SpatialFullDistortionKernel*_SSSE3_INTRIN: squared 8-bit WRAPPED difference, e = |int8(a - b)| with -128 -> 128
 * (ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c:497-623). */
static inline uint32_t wrap_sq(int a, int b)
{
    int d = (a - b) & 255;
    int e = d > 128 ? 256 - d : d;
    return (uint32_t)(e * e);
}

uint32_t orc_ssd_wrapped(const uint8_t *src, uint32_t src_stride, const uint8_t *rec, uint32_t rec_stride, uint32_t w, uint32_t h)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) s += wrap_sq(src[y * src_stride + x], rec[y * rec_stride + x]);
    return s;
}

/* Rows the half-pel stage's SSD covers for a w x h PU.  PU_HalfPelRefinement picks the SSD leaf by WIDTH only:
 * SpatialFullDistortionKernel_funcPtrArray[asm_type][Log2f(pu_width) - 2] (Codec/EbMotionEstimation.c:1912, :1929, :1964 ...;
 * table Codec/EbPictureOperators.h:532-559).  Width 8 selects SpatialFullDistortionKernel8x8_SSSE3_INTRIN, which always
 * runs 8 rows and ignores areaHeight (ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c:534-571), so the 8x16 and 8x32 PUs
 * of the 209-PU mode are compared on their top 8 rows only; widths 16/32/64 select the 16MxN kernel (all rows). */
static inline int halfpel_ssd_rows(int w, int h) { return w == 8 ? 8 : h; }

/* the half-pel stage's SSD exactly as PU_HalfPelRefinement dispatches it for a (w,h) PU */
uint32_t orc_halfpel_ssd_dispatched(const uint8_t *src, uint32_t src_stride, const uint8_t *rec, uint32_t rec_stride, uint32_t w, uint32_t h)
{
    return orc_ssd_wrapped(src, src_stride, rec, rec_stride, w, (uint32_t)halfpel_ssd_rows((int)w, (int)h));
}

/* the SAD stored on improvement: NxMSadKernel_funcPtrArray[asm_type][pu_width >> 3](.., pu_height, pu_width), all rows
 * (:1943; table Codec/EbComputeSAD.h:126-152: FastLoop_NxMSadKernel for every width in the ASM_NON_AVX2 row) */
uint32_t orc_halfpel_sad_dispatched(const uint8_t *src, uint32_t src_stride, const uint8_t *rec, uint32_t rec_stride, uint32_t w, uint32_t h)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int d = (int)src[y * src_stride + x] - (int)rec[y * rec_stride + x];
            s += (uint32_t)(d < 0 ? -d : d);
        }
    return s;
}

/* quarter-pel stage: CombinedAveragingSSD (true SSD, all rows, :2792-2817, call :2914) and the stored
 * NxMSadAveragingKernel_funcPtrArray[asm_type][pu_width >> 3] = CombinedAveragingSAD (all rows, :2929) */
uint32_t orc_quarterpel_ssd_dispatched(const uint8_t *src, uint32_t src_stride, const uint8_t *r1, uint32_t r1_stride, const uint8_t *r2,
                                       uint32_t r2_stride, uint32_t w, uint32_t h)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int e = (int)src[y * src_stride + x] - ((r1[y * r1_stride + x] + r2[y * r2_stride + x] + 1) >> 1);
            s += (uint32_t)(e * e);
        }
    return s;
}

uint32_t orc_quarterpel_sad_dispatched(const uint8_t *src, uint32_t src_stride, const uint8_t *r1, uint32_t r1_stride, const uint8_t *r2,
                                       uint32_t r2_stride, uint32_t w, uint32_t h)
{
    uint32_t s = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int e = (int)src[y * src_stride + x] - ((r1[y * r1_stride + x] + r2[y * r2_stride + x] + 1) >> 1);
            s += (uint32_t)(e < 0 ? -e : e);
        }
    return s;
}

/* 0 = every half-pel SSD over all rows (the round-1 restatement, kept ONLY so that tests can show the 8-wide classes change);
 * 1 = as the reference dispatches it (default) */
static int g_halfpel_dispatch_exact = 1;
void orc_set_halfpel_dispatch_exact(int on) { g_halfpel_dispatch_exact = on; }

#define DIR_TL 0
#define DIR_T 1
#define DIR_TR 2
#define DIR_R 3
#define DIR_BR 4
#define DIR_B 5
#define DIR_BL 6
#define DIR_L 7

/* candidate order of PU_HalfPelRefinement: L, R, T, B, TL, TR, BR, BL  -> (plane, dx, dy, mvdx, mvdy) */
static const int8_t kHalf[8][5] = {{1, 0, 0, -2, 0}, {1, 1, 0, 2, 0},  {2, 0, 0, 0, -2}, {2, 0, 1, 0, 2},
                                   {3, 0, 0, -2, -2}, {3, 1, 0, 2, -2}, {3, 1, 1, 2, 2},  {3, 0, 1, -2, 2}};

/* MeContext_t::fractionalSearchMethod (Codec/EbDefinitions.h:1846-1848): the distortion the refinement compares.  MotionEstimateLcu
 * hard-wires SSD_SEARCH (:6254); the two SAD methods run the SAME statements with other leaves and are what the reference itself
 * can execute in this image (oracle/ref_subpel_search_driver.c), which is how the control flow below is pinned. */
#define METHOD_SUB_SAD 0  /* 2 x SAD over every second row (strides << 1, height >> 1) */
#define METHOD_FULL_SAD 1 /* SAD over all rows */
#define METHOD_SSD 2      /* wrapped 8-bit SSD in the half-pel stage, true SSD in the quarter-pel stage; the stored SAD is the full SAD */

static void pu_half_pel(const uint8_t *src, int src_stride, const RefView *r, int px, int py, int w, int h, int xo, int yo, int method,
                        uint32_t *best_sad, uint32_t *best_mv, uint32_t *best_ssd, uint8_t *dir)
{
    const int16_t x_mv = (int16_t)(*best_mv & 0xffff), y_mv = (int16_t)(*best_mv >> 16);
    const int xs = (x_mv >> 2) - xo, ys = (y_mv >> 2) - yo;
    const uint8_t *s = src + py * src_stride + px;
    const int hs = g_halfpel_dispatch_exact ? halfpel_ssd_rows(w, h) : h; /* rows the width-keyed SSD leaf covers */
    if (method == METHOD_SSD) {
        uint32_t ssd = 0;
        for (int y = 0; y < hs; y++)
            for (int x = 0; x < w; x++) ssd += wrap_sq(s[y * src_stride + x], A_(r, xs + px + x, ys + py + y));
        *best_ssd = ssd; /* :1911-1919 */
    }
    uint64_t dist[8];
    for (int k = 0; k < 8; k++) {
        uint32_t d = 0, sad = 0, sad_even = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int p = sample(r, kHalf[k][0], xs + px + x + kHalf[k][1], ys + py + y + kHalf[k][2]);
                const int sv = s[y * src_stride + x];
                const uint32_t ad = (uint32_t)(sv > p ? sv - p : p - sv);
                if (y < hs) d += wrap_sq(sv, p);
                sad += ad;
                if (!(y & 1)) sad_even += ad;
            }
        if (method == METHOD_SSD) {
            dist[k] = d;
            if (d < *best_ssd) { /* strict '<', :1942 */
                *best_sad = sad;  /* true SAD over all rows, :1943 */
                *best_mv = ((uint32_t)(uint16_t)(y_mv + kHalf[k][4]) << 16) | (uint16_t)(x_mv + kHalf[k][3]);
                *best_ssd = d;
            }
        } else {
            /* :1930-1932: NxMSadKernel(src, stride << 1, ref, stride << 1, height >> 1, width) << 1, or all rows; compared with and stored
             * as the best SAD (:1948-1953) */
            const uint32_t dd = method == METHOD_SUB_SAD ? sad_even << 1 : sad;
            dist[k] = dd;
            if (dd < *best_sad) {
                *best_sad = dd;
                *best_mv = ((uint32_t)(uint16_t)(y_mv + kHalf[k][4]) << 16) | (uint16_t)(x_mv + kHalf[k][3]);
            }
        }
    }
    /* direction: first match in the order L, R, T, B, TL, TR, BL, BR (:2209-2238; note BL before BR) */
    uint64_t m = dist[0];
    for (int k = 1; k < 8; k++)
        if (dist[k] < m) m = dist[k];
    if (m == dist[0]) *dir = DIR_L;
    else if (m == dist[1]) *dir = DIR_R;
    else if (m == dist[2]) *dir = DIR_T;
    else if (m == dist[3]) *dir = DIR_B;
    else if (m == dist[4]) *dir = DIR_TL;
    else if (m == dist[5]) *dir = DIR_TR;
    else if (m == dist[7]) *dir = DIR_BL;
    else *dir = DIR_BR;
}

/* SetQuarterPelRefinementInputsOnTheFly (:3271-3323): [method][position L,R,T,B,TL,TR,BR,BL][buf1/buf2][plane,dx,dy],
 * plane 0 = integer (A), 1 = b, 2 = h, 3 = j */
static const int8_t kQuarter[4][8][2][3] = {
    /* EB_QUARTER_IN_FULL */
    {{{1, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {1, 1, 0}}, {{2, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {2, 0, 1}},
     {{1, 0, 0}, {2, 0, 0}}, {{2, 0, 0}, {1, 1, 0}}, {{2, 0, 1}, {1, 1, 0}}, {{1, 0, 0}, {2, 0, 1}}},
    /* EB_QUARTER_IN_HALF_HORIZONTAL */
    {{{0, -1, 0}, {1, 0, 0}}, {{1, 0, 0}, {0, 0, 0}}, {{3, 0, 0}, {1, 0, 0}}, {{1, 0, 0}, {3, 0, 1}},
     {{2, -1, 0}, {1, 0, 0}}, {{1, 0, 0}, {2, 0, 0}}, {{1, 0, 0}, {2, 0, 1}}, {{2, -1, 1}, {1, 0, 0}}},
    /* EB_QUARTER_IN_HALF_VERTICAL */
    {{{3, 0, 0}, {2, 0, 0}}, {{2, 0, 0}, {3, 1, 0}}, {{0, 0, -1}, {2, 0, 0}}, {{2, 0, 0}, {0, 0, 0}},
     {{1, 0, -1}, {2, 0, 0}}, {{2, 0, 0}, {1, 1, -1}}, {{2, 0, 0}, {1, 1, 0}}, {{1, 0, 0}, {2, 0, 0}}},
    /* EB_QUARTER_IN_HALF_DIAGONAL */
    {{{2, -1, 0}, {3, 0, 0}}, {{3, 0, 0}, {2, 0, 0}}, {{1, 0, -1}, {3, 0, 0}}, {{3, 0, 0}, {1, 0, 0}},
     {{2, -1, 0}, {1, 0, -1}}, {{1, 0, -1}, {2, 0, 0}}, {{1, 0, 0}, {2, 0, 0}}, {{2, -1, 0}, {1, 0, 0}}}};

/* accessor for tests/test_subpel_tables.py: the six numbers of one table entry (buf1 plane, dx, dy, buf2 plane, dx, dy) */
void orc_quarter_table_entry(int method, int pos, int8_t *out6)
{
    for (int b = 0; b < 2; b++)
        for (int k = 0; k < 3; k++) out6[3 * b + k] = kQuarter[method][pos][b][k];
}

static const int8_t kQmv[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};

static void pu_quarter_pel(const uint8_t *src, int src_stride, const RefView *r, int px, int py, int w, int h, int xo, int yo, int fsm,
                           uint32_t *best_sad, uint32_t *best_mv, uint32_t *best_ssd, uint8_t d)
{
    const int16_t x_mv = (int16_t)(*best_mv & 0xffff), y_mv = (int16_t)(*best_mv >> 16);
    const int xs = ((x_mv + 2) >> 2) - xo, ys = ((y_mv + 2) >> 2) - yo; /* :2847-2848 */
    const int method = (y_mv & 2) + ((x_mv & 2) >> 1);
    int valid[8]; /* L, R, T, B, TL, TR, BR, BL */
    if (method) { /* :2859-2869 */
        valid[4] = (d == DIR_R || d == DIR_BR || d == DIR_B);
        valid[2] = (d == DIR_BR || d == DIR_B || d == DIR_BL);
        valid[5] = (d == DIR_B || d == DIR_BL || d == DIR_L);
        valid[1] = (d == DIR_BL || d == DIR_L || d == DIR_TL);
        valid[6] = (d == DIR_L || d == DIR_TL || d == DIR_T);
        valid[3] = (d == DIR_TL || d == DIR_T || d == DIR_TR);
        valid[7] = (d == DIR_T || d == DIR_TR || d == DIR_R);
        valid[0] = (d == DIR_TR || d == DIR_R || d == DIR_BR);
    } else { /* :2873-2880 */
        valid[4] = (d == DIR_L || d == DIR_TL || d == DIR_T);
        valid[2] = (d == DIR_TL || d == DIR_T || d == DIR_TR);
        valid[5] = (d == DIR_T || d == DIR_TR || d == DIR_R);
        valid[1] = (d == DIR_TR || d == DIR_R || d == DIR_BR);
        valid[6] = (d == DIR_R || d == DIR_BR || d == DIR_B);
        valid[3] = (d == DIR_BR || d == DIR_B || d == DIR_BL);
        valid[7] = (d == DIR_B || d == DIR_BL || d == DIR_L);
        valid[0] = (d == DIR_BL || d == DIR_L || d == DIR_TL);
    }
    const uint8_t *s = src + py * src_stride + px;
    for (int k = 0; k < 8; k++) {
        if (!valid[k]) continue;
        const int8_t(*q)[3] = kQuarter[method][k];
        uint32_t ssd = 0, sad = 0, sad_even = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int p1 = sample(r, q[0][0], xs + px + x + q[0][1], ys + py + y + q[0][2]);
                const int p2 = sample(r, q[1][0], xs + px + x + q[1][1], ys + py + y + q[1][2]);
                const int avg = (p1 + p2 + 1) >> 1;
                const int sv = s[y * src_stride + x];
                const int e = sv - avg;
                ssd += (uint32_t)(e * e); /* CombinedAveragingSSD: true SSD (:2792-2817) */
                sad += (uint32_t)(e < 0 ? -e : e);
                if (!(y & 1)) sad_even += (uint32_t)(e < 0 ? -e : e);
            }
        if (fsm == METHOD_SSD) {
            if (ssd < *best_ssd) {
                *best_sad = sad;
                *best_mv = ((uint32_t)(uint16_t)(y_mv + kQmv[k][1]) << 16) | (uint16_t)(x_mv + kQmv[k][0]);
                *best_ssd = ssd;
            }
        } else { /* :2915-2917, :2934-2939: NxMSadAveragingKernel on every second row << 1, or on all rows */
            const uint32_t dd = fsm == METHOD_SUB_SAD ? sad_even << 1 : sad;
            if (dd < *best_sad) {
                *best_sad = dd;
                *best_mv = ((uint32_t)(uint16_t)(y_mv + kQmv[k][1]) << 16) | (uint16_t)(x_mv + kQmv[k][0]);
            }
        }
    }
}

static const uint8_t kTab16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* Codec/EbMotionEstimation.h:89-94 */
static const uint8_t kTab8[64] = {0,  1,  4,  5,  16, 17, 20, 21, 2,  3,  6,  7,  18, 19, 22, 23, 8,  9,  12, 13, 24, 25,
                                  28, 29, 10, 11, 14, 15, 26, 27, 30, 31, 32, 33, 36, 37, 48, 49, 52, 53, 34, 35, 38, 39,
                                  50, 51, 54, 55, 40, 41, 44, 45, 56, 57, 60, 61, 42, 43, 46, 47, 58, 59, 62, 63};

/* Sub-pel refinement of the 85 square PUs of one SB against one list (use_subpel_flag = 1, enc modes M0/M1:
 * half-pel on every PU size, quarter-pel enabled, fractional_search64x64 = 1; :6857-6964).
 *   src   : SB top-left in the padded source plane;  ref00: reference sample at search position (0,0)
 *   best_*: [85] ME-buffer order, in/out;  out_ssd/out_dir: optional [85] (final SSD, half-pel direction)
 *   disable_8x8: cu8x8_mode == CU_8x8_MODE_1 (8x8 PUs keep their full-pel result) */
static void subpel_refine_85pu_m(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                                 int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, int method, uint32_t *best_sad,
                                 uint32_t *best_mv, uint32_t *out_ssd, uint8_t *out_dir)
{
    RefView r = {ref00, (int)ref_stride};
    const int xo = x_search_area_origin, yo = y_search_area_origin, ss = (int)src_stride;
    uint32_t ssd[85];
    uint8_t dir[85];
    for (int i = 0; i < 85; i++) { ssd[i] = 0; dir[i] = 0; }
    /* HalfPelSearch_LCU order: 64x64, 32x32[0..3], 16x16[raster 0..15], 8x8[raster 0..63] */
    pu_half_pel(src, ss, &r, 0, 0, 64, 64, xo, yo, method, &best_sad[0], &best_mv[0], &ssd[0], &dir[0]);
    for (int p = 0; p < 4; p++)
        pu_half_pel(src, ss, &r, (p & 1) << 5, (p >> 1) << 5, 32, 32, xo, yo, method, &best_sad[1 + p], &best_mv[1 + p], &ssd[1 + p], &dir[1 + p]);
    for (int p = 0; p < 16; p++) {
        const int i = 5 + kTab16[p];
        pu_half_pel(src, ss, &r, (p & 3) << 4, (p >> 2) << 4, 16, 16, xo, yo, method, &best_sad[i], &best_mv[i], &ssd[i], &dir[i]);
    }
    if (!disable_8x8)
        for (int p = 0; p < 64; p++) {
            const int i = 21 + kTab8[p];
            pu_half_pel(src, ss, &r, (p & 7) << 3, (p >> 3) << 3, 8, 8, xo, yo, method, &best_sad[i], &best_mv[i], &ssd[i], &dir[i]);
        }
    /* QuarterPelSearch_LCU: the 64x64 PU is refined with a 32x32 block at the SB origin (:3395-3409) */
    pu_quarter_pel(src, ss, &r, 0, 0, 32, 32, xo, yo, method, &best_sad[0], &best_mv[0], &ssd[0], dir[0]);
    for (int p = 0; p < 4; p++)
        pu_quarter_pel(src, ss, &r, (p & 1) << 5, (p >> 1) << 5, 32, 32, xo, yo, method, &best_sad[1 + p], &best_mv[1 + p], &ssd[1 + p], dir[1 + p]);
    for (int p = 0; p < 16; p++) {
        const int i = 5 + kTab16[p];
        pu_quarter_pel(src, ss, &r, (p & 3) << 4, (p >> 2) << 4, 16, 16, xo, yo, method, &best_sad[i], &best_mv[i], &ssd[i], dir[i]);
    }
    if (!disable_8x8)
        for (int p = 0; p < 64; p++) {
            const int i = 21 + kTab8[p];
            pu_quarter_pel(src, ss, &r, (p & 7) << 3, (p >> 3) << 3, 8, 8, xo, yo, method, &best_sad[i], &best_mv[i], &ssd[i], dir[i]);
        }
    if (out_ssd)
        for (int i = 0; i < 85; i++) out_ssd[i] = ssd[i];
    if (out_dir)
        for (int i = 0; i < 85; i++) out_dir[i] = dir[i];
}

void orc_subpel_refine_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                            int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, uint32_t *best_sad,
                            uint32_t *best_mv, uint32_t *out_ssd, uint8_t *out_dir)
{
    subpel_refine_85pu_m(src, src_stride, ref00, ref_stride, x_search_area_origin, y_search_area_origin, disable_8x8, METHOD_SSD, best_sad,
                         best_mv, out_ssd, out_dir);
}

void orc_subpel_refine_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane, uint32_t ref_stride,
                             const int32_t *desc, uint32_t n_sb, int disable_8x8, uint32_t *best_sad, uint32_t *best_mv,
                             uint32_t *out_ssd, uint8_t *out_dir)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d = desc + 6 * i;
        orc_subpel_refine_85pu(src_plane + d[0], src_stride, ref_plane + d[1], ref_stride, (int16_t)d[2], (int16_t)d[3], disable_8x8,
                               best_sad + 85 * i, best_mv + 85 * i, out_ssd ? out_ssd + 85 * i : 0, out_dir ? out_dir + 85 * i : 0);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Geometry of the 209 PUs of the all-partition mode (pic_depth_mode <= PIC_ALL_C_DEPTH_MODE).
 *
 * me_results / BiPredictionCompensation index PUs in RASTER order within each shape class (partitionWidth /
 * partitionHeight / puSearchIndexMap, Codec/EbMotionEstimation.h:177-322); the ME buffers (p_sb_best_sad / _mv) hold
 * each class in the order ExtSadCalculation* fills it (z-order of the constituent squares), and HalfPelSearch_LCU /
 * QuarterPelSearch_LCU / the packing loop translate with tab16x16 .. tab8x32 (EbMotionEstimation.h:89-171).  The maps
 * are DERIVED here from the buffer order the full-pel stage produces (orc_fullpel_search_209pu above, itself pinned)
 * rather than restated as tables; tests/test_hme_vs_ref.py checks the result against the reference's own
 * MotionEstimateLcu run in 209-PU mode.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct { uint8_t w, h, px, py; uint8_t me; } PuGeom; /* indexed by raster PU index; me = ME-buffer index */

static int z4(int col, int row) { return ((row >> 1) * 2 + (col >> 1)) * 4 + (row & 1) * 2 + (col & 1); } /* 4x4 grid z-order */

static const PuGeom *pu_geom209(void)
{
    static PuGeom g[209];
    static int ready = 0;
    if (ready) return g;
    /* class: base, count, w, h, columns */
    static const int cls[14][5] = {{0, 1, 64, 64, 1},   {1, 4, 32, 32, 2},   {5, 16, 16, 16, 4},  {21, 64, 8, 8, 8},  {85, 2, 64, 32, 1},
                                   {87, 8, 32, 16, 2},  {95, 32, 16, 8, 4},  {127, 2, 32, 64, 2}, {129, 8, 16, 32, 4}, {137, 32, 8, 16, 8},
                                   {169, 16, 32, 8, 2}, {185, 16, 8, 32, 8}, {201, 4, 64, 16, 1}, {205, 4, 16, 64, 4}};
    for (int c = 0; c < 14; c++)
        for (int p = 0; p < cls[c][1]; p++) {
            const int base = cls[c][0], w = cls[c][2], h = cls[c][3], col = p % cls[c][4], row = p / cls[c][4];
            int i;
            switch (base) {
            case 5: i = z4(col, row); break;                                              /* 16x16: z-order */
            case 21: i = 4 * z4(col >> 1, row >> 1) + (row & 1) * 2 + (col & 1); break;   /* 8x8: raster inside its 16x16 */
            case 87: i = 2 * ((row >> 1) * 2 + col) + (row & 1); break;                   /* 32x16: (quadrant, upper/lower) */
            case 95: i = 2 * z4(col, row >> 1) + (row & 1); break;                        /* 16x8: (16x16 z, upper/lower) */
            case 137: i = 2 * z4(col >> 1, row) + (col & 1); break;                       /* 8x16: (16x16 z, left/right) */
            case 169: i = 4 * ((row >> 2) * 2 + col) + (row & 3); break;                  /* 32x8: (quadrant, row of 8) */
            default: i = p; break; /* 64x64, 32x32, 64x32, 32x64, 16x32, 8x32, 64x16, 16x64: buffer order = raster */
            }
            g[base + p] = (PuGeom){(uint8_t)w, (uint8_t)h, (uint8_t)(col * w), (uint8_t)(row * h), (uint8_t)(base + i)};
        }
    ready = 1;
    return g;
}

/* [209][5] = w, h, px, py, ME-buffer index, by raster PU index (for the tests) */
void orc_pu_geometry209(uint8_t *out)
{
    const PuGeom *g = pu_geom209();
    for (int i = 0; i < 209; i++) { out[5 * i] = g[i].w; out[5 * i + 1] = g[i].h; out[5 * i + 2] = g[i].px; out[5 * i + 3] = g[i].py; out[5 * i + 4] = g[i].me; }
}

/* Sub-pel refinement of all 209 PUs of one SB against one list: the 85 squares as above, then the rectangular PUs
 * (HalfPelSearch_LCU :2418-2786, QuarterPelSearch_LCU :3580-4114; every PU is refined independently, so the order of
 * the calls does not matter).  Arrays are [209] in ME-buffer order.  Same pinning status as the 85-PU function. */
static void subpel_refine_209pu_m(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                                  int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, int method, uint32_t *best_sad,
                                  uint32_t *best_mv, uint8_t *out_dir)
{
    subpel_refine_85pu_m(src, src_stride, ref00, ref_stride, x_search_area_origin, y_search_area_origin, disable_8x8, method, best_sad, best_mv,
                         0, out_dir);
    RefView r = {ref00, (int)ref_stride};
    const PuGeom *g = pu_geom209();
    for (int pu = 85; pu < 209; pu++) {
        const int n = g[pu].me;
        uint32_t ssd = 0;
        uint8_t dir = 0;
        pu_half_pel(src, (int)src_stride, &r, g[pu].px, g[pu].py, g[pu].w, g[pu].h, x_search_area_origin, y_search_area_origin, method,
                    &best_sad[n], &best_mv[n], &ssd, &dir);
        pu_quarter_pel(src, (int)src_stride, &r, g[pu].px, g[pu].py, g[pu].w, g[pu].h, x_search_area_origin, y_search_area_origin, method,
                       &best_sad[n], &best_mv[n], &ssd, dir);
        if (out_dir) out_dir[n] = dir;
    }
}

void orc_subpel_refine_209pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride,
                             int16_t x_search_area_origin, int16_t y_search_area_origin, int disable_8x8, uint32_t *best_sad,
                             uint32_t *best_mv)
{
    subpel_refine_209pu_m(src, src_stride, ref00, ref_stride, x_search_area_origin, y_search_area_origin, disable_8x8, METHOD_SSD, best_sad,
                          best_mv, 0);
}

/* The refinement of one SB and list under any of the reference's three fractional search methods (0 SUB_SAD_SEARCH, 1 FULL_SAD_SEARCH,
 * 2 SSD_SEARCH): 85 or 209 PUs in ME-buffer order, in/out; out_dir (optional): half-pel direction per PU, same order, reference codes
 * (Codec/EbMotionEstimation.h:62-69).  Methods 0 and 1 are checked against the reference EXECUTING the same search
 * (tests/test_subpel_vs_ref.py, oracle/ref_subpel_search_driver.c). */
void orc_subpel_refine_method(const uint8_t *src, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride, int16_t x_search_area_origin,
                              int16_t y_search_area_origin, int disable_8x8, int all_pu, int method, uint32_t *best_sad, uint32_t *best_mv,
                              uint8_t *out_dir)
{
    if (all_pu)
        subpel_refine_209pu_m(src, src_stride, ref00, ref_stride, x_search_area_origin, y_search_area_origin, disable_8x8, method, best_sad,
                              best_mv, out_dir);
    else
        subpel_refine_85pu_m(src, src_stride, ref00, ref_stride, x_search_area_origin, y_search_area_origin, disable_8x8, method, best_sad,
                             best_mv, 0, out_dir);
}

void orc_subpel_refine209_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref_plane, uint32_t ref_stride,
                                const int32_t *desc, uint32_t n_sb, int disable_8x8, uint32_t *best_sad, uint32_t *best_mv)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d = desc + 6 * i;
        orc_subpel_refine_209pu(src_plane + d[0], src_stride, ref_plane + d[1], ref_stride, (int16_t)d[2], (int16_t)d[3], disable_8x8,
                                best_sad + 209 * i, best_mv + 209 * i);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Bi-prediction SAD and result packing (Codec/EbMotionEstimation.c:6973-7146, BiPredictionSearch :5261-5342,
 * BiPredictionCompensation :5090-5254, BiPredAverging :4933-5081, SelectBuffer :4762-4815,
 * QuarterPelCompensation :4818-4920, Sort3Elements :5434-5463).
 *
 * Pinned against the reference's own MotionEstimateLcu for integer MVs (sub-pel disabled) in
 * tests/test_hme_vs_ref.py; the fractional-position buffer selection below is pinned against the reference's own
 * BiPredictionSearch for arbitrary quarter-pel vectors in tests/test_bipred_vs_ref.py (oracle/ref_bipred_driver.c).
 * ------------------------------------------------------------------------------------------------------------ */

/* prediction sample of one list at fractional position `frac` = (x_mv & 3) + ((y_mv & 3) << 2), integer position (x,y)
 * in search-region coordinates: F = A(x,y), Bq = b(x+1,y), Hq = h(x,y+1), Jq = j(x+1,y+1) are what BiPredictionCompensation's
 * buffer indices select (:5155-5158). */
static inline int bipred_sample(const RefView *r, int frac, int x, int y)
{
#define AV(a, b) (((a) + (b) + 1) >> 1)
    switch (frac) {
    case 0: return A_(r, x, y);
    case 2: return B_(r, x + 1, y);
    case 8: return H_(r, x, y + 1);
    case 10: return J_(r, x + 1, y + 1);
    case 1: return AV(A_(r, x, y), B_(r, x + 1, y));                 /* a */
    case 3: return AV(B_(r, x + 1, y), A_(r, x + 1, y));             /* c */
    case 4: return AV(A_(r, x, y), H_(r, x, y + 1));                 /* d */
    case 5: return AV(B_(r, x + 1, y), H_(r, x, y + 1));             /* e */
    case 6: return AV(B_(r, x + 1, y), J_(r, x + 1, y + 1));         /* f */
    case 7: return AV(B_(r, x + 1, y), H_(r, x + 1, y + 1));         /* g */
    case 9: return AV(H_(r, x, y + 1), J_(r, x + 1, y + 1));         /* i */
    case 11: return AV(J_(r, x + 1, y + 1), H_(r, x + 1, y + 1));    /* k */
    case 12: return AV(H_(r, x, y + 1), A_(r, x, y + 1));            /* L */
    case 13: return AV(H_(r, x, y + 1), B_(r, x + 1, y + 1));        /* m */
    case 14: return AV(J_(r, x + 1, y + 1), B_(r, x + 1, y + 1));    /* n */
    default: return AV(H_(r, x + 1, y + 1), B_(r, x + 1, y + 1));    /* 15: o */
    }
#undef AV
}

static uint32_t bipred_sad(const uint8_t *src, int ss, const RefView *r0, int xo0, int yo0, uint32_t mv0, const RefView *r1,
                           int xo1, int yo1, uint32_t mv1, int px, int py, int w, int h)
{
    const int16_t x0 = (int16_t)(mv0 & 0xffff), y0 = (int16_t)(mv0 >> 16), x1 = (int16_t)(mv1 & 0xffff), y1 = (int16_t)(mv1 >> 16);
    const int f0 = ((uint8_t)x0 & 3) + (((uint8_t)y0 & 3) << 2), f1 = ((uint8_t)x1 & 3) + (((uint8_t)y1 & 3) << 2);
    const int ix0 = (x0 >> 2) - xo0 + px, iy0 = (y0 >> 2) - yo0 + py, ix1 = (x1 >> 2) - xo1 + px, iy1 = (y1 >> 2) - yo1 + py;
    uint32_t sad = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int p0 = bipred_sample(r0, f0, ix0 + x, iy0 + y), p1 = bipred_sample(r1, f1, ix1 + x, iy1 + y);
            const int avg = (p0 + p1 + 1) >> 1; /* NxMSadAveragingKernel / CombinedAveragingSAD, all rows (SSD_SEARCH mode) */
            const int sv = src[(py + y) * ss + px + x];
            sad += (uint32_t)(sv > avg ? sv - avg : avg - sv);
        }
    return sad;
}

/* n_lists = 1 (P) or 2 (B).  n_pu = 85 or 209.  sad/mv arrays are ME-buffer ordered [n_pu]; out = n_pu svthip_me_cu_result
 * in raster PU order (ME-buffer index per raster PU: :6980-7015).
 * bipred_8x8 = (cu8x8_mode == CU_8x8_MODE_0): 8x8 PUs get a bi-pred candidate too; in the 209-PU mode every PU gets one
 * whatever cu8x8_mode is (:7028). */
static void bipred_pack_sb(const uint8_t *src, uint32_t src_stride, const uint8_t *ref0_00, uint32_t ref0_stride, int16_t xo0, int16_t yo0,
                           const uint8_t *ref1_00, uint32_t ref1_stride, int16_t xo1, int16_t yo1, const uint32_t *sad0,
                           const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1, int n_lists, int bipred_8x8, int n_pu,
                           svthip_me_cu_result *out)
{
    RefView r0 = {ref0_00, (int)ref0_stride}, r1 = {ref1_00, (int)ref1_stride};
    const PuGeom *g = pu_geom209();
    for (int pu = 0; pu < n_pu; pu++) {
        const int n = g[pu].me;
        svthip_me_cu_result *o = &out[pu];
        int total = n_lists;
        uint32_t bi = 0;
        if (n_lists == 2 && (bipred_8x8 || pu < 21 || n_pu == 209)) {
            bi = bipred_sad(src, (int)src_stride, &r0, xo0, yo0, mv0[n], &r1, xo1, yo1, mv1[n], g[pu].px, g[pu].py, g[pu].w, g[pu].h);
            total = 3;
        }
        o->xMvL0 = (int16_t)(mv0[n] & 0xffff);
        o->yMvL0 = (int16_t)(mv0[n] >> 16);
        o->xMvL1 = n_lists == 2 ? (int16_t)(mv1[n] & 0xffff) : 0; /* P pictures: list-1 fields are don't-care in the reference */
        o->yMvL1 = n_lists == 2 ? (int16_t)(mv1[n] >> 16) : 0;
        o->totalMeCandidateIndex = (uint8_t)total;
        for (int k = 0; k < 3; k++) { o->distortion[k] = 0; o->direction[k] = 0; }
        const uint32_t a = sad0[n], b = n_lists == 2 ? sad1[n] : 0, c = bi;
        if (total == 3) {
            /* Sort3Elements (:5434-5463): '<=' everywhere, ties favour L0, then L1 */
            int order[3];
            if (a <= b && a <= c) { order[0] = 0; if (b <= c) { order[1] = 1; order[2] = 2; } else { order[1] = 2; order[2] = 1; } }
            else if (b <= a && b <= c) { order[0] = 1; if (a <= c) { order[1] = 0; order[2] = 2; } else { order[1] = 2; order[2] = 0; } }
            else if (a <= b) { order[0] = 2; order[1] = 0; order[2] = 1; }
            else { order[0] = 2; order[1] = 1; order[2] = 0; }
            const uint32_t v[3] = {a, b, c};
            for (int k = 0; k < 3; k++) { o->distortion[k] = v[order[k]]; o->direction[k] = (uint8_t)order[k]; }
        } else if (total == 2) {
            if (a <= b) { o->distortion[0] = a; o->direction[0] = 0; o->distortion[1] = b; o->direction[1] = 1; }
            else { o->distortion[0] = b; o->direction[0] = 1; o->distortion[1] = a; o->direction[1] = 0; }
        } else {
            o->distortion[0] = a;
            o->direction[0] = 0;
        }
    }
}

void orc_bipred_pack_85pu(const uint8_t *src, uint32_t src_stride, const uint8_t *ref0_00, uint32_t ref0_stride, int16_t xo0,
                          int16_t yo0, const uint8_t *ref1_00, uint32_t ref1_stride, int16_t xo1, int16_t yo1,
                          const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1, int n_lists,
                          int bipred_8x8, svthip_me_cu_result *out)
{
    bipred_pack_sb(src, src_stride, ref0_00, ref0_stride, xo0, yo0, ref1_00, ref1_stride, xo1, yo1, sad0, mv0, sad1, mv1, n_lists, bipred_8x8, 85, out);
}

/* the same over [n_sb][n_pu] arrays, n_pu = 85 or 209 */
void orc_bipred_pack_batch_npu(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref0_plane, uint32_t ref0_stride,
                               const int32_t *desc0, const uint8_t *ref1_plane, uint32_t ref1_stride, const int32_t *desc1,
                               uint32_t n_sb, const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1,
                               int n_lists, int bipred_8x8, int n_pu, svthip_me_cu_result *out)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d0 = desc0 + 6 * i, *d1 = n_lists == 2 ? desc1 + 6 * i : d0;
        bipred_pack_sb(src_plane + d0[0], src_stride, ref0_plane + d0[1], ref0_stride, (int16_t)d0[2], (int16_t)d0[3],
                       (n_lists == 2 ? ref1_plane : ref0_plane) + d1[1], n_lists == 2 ? ref1_stride : ref0_stride, (int16_t)d1[2],
                       (int16_t)d1[3], sad0 + (size_t)n_pu * i, mv0 + (size_t)n_pu * i, n_lists == 2 ? sad1 + (size_t)n_pu * i : sad0 + (size_t)n_pu * i,
                       n_lists == 2 ? mv1 + (size_t)n_pu * i : mv0 + (size_t)n_pu * i, n_lists, bipred_8x8, n_pu, out + (size_t)n_pu * i);
    }
}

void orc_bipred_pack_batch(const uint8_t *src_plane, uint32_t src_stride, const uint8_t *ref0_plane, uint32_t ref0_stride,
                           const int32_t *desc0, const uint8_t *ref1_plane, uint32_t ref1_stride, const int32_t *desc1,
                           uint32_t n_sb, const uint32_t *sad0, const uint32_t *mv0, const uint32_t *sad1, const uint32_t *mv1,
                           int n_lists, int bipred_8x8, svthip_me_cu_result *out)
{
    for (uint32_t i = 0; i < n_sb; i++) {
        const int32_t *d0 = desc0 + 6 * i, *d1 = n_lists == 2 ? desc1 + 6 * i : d0;
        orc_bipred_pack_85pu(src_plane + d0[0], src_stride, ref0_plane + d0[1], ref0_stride, (int16_t)d0[2], (int16_t)d0[3],
                             (n_lists == 2 ? ref1_plane : ref0_plane) + d1[1], n_lists == 2 ? ref1_stride : ref0_stride,
                             (int16_t)d1[2], (int16_t)d1[3], sad0 + 85 * i, mv0 + 85 * i, n_lists == 2 ? sad1 + 85 * i : sad0 + 85 * i,
                             n_lists == 2 ? mv1 + 85 * i : mv0 + 85 * i, n_lists, bipred_8x8, out + 85 * i);
    }
}
