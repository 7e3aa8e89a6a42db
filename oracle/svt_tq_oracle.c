/*
 * oracle/svt_tq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 * Plain-C restatement of the reference's quantisation (Source/Lib/Codec/EbFullLoop.c) and, below, its forward
 * transforms (Source/Lib/Codec/EbTransforms.c).  Pinned against the reference's own C functions in
 * oracle/_ref/libsvtref_tq.so by tests/test_tq_vs_ref.py.
 */
#include "svt_me_oracle.h"

#include <string.h>

static inline int32_t rpot(int32_t v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); } /* ROUND_POWER_OF_TWO */

/* quantize_b_helper_c_II (:46-108, 8-bit: |coeff|+round is clamped to int16) and highbd_quantize_b_helper_c (:242-299, no
 * clamp) with the flat quantisation matrix the encoder uses (qm_ptr == NULL -> wt = iwt = 1 << AOM_QM_BITS).
 * The pre-scan passes of both only skip coefficients inside the dead zone, which the per-coefficient test repeats, so the
 * result is a pure per-coefficient function plus eob = 1 + last scan position with a non-zero level.
 * qp = {zbin[2], round[2], quant[2], quant_shift[2], dequant[2]} (index 0 = DC, 1 = AC). */
void orc_quantize_b(const int32_t *coeff, int32_t n_coeffs, const int16_t *qp, const int16_t *scan, int log_scale, int highbd,
                    int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob_ptr)
{
    const int16_t *zbin = qp, *round = qp + 2, *quant = qp + 4, *quant_shift = qp + 6, *dequant = qp + 8;
    const int32_t zbins[2] = {rpot(zbin[0], log_scale), rpot(zbin[1], log_scale)};
    int32_t eob = -1;
    memset(qcoeff, 0, sizeof(int32_t) * (size_t)n_coeffs);
    memset(dqcoeff, 0, sizeof(int32_t) * (size_t)n_coeffs);
    for (int32_t i = 0; i < n_coeffs; i++) {
        const int32_t rc = scan[i];
        const int ac = rc != 0;
        const int32_t c = coeff[rc];
        const int32_t sign = c >> 31;
        const int32_t abs_c = (c ^ sign) - sign;
        if (abs_c < zbins[ac]) continue;
        int64_t tmp = (int64_t)abs_c + rpot(round[ac], log_scale);
        if (!highbd) tmp = tmp < -32768 ? -32768 : (tmp > 32767 ? 32767 : tmp); /* clamp(.., INT16_MIN, INT16_MAX), :88-90 */
        tmp *= 32;                                                              /* wt = 1 << AOM_QM_BITS */
        const int32_t level = (int32_t)(((((tmp * quant[ac]) >> 16) + tmp) * quant_shift[ac]) >> (16 - log_scale + 5));
        qcoeff[rc] = (level ^ sign) - sign;
        const int32_t dq = (dequant[ac] * 32 + 16) >> 5;
        const int32_t abs_dq = (int32_t)((uint32_t)level * (uint32_t)dq) >> log_scale; /* int32 product as in the reference */
        dqcoeff[rc] = (abs_dq ^ sign) - sign;
        if (level) eob = i;
    }
    *eob_ptr = (uint16_t)(eob + 1);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Forward 2-D transforms (row a16).  Restates Av1TranformTwoDCore_c (EbTransforms.c:3701-3780) with the configuration of
 * Av1TransformConfig (:3847-3867) and the tables of EbTransforms.h:88-160.  The 1-D networks are written from their
 * structure (recursive even/odd split for the DCT, layered pair rotations for the ADST), not stage by stage; every
 * rotation rounds exactly where half_btf (:1292-1299) rounds, so the results are bit-identical to
 * av1_fdct{4..64}_new (:1314-2762), av1_fadst{4,8,16}_new (:2764-3183) and av1_fidentity*_c.
 * ------------------------------------------------------------------------------------------------------------------ */
#include <math.h>

#define ORC_SQRT2 5793 /* NewSqrt2, NewSqrt2Bits = 12 */

static int32_t g_cospi[7][64]; /* [bit-10][j] = round(cos(pi j/128) 2^bit), :1242-1287; checked against the reference table */
static int g_cospi_ready;
/* av1_sinpi_arr_data (:1303-1311): round(sqrt(2) sin(j pi/9) 2/3 2^bit), adjusted so that [1]+[2] == [4] */
static const int32_t g_sinpi[7][5] = {{0, 330, 621, 836, 951},        {0, 660, 1241, 1672, 1901},     {0, 1321, 2482, 3344, 3803},
                                      {0, 2642, 4964, 6689, 7606},    {0, 5283, 9929, 13377, 15212},  {0, 10566, 19858, 26755, 30424},
                                      {0, 21133, 39716, 53510, 60849}};

static void cospi_init(void)
{
    if (g_cospi_ready) return;
    for (int b = 0; b < 7; b++)
        for (int j = 0; j < 64; j++) g_cospi[b][j] = (int32_t)floor(cos(M_PI * j / 128.0) * (double)(1 << (b + 10)) + 0.5);
    g_cospi_ready = 1;
}
const int32_t *orc_cospi_table(int bit) { cospi_init(); return g_cospi[bit - 10]; }
const int32_t *orc_sinpi_table(int bit) { return g_sinpi[bit - 10]; }

static inline int32_t rs64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); } /* int32 product, wraps */
static inline int32_t hb(int32_t w0, int32_t in0, int32_t w1, int32_t in1, int bit)
{
    return rs64((int64_t)wmul(w0, in0) + (int64_t)wmul(w1, in1), bit);
}
static int brev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* mirrored add/sub layer over blocks of `span`: even blocks (lo+hi, lo-hi), odd blocks (hi-lo, hi+lo) */
static void odd_bfly(int32_t *a, int M, int span)
{
    for (int base = 0, blk = 0; base < M; base += span, blk++)
        for (int t = 0; t < span / 2; t++) {
            const int i = base + t, j = base + span - 1 - t;
            const int32_t lo = a[i], hi = a[j];
            if (!(blk & 1)) { a[i] = lo + hi; a[j] = lo - hi; }
            else            { a[i] = hi - lo; a[j] = hi + lo; }
        }
}
/* rotation layer j of the odd part: pairs (i, M-1-i) */
static void odd_rot(int32_t *a, int M, int j, const int32_t *c, int bit)
{
    if (j == 1) {
        for (int i = M / 4; i < M / 2; i++) {
            const int k = M - 1 - i;
            const int32_t x = a[i], y = a[k];
            a[i] = hb(-c[32], x, c[32], y, bit);
            a[k] = hb(c[32], y, c[32], x, bit);
        }
        return;
    }
    const int nb = 1 << (j - 2), L = (M / 2) / nb;
    for (int b = 0; b < nb; b++) {
        const int al = (16 / nb) * (1 + 4 * brev(b, j - 2));
        for (int t = L / 4; t < 3 * L / 4; t++) {
            const int i = b * L + t, k = M - 1 - i;
            const int32_t x = a[i], y = a[k];
            if (t < L / 2) { a[i] = hb(-c[al], x, c[64 - al], y, bit);      a[k] = hb(c[al], y, c[64 - al], x, bit); }
            else           { a[i] = hb(-c[64 - al], x, -c[al], y, bit);     a[k] = hb(c[64 - al], y, -c[al], x, bit); }
        }
    }
}
/* odd half of an N = 2M point DCT: a[t] = x[M-1-t] - x[M+t]; result y[k] is coefficient 1 + 2 brev(k) */
static void dct_odd(int32_t *a, int M, const int32_t *c, int bit)
{
    const int m = ilog2(M);
    for (int j = 1; j < m; j++) { odd_rot(a, M, j, c, bit); odd_bfly(a, M, M >> j); }
    for (int k = 0; k < M / 2; k++) {
        const int al = (32 / M) * (1 + 4 * brev(k, m - 1)), q = M - 1 - k;
        const int32_t x = a[k], y = a[q];
        a[k] = hb(c[64 - al], x, c[al], y, bit);
        a[q] = hb(c[64 - al], y, -c[al], x, bit);
    }
}
static void fdct_rec(const int32_t *x, int32_t *out, int n, int ostride, const int32_t *c, int bit)
{
    if (n == 2) {
        out[0] = hb(c[32], x[0], c[32], x[1], bit);
        out[ostride] = hb(-c[32], x[1], c[32], x[0], bit);
        return;
    }
    int32_t s[32] = {0}, a[32];
    const int M = n / 2, m = ilog2(M);
    for (int i = 0; i < M; i++) { s[i] = x[i] + x[n - 1 - i]; a[i] = x[M - 1 - i] - x[M + i]; }
    fdct_rec(s, out, M, 2 * ostride, c, bit);
    dct_odd(a, M, c, bit);
    for (int k = 0; k < M; k++) out[(1 + 2 * brev(k, m)) * ostride] = a[k];
}

static void rotP(int32_t *p, int al, const int32_t *c, int bit)
{
    const int32_t x = p[0], y = p[1];
    p[0] = hb(c[al], x, c[64 - al], y, bit);
    p[1] = hb(c[64 - al], x, -c[al], y, bit);
}
static void rotQ(int32_t *p, int al, const int32_t *c, int bit)
{
    const int32_t x = p[0], y = p[1];
    p[0] = hb(-c[64 - al], x, c[al], y, bit);
    p[1] = hb(c[al], x, c[64 - al], y, bit);
}
static void span_bfly(int32_t *f, int n, int span)
{
    for (int base = 0; base < n; base += 2 * span)
        for (int t = 0; t < span; t++) {
            const int32_t x = f[base + t], y = f[base + span + t];
            f[base + t] = x + y;
            f[base + span + t] = x - y;
        }
}
static void fadst4(const int32_t *x, int32_t *out, int bit)
{
    const int32_t *s = g_sinpi[bit - 10];
    if (!(x[0] | x[1] | x[2] | x[3])) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    const int32_t p = wmul(s[1], x[0]) + wmul(s[2], x[1]) + wmul(s[4], x[3]);
    const int32_t q = wmul(s[4], x[0]) - wmul(s[1], x[1]) + wmul(s[2], x[3]);
    const int32_t r = wmul(s[3], x[2]);
    out[0] = rs64((int32_t)((uint32_t)p + (uint32_t)r), bit);
    out[1] = rs64(wmul(s[3], x[0] + x[1] - x[3]), bit);
    out[2] = rs64((int32_t)((uint32_t)q - (uint32_t)r), bit);
    out[3] = rs64((int32_t)((uint32_t)q - (uint32_t)p + (uint32_t)r), bit);
}
static void fadst_n(const int32_t *x, int32_t *out, int n, const int32_t *c, int bit)
{
    static const int8_t idx8[8] = {0, 7, 3, 4, 1, 6, 2, 5}, neg8[8] = {0, 1, 1, 0, 1, 0, 0, 1};
    static const int8_t idx16[16] = {0, 15, 7, 8, 3, 12, 4, 11, 1, 14, 6, 9, 2, 13, 5, 10};
    static const int8_t neg16[16] = {0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0};
    int32_t f[16];
    for (int i = 0; i < n; i++) {
        const int32_t v = x[n == 8 ? idx8[i] : idx16[i]];
        f[i] = (n == 8 ? neg8[i] : neg16[i]) ? -v : v;
    }
    for (int g = 0; g < n; g += 4) rotP(f + g + 2, 32, c, bit);
    span_bfly(f, n, 2);
    for (int g = 0; g < n; g += 8) { rotP(f + g + 4, 16, c, bit); rotQ(f + g + 6, 16, c, bit); }
    span_bfly(f, n, 4);
    if (n == 16) {
        rotP(f + 8, 8, c, bit); rotP(f + 10, 40, c, bit); rotQ(f + 12, 8, c, bit); rotQ(f + 14, 40, c, bit);
        span_bfly(f, n, 8);
    }
    for (int k = 0; k < n / 2; k++) rotP(f + 2 * k, n == 8 ? 4 + 16 * k : 2 + 8 * k, c, bit);
    for (int i = 0; i < n / 2; i++) { out[2 * i] = f[2 * i + 1]; out[2 * i + 1] = f[n - 2 - 2 * i]; }
}
static void fidentity(const int32_t *x, int32_t *out, int n)
{
    for (int i = 0; i < n; i++) switch (n) {
        case 4:  out[i] = rs64((int64_t)x[i] * ORC_SQRT2, 12); break;
        case 8:  out[i] = x[i] * 2; break;
        case 16: out[i] = rs64((int64_t)x[i] * 2 * ORC_SQRT2, 12); break;
        case 32: out[i] = x[i] * 4; break;
        default: out[i] = rs64((int64_t)x[i] * 4 * ORC_SQRT2, 12); break;
    }
}
/* 1-D kinds as TX_TYPE_1D: 0 DCT, 1 ADST, 2 FLIPADST (same network, flip applied by the caller), 3 IDTX */
static void txfm1d(int kind, const int32_t *x, int32_t *out, int n, int bit)
{
    const int32_t *c = orc_cospi_table(bit);
    if (kind == 0) fdct_rec(x, out, n, 1, c, bit);
    else if (kind == 3) fidentity(x, out, n);
    else if (n == 4) fadst4(x, out, bit);
    else fadst_n(x, out, n, c, bit);
}
static void shift_array(int32_t *v, int n, int bit) /* av1_round_shift_array_c (:3683-3698) */
{
    if (bit == 0) return;
    for (int i = 0; i < n; i++) v[i] = bit > 0 ? rs64(v[i], bit) : v[i] * (1 << -bit);
}

static const int8_t k_vtx[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3}; /* vtx_tab, EbTransforms.h:88 */
static const int8_t k_htx[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2}; /* htx_tab, :93 */
static const int8_t k_cos_col[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t k_cos_row[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
/* fwd_shift_WxH (:121-139), indexed [log2 w - 2][log2 h - 2] */
static const int8_t k_shift[5][5][3] = {
    {{2, 0, 0}, {2, -1, 0}, {2, -1, 0}, {0, 0, 0}, {0, 0, 0}},
    {{2, -1, 0}, {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, 0, 0}},
    {{2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, 0}},
    {{0, 0, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0}, {0, -2, -2}},
    {{0, 0, 0}, {0, 0, 0}, {2, -4, 0}, {2, -4, -2}, {0, -2, -2}}};

/* returns 0 when (w, h, tx_type) is a combination the reference can run, -1 otherwise (no 1-D network for it) */
int orc_fwd_txfm2d_valid(int w, int h, int tx_type)
{
    if (tx_type < 0 || tx_type > 15) return -1;
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    if (wi < 0 || wi > 4 || hi < 0 || hi > 4 || (1 << (wi + 2)) != w || (1 << (hi + 2)) != h) return -1;
    if (wi - hi > 2 || hi - wi > 2) return -1;
    const int kc = k_vtx[tx_type], kr = k_htx[tx_type];
    if ((kc == 1 || kc == 2) && h > 16) return -1;
    if ((kr == 1 || kr == 2) && w > 16) return -1;
    if (kc == 3 && h > 32) return -1;
    if (kr == 3 && w > 32) return -1;
    return 0;
}

void orc_fwd_txfm2d(const int16_t *input, int32_t stride, int w, int h, int tx_type, int32_t *output)
{
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    const int kc = k_vtx[tx_type], kr = k_htx[tx_type];
    const int ud = (kc == 2), lr = (kr == 2); /* get_flip_cfg (:3782-3820) */
    const int8_t *sh = k_shift[wi][hi];
    const int cos_col = k_cos_col[wi][hi], cos_row = k_cos_row[wi][hi];
    const int rect = wi > hi ? wi - hi : hi - wi;
    int32_t col[64], tmp[64];
    int32_t *buf = (int32_t *)__builtin_alloca(sizeof(int32_t) * (size_t)(w * h));
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) col[r] = input[(ud ? h - 1 - r : r) * stride + c];
        shift_array(col, h, -sh[0]);
        txfm1d(kc, col, tmp, h, cos_col);
        shift_array(tmp, h, -sh[1]);
        for (int r = 0; r < h; r++) buf[r * w + (lr ? w - 1 - c : c)] = tmp[r];
    }
    for (int r = 0; r < h; r++) {
        int32_t *o = output + r * w;
        txfm1d(kr, buf + r * w, o, w, cos_row);
        shift_array(o, w, -sh[2]);
        if (rect == 1)
            for (int c = 0; c < w; c++) o[c] = rs64((int64_t)o[c] * ORC_SQRT2, 12);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Inverse 2-D transforms + reconstruction (row a17).  Restates inv_txfm2d_add_c (EbTransforms.c:7617-7700) behind
 * av1_inv_txfm2d_add_{WxH}_c (:7714-7900) as configured by av1_get_inv_txfm_cfg (:7590-7616): rows first (input clamped to
 * bd+8 bits, 2:1 rectangles pre-scaled by 1/sqrt(2)), round shift, columns (input clamped to max(bd+6,16) bits), round
 * shift by 4, flips, add to the prediction and clip.  Every add/sub stage of the inverse DCT/ADST networks clamps to the
 * pass's stage range (av1_gen_inv_stage_range, :4841-4893: bd+8 for rows, 16 for columns at 8/10 bit); the rotations are
 * the transposes of the forward ones.  64-point dimensions read a 32-wide/32-high packed input, rest zero (:7736-7760).
 * ------------------------------------------------------------------------------------------------------------------ */
#define ORC_INV_SQRT2 2896 /* NewInvSqrt2 */

static inline int32_t clampv(int32_t v, int bit) /* clamp_value (:4895-4900) */
{
    if (bit <= 0) return v;
    const int64_t mx = ((int64_t)1 << (bit - 1)) - 1, mn = -((int64_t)1 << (bit - 1));
    return (int32_t)(v < mn ? mn : (v > mx ? mx : v));
}
static inline int32_t addc(int32_t a, int32_t b, int R) { return clampv((int32_t)((uint32_t)a + (uint32_t)b), R); }
static inline int32_t subc(int32_t a, int32_t b, int R) { return clampv((int32_t)((uint32_t)a - (uint32_t)b), R); }

static void odd_bfly_c(int32_t *a, int M, int span, int R)
{
    for (int base = 0, blk = 0; base < M; base += span, blk++)
        for (int t = 0; t < span / 2; t++) {
            const int i = base + t, j = base + span - 1 - t;
            const int32_t lo = a[i], hi = a[j];
            if (!(blk & 1)) { a[i] = addc(lo, hi, R); a[j] = subc(lo, hi, R); }
            else            { a[i] = subc(hi, lo, R); a[j] = addc(hi, lo, R); }
        }
}
static void idct_rec(const int32_t *x, int xs, int32_t *out, int n, const int32_t *c, int bit, int R)
{
    if (n == 2) {
        out[0] = hb(c[32], x[0], c[32], x[xs], bit);
        out[1] = hb(c[32], x[0], -c[32], x[xs], bit);
        return;
    }
    int32_t e[32] = {0}, d[32];
    const int M = n / 2, m = ilog2(M);
    idct_rec(x, 2 * xs, e, M, c, bit, R);
    for (int k = 0; k < M; k++) d[k] = x[(1 + 2 * brev(k, m)) * xs];
    for (int k = 0; k < M / 2; k++) { /* transposed final rotations of the forward odd part */
        const int al = (32 / M) * (1 + 4 * brev(k, m - 1)), q = M - 1 - k;
        const int32_t u = d[k], v = d[q];
        d[k] = hb(c[64 - al], u, -c[al], v, bit);
        d[q] = hb(c[al], u, c[64 - al], v, bit);
    }
    for (int j = m - 1; j >= 1; j--) { odd_bfly_c(d, M, M >> j, R); odd_rot(d, M, j, c, bit); }
    for (int i = 0; i < M; i++) { out[i] = addc(e[i], d[M - 1 - i], R); out[n - 1 - i] = subc(e[i], d[M - 1 - i], R); }
}
static void span_bfly_c(int32_t *f, int n, int span, int R)
{
    for (int base = 0; base < n; base += 2 * span)
        for (int t = 0; t < span; t++) {
            const int32_t x = f[base + t], y = f[base + span + t];
            f[base + t] = addc(x, y, R);
            f[base + span + t] = subc(x, y, R);
        }
}
static void iadst4(const int32_t *x, int32_t *out, int bit)
{
    const int32_t *s = g_sinpi[bit - 10];
    if (!(x[0] | x[1] | x[2] | x[3])) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    const uint32_t A = (uint32_t)wmul(s[1], x[0]) + (uint32_t)wmul(s[4], x[2]) + (uint32_t)wmul(s[2], x[3]);
    const uint32_t B = (uint32_t)wmul(s[2], x[0]) - (uint32_t)wmul(s[1], x[2]) - (uint32_t)wmul(s[4], x[3]);
    const uint32_t Cc = (uint32_t)wmul(s[3], x[1]);
    const uint32_t x023 = (uint32_t)x[0] - (uint32_t)x[2] + (uint32_t)x[3];
    out[0] = rs64((int32_t)(A + Cc), bit);
    out[1] = rs64((int32_t)(B + Cc), bit);
    out[2] = rs64(wmul(s[3], (int32_t)x023), bit);
    out[3] = rs64((int32_t)(A + B - Cc), bit);
}
static void iadst_n(const int32_t *x, int32_t *out, int n, const int32_t *c, int bit, int R)
{
    static const int8_t o8[8] = {0, 4, 6, 2, 3, 7, 5, 1};
    static const int8_t o16[16] = {0, 8, 12, 4, 6, 14, 10, 2, 3, 11, 15, 7, 5, 13, 9, 1};
    int32_t f[16];
    for (int i = 0; i < n / 2; i++) { f[2 * i] = x[n - 1 - 2 * i]; f[2 * i + 1] = x[2 * i]; }
    for (int k = 0; k < n / 2; k++) rotP(f + 2 * k, n == 8 ? 4 + 16 * k : 2 + 8 * k, c, bit);
    span_bfly_c(f, n, n / 2, R);
    if (n == 16) {
        rotP(f + 8, 8, c, bit); rotP(f + 10, 40, c, bit); rotQ(f + 12, 8, c, bit); rotQ(f + 14, 40, c, bit);
        span_bfly_c(f, n, 4, R);
    }
    for (int g = 0; g < n; g += 8) { rotP(f + g + 4, 16, c, bit); rotQ(f + g + 6, 16, c, bit); }
    span_bfly_c(f, n, 2, R);
    for (int g = 0; g < n; g += 4) rotP(f + g + 2, 32, c, bit);
    for (int i = 0; i < n; i++) {
        const int32_t v = f[n == 8 ? o8[i] : o16[i]];
        out[i] = (i & 1) ? (int32_t)(0u - (uint32_t)v) : v;
    }
}
static void iidentity(const int32_t *x, int32_t *out, int n)
{
    for (int i = 0; i < n; i++) switch (n) {
        case 4:  out[i] = rs64((int64_t)ORC_SQRT2 * x[i], 12); break;
        case 8:  out[i] = (int32_t)((int64_t)x[i] * 2); break;
        case 16: out[i] = rs64((int64_t)ORC_SQRT2 * 2 * x[i], 12); break;
        case 32: out[i] = (int32_t)((int64_t)x[i] * 4); break;
        default: out[i] = rs64((int64_t)ORC_SQRT2 * 4 * x[i], 12); break;
    }
}
static void itxfm1d(int kind, const int32_t *x, int32_t *out, int n, int bit, int R)
{
    const int32_t *c = orc_cospi_table(bit);
    if (kind == 0) idct_rec(x, 1, out, n, c, bit, R);
    else if (kind == 3) iidentity(x, out, n);
    else if (n == 4) iadst4(x, out, bit);
    else iadst_n(x, out, n, c, bit, R);
}
/* inv_shift_WxH[0] (EbTransforms.h:255-273), indexed [log2 w - 2][log2 h - 2]; shift[1] is -4 for every size */
static const int8_t k_inv_shift0[5][5] = {{0, 0, -1, 0, 0}, {0, -1, -1, -2, 0}, {-1, -1, -2, -1, -2}, {0, -2, -1, -2, -1}, {0, 0, -2, -1, -2}};

/* residual core: input rows at `in_stride`, only the top-left min(w,32) x min(h,32) coefficients are read (the rest of a
 * 64-point dimension is zero, :7736-7760 / :7455-7476); res = w x h residuals (after the final round shift and flips) */
static void inv_txfm2d_core(const int32_t *input, int32_t in_stride, int w, int h, int tx_type, int bd, int32_t *res)
{
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    const int kc = k_vtx[tx_type], kr = k_htx[tx_type];
    const int ud = (kc == 2), lr = (kr == 2);
    const int rect = wi > hi ? wi - hi : hi - wi;
    const int sh0 = k_inv_shift0[wi][hi];
    const int R_row = bd + 8, R_col = (bd + 6 > 16) ? bd + 6 : 16;
    const int win = w > 32 ? 32 : w, hin = h > 32 ? 32 : h;
    int32_t tin[64], tout[64];
    int32_t *buf = (int32_t *)__builtin_alloca(sizeof(int32_t) * (size_t)(w * h));
    for (int r = 0; r < h; r++) {
        for (int c = 0; c < w; c++) {
            const int32_t v = (r < hin && c < win) ? input[r * in_stride + c] : 0;
            tin[c] = clampv(rect == 1 ? rs64((int64_t)v * ORC_INV_SQRT2, 12) : v, bd + 8);
        }
        itxfm1d(kr, tin, buf + r * w, w, 12, R_row);
        shift_array(buf + r * w, w, -sh0);
    }
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = clampv(buf[r * w + (lr ? w - 1 - c : c)], R_col);
        itxfm1d(kc, tin, tout, h, 12, R_col);
        shift_array(tout, h, 4);
        for (int r = 0; r < h; r++) res[r * w + c] = tout[ud ? h - 1 - r : r];
    }
}

/* Av1InverseTransformTwoD_{NxN}_c (:7343-7500): residual out, no prediction */
void orc_inv_txfm2d(const int32_t *input, int32_t in_stride, int w, int h, int tx_type, int bd, int32_t *output, int32_t out_stride)
{
    int32_t *res = (int32_t *)__builtin_alloca(sizeof(int32_t) * (size_t)(w * h));
    inv_txfm2d_core(input, in_stride, w, h, tx_type, bd, res);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) output[r * out_stride + c] = res[r * w + c];
}

/* av1_inv_txfm2d_add_{WxH}_c: input min(w,32) x min(h,32) coefficients, row stride min(w,32); output: uint16 prediction,
 * reconstructed in place */
void orc_inv_txfm2d_add(const int32_t *input, uint16_t *output, int32_t stride, int w, int h, int tx_type, int bd)
{
    int32_t *res = (int32_t *)__builtin_alloca(sizeof(int32_t) * (size_t)(w * h));
    inv_txfm2d_core(input, w > 32 ? 32 : w, w, h, tx_type, bd, res);
    const int64_t int_max = ((int64_t)1 << (7 + bd)) - 1 + (914 << (bd - 7)); /* check_range (:7135-7147) */
    const int pix_max = (1 << bd) - 1;
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            int64_t t = res[r * w + c];
            t = t < -int_max - 1 ? -int_max - 1 : (t > int_max ? int_max : t);
            const int32_t p = (int32_t)output[r * stride + c] + (int32_t)t;
            output[r * stride + c] = (uint16_t)(p < 0 ? 0 : (p > pix_max ? pix_max : p));
        }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Glue of the encode pass (row a20) and the per-TU chain of Av1EncodeLoop (Source/Lib/Codec/EbCodingLoop.c:552-760):
 * ResidualKernel -> Av1EstimateTransform -> Av1QuantizeInvQuantize -> Av1InvTransformRecon8bit.
 * ------------------------------------------------------------------------------------------------------------------ */
/* Av1EstimateTransform's handling of 64-point dimensions (EbTransforms.c:4440-4470, :4640-4660; HandleTransform*_c
 * :3894-3926, :4072-4222): energy of the coefficients outside the top-left min(w,32) x min(h,32), which are dropped, and
 * re-packing of the kept part to row stride min(w,32).  full: w x h transform output; packed: min(w,32) x min(h,32). */
uint64_t orc_pack_transform(const int32_t *full, int w, int h, int32_t *packed)
{
    const int win = w > 32 ? 32 : w, hin = h > 32 ? 32 : h;
    uint64_t energy = 0;
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const int64_t v = full[r * w + c];
            if (r < hin && c < win) packed[r * win + c] = (int32_t)v;
            else energy += (uint64_t)(v * v);
        }
    return energy;
}
int orc_tx_log_scale(int w, int h) { return (w * h > 256) + (w * h > 1024); } /* av1_get_tx_scale, EbTransforms.h:312-316 */

/* FullDistortionKernel32Bits (EbPictureOperators.c:374-404) on a dense n-coefficient block */
void orc_full_distortion(const int32_t *coeff, const int32_t *recon_coeff, int n, uint64_t dist[2])
{
    uint64_t res = 0, pred = 0;
    for (int i = 0; i < n; i++) {
        const int64_t d = (int64_t)coeff[i] - recon_coeff[i], c = coeff[i];
        res += (uint64_t)(d * d);
        pred += (uint64_t)(c * c);
    }
    dist[0] = res;
    dist[1] = pred;
}

void orc_encode_tu(const uint8_t *src, int32_t src_stride, const uint8_t *pred, int32_t pred_stride, uint8_t *recon,
                   int32_t recon_stride, int w, int h, int tx_type, const int16_t *qp, const int16_t *scan, int32_t *coeff,
                   int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, uint64_t *three_quad_energy, uint64_t dist[2])
{
    static __thread int16_t residual[64 * 64];
    static __thread int32_t full[64 * 64];
    static __thread uint16_t rec16[64 * 64];
    const int win = w > 32 ? 32 : w, hin = h > 32 ? 32 : h;
    for (int r = 0; r < h; r++) /* ResidualKernel (EbPictureOperators.c:257-285) */
        for (int c = 0; c < w; c++) residual[r * w + c] = (int16_t)((int)src[r * src_stride + c] - (int)pred[r * pred_stride + c]);
    orc_fwd_txfm2d(residual, w, w, h, tx_type, full);
    *three_quad_energy = orc_pack_transform(full, w, h, coeff);
    orc_quantize_b(coeff, win * hin, qp, scan, orc_tx_log_scale(w, h), 0, qcoeff, dqcoeff, eob);
    orc_full_distortion(coeff, dqcoeff, win * hin, dist);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) rec16[r * w + c] = pred[r * pred_stride + c];
    if (*eob) orc_inv_txfm2d_add(dqcoeff, rec16, w, w, h, tx_type, 8); /* only when the TU has coefficients (:717-727) */
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) recon[r * recon_stride + c] = (uint8_t)rec16[r * w + c];
}

/* the same chain for 10-bit samples in 16-bit planes: ResidualKernel16bit (EbPictureOperators.c:225-255), high-bit-depth
 * quantiser, Av1InvTransformRecon with bd = 10 (EbTransforms.c:8344-8372) */
void orc_encode_tu16(const uint16_t *src, int32_t src_stride, const uint16_t *pred, int32_t pred_stride, uint16_t *recon,
                     int32_t recon_stride, int w, int h, int tx_type, const int16_t *qp, const int16_t *scan, int32_t *coeff,
                     int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, uint64_t *three_quad_energy, uint64_t dist[2])
{
    static __thread int16_t residual[64 * 64];
    static __thread int32_t full[64 * 64];
    static __thread uint16_t rec16[64 * 64];
    const int win = w > 32 ? 32 : w, hin = h > 32 ? 32 : h;
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) residual[r * w + c] = (int16_t)((int)src[r * src_stride + c] - (int)pred[r * pred_stride + c]);
    orc_fwd_txfm2d(residual, w, w, h, tx_type, full);
    *three_quad_energy = orc_pack_transform(full, w, h, coeff);
    orc_quantize_b(coeff, win * hin, qp, scan, orc_tx_log_scale(w, h), 1, qcoeff, dqcoeff, eob);
    orc_full_distortion(coeff, dqcoeff, win * hin, dist);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) rec16[r * w + c] = pred[r * pred_stride + c];
    if (*eob) orc_inv_txfm2d_add(dqcoeff, rec16, w, w, h, tx_type, 10);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) recon[r * recon_stride + c] = rec16[r * w + c];
}
