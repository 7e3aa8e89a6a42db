/*
 * oracle/svt_ois_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's open-loop intra search:
 *   OpenLoopIntraSearchLcu                Source/Lib/Codec/EbMotionEstimation.c:8047-8355
 *   UpdateNeighborSamplesArrayOpenLoop    Codec/EbIntraPrediction.c:5233-5348
 *   IntraPredictionOpenLoop               Codec/EbIntraPrediction.c:5353-5446  (35 HEVC-style luma modes, unfiltered
 *                                         neighbours taken from the SOURCE picture)
 *   leaf predictors                       ASM_SSE2/EbIntraPrediction_Intrinsic_SSE2.c (planar :1826, DC :749, vertical :100,
 *                                         horizontal :306, modes 2 / 18 / 34 :1481 / :1712 / :1596),
 *                                         ASM_SSSE3/EbIntraPrediction_Intrinsic_SSSE3.c:15 / :141 (angular kernels),
 *                                         mode-range wrappers Codec/EbIntraPrediction.c:2502-2676
 *   candidate selection                   Codec/EbMotionEstimation.c:7419-7900 (IntraOpenLoopSearchTheseModesOutputBest,
 *                                         InjectIntraCandidatesBasedOnBestMode[Islice], GetInterIntraSadDistance, GetOisPoint,
 *                                         SortIntraModesOpenLoop, SortOisCandidateOpenLoop), tables :28-85
 * Pinned against the reference's own functions (oracle/ref_ois_driver.c in oracle/_ref/libsvtref_me.so) by
 * tests/test_ois_vs_ref.py: every mode x size x neighbour-availability case of the predictors and whole pictures through
 * every branch of the search.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OIS_MAX_CAND 18 /* MAX_OPEN_LOOP_INTRA_CANDIDATES, Codec/EbCodingUnit.h:42 */

/* [left top->bottom (2s)] [top-left] [top (2s)]; samples outside the picture (and everything when the CU touches the
 * picture's left / top edge) are 128 -- there is no substitution process (:5262-5316). */
void orc_ois_neighbours(const uint8_t *plane, int stride, int origin, int width, int height, int cu_x, int cu_y, int size,
                        uint8_t *refs)
{
    const uint8_t *src = plane + (size_t)(cu_y + origin) * stride + cu_x + origin;
    const int n = 2 * size;
    memset(refs, 128, (size_t)4 * size + 1);
    if (cu_x != 0)
        for (int k = 0; k < n && cu_y + k < height; k++) refs[k] = src[(ptrdiff_t)k * stride - 1];
    if (cu_x != 0 && cu_y != 0) refs[n] = src[-stride - 1];
    if (cu_y != 0)
        for (int k = 0; k < n && cu_x + k < width; k++) refs[n + 1 + k] = src[-stride + k];
}

static const int8_t kAngle[35] = {0,  0,  32, 26, 21, 17, 13, 9,  5,  2,  0,  -2, -5, -9, -13, -17, -21, -26,
                                  -32, -26, -21, -17, -13, -9, -5, -2, 0,  2,  5,  9,  13, 17, 21, 26, 32};

static int inv_angle(int a)
{
    switch (a < 0 ? -a : a) {
    case 2: return 4096;
    case 5: return 1638;
    case 9: return 910;
    case 13: return 630;
    case 17: return 482;
    case 21: return 390;
    case 26: return 315;
    case 32: return 256;
    }
    return 0;
}

static uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void orc_ois_predict(const uint8_t *refs, int s, int mode, uint8_t *pred)
{
    const uint8_t *left = refs, *top = refs + 2 * s + 1;
    const int tl = refs[2 * s];
    int lg = 0;
    while ((1 << lg) < s) lg++;
    if (mode == 0) { /* planar */
        for (int y = 0; y < s; y++)
            for (int x = 0; x < s; x++)
                pred[y * s + x] = (uint8_t)(((s - 1 - x) * left[y] + (x + 1) * top[s] + (s - 1 - y) * top[x] + (y + 1) * left[s] + s) >> (lg + 1));
        return;
    }
    if (mode == 1) { /* DC with the luma edge smoothing below 32x32 */
        int sum = s;
        for (int k = 0; k < s; k++) sum += top[k] + left[k];
        const int dc = sum >> (lg + 1);
        for (int i = 0; i < s * s; i++) pred[i] = (uint8_t)dc;
        if (s < 32) {
            pred[0] = (uint8_t)((left[0] + 2 * dc + top[0] + 2) >> 2);
            for (int x = 1; x < s; x++) pred[x] = (uint8_t)((top[x] + 3 * dc + 2) >> 2);
            for (int y = 1; y < s; y++) pred[y * s] = (uint8_t)((left[y] + 3 * dc + 2) >> 2);
        }
        return;
    }
    const int vertical = mode >= 18;
    const int angle = kAngle[mode];
    const uint8_t *mainr = vertical ? top : left, *side = vertical ? left : top;
    uint8_t buf[3 * 64 + 2];
    uint8_t *m = buf + 64; /* m[0] = top-left, m[k] = main[k-1], m[-k] projected from the side */
    m[0] = (uint8_t)tl;
    for (int k = 1; k <= 2 * s; k++) m[k] = mainr[k - 1];
    if (angle < 0) {
        const int inv = inv_angle(angle);
        for (int k = -1; k > ((s * angle) >> 5); k--) {
            const int idx = (128 + inv * (-k)) >> 8;
            m[k] = idx == 0 ? (uint8_t)tl : side[idx - 1];
        }
    }
    for (int a = 0; a < s; a++) { /* a = row for vertical modes, column for horizontal ones */
        const int d = (a + 1) * angle, i = d >> 5, f = d & 31;
        for (int b = 0; b < s; b++) {
            const int v = ((32 - f) * m[b + i + 1] + f * m[b + i + 2] + 16) >> 5;
            if (vertical)
                pred[a * s + b] = (uint8_t)v;
            else
                pred[b * s + a] = (uint8_t)v;
        }
    }
    if (angle == 0 && s < 32) { /* pure vertical / horizontal: gradient filter on the first column / row */
        for (int k = 0; k < s; k++) {
            const uint8_t v = clip8(mainr[0] + ((side[k] - tl) >> 1));
            if (vertical)
                pred[k * s] = v;
            else
                pred[k] = v;
        }
    }
}

static uint32_t sad_block(const uint8_t *src, int stride, const uint8_t *pred, int s)
{
    uint32_t sad = 0;
    for (int y = 0; y < s; y++)
        for (int x = 0; x < s; x++) sad += (uint32_t)abs((int)src[(size_t)y * stride + x] - (int)pred[y * s + x]);
    return sad;
}

/* ---------------------------------------------------------------------------------------------------------------- */
enum { OP_SLICE_I, OP_TEMPORAL_LAYER, OP_IS_REF, OP_RES_4K, OP_LIMIT_DC, OP_CU8X8_MODE, OP_ENC_MODE, OP_COUNT };

static const int32_t kOisPointTh[3][6][4] = {
    {{-20, 50, 150, 200}, {-20, 50, 150, 200}, {-20, 50, 100, 150}, {-20, 50, 200, 300}, {-20, 50, 200, 300}, {-20, 50, 200, 300}},
    {{-150, 0, 150, 200}, {-150, 0, 150, 200}, {-125, 0, 100, 150}, {-50, 50, 200, 300}, {-50, 50, 200, 300}, {-50, 50, 200, 300}},
    {{-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}}};
static const uint8_t kNumModes[5] = {1, 3, 5, 7, 9};
static const uint8_t kISliceModes[7] = {0, 1, 10, 26, 2, 18, 34};
static const uint8_t kStage1Modes[9] = {10, 26, 2, 18, 34, 6, 14, 22, 30};

/* the 9 modes InjectIntraCandidatesBasedOnBestMode writes for each stage-1 winner (:7529-7775), in kStage1Modes order */
static const uint8_t kInject[9][9] = {
    {10, 1, 0, 9, 11, 8, 12, 7, 13},    {26, 1, 0, 25, 27, 24, 28, 23, 29}, {2, 1, 0, 3, 4, 5, 7, 8, 9},
    {18, 1, 0, 17, 19, 16, 20, 15, 21}, {34, 1, 0, 33, 32, 29, 31, 27, 28}, {6, 1, 0, 7, 5, 4, 8, 3, 9},
    {14, 1, 0, 13, 15, 12, 16, 11, 17}, {22, 1, 0, 21, 23, 20, 24, 19, 25}, {30, 1, 0, 29, 31, 28, 32, 27, 33}};

static uint32_t cand_word(uint32_t dist, int valid, int mode) /* OisCandidate_t, Codec/EbCodingUnit.h:303-313 */
{
    return (dist & 0xfffffu) | ((uint32_t)(valid & 1) << 20) | ((uint32_t)mode << 24);
}

static void cu_geometry(int cu, int *x, int *y, int *s, int *depth)
{
    if (cu < 5) { *s = 32; *depth = 1; *x = ((cu - 1) & 1) * 32; *y = ((cu - 1) >> 1) * 32; }
    else if (cu < 21) { *s = 16; *depth = 2; *x = ((cu - 5) & 3) * 16; *y = ((cu - 5) >> 2) * 16; }
    else { *s = 8; *depth = 3; *x = ((cu - 21) & 7) * 8; *y = ((cu - 21) >> 3) * 8; }
}

/* SADs of all 35 modes of every CU of every SB: out_sad [n_sb][85][35] (0 where the CU is not wholly inside the picture). */
int orc_ois_sad_table(const uint8_t *plane, int stride, int origin, int width, int height, uint32_t *out_sad)
{
    const int nx = (width + 63) / 64, ny = (height + 63) / 64;
    uint8_t refs[4 * 32 + 1], pred[32 * 32];
    memset(out_sad, 0, (size_t)nx * ny * 85 * 35 * 4);
    for (int sb = 0; sb < nx * ny; sb++)
        for (int cu = 1; cu < 85; cu++) {
            int cx, cy, s, depth;
            cu_geometry(cu, &cx, &cy, &s, &depth);
            cx += (sb % nx) * 64;
            cy += (sb / nx) * 64;
            if (cx + s > width || cy + s > height) continue;
            orc_ois_neighbours(plane, stride, origin, width, height, cx, cy, s, refs);
            const uint8_t *src = plane + (size_t)(cy + origin) * stride + cx + origin;
            for (int m = 0; m < 35; m++) {
                orc_ois_predict(refs, s, m, pred);
                out_sad[((size_t)sb * 85 + cu) * 35 + m] = sad_block(src, stride, pred, s);
            }
        }
    return 0;
}

/* OpenLoopIntraSearchLcu for every SB; out_cand [n_sb][85][18] OisCandidate_t words, out_total [n_sb][85]; fields the
 * reference leaves untouched are zero (the reference keeps whatever an earlier picture stored there). */
int orc_ois_search_picture(const uint8_t *plane, int stride, int origin, int width, int height, const int32_t *op,
                           const uint32_t *me_dist, uint32_t *out_cand, uint8_t *out_total)
{
    const int nx = (width + 63) / 64, ny = (height + 63) / 64, nsb = nx * ny;
    uint8_t refs[4 * 32 + 1], pred[32 * 32];
    memset(out_cand, 0, (size_t)nsb * 85 * OIS_MAX_CAND * 4);
    memset(out_total, 0, (size_t)nsb * 85);
    const int tl = op[OP_TEMPORAL_LAYER];
    const int th_set = op[OP_RES_4K] ? ((tl == 0 || op[OP_IS_REF]) ? 2 : 1) : 2; /* :8151-8172 */
    for (int sb = 0; sb < nsb; sb++)
        for (int cu = 1; cu < 85; cu++) {
            int cx, cy, s, depth;
            cu_geometry(cu, &cx, &cy, &s, &depth);
            cx += (sb % nx) * 64;
            cy += (sb / nx) * 64;
            uint32_t *o = out_cand + ((size_t)sb * 85 + cu) * OIS_MAX_CAND;
            uint8_t *total = out_total + (size_t)sb * 85 + cu;
            if (cx + s > width || cy + s > height) continue;
            if (!op[OP_SLICE_I] && op[OP_CU8X8_MODE] && s == 8) continue;
            const uint8_t *src = plane + (size_t)(cy + origin) * stride + cx + origin;
            orc_ois_neighbours(plane, stride, origin, width, height, cx, cy, s, refs);
            uint32_t sad[35];
#define SAD_OF(m) (orc_ois_predict(refs, s, (m), pred), sad_block(src, stride, pred, s))
            if (op[OP_SLICE_I]) { /* :8076-8153 */
                if (s == 32) {
                    o[0] = cand_word(SAD_OF(0), 1, 0);
                    continue; /* total_intra_luma_mode is not written for 32x32 CUs of I pictures */
                }
                uint32_t best_sad = 32 * 32 * 255;
                int best = 0;
                for (int k = 0; k < 7; k++) {
                    sad[k] = SAD_OF(kISliceModes[k]);
                    if (sad[k] < best_sad) { best_sad = sad[k]; best = kISliceModes[k]; }
                }
                int n = 0;
                uint8_t modes[5];
                modes[n++] = 0;
                modes[n++] = 1;
                switch (best) { /* InjectIntraCandidatesBasedOnBestModeIslice :7465-7523 */
                case 0: case 1: break;
                case 2: modes[n++] = 2; modes[n++] = 4; modes[n++] = 6; break;
                case 10: modes[n++] = 10; modes[n++] = 6; modes[n++] = 14; break;
                case 18: modes[n++] = 18; modes[n++] = 14; modes[n++] = 22; break;
                case 26: modes[n++] = 26; modes[n++] = 22; modes[n++] = 30; break;
                default: modes[n++] = 34; modes[n++] = 32; modes[n++] = 30; break;
                }
                o[0] = cand_word(sad[0], 1, modes[0]);
                for (int k = 1; k < n; k++) o[k] = cand_word(0, 0, modes[k]);
                *total = (uint8_t)n;
                continue;
            }
            if (tl == 0 && !op[OP_RES_4K]) { /* all 35 modes, best 18 kept and sorted (:8219-8270) */
                uint32_t d[OIS_MAX_CAND];
                uint8_t md[OIS_MAX_CAND];
                for (int m = 0; m < 35; m++) {
                    const uint32_t v = SAD_OF(m);
                    if (m < OIS_MAX_CAND) { d[m] = v; md[m] = (uint8_t)m; continue; }
                    int worst = 0; /* SortIntraModesOpenLoop :7872-7900 */
                    for (int k = 1; k < OIS_MAX_CAND; k++)
                        if (d[k] > d[worst]) worst = k;
                    if (v < d[worst]) { d[worst] = v; md[worst] = (uint8_t)m; }
                }
                for (int a = 0; a < OIS_MAX_CAND; a++) /* SortOisCandidateOpenLoop :7840-7866 */
                    for (int b = a; b < OIS_MAX_CAND; b++)
                        if (d[a] > d[b]) {
                            const uint32_t t = d[a]; d[a] = d[b]; d[b] = t;
                            const uint8_t u = md[a]; md[a] = md[b]; md[b] = u;
                        }
                for (int k = 0; k < OIS_MAX_CAND; k++) o[k] = cand_word(d[k], 0, md[k]);
                *total = OIS_MAX_CAND;
                continue;
            }
            const uint32_t dc_sad = SAD_OF(1);
            if (op[OP_LIMIT_DC]) { /* OpenLoopIntraDC :7951-8022 */
                o[0] = cand_word(dc_sad, 1, 1);
                *total = 1;
                continue;
            }
            const uint32_t me_sad = me_dist[(size_t)sb * 85 + cu];
            const int32_t diff = (int32_t)(me_sad - dc_sad) * 100; /* GetInterIntraSadDistance :7779-7812 */
            const int32_t distance = dc_sad ? diff / (int32_t)dc_sad : 0;
            int point = 4; /* GetOisPoint :7826-7852 */
            const int32_t *th = kOisPointTh[th_set][tl];
            if (dc_sad == 0 || me_sad == 0 || distance <= th[0]) point = 0;
            else if (distance <= th[1]) point = 1;
            else if (distance <= th[2]) point = 2;
            else if (distance <= th[3]) point = 3;
            if (point == 0) {
                o[0] = cand_word(dc_sad, 0, 1);
                *total = 1;
                continue;
            }
            uint32_t best_sad = 32 * 32 * 255;
            int best = 8;
            for (int k = 0; k < kNumModes[point]; k++) {
                sad[k] = SAD_OF(kStage1Modes[k]);
                if (sad[k] < best_sad) { best_sad = sad[k]; best = k; }
            }
            /* the vertical winner's flag depends on enc_mode, the others do not under ENCODER_MODE_CLEANUP (:7609-7613) */
            const int valid = (best == 1 && op[OP_ENC_MODE] > 1) ? 1 : (tl > 1);
            o[0] = cand_word(sad[best], valid, kInject[best][0]);
            for (int k = 1; k < 9; k++) o[k] = cand_word(0, 0, kInject[best][k]);
            *total = kNumModes[point]; /* intraSearchInMd[point][depth], identical for depths 1..3 (:74-81) */
#undef SAD_OF
        }
    return 0;
}
