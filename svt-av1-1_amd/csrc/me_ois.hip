// svt-av1-1_amd/csrc/me_ois.hip
//
// Open-loop intra search of a batch of superblocks on gfx950: the per-SB body of OpenLoopIntraSearchLcu
// (reference: Source/Lib/Codec/EbMotionEstimation.c:8047-8355) with everything it calls --
//   UpdateNeighborSamplesArrayOpenLoop / IntraPredictionOpenLoop     Codec/EbIntraPrediction.c:5233-5446
//   the 35 HEVC-style luma predictors                                ASM_SSE2/EbIntraPrediction_Intrinsic_SSE2.c,
//                                                                    ASM_SSSE3/EbIntraPrediction_Intrinsic_SSSE3.c:15-, Codec/EbIntraPrediction.c:2502-2676
//   plain SAD (NxMSadKernel row ASM_NON_AVX2 = FastLoop_NxMSadKernel)   Codec/EbComputeSAD.h:125-138
//   candidate selection                                              Codec/EbMotionEstimation.c:7419-7900, tables :28-85
// for the 4 + 16 + 64 CUs (32x32, 16x16, 8x8) of every SB.
//
// Mapping: one 256-thread workgroup per (picture, SB).
//   * The SB and the samples around it that any CU's neighbour arrays can touch (row / column -1 .. 95: top-right and bottom-left
//     extensions reach 32 samples past the SB) are staged once into LDS, row-major (T) and transposed (TT); samples outside the picture
//     are stored as 128, which is exactly the reference's "memset 128, copy what exists" neighbour rule, so a CU's top row is a tile row
//     and its left column a row of the transposed tile: no per-CU neighbour array is built.
//   * Phase 1: SADs of every mode the picture's branch can ask for (7 / 10 / 1 / 35 modes).  A wave pass covers 64 lanes x 4 pixels:
//     four 8x8 CUs, one 16x16 CU or a quarter of a 32x32 CU.  Horizontal-class modes (2..17) run as their vertical twins (36 - mode)
//     on the transposed tile -- SAD does not care about transposition.  Angular taps are a 2-pixel blend at a per-row offset: five
//     bytes from two LDS dwords, blended four pixels at a time in two 16-bit SWAR lanes per dword, then v_sad_u8 against the source.
//     Negative angles first project the side samples into a small per-wave array (the reference's refAbove / refLeft extension).
//   * Phase 2: one lane per CU walks the reference's decision code on the SAD table (stage-1 best, OIS point from the inter / intra
//     distance, injected mode lists, or the sorted best-18 list of base-layer pictures) and the rows are copied out coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace svthip {

namespace {

constexpr int kTP = 100;                 // tile pitch in bytes: 25 dwords (odd => rows spread over the banks), sample x at byte x + 4
constexpr int kTileBytes = 98 * kTP;     // rows -1 .. 95 (+1 row of slack for the 8-byte fetches at the end)
constexpr int kSadStride = 36;           // SAD table [85][36]
constexpr int kScr = 112;                // projected main array of one sub-item: m[k] at byte 36 + k, k = -32 .. 65
constexpr int kCand = 18;                // MAX_OPEN_LOOP_INTRA_CANDIDATES

__constant__ const uint8_t kOisModeList[4][36] = {
    {7, 0, 1, 10, 26, 2, 18, 34},                   // I pictures: PL, DC, H, V, 2, 18, 34
    {35, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
     18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34},
    {1, 1},                                         // limit_ois_to_dc_mode_flag
    {10, 1, 10, 26, 2, 18, 34, 6, 14, 22, 30}};     // DC + stage1ModesArray
// What a mode-loop iteration needs, ONE dword per (mode list, entry) in the constant address space: mode | angle of its vertical twin << 8 |
// inverse angle << 16; entry 0 = the number of modes.  The index is wave-uniform, so the look-up is a scalar load.  As three byte / halfword
// tables (mode list, angle, inverse angle) every iteration issued three dependent vector loads -- 280 per wave, the texture path busy 0.7 of
// the launch and the waves waiting 57 % of their time (profiles/r03_pmcx_ois_before.txt).
struct OisModeInfo {
    uint32_t v[4][36];
};
constexpr OisModeInfo ois_make_mode_info()
{
    constexpr int8_t angle_of[17] = {-32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
    constexpr uint16_t inv_of[8] = {256, 315, 390, 482, 630, 910, 1638, 4096};
    constexpr uint8_t lists[4][36] = {{7, 0, 1, 10, 26, 2, 18, 34},
                                      {35, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                                       18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34},
                                      {1, 1},
                                      {10, 1, 10, 26, 2, 18, 34, 6, 14, 22, 30}};
    OisModeInfo t{};
    for (int p = 0; p < 4; p++) {
        t.v[p][0] = lists[p][0];
        for (int i = 1; i <= lists[p][0]; i++) {
            const int m = lists[p][i];
            const bool frame_v = m >= 18 || m < 2;
            const int vm = m < 2 ? m : frame_v ? m : 36 - m;
            const int angle = vm >= 18 ? angle_of[vm - 18] : 0;
            const uint32_t inv = (vm >= 18 && angle < 0) ? inv_of[vm - 18] : 0u;
            t.v[p][i] = (uint32_t)m | ((uint32_t)(uint8_t)(int8_t)angle << 8) | (inv << 16);
        }
    }
    return t;
}
__constant__ const OisModeInfo kOisModeInfo = ois_make_mode_info();
__constant__ const int16_t kOisPointTh[3][6][4] = {
    {{-20, 50, 150, 200}, {-20, 50, 150, 200}, {-20, 50, 100, 150}, {-20, 50, 200, 300}, {-20, 50, 200, 300}, {-20, 50, 200, 300}},
    {{-150, 0, 150, 200}, {-150, 0, 150, 200}, {-125, 0, 100, 150}, {-50, 50, 200, 300}, {-50, 50, 200, 300}, {-50, 50, 200, 300}},
    {{-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}, {-400, -300, -200, 0}}};
// InjectIntraCandidatesBasedOnBestMode (:7525-7776): the nine modes written for each stage-1 winner, in stage1ModesArray order
__constant__ const uint8_t kOisInject[9][9] = {
    {10, 1, 0, 9, 11, 8, 12, 7, 13},    {26, 1, 0, 25, 27, 24, 28, 23, 29}, {2, 1, 0, 3, 4, 5, 7, 8, 9},
    {18, 1, 0, 17, 19, 16, 20, 15, 21}, {34, 1, 0, 33, 32, 29, 31, 27, 28}, {6, 1, 0, 7, 5, 4, 8, 3, 9},
    {14, 1, 0, 13, 15, 12, 16, 11, 17}, {22, 1, 0, 21, 23, 20, 24, 19, 25}, {30, 1, 0, 29, 31, 28, 32, 27, 33}};

struct OisLds {
    uint8_t T[kTileBytes];
    uint8_t TT[kTileBytes];
    uint32_t sad[85 * kSadStride];
    uint8_t scr[4 * 4 * kScr];
    uint32_t cand[85 * kCand];
    uint8_t total[88];
};

__device__ __forceinline__ uint32_t cand_word(uint32_t dist, uint32_t valid, uint32_t mode)  // OisCandidate_t, Codec/EbCodingUnit.h:303-313
{
    return (dist & 0xfffffu) | (valid << 20) | (mode << 24);
}

// bytes [addr, addr + 4) -> a, [addr + 1, addr + 5) -> b, from the two dwords that hold them
__device__ __forceinline__ void fetch5(const uint8_t* base, int addr, uint32_t& a, uint32_t& b)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (addr & ~3));
    const uint32_t lo = p[0], hi = p[1];
    const uint32_t sh = ((uint32_t)addr & 3u) * 8u;
    a = __builtin_amdgcn_alignbit(hi, lo, sh);
    b = __builtin_amdgcn_alignbit(hi >> sh, a, 8);
}

// ((32 - f) * a + f * b + 16) >> 5 on four byte pairs: even and odd bytes as two 16-bit lanes each (255 * 32 + 16 < 2^16)
__device__ __forceinline__ uint32_t blend4(uint32_t a, uint32_t b, uint32_t f)
{
    const uint32_t m = 0x00ff00ffu, w0 = 32u - f;
    const uint32_t e = __umul24(a & m, w0) + __umul24(b & m, f) + 0x00100010u;
    const uint32_t o = __umul24((a >> 8) & m, w0) + __umul24((b >> 8) & m, f) + 0x00100010u;
    return ((e >> 5) & m) | (((o >> 5) & m) << 8);
}

template <int LPS>
__device__ __forceinline__ uint32_t sub_sum(uint32_t v)  // sum over the LPS consecutive lanes of a sub-item
{
#pragma unroll
    for (int m = 1; m < LPS; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// SADs of the modes in `list` for the CUs first_cu .. first_cu + NSUB - 1 (S x S each), one wave.
//   S = 8 : 4 CUs per pass (16 lanes each);  S = 16 : one CU per pass;  S = 32 : one CU in four passes of 8 rows.
template <int S>
__device__ __forceinline__ void ois_unit(OisLds& L, int first_cu, int level_first, const uint32_t* list, uint8_t* scr_wave)
{
    constexpr int LPS = (S == 8) ? 16 : 64;        // lanes per CU
    constexpr int IT = (S == 32) ? 4 : 1;          // passes per mode
    constexpr int LPR = S / 4;                     // lanes per row
    constexpr int RPP = LPS / LPR;                 // rows per pass
    constexpr int LG = (S == 8) ? 3 : (S == 16) ? 4 : 5;
    constexpr int PER_ROW = 64 / S;                // CUs per SB row at this level
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPS, l = lane % LPS;
    const int cu = first_cu + sub;
    const int ci = cu - level_first;
    const int cx = (ci % PER_ROW) * S, cy = (ci / PER_ROW) * S;
    const int r0 = l / LPR, c0 = (l % LPR) * 4;
    uint8_t* scr = scr_wave + sub * kScr;

    // frame V: rows are picture rows (tile T, side = TT); frame H: rows are picture columns (tile TT, side = T)
    const int mbV = cy * kTP + cx + 3;   // offset of m[0] (top-left) in the main tile; main[k] = m[k + 1]
    const int mbH = cx * kTP + cy + 3;
    uint32_t srcV[IT], srcH[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int r = it * RPP + r0;
        srcV[it] = *reinterpret_cast<const uint32_t*>(L.T + (cy + r + 1) * kTP + cx + c0 + 4);
        srcH[it] = *reinterpret_cast<const uint32_t*>(L.TT + (cx + r + 1) * kTP + cy + c0 + 4);
    }
    // DC value (mode 1): (sum of S top + S left + S) >> (log2 S + 1)
    uint32_t dc;
    {
        uint32_t v = 0;
        if (l < LPR) {
            uint32_t a, b;
            fetch5(L.T, mbV + 1 + 4 * l, a, b);
            v = __builtin_amdgcn_sad_u8(a, 0u, 0u);
            fetch5(L.TT, mbH + 1 + 4 * l, a, b);
            v = __builtin_amdgcn_sad_u8(a, 0u, v);
        }
        v += __shfl_xor(v, 1);
        if (LPR >= 4) v += __shfl_xor(v, 2);
        if (LPR >= 8) v += __shfl_xor(v, 4);
        dc = ((uint32_t)__shfl((int)v, sub * LPS) + S) >> (LG + 1);
    }

    const int n_modes = (int)list[0];
    for (int mi = 0; mi < n_modes; mi++) {
        const uint32_t info = list[1 + mi];  // wave-uniform: a scalar load
        const int mode = (int)(info & 0xffu);
        const bool frame_v = (mode >= 18) || (mode < 2);
        const uint8_t* A = frame_v ? L.T : L.TT;
        const uint8_t* B = frame_v ? L.TT : L.T;
        const int mb = frame_v ? mbV : mbH;    // m[0] in A
        const int sbo = frame_v ? mbH : mbV;   // side[-1] (top-left) in B; side[j] = B[sbo + 1 + j]
        const int vm = (mode < 2) ? mode : (frame_v ? mode : 36 - mode);
        uint32_t acc = 0;
        if (vm >= 18) {
            const int angle = (int)(int8_t)(info >> 8);
            const uint8_t* mp = A;
            int mo = mb;
            if (angle < 0) {
                const int inv = (int)(info >> 16);
                for (int t = l; t <= 2 * S; t += LPS) {
                    const int k = t - S;
                    uint8_t v;
                    if (k >= 0) {
                        v = A[mb + k];
                    } else {
                        int idx = (128 + inv * (-k)) >> 8;
                        idx = idx > 2 * S ? 2 * S : idx;  // entries below the reference's fill limit are never read
                        v = B[sbo + idx];
                    }
                    scr[36 + k] = v;
                }
                __builtin_amdgcn_wave_barrier();
                mp = scr;
                mo = 36;
            }
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int r = it * RPP + r0;
                const int d = (r + 1) * angle;
                uint32_t a, b;
                fetch5(mp, mo + c0 + (d >> 5) + 1, a, b);
                uint32_t p = blend4(a, b, (uint32_t)d & 31u);
                if (S < 32 && angle == 0 && c0 == 0) {  // boundary gradient of the pure vertical / horizontal mode
                    int e = (int)A[mb + 1] + (((int)B[sbo + 1 + r] - (int)A[mb]) >> 1);
                    e = e < 0 ? 0 : e > 255 ? 255 : e;
                    p = (p & 0xffffff00u) | (uint32_t)e;
                }
                acc = __builtin_amdgcn_sad_u8(p, frame_v ? srcV[it] : srcH[it], acc);
            }
            __builtin_amdgcn_wave_barrier();
        } else if (vm == 1) {
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int r = it * RPP + r0;
                uint32_t p = dc * 0x01010101u;
                if (S < 32) {
                    if (r == 0) {
                        uint32_t a, b;
                        fetch5(L.T, mbV + 1 + c0, a, b);
                        const uint32_t m = 0x00ff00ffu, k3 = (3u * dc + 2u) * 0x00010001u;
                        const uint32_t e = (((a & m) + k3) >> 2) & m, o = ((((a >> 8) & m) + k3) >> 2) & m;
                        p = e | (o << 8);
                        if (c0 == 0) p = (p & 0xffffff00u) | (((uint32_t)L.TT[mbH + 1] + 2u * dc + (a & 0xffu) + 2u) >> 2);
                    } else if (c0 == 0) {
                        p = (p & 0xffffff00u) | (((uint32_t)L.TT[mbH + 1 + r] + 3u * dc + 2u) >> 2);
                    }
                }
                acc = __builtin_amdgcn_sad_u8(p, srcV[it], acc);
            }
        } else {  // planar
            const int tr = L.T[mbV + S + 1], bl = L.TT[mbH + S + 1];
            uint32_t a, b;
            fetch5(L.T, mbV + 1 + c0, a, b);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int r = it * RPP + r0;
                const int lf = L.TT[mbH + 1 + r];
                const int base = (r + 1) * bl + S + lf * (S - 1 - c0) + tr * (c0 + 1);
                const int wy = S - 1 - r, dx = tr - lf;
                uint32_t p = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (base + j * dx + wy * (int)((a >> (8 * j)) & 0xffu)) >> (LG + 1);
                    p |= (uint32_t)v << (8 * j);
                }
                acc = __builtin_amdgcn_sad_u8(p, srcV[it], acc);
            }
        }
        acc = sub_sum<LPS>(acc);
        if (l == 0) L.sad[cu * kSadStride + mode] = acc;
    }
}

__device__ __forceinline__ void ois_decide(OisLds& L, const svthip_ois_params& P, int path, int cu, bool cu_valid, uint32_t me_sad)
{
    uint32_t* o = L.cand + cu * kCand;
    const uint32_t* sad = L.sad + cu * kSadStride;
    const int s = cu < 5 ? 32 : cu < 21 ? 16 : 8;
    for (int k = 0; k < kCand; k++) o[k] = 0;
    uint32_t total = 0;
    if (cu_valid && !(path != 0 && P.cu8x8_mode && s == 8)) {
        if (path == 0) {  // I pictures (:8076-8153)
            if (s == 32) {
                o[0] = cand_word(sad[0], 1, 0);
            } else {
                uint32_t best_sad = 32 * 32 * 255;
                int best = 0;
                for (int k = 0; k < 7; k++) {
                    const int m = kOisModeList[0][1 + k];
                    if (sad[m] < best_sad) { best_sad = sad[m]; best = m; }
                }
                o[0] = cand_word(sad[0], 1, 0);
                o[1] = cand_word(0, 0, 1);
                total = 2;
                if (best >= 2) {  // InjectIntraCandidatesBasedOnBestModeIslice (:7465-7523)
                    const int a = best, b = best == 2 ? 4 : best == 10 ? 6 : best == 18 ? 14 : best == 26 ? 22 : 32,
                              c = best == 2 ? 6 : best == 10 ? 14 : best == 18 ? 22 : 30;
                    o[2] = cand_word(0, 0, a);
                    o[3] = cand_word(0, 0, b);
                    o[4] = cand_word(0, 0, c);
                    total = 5;
                }
            }
        } else if (path == 1) {  // base-layer pictures below 4K: best 18 of all 35 modes, sorted (:8219-8270)
            for (int m = 0; m < kCand; m++) o[m] = (sad[m] << 8) | (uint32_t)m;
            for (int m = kCand; m < 35; m++) {  // SortIntraModesOpenLoop: replace the first worst entry when better
                int worst = 0;
                uint32_t wd = o[0] >> 8;
                for (int k = 1; k < kCand; k++) {
                    const uint32_t d = o[k] >> 8;
                    if (d > wd) { wd = d; worst = k; }
                }
                if (sad[m] < wd) o[worst] = (sad[m] << 8) | (uint32_t)m;
            }
            for (int a = 0; a < kCand; a++) {  // SortOisCandidateOpenLoop: exchange sort, strict '>' on the distortion
                uint32_t ka = o[a];
                for (int b = a + 1; b < kCand; b++) {
                    const uint32_t kb = o[b];
                    if ((ka >> 8) > (kb >> 8)) { o[b] = ka; ka = kb; }
                }
                o[a] = ka;
            }
            for (int k = 0; k < kCand; k++) o[k] = cand_word(o[k] >> 8, 0, o[k] & 0xffu);
            total = kCand;
        } else if (path == 2) {  // OpenLoopIntraDC (:7951-8022)
            o[0] = cand_word(sad[1], 1, 1);
            total = 1;
        } else {
            const uint32_t dc_sad = sad[1];
            const int32_t diff = (int32_t)(me_sad - dc_sad) * 100;            // GetInterIntraSadDistance (:7779-7812)
            const int32_t distance = dc_sad ? diff / (int32_t)dc_sad : 0;
            const int tl = P.temporal_layer_index;
            const int th_set = P.input_resolution_4k ? ((tl == 0 || P.is_used_as_reference_flag) ? 2 : 1) : 2;  // (:8151-8172)
            const int16_t* th = kOisPointTh[th_set][tl];
            int point = 4;                                                     // GetOisPoint (:7826-7852)
            if (dc_sad == 0 || me_sad == 0 || distance <= th[0]) point = 0;
            else if (distance <= th[1]) point = 1;
            else if (distance <= th[2]) point = 2;
            else if (distance <= th[3]) point = 3;
            total = 2 * point + 1;  // numberOfOisModePoints == intraSearchInMd[point][1..3]
            if (point == 0) {
                o[0] = cand_word(dc_sad, 0, 1);
            } else {
                uint32_t best_sad = 32 * 32 * 255;
                int best = 8;
                for (int k = 0; k < 2 * point + 1; k++) {
                    const uint32_t v = sad[kOisModeList[3][2 + k]];
                    if (v < best_sad) { best_sad = v; best = k; }
                }
                const uint32_t valid = (best == 1 && P.enc_mode > 1) ? 1u : (tl > 1 ? 1u : 0u);  // (:7609-7613)
                o[0] = cand_word(best_sad, valid, kOisInject[best][0]);
                for (int k = 1; k < 9; k++) o[k] = cand_word(0, 0, kOisInject[best][k]);
            }
        }
    }
    L.total[cu] = (uint8_t)total;
}

}  // namespace

__global__ void __launch_bounds__(256) ois_kernel(const uint8_t* __restrict__ pool, PaJobTable jobs, svthip_ois_params P,
                                                  const svthip_sb_origin* __restrict__ sbs, uint32_t n_sb, uint32_t n_jobs,
                                                  const svthip_me_cu_result* __restrict__ me, uint32_t me_stride,
                                                  uint32_t* __restrict__ out_cand, uint8_t* __restrict__ out_total)
{
    __shared__ __attribute__((aligned(16))) OisLds L;
    const uint32_t item = xcd_item(blockIdx.x, n_sb * n_jobs);
    if (item >= n_sb * n_jobs) return;
    const uint32_t job = item / n_sb, sb_local = item - job * n_sb;
    const svthip_pa_picture pic = jobs.pic[job];
    const int ox = sbs[sb_local].x, oy = sbs[sb_local].y;
    const int width = pic.width, height = pic.height;
    const int tid = threadIdx.x;

    // ---- stage rows / columns -1 .. 95 around the SB; 128 where the picture has no sample ----
    const uint8_t* plane = pool + pic.full_offset + (int64_t)68 * pic.full_stride + 68;
    for (int i = tid; i < 97 * 25; i += 256) {
        const int row = i / 25, dwi = i - row * 25;
        const int y = row - 1, x = dwi * 4 - 4;  // dword 0 holds x = -4 .. -1 (only -1 is used)
        const int py = oy + y, px = ox + x;
        uint32_t v = 0x80808080u;
        if (py >= 0 && py < height && px >= 0 && px < width)
            v = *reinterpret_cast<const uint32_t*>(plane + (int64_t)py * pic.full_stride + px);
        *reinterpret_cast<uint32_t*>(L.T + row * kTP + dwi * 4) = v;
        if (x >= 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) L.TT[(x + j + 1) * kTP + y + 4] = (uint8_t)(v >> (8 * j));
        } else {
            L.TT[0 * kTP + y + 4] = (uint8_t)(v >> 24);
        }
    }
    for (int i = tid; i < 85 * kSadStride; i += 256) L.sad[i] = 0;
    __syncthreads();

    const int path = P.slice_is_intra ? 0 : (P.temporal_layer_index == 0 && !P.input_resolution_4k) ? 1 : P.limit_ois_to_dc_mode_flag ? 2 : 3;
    const uint32_t* list = kOisModeInfo.v[path];
    const int wave = tid >> 6;
    uint8_t* scr_wave = L.scr + wave * 4 * kScr;
    // per wave: one 32x32 CU, four 16x16 CUs, four groups of four 8x8 CUs (48 equal pass-units per SB and mode)
    ois_unit<32>(L, 1 + wave, 1, list, scr_wave);
    for (int k = 0; k < 4; k++) ois_unit<16>(L, 5 + wave * 4 + k, 5, list, scr_wave);
    for (int k = 0; k < 4; k++) ois_unit<8>(L, 21 + (wave * 4 + k) * 4, 21, list, scr_wave);
    __syncthreads();

    // ---- decisions: one lane per CU ----
    if (tid < 85) {
        if (tid == 0) {
            for (int k = 0; k < kCand; k++) L.cand[k] = 0;
            L.total[0] = 0;
        } else {
            const int cu = tid;
            const int s = cu < 5 ? 32 : cu < 21 ? 16 : 8;
            const int ci = cu < 5 ? cu - 1 : cu < 21 ? cu - 5 : cu - 21;
            const int per_row = 64 / s;
            const int cx = (ci % per_row) * s, cy = (ci / per_row) * s;
            const bool cu_valid = (ox + cx + s <= width) && (oy + cy + s <= height);
            const uint32_t me_sad = (path == 3 && me) ? me[(size_t)item * me_stride + cu].distortion[0] : 0u;
            ois_decide(L, P, path, cu, cu_valid, me_sad);
        }
    }
    __syncthreads();
    uint32_t* oc = out_cand + (size_t)item * 85 * kCand;
    for (int i = tid; i < 85 * kCand; i += 256) oc[i] = L.cand[i];
    if (tid < 85) out_total[(size_t)item * 85 + tid] = L.total[tid];
}

}  // namespace svthip
