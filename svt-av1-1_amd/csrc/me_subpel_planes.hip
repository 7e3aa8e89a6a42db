// svt-av1-1_amd/csrc/me_subpel_planes.hip
//
// Half-pel + quarter-pel refinement of all PUs (85 squares, or 209 with the rectangles) of a batch of superblocks against one list,
// gfx950 -- the default sub-pel path (search areas whose planes fit the LDS; me_subpel.hip keeps the per-PU-tile kernels for the rest).
// Replaces InterpolateSearchRegionAVC (Source/Lib/Codec/EbMotionEstimation.c:1707-1835), HalfPelSearch_LCU / PU_HalfPelRefinement
// (:2246-2786 / :1842-2240) and QuarterPelSearch_LCU / SetQuarterPelRefinementInputsOnTheFly / PU_QuarterPelRefinementOnTheFly /
// CombinedAveragingSSD (:3337-4114 / :3246-3331 / :2824-3239 / :2792-2817), SSD_SEARCH metric, every PU size refined.
//
// Round 1 interpolated b / h / j tiles per PU: every one of the 14 shape classes re-interpolated the whole superblock area (14x in the
// 209-PU mode) and the small per-group tiles collided on the LDS banks.  Here, like the reference, the three half-pel planes are
// interpolated ONCE per (superblock, list) -- but only over the bounding box of the positions the PUs' full-pel vectors can touch, and
// into LDS, never memory.  Then every PU is cut into 8x8 cells: a group of 8 lanes owns a chunk of 4 cells (lane = one 8-pixel row of a
// cell), reads its candidate rows as three aligned dwords + v_alignbyte, and sums the nine half-pel candidates (wrapped SSD + SAD) or the
// three quarter-pel candidates (true SSD + SAD) for its cells; the groups of one PU are adjacent lanes of ONE wave and combine with DPP /
// bpermute adds, so a wave runs half-pel -> decision -> quarter-pel -> decision -> write-back for its PUs without any workgroup barrier
// (only the 64x64 PU spans two waves: one LDS hand-shake).  Every class is 64 cells = 16 chunks = 2 waves, so the load is even.
//
// Arithmetic is the round-1 code (me_subpel_common.h): {-2,18,18,-2} + 16 >> 5 with clip, j from the ROUNDED b (quirk 8), 8-bit wrapped
// SSD in the half-pel stage (quirk 4) over 8 rows only for 8-wide PUs (quirk 11), true SSD in the quarter-pel stage, the 64x64 PU
// quarter-pel refined on a 32x32 block (quirk 5), L,R,T,B,TL,TR,BR,BL order with strict '<'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace svthip {

namespace {

#include "me_subpel_common.h"

// planes in LDS, search-region coordinates (x, y): integer sample A(x, y) = ref[ref_off + y * stride + x]
//   A : x in [-3, sw + 65], y in [-3, sh + 65]        b : x in [-1, sw + 64], y in [-3, sh + 65]
//   h, j : x in [-1, sw + 64], y in [-1, sh + 64]
// All four planes have the SAME geometry -- sample (x, y) at byte (y + 3) * pitch + x + 3 of its plane, planes D bytes apart -- so the
// address of sample (x, y) of plane p is base + p * D + y * pitch + x: one multiply-add instead of a per-plane choice of origin and pitch
// (b / h / j carry two unused columns and rows; 64 x 64 areas: 74.7 KB instead of 71 KB, still two workgroups per CU).
// Row pitch: an ODD number of dwords.  A lane reads row r of a cell; the 4 lane groups of a 32-lane LDS pass are cells stacked vertically
// (rows 8g + r) or side by side (+8 dwords), so with an odd pitch the 32 lanes of a pass fall on 32 different banks.  (A pitch of 4
// dwords mod 32 was tried: rows 8 apart alias and the groups of one PU collide 4-way, SQ_LDS_BANK_CONFLICT doubled.)
__host__ __device__ constexpr int subpel_plane_pitch(int cols) { return 4 * (((cols + 3) / 4) | 1); }

struct Planes {
    lds_u8* A;      // first byte of plane A = sample (-3, -3); b, h, j follow at D, 2 D, 3 D
    uint32_t base;  // LDS byte address of plane A's sample (0, 0)
    int P;          // row pitch in bytes (a multiple of 4)
    int D;          // plane distance in bytes (a multiple of 16: every plane has the same dword phase)
};

// The half-pel grid in QUARTER-pel units: a sample at (sx, sy), both even, lives in plane (sx & 2 ? 1 : 0) | (sy & 2) (A: both integer,
// b: x half, h: y half, j: both) at the index rounded UP to the next integer position (b[x] is the half-pel sample at x - 1/2).  Its LDS
// address is Planes::base + col_term(sx) + row_term(sy): the plane choice splits into an x part (D) and a y part (2 D).
__device__ __forceinline__ int col_term(const Planes& P, int sx)
{
    const int t = sx & 2;
    return ((sx + t) >> 2) + (t >> 1) * P.D;
}
__device__ __forceinline__ int row_term(const Planes& P, int sy)
{
    const int t = sy & 2;
    return ((sy + t) >> 2) * P.P + t * P.D;
}

// 8 bytes starting `s` bytes into the aligned dword at LDS address `qa`: three aligned dwords + two v_alignbyte
__device__ __forceinline__ void rd8(uint32_t qa, uint32_t s, uint32_t& lo, uint32_t& hi)
{
    const lds_u32* q = reinterpret_cast<const lds_u32*>((uintptr_t)qa);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    lo = __builtin_amdgcn_alignbyte(d1, d0, s);
    hi = __builtin_amdgcn_alignbyte(d2, d1, s);
}
// the same and the 8 bytes one byte further on, from the same three dwords (byte 8 of the second window is still in the third dword)
__device__ __forceinline__ void rd8x2(uint32_t qa, uint32_t s, uint32_t& lo0, uint32_t& hi0, uint32_t& lo1, uint32_t& hi1)
{
    const lds_u32* q = reinterpret_cast<const lds_u32*>((uintptr_t)qa);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    lo0 = __builtin_amdgcn_alignbyte(d1, d0, s);
    hi0 = __builtin_amdgcn_alignbyte(d2, d1, s);
    lo1 = __builtin_amdgcn_alignbyte(hi0, lo0, 1);
    hi1 = __builtin_amdgcn_alignbyte(d2 >> (8 * s), hi0, 1);
}

// class: base raster PU index, PUs, width, height (PUs of a class in raster order; kPu gives position and ME-buffer index)
struct ClassInfo {
    uint8_t base, count, w, h;
};
__device__ constexpr ClassInfo kClass[14] = {{0, 1, 64, 64},   {1, 4, 32, 32},   {5, 16, 16, 16},  {21, 64, 8, 8},    {85, 2, 64, 32},
                                             {87, 8, 32, 16},  {95, 32, 16, 8},  {127, 2, 32, 64}, {129, 8, 16, 32},  {137, 32, 8, 16},
                                             {169, 16, 32, 8}, {185, 16, 8, 32}, {201, 4, 64, 16}, {205, 4, 16, 64}};

// geometry by ME-buffer index: px | py << 8 | w << 16 | h << 24
struct MeGeom {
    uint32_t v[209];
};
constexpr MeGeom make_me_geom()
{
    MeGeom g{};
    for (int c = 0; c < 14; c++)
        for (int p = 0; p < kClass[c].count; p++) {
            const int pu = kClass[c].base + p;
            g.v[kPu.me[pu]] = (uint32_t)kPu.px[pu] | ((uint32_t)kPu.py[pu] << 8) | ((uint32_t)kClass[c].w << 16) | ((uint32_t)kClass[c].h << 24);
        }
    return g;
}
__device__ constexpr MeGeom kMeGeom = make_me_geom();

// raster PU index -> ME-buffer index | px << 8 | py << 16 (one load per PU instead of three)
struct PuPacked {
    uint32_t v[209];
};
constexpr PuPacked make_pu_packed()
{
    PuPacked t{};
    for (int i = 0; i < 209; i++) t.v[i] = (uint32_t)kPu.me[i] | ((uint32_t)kPu.px[i] << 8) | ((uint32_t)kPu.py[i] << 16);
    return t;
}
__device__ constexpr PuPacked kPuPacked = make_pu_packed();

// Quarter-pel positions (bits L, R, T, B, TL, TR, BR, BL = 0..7) tried next to the winning half-pel direction (:2859-2881; a best vector
// that sits on a half-pel position uses the mirrored set), one byte per direction in the order the reference tests for the first match:
// L, R, T, B, TL, TR, BL, BR (:2209-2238) -- the rank that the keyed minimum in refine_class_half delivers.
constexpr uint64_t make_quarter_valid(bool mirrored)
{
    const int dir_of_rank[8] = {DIR_L, DIR_R, DIR_T, DIR_B, DIR_TL, DIR_TR, DIR_BL, DIR_BR};
    uint64_t tab = 0;
    for (int rank = 0; rank < 8; rank++) {
        const int d = dir_of_rank[rank];
        bool v[8] = {};
        if (mirrored) {
            v[4] = (d == DIR_R || d == DIR_BR || d == DIR_B);  v[2] = (d == DIR_BR || d == DIR_B || d == DIR_BL);
            v[5] = (d == DIR_B || d == DIR_BL || d == DIR_L);  v[1] = (d == DIR_BL || d == DIR_L || d == DIR_TL);
            v[6] = (d == DIR_L || d == DIR_TL || d == DIR_T);  v[3] = (d == DIR_TL || d == DIR_T || d == DIR_TR);
            v[7] = (d == DIR_T || d == DIR_TR || d == DIR_R);  v[0] = (d == DIR_TR || d == DIR_R || d == DIR_BR);
        } else {
            v[4] = (d == DIR_L || d == DIR_TL || d == DIR_T);  v[2] = (d == DIR_TL || d == DIR_T || d == DIR_TR);
            v[5] = (d == DIR_T || d == DIR_TR || d == DIR_R);  v[1] = (d == DIR_TR || d == DIR_R || d == DIR_BR);
            v[6] = (d == DIR_R || d == DIR_BR || d == DIR_B);  v[3] = (d == DIR_BR || d == DIR_B || d == DIR_BL);
            v[7] = (d == DIR_B || d == DIR_BL || d == DIR_L);  v[0] = (d == DIR_BL || d == DIR_L || d == DIR_TL);
        }
        uint64_t m = 0;
        for (int k = 0; k < 8; k++) m |= v[k] ? (1ull << k) : 0ull;
        tab |= m << (8 * rank);
    }
    return tab;
}
constexpr uint64_t kQuarterValid0 = make_quarter_valid(false), kQuarterValid1 = make_quarter_valid(true);

struct Ctx {
    const uint8_t* src;   // SB top-left in the source plane
    uint32_t src_stride;
    Planes P;
    int xo, yo;
    uint32_t *sad_io, *mv_io;  // this SB's [n_pu] arrays
    const lds_u32* in_tab;     // LDS copy of what a PU's refinement starts from: [0,209) PU table (ME-buffer index | px << 8 | py << 16 in class
                               // order), [256, 256 + n_pu) full-pel SADs, [512, 512 + n_pu) full-pel vectors (ME-buffer order)
    uint32_t* pred;            // this SB's prediction slots ([slots][1024 dwords], slots in the class order above), or null
    lds_u32* shake;            // hand-shake area of the 64x64 PU (two waves): 17 half-pel sums, 6 quarter-pel sums, 2 arrival counters
    int lane;
    int method;                // MeContext_t::fractionalSearchMethod: 0 SUB_SAD_SEARCH, 1 FULL_SAD_SEARCH, 2 SSD_SEARCH (wave-uniform)
};

// All PUs of one class, 16 chunks of 4 cells, chunks [8 * half, 8 * half + 8) on this wave: lane group g = lane >> 3 owns chunk
// 8 * half + g, lane & 7 = the row inside a cell.  The 4 cells of a chunk sit at compile-time offsets from the chunk's first cell
// (CW >= 4: a row of four; CW = 2: 2 x 2; CW = 1: a column), so every LDS address is `row base + immediate (+ k * 8 pitches)` and the
// byte shift of an unaligned row is the same for all cells of a PU (cell offsets are multiples of 8 bytes and of 8 pitches).
// SADM: one of the SAD search methods (c.method 0 / 1) instead of SSD_SEARCH -- a template parameter so that the SSD instantiation, the
// one MotionEstimateLcu uses, carries none of the other methods' code (as a run-time branch it cost 136 bytes of scratch and 3-5 %).
template <int W, int H, int CLS, bool SADM>
__device__ void refine_class_half(const Ctx& c, int half, bool refine, int u_first, int u_count)
{
    constexpr int CW = W / 8, CPP = CW * (H / 8);                 // cells per PU
    constexpr int UNITS = CPP >= 4 ? 1 : 4 / CPP;                  // PUs per chunk
    constexpr int CPU = 4 / UNITS;                                 // cells per PU handled by this chunk
    constexpr int GPP = CPP >= 4 ? (CPP / 4 > 8 ? 8 : CPP / 4) : 1;  // lane groups of this wave that share a PU
    constexpr int LPP = 8 * GPP;
    constexpr bool TWO_WAVES = CPP > 32;                           // 64x64: the other half of the PU is on the partner wave
    const int g = c.lane >> 3, r = c.lane & 7, chunk = 8 * half + g;
    const int p8 = 8 * c.P.P;
    // u_first / u_count: the slice of the chunk's UNITS PUs this task covers (classes with one PU per chunk: 0 / 1)
#pragma unroll 1
    for (int u = (UNITS > 1 ? u_first : 0); u < (UNITS > 1 ? u_first + u_count : 1); u++) {
        constexpr int CQ = CPP >= 4 ? CPP / 4 : 1;  // chunks per PU
        const int pu_in_class = CPP >= 4 ? chunk / CQ : chunk * UNITS + u;
        const int cell0 = CPP >= 4 ? (chunk % CQ) * 4 : 0;
        const int cx0 = cell0 % CW, cy0 = cell0 / CW;
        // from LDS (staged by the workgroup before the plane phases): as a table look-up in memory followed by two loads that depend on it,
        // every unit of every wave began with two dependent memory round trips
        const uint32_t ppk = c.in_tab[kClass[CLS].base + pu_in_class];
        const int me = (int)(ppk & 255u), px = (int)((ppk >> 8) & 255u), py = (int)(ppk >> 16);
        uint32_t best_sad = c.in_tab[256 + me], best_mv = c.in_tab[512 + me], best_ssd = 0;
        const uint8_t* srow = c.src + (size_t)(py + 8 * cy0 + r) * c.src_stride + px + 8 * cx0;  // row r of the chunk's first cell
        const size_t s8 = 8 * (size_t)c.src_stride;
        if (refine) {
            const int x_mv = (int)(int16_t)(best_mv & 0xffffu), y_mv = (int)(int16_t)(best_mv >> 16);
            const int bx = (x_mv >> 2) - c.xo + px + 8 * cx0, by = (y_mv >> 2) - c.yo + py + 8 * cy0 + r;
            // ---- half-pel: full-pel SSD + 8 candidates, wrapped SSD and SAD ----
            uint32_t ssd[9], sad[8];
#pragma unroll
            for (int k = 0; k < 9; k++) ssd[k] = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) sad[k] = 0;
            // the four planes share their geometry: one address, one byte shift, the planes D apart
            const uint32_t aA = c.P.base + (uint32_t)(by * c.P.P + bx);
            const uint32_t sA = aA & 3u, qA = aA & ~3u, qB = qA + (uint32_t)c.P.D, qH = qB + (uint32_t)c.P.D, qJ = qH + (uint32_t)c.P.D;
#pragma unroll
            for (int ci = 0; ci < CPU; ci++) {
                constexpr int dummy = 0;
                (void)dummy;
                const int dx = CW >= 4 ? ci : (CW == 2 ? (ci & 1) : 0), dy = CW >= 4 ? 0 : (CW == 2 ? (ci >> 1) : ci);
                const uint2 sv = *reinterpret_cast<const uint2*>(srow + dy * s8 + 8 * dx);
                const uint32_t oa = 8 * dx + dy * p8;
                uint32_t cand[9][2];  // L, R, T, B, TL, TR, BR, BL, full
                rd8(qA + oa, sA, cand[8][0], cand[8][1]);
                rd8x2(qB + oa, sA, cand[0][0], cand[0][1], cand[1][0], cand[1][1]);
                rd8(qH + oa, sA, cand[2][0], cand[2][1]);
                rd8(qH + oa + c.P.P, sA, cand[3][0], cand[3][1]);
                rd8x2(qJ + oa, sA, cand[4][0], cand[4][1], cand[5][0], cand[5][1]);
                rd8x2(qJ + oa + c.P.P, sA, cand[7][0], cand[7][1], cand[6][0], cand[6][1]);
                // the SSD leaf is keyed by width: 8-wide PUs are compared on their top 8 rows only (quirk 11); SAD covers all rows
                const bool in_ssd = (W != 8) || (cy0 + dy == 0);
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    if (in_ssd) ssd[k] = wssd8(sv.x, sv.y, cand[k][0], cand[k][1], ssd[k]);
                    if (k < 8) sad[k] = __builtin_amdgcn_sad_u8(sv.y, cand[k][1], __builtin_amdgcn_sad_u8(sv.x, cand[k][0], sad[k]));
                }
            }
            // The SAD search methods compare (and store) SADs instead: NxMSadKernel over every row, or over every second row (this
            // lane's row r is even) doubled (:1930-1932).  Same reductions, same decision code below; only the metric differs.
            constexpr bool sad_method = SADM;
            if (sad_method) {
#pragma unroll
                for (int k = 0; k < 8; k++) ssd[k] = (c.method == 0 && (r & 1)) ? 0u : sad[k];
            }
#pragma unroll
            for (int k = 0; k < 9; k++) ssd[k] = gsum<LPP>(ssd[k]);
#pragma unroll
            for (int k = 0; k < 8; k++) sad[k] = gsum<LPP>(sad[k]);
            if (TWO_WAVES) {  // add the partner wave's half of the PU
                if (c.lane == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) __hip_atomic_fetch_add(c.shake + k, ssd[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (int k = 0; k < 8; k++) __hip_atomic_fetch_add(c.shake + 9 + k, sad[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c.shake + 23, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                while (__hip_atomic_load(c.shake + 23, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < 9; k++) ssd[k] = __hip_atomic_load(c.shake + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int k = 0; k < 8; k++) sad[k] = __hip_atomic_load(c.shake + 9 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const int mvdx[8] = {-2, 2, 0, 0, -2, 2, 2, -2}, mvdy[8] = {0, 0, -2, 2, -2, -2, 2, 2};
            if (sad_method) {
                // the candidate's distortion is compared with, and stored as, the best SAD (:1948-1953); there is no SSD state
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (c.method == 0) ssd[k] <<= 1;
                    if (ssd[k] < best_sad) {
                        best_sad = ssd[k];
                        best_mv = ((uint32_t)(uint16_t)(y_mv + mvdy[k]) << 16) | (uint32_t)(uint16_t)(x_mv + mvdx[k]);
                    }
                }
            } else {
                best_ssd = ssd[8];  // SSD of the best full-pel candidate (:1912)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (ssd[k] < best_ssd) {  // strict '<' (:1942)
                        best_sad = sad[k];
                        best_mv = ((uint32_t)(uint16_t)(y_mv + mvdy[k]) << 16) | (uint32_t)(uint16_t)(x_mv + mvdx[k]);
                        best_ssd = ssd[k];
                    }
                }
            }
            // Winning direction = the first minimum in the order L, R, T, B, TL, TR, BL, BR (:2209-2238): one keyed minimum,
            // (distortion << 3) | rank -- a distortion is < 2^27 (64x64: 4096 x 128^2 = 2^26) -- instead of a minimum and a chain of
            // compares, which the compiler turned into eight nested exec-mask regions.
            uint32_t rank;
            {
                const uint32_t k01 = min((ssd[0] << 3) | 0u, (ssd[1] << 3) | 1u), k23 = min((ssd[2] << 3) | 2u, (ssd[3] << 3) | 3u);
                const uint32_t k45 = min((ssd[4] << 3) | 4u, (ssd[5] << 3) | 5u), k67 = min((ssd[7] << 3) | 6u, (ssd[6] << 3) | 7u);
                rank = min(min(k01, k23), min(k45, k67)) & 7u;
            }

            // ---- quarter-pel: the three positions next to the winning half-pel direction, true SSD and SAD ----
            const int hx = (int)(int16_t)(best_mv & 0xffffu), hy = (int)(int16_t)(best_mv >> 16);
            const int method = (hy & 2) + ((hx & 2) >> 1);
            // valid positions (L, R, T, B, TL, TR, BR, BL = bits 0..7) of that direction, :2859-2881; `method != 0` uses the mirrored set
            uint32_t vmask = (uint32_t)((method ? kQuarterValid1 : kQuarterValid0) >> (8u * rank)) & 0xffu;
            // Per candidate the two samples whose rounded mean is the quarter-pel sample (SetQuarterPelRefinementInputsOnTheFly,
            // :3271-3323), resolved ONCE for row r of the chunk's first cell.  The reference's 4 x 8 x 2 table is the H.264 rule and is
            // computed instead of looked up (a dependent table read per candidate sat in the middle of every PU): with the best half-pel
            // point (hx, hy) and the candidate (hx + qdx, hy + qdy), the pair is {(hx, hy), (hx + 2 qdx, hy + 2 qdy)} -- the two
            // neighbours along the odd axis, or the diagonal -- except that a diagonal candidate of a point with hx, hy both integer or
            // both half takes the other diagonal, {(hx, hy + 2 qdy), (hx + 2 qdx, hy)}.  (tests/test_subpel_tables.py compares the rule
            // with the reference's table entry by entry.)
            const int HX = 4 * (px + 8 * cx0 - c.xo) + hx, HY = 4 * (py + 8 * cy0 + r - c.yo) + hy;  // this lane's row, quarter units
            const int cx_0 = col_term(c.P, HX), cy_0 = row_term(c.P, HY);
            const bool same_phase = ((hx ^ hy) & 2) == 0;
            int qk[3];
            uint32_t qa1[3], qs1[3], qa2[3], qs2[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {  // ascending position index = the reference's evaluation order
                qk[t] = __builtin_ctz(vmask);
                vmask &= vmask - 1;
                // L, R, T, B, TL, TR, BR, BL: dx = {-1, 1, 0, 0, -1, 1, 1, -1}, dy = {0, 0, -1, 1, -1, -1, 1, 1}, two bits each (+1)
                const int qdx = (int)((0x2858u >> (2 * qk[t])) & 3u) - 1, qdy = (int)((0xA085u >> (2 * qk[t])) & 3u) - 1;
                const int cx_n = col_term(c.P, HX + 2 * qdx), cy_n = row_term(c.P, HY + 2 * qdy);
                const bool cross = same_phase && qdx != 0 && qdy != 0;
                const uint32_t a1 = c.P.base + (uint32_t)(cx_0 + (cross ? cy_n : cy_0));
                const uint32_t a2 = c.P.base + (uint32_t)(cx_n + (cross ? cy_0 : cy_n));
                qa1[t] = a1 & ~3u; qs1[t] = a1 & 3u;
                qa2[t] = a2 & ~3u; qs2[t] = a2 & 3u;
            }
            uint32_t qssd[3] = {0, 0, 0}, qsad[3] = {0, 0, 0}, qsv[3] = {0, 0, 0}, qsrc2 = 0;
#pragma unroll
            for (int ci = 0; ci < CPU; ci++) {
                const int dx = CW >= 4 ? ci : (CW == 2 ? (ci & 1) : 0), dy = CW >= 4 ? 0 : (CW == 2 ? (ci >> 1) : ci);
                // the 64x64 PU is quarter-pel refined on the 32x32 block at the SB origin (:3395-3409)
                if (W == 64 && H == 64 && (cx0 + dx >= 4 || cy0 + dy >= 4)) continue;
                const uint2 sv = *reinterpret_cast<const uint2*>(srow + dy * s8 + 8 * dx);
                qsrc2 = __builtin_amdgcn_udot4(sv.y, sv.y, __builtin_amdgcn_udot4(sv.x, sv.x, qsrc2, false), false);
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    uint32_t a0, a1, b0, b1;
                    rd8(qa1[t] + 8 * dx + dy * p8, qs1[t], a0, a1);
                    rd8(qa2[t] + 8 * dx + dy * p8, qs2[t], b0, b1);
                    const uint32_t v0 = avg_u8x4(a0, b0), v1 = avg_u8x4(a1, b1);
                    // CombinedAveragingSSD: true SSD (:2792-2817) = sum s^2 (shared by the three candidates) + sum v^2 - 2 sum s v
                    qssd[t] = __builtin_amdgcn_udot4(v1, v1, __builtin_amdgcn_udot4(v0, v0, qssd[t], false), false);
                    qsv[t] = __builtin_amdgcn_udot4(sv.y, v1, __builtin_amdgcn_udot4(sv.x, v0, qsv[t], false), false);
                    qsad[t] = __builtin_amdgcn_sad_u8(sv.y, v1, __builtin_amdgcn_sad_u8(sv.x, v0, qsad[t]));
                }
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                // SSD method: per lane a sum of squares, never negative; SAD methods: NxMSadAveragingKernel on all / every second row (:2915-2917)
                const uint32_t lane_metric = sad_method ? ((c.method == 0 && (r & 1)) ? 0u : qsad[t]) : qsrc2 + qssd[t] - 2u * qsv[t];
                qssd[t] = gsum<LPP>(lane_metric);
                qsad[t] = gsum<LPP>(qsad[t]);
            }
            if (TWO_WAVES) {
                if (c.lane == 0) {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        __hip_atomic_fetch_add(c.shake + 17 + t, qssd[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c.shake + 20 + t, qsad[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __hip_atomic_fetch_add(c.shake + 24, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                while (__hip_atomic_load(c.shake + 24, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    qssd[t] = __hip_atomic_load(c.shake + 17 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    qsad[t] = __hip_atomic_load(c.shake + 20 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                // L, R, T, B, TL, TR, BR, BL: dx = {-1, 1, 0, 0, -1, 1, 1, -1}, dy = {0, 0, -1, 1, -1, -1, 1, 1}, two bits each (+1)
                const int qdx = (int)((0x2858u >> (2 * qk[t])) & 3u) - 1, qdy = (int)((0xA085u >> (2 * qk[t])) & 3u) - 1;
                if (sad_method) {
                    const uint32_t dist = c.method == 0 ? qssd[t] << 1 : qssd[t];
                    if (dist < best_sad) {  // :2934-2939
                        best_sad = dist;
                        best_mv = ((uint32_t)(uint16_t)(hy + qdy) << 16) | (uint32_t)(uint16_t)(hx + qdx);
                    }
                } else if (qssd[t] < best_ssd) {
                    best_sad = qsad[t];
                    best_mv = ((uint32_t)(uint16_t)(hy + qdy) << 16) | (uint32_t)(uint16_t)(hx + qdx);
                    best_ssd = qssd[t];
                }
            }
            if ((c.lane & (LPP - 1)) == 0 && !(TWO_WAVES && half)) {
                c.sad_io[me] = best_sad;
                c.mv_io[me] = best_mv;
            }
        }
        if (c.pred) {
            // the prediction block at the final MV as BiPredictionCompensation builds it (kBiFrac), for the stored-prediction bi-pred
            const int fx = (int)(int16_t)(best_mv & 0xffffu), fy = (int)(int16_t)(best_mv >> 16);
            // the same rule for any final vector: an even coordinate keeps its column / row, an odd one takes its two neighbours, the one
            // on the half-pel grid (== 2 mod 4) and the one on the integer grid; a diagonal pairs (half, integer) with (integer, half)
            const int FX = 4 * (px + 8 * cx0 - c.xo) + fx, FY = 4 * (py + 8 * cy0 + r - c.yo) + fy;
            const int ux = 1 - (FX & 2), uy = 1 - (FY & 2);
            const bool ox = FX & 1, oy = FY & 1;
            const int x1 = ox ? FX + ux : FX, x2 = ox ? FX - ux : FX;
            const int y1 = oy ? (ox ? FY - uy : FY + uy) : FY, y2 = oy ? (ox ? FY + uy : FY - uy) : FY;
            const uint32_t a1 = c.P.base + (uint32_t)(col_term(c.P, x1) + row_term(c.P, y1));
            const uint32_t a2 = c.P.base + (uint32_t)(col_term(c.P, x2) + row_term(c.P, y2));
            uint32_t* out = c.pred + CLS * 1024 + pu_in_class * (W * H / 4) + ((8 * cy0 + r) * W + 8 * cx0) / 4;
#pragma unroll
            for (int ci = 0; ci < CPU; ci++) {
                const int dx = CW >= 4 ? ci : (CW == 2 ? (ci & 1) : 0), dy = CW >= 4 ? 0 : (CW == 2 ? (ci >> 1) : ci);
                uint32_t a0, a1h, b0, b1;
                rd8((a1 & ~3u) + 8 * dx + dy * p8, a1 & 3u, a0, a1h);
                rd8((a2 & ~3u) + 8 * dx + dy * p8, a2 & 3u, b0, b1);
                uint2 v;
                v.x = avg_u8x4(a0, b0);
                v.y = avg_u8x4(a1h, b1);
                *reinterpret_cast<uint2*>(out + (8 * dy * W + 8 * dx) / 4) = v;
            }
        }
    }
}

template <int CLS>
__device__ __forceinline__ void run_class(const Ctx& c, int half, bool refine, int u_first, int u_count)
{
    if (c.method == 2) refine_class_half<kClass[CLS].w, kClass[CLS].h, CLS, false>(c, half, refine, u_first, u_count);
    else refine_class_half<kClass[CLS].w, kClass[CLS].h, CLS, true>(c, half, refine, u_first, u_count);
}

// Which wave refines what.  A task = (class, half, slice of the chunk's PUs) = 8 chunks of 4 cells, i.e. the same pixels whatever the class --
// but not the same time: a chunk of a class with small PUs runs the per-PU part (vector / table reads, 17 + 6 lane-group sums, decisions,
// quarter-pel set-up) once per PU, and per-wave time stamps (tools/subpel_stamps_probe.py) showed the two 8x8 waves of the 85-PU mode at
// 23.8 k ticks against 13 - 15 k for the other six, with the workgroup (and its 71 KB of LDS) waiting for them.  So the classes with 4 / 2
// PUs per chunk (8x8; 16x8, 8x16) are cut into per-PU slices and dealt out by measured cost.  Entry: class | half << 4 | first << 5 |
// count << 7, 0xffff ends a wave's list.  The two halves of the 64x64 PU (which hand-shake through the LDS) lead the lists of two waves.
// (dwords in the constant address space: the index is wave-uniform, so a wave's next task is a scalar load, not a vector load + readfirstlane)
#define T(cls, half, first, count) (uint32_t)((cls) | ((half) << 4) | ((first) << 5) | ((count) << 7))
#define TEND (uint32_t)0xffff
__constant__ const uint32_t kTasks85[8][4] = {
    {T(0, 0, 0, 1), TEND, TEND, TEND},          {T(0, 1, 0, 1), TEND, TEND, TEND},
    {T(1, 0, 0, 1), T(3, 0, 0, 1), TEND, TEND}, {T(1, 1, 0, 1), T(3, 1, 0, 1), TEND, TEND},
    {T(2, 0, 0, 1), TEND, TEND, TEND},          {T(2, 1, 0, 1), TEND, TEND, TEND},
    {T(3, 0, 1, 3), TEND, TEND, TEND},          {T(3, 1, 1, 3), TEND, TEND, TEND}};
__constant__ const uint32_t kTasks209[7][6] = {
    {T(0, 0, 0, 1), T(3, 1, 0, 3), T(7, 0, 0, 1), T(10, 1, 0, 1), TEND, TEND},
    {T(0, 1, 0, 1), T(4, 0, 0, 1), T(7, 1, 0, 1), T(11, 0, 0, 1), T(3, 1, 3, 1), TEND},
    {T(1, 0, 0, 1), T(4, 1, 0, 1), T(8, 0, 0, 1), T(11, 1, 0, 1), T(3, 0, 2, 2), TEND},
    {T(1, 1, 0, 1), T(5, 0, 0, 1), T(8, 1, 0, 1), T(12, 0, 0, 1), T(9, 0, 1, 1), TEND},
    {T(2, 0, 0, 1), T(5, 1, 0, 1), T(9, 0, 0, 1), T(12, 1, 0, 1), TEND, TEND},
    {T(2, 1, 0, 1), T(6, 0, 0, 2), T(9, 1, 0, 2), T(13, 0, 0, 1), TEND, TEND},
    {T(3, 0, 0, 2), T(6, 1, 0, 2), T(10, 0, 0, 1), T(13, 1, 0, 1), TEND, TEND}};
#undef T
#undef TEND

}  // namespace

// Optional per-phase time stamps (tools/subpel_stamps_probe.py builds a variant with -DSVTHIP_SUBPEL_STAMPS): s_memtime of wave 0's
// lane 0 at the phase boundaries, [superblock][8].
#ifdef SVTHIP_SUBPEL_STAMPS
__device__ unsigned long long g_subpel_stamps[8192 * 8];
__device__ unsigned long long g_subpel_wave_end[8192 * 8];  // every wave's own end of the PU phase
#define SUBPEL_STAMP(i)                                                                                          \
    do {                                                                                                         \
        if (tid == 0 && sb < 8192u) g_subpel_stamps[(size_t)sb * 8 + (i)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#else
#define SUBPEL_STAMP(i) do { } while (0)
#endif

// One workgroup per (SB, list): 512 threads (8 wave tasks) for the 85 squares, 448 threads (7 waves x 4 tasks) for all 209 PUs.
// io arrays [n_sb][n_pu] in ME-buffer order, refined in place; pred_out (optional) [n_sb][slots][1024 dwords].
__global__ void __launch_bounds__(512, 4) subpel_planes_kernel(const uint8_t* __restrict__ src_plane, uint32_t src_stride,
                                                            const uint8_t* __restrict__ ref_plane, uint32_t ref_stride,
                                                            const int32_t* __restrict__ desc, uint32_t n_sb, int disable_8x8, int n_pu,
                                                            uint32_t* __restrict__ io_sad, uint32_t* __restrict__ io_mv, uint32_t* __restrict__ pred_out,
                                                            int method)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
#ifdef SVTHIP_SUBPEL_PERSISTENT
    for (uint32_t vb = blockIdx.x; vb < xcd_grid(n_sb); vb += gridDim.x) {
    const uint32_t sb = xcd_item(vb, n_sb);
    if (sb >= n_sb) continue;
    __syncthreads();  // the previous superblock's readers are done with the planes and the hand-shake words
#else
    const uint32_t sb = xcd_item(blockIdx.x, n_sb);  // raster neighbours share an XCD's L2 (me_kernels.h)
    if (sb >= n_sb) return;
#endif
    const int32_t* d = desc + 6 * sb;
    const int src_off = d[0], ref_off = d[1], xo = d[2], yo = d[3], sw = d[4], sh = d[5];
    uint32_t* sad_io = io_sad + (size_t)n_pu * sb;
    uint32_t* mv_io = io_mv + (size_t)n_pu * sb;

    // LDS: [4 spare words + hand-shake 25 words -> 128 B][A][b][h][j]
    lds_u32* ctl = (lds_u32*)smem;
    Planes P;
    P.P = subpel_plane_pitch(sw + 69);
    const int rows_a = sh + 69;
    P.A = (lds_u8*)smem + 128;
    P.D = (P.P * rows_a + 15) & ~15;
    P.base = (uint32_t)reinterpret_cast<uintptr_t>(P.A) + (uint32_t)(3 * P.P + 3);
    lds_u8* const PB = P.A + P.D;
    lds_u8* const PH = PB + P.D;
    lds_u8* const PJ = PH + P.D;
    // behind the planes (and their 32 bytes of read slack): the PUs' starting values, visible after the barriers of the plane phases
    lds_u32* const in_tab = (lds_u32*)(PJ + P.D + 32);
    for (int i = tid; i < 209; i += nthr) in_tab[i] = kPuPacked.v[i];
    for (int i = tid; i < n_pu; i += nthr) {
        in_tab[256 + i] = sad_io[i];
        in_tab[512 + i] = mv_io[i];
    }

    // The plane phases are a chain of short, dependent steps (vectors -> box -> window -> b, h -> j, four barriers) that the whole workgroup
    // waits for, while the PU phase of the CU's other workgroup is a long stream of independent vector work: run the chain at raised wave
    // priority so that its few instructions issue ahead of that stream instead of queueing behind it (SVTHIP_SUBPEL_NO_PRIO: A/B only).
#ifndef SVTHIP_SUBPEL_NO_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    SUBPEL_STAMP(0);
    // ---- bounding box of the samples the PUs can touch: x in [bx - 1, bx + W + 1], y in [by - 1, by + H + 1] ----
    // Every wave computes it for itself (two independent coalesced loads per lane, a DPP / bpermute min-max over the wave): no LDS, no
    // workgroup barrier, one memory latency.  (First version: LDS atomics by the first n_pu threads between two barriers with a
    // dependent table -> vector chain: 84 us of a 420 us launch.)
    if (tid < 32) ctl[tid] = 0u;  // hand-shake area of the 64x64 PU; ordered before its use by the barriers below
    int X0 = 0x7fffffff, X1 = -0x7fffffff, Y0 = 0x7fffffff, Y1 = -0x7fffffff;
    for (int m = lane; m < n_pu; m += 64) {
        if (disable_8x8 && m >= 21 && m < 85 && !pred_out) continue;  // 8x8 PUs keep their full-pel result and read no plane
        const uint32_t mv = mv_io[m], gm = kMeGeom.v[m];               // ME-buffer index -> px | py << 8 | w << 16 | h << 24
        const int bx = ((int)(int16_t)(mv & 0xffffu) >> 2) - xo + (int)(gm & 255u), by = ((int)(int16_t)(mv >> 16) >> 2) - yo + (int)((gm >> 8) & 255u);
        X0 = min(X0, bx - 1);
        X1 = max(X1, bx + (int)((gm >> 16) & 255u) + 1);
        Y0 = min(Y0, by - 1);
        Y1 = max(Y1, by + (int)(gm >> 24) + 1);
    }
    {
        // the four extremes as two packed pairs of 16-bit minima (|coordinate| < 2^9): (X0, -X1) and (Y0, -Y1), one v_pk_min_i16 per pair and
        // step -- 12 cross-lane moves instead of 24 on the serial path every wave walks before anything else can start
        typedef short v2s16 __attribute__((ext_vector_type(2)));
        // (a lane without a PU still holds the initial +-0x7fffffff: clamped to 32000, it changes no minimum)
        v2s16 px = {(short)min(X0, 32000), (short)min(-X1, 32000)}, py = {(short)min(Y0, 32000), (short)min(-Y1, 32000)};
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            px = __builtin_elementwise_min(px, __builtin_bit_cast(v2s16, __shfl_xor(__builtin_bit_cast(int, px), sft)));
            py = __builtin_elementwise_min(py, __builtin_bit_cast(v2s16, __shfl_xor(__builtin_bit_cast(int, py), sft)));
        }
        X0 = px.x; X1 = -(int)px.y; Y0 = py.x; Y1 = -(int)py.y;
    }
    // clamp to what the planes hold (a vector outside its search area would be a caller error; never index outside the LDS)
    X0 = __builtin_amdgcn_readfirstlane(max(X0, -1));
    X1 = __builtin_amdgcn_readfirstlane(min(X1, sw + 64));
    Y0 = __builtin_amdgcn_readfirstlane(max(Y0, -1));
    Y1 = __builtin_amdgcn_readfirstlane(min(Y1, sh + 64));
    // dword columns of the planes to fill (plane column = x + 3: dword w holds x = 4 w - 3 .. 4 w) and rows.  Every phase walks ONE flat
    // list of (row, dword column) items, thread t taking items t, t + threads, ...: the typical box is ~72 samples = 18 dword columns wide,
    // and rows-to-waves with columns-to-lanes (round 2) kept 36 of 64 lanes busy and needed 5 + 5 + 5 wave passes where the flat lists need
    // 6 + 3.  Row of an item by a reciprocal multiply (exact: items < 2^15, columns <= 51, so item * (columns - 1) < 2^20).
    SUBPEL_STAMP(1);
    const int c0 = (X0 + 3) >> 2, c1 = (X1 + 3) >> 2, ncol = c1 - c0 + 1, p4 = P.P >> 2;
#ifdef SVTHIP_SUBPEL_EXPERIMENT_NO_PLANES  // timing experiments only (tools/build_variant.sh): results are wrong
    if (false) {
#else
    if (ncol > 0 && Y1 >= Y0) {
#endif
        // ---- A: rows Y0 - 2 .. Y1 + 1, dword columns c0 - 1 .. c1 + 1 (b's four samples of dword w read A bytes 4 w - 2 .. 4 w + 4) ----
        {
            const uint8_t* base = ref_plane + ref_off - 3;  // plane column 0 = x = -3
            const int ra0 = Y0 - 2, nra = Y1 - Y0 + 4;
            // the dword left of column 0 does not exist and the one right of the last column may not (only don't-care b samples read them)
            const int ca0 = max(c0 - 1, 0), ncol_a = min(c1 + 1, p4 - 1) - ca0 + 1;
            const uint32_t inv_a = ((1u << 20) + (uint32_t)ncol_a - 1u) / (uint32_t)ncol_a;
            const int n_a = nra * ncol_a;
            for (int i0 = tid; i0 < n_a; i0 += 4 * nthr) {
                // one dword per lane at the window's own byte alignment (global loads need no alignment on this target), four in flight
                struct __attribute__((packed, aligned(1))) u1 { uint32_t v; };
                uint32_t val[4];
                int at[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * nthr;
                    const int rr = (int)(((uint32_t)i * inv_a) >> 20), cc = ca0 + i - rr * ncol_a;
                    at[u] = (ra0 + rr + 3) * p4 + cc;
                    if (i < n_a)
                        val[u] = ((const __attribute__((address_space(1))) u1*)reinterpret_cast<uintptr_t>(base + (int64_t)(ra0 + rr) * ref_stride + 4 * cc))->v;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * nthr < n_a) reinterpret_cast<lds_u32*>(P.A)[at[u]] = val[u];
            }
        }
        __syncthreads();
        SUBPEL_STAMP(2);
        const uint32_t inv_c = ((1u << 20) + (uint32_t)ncol - 1u) / (uint32_t)ncol;
        // ---- b rows Y0 - 2 .. Y1 + 1, then h rows Y0 .. Y1, as one list ----
        {
            const int nrb = Y1 - Y0 + 4, nrh = Y1 - Y0 + 1, n_bh = (nrb + nrh) * ncol;
            for (int i = tid; i < n_bh; i += nthr) {
                const int rr = (int)(((uint32_t)i * inv_c) >> 20), cc = c0 + i - rr * ncol;
                if (rr < nrb) {
                    // b(x, y) from A(x - 2 .. x + 1, y), x = 4 cc - 3 .. 4 cc: A bytes 4 cc - 2 .. 4 cc + 4 of the row = the last two bytes of
                    // dword cc - 1, dword cc and the first byte of dword cc + 1 (column 0: the bytes before the row feed x = -3, -2 only)
                    const int y = Y0 - 2 + rr;
                    const lds_u32* q = reinterpret_cast<const lds_u32*>(P.A) + (y + 3) * p4 + cc;
                    const uint32_t em = q[-1], e0 = q[0], e1 = q[1];
                    reinterpret_cast<lds_u32*>(PB)[(y + 3) * p4 + cc] =
                        hfilt1(__builtin_amdgcn_alignbyte(e0, em, 2)) | (hfilt1(__builtin_amdgcn_alignbyte(e0, em, 3)) << 8) | (hfilt1(e0) << 16) |
                        (hfilt1(__builtin_amdgcn_alignbyte(e1, e0, 1)) << 24);
                } else {
                    // h(x, y) from A(x, y - 2 .. y + 1): the same dword column of four rows
                    const int y = Y0 + rr - nrb;
                    const lds_u32* q = reinterpret_cast<const lds_u32*>(P.A) + (y - 2 + 3) * p4 + cc;
                    reinterpret_cast<lds_u32*>(PH)[(y + 3) * p4 + cc] = vfilt4(q[0], q[p4], q[2 * p4], q[3 * p4]);
                }
            }
        }
        __syncthreads();
        SUBPEL_STAMP(3);
        // ---- j rows Y0 .. Y1 from the ROUNDED b rows y - 2 .. y + 1 ----
        {
            const int nrh = Y1 - Y0 + 1, n_j = nrh * ncol;
            for (int i = tid; i < n_j; i += nthr) {
                const int rr = (int)(((uint32_t)i * inv_c) >> 20), cc = c0 + i - rr * ncol;
                const int y = Y0 + rr;
                const lds_u32* q = reinterpret_cast<const lds_u32*>(PB) + (y - 2 + 3) * p4 + cc;
                reinterpret_cast<lds_u32*>(PJ)[(y + 3) * p4 + cc] = vfilt4(q[0], q[p4], q[2 * p4], q[3 * p4]);
            }
        }
    }
    __syncthreads();
#ifndef SVTHIP_SUBPEL_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    SUBPEL_STAMP(4);

    Ctx c;
    c.src = src_plane + src_off;
    c.src_stride = src_stride;
    c.P = P;
    c.xo = xo;
    c.yo = yo;
    c.sad_io = sad_io;
    c.mv_io = mv_io;
    c.in_tab = in_tab;
    c.pred = pred_out ? pred_out + (size_t)sb * (n_pu == 209 ? 14 : 4) * 1024 : nullptr;
    c.shake = ctl + 4;
    c.lane = lane;
    c.method = __builtin_amdgcn_readfirstlane(method);
#ifndef SVTHIP_SUBPEL_EXPERIMENT_NO_PU
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t* tasks = n_pu == 209 ? kTasks209[wave_u < 7 ? wave_u : 0] : kTasks85[wave_u & 7];
    const int max_tasks = (n_pu == 209 ? 6 : 4) * ((n_pu == 209 ? wave < 7 : wave < 8) ? 1 : 0);
#pragma unroll 1
    for (int k = 0; k < max_tasks; k++) {
        const int t = __builtin_amdgcn_readfirstlane((int)tasks[k]);
        if (t == 0xffff) break;
        const int cls = t & 15, half = (t >> 4) & 1, u_first = (t >> 5) & 3, u_count = (t >> 7) & 7;
        const bool refine = !(cls == 3 && disable_8x8);
        switch (cls) {
        case 0: run_class<0>(c, half, refine, u_first, u_count); break;
        case 1: run_class<1>(c, half, refine, u_first, u_count); break;
        case 2: run_class<2>(c, half, refine, u_first, u_count); break;
        case 3: run_class<3>(c, half, refine, u_first, u_count); break;
        case 4: run_class<4>(c, half, refine, u_first, u_count); break;
        case 5: run_class<5>(c, half, refine, u_first, u_count); break;
        case 6: run_class<6>(c, half, refine, u_first, u_count); break;
        case 7: run_class<7>(c, half, refine, u_first, u_count); break;
        case 8: run_class<8>(c, half, refine, u_first, u_count); break;
        case 9: run_class<9>(c, half, refine, u_first, u_count); break;
        case 10: run_class<10>(c, half, refine, u_first, u_count); break;
        case 11: run_class<11>(c, half, refine, u_first, u_count); break;
        case 12: run_class<12>(c, half, refine, u_first, u_count); break;
        default: run_class<13>(c, half, refine, u_first, u_count); break;
        }
    }
#endif
    SUBPEL_STAMP(5);
#ifdef SVTHIP_SUBPEL_STAMPS
    if (lane == 0 && sb < 8192u) g_subpel_wave_end[(size_t)sb * 8 + wave] = __builtin_amdgcn_s_memtime();
#endif
#ifdef SVTHIP_SUBPEL_PERSISTENT
    }
#endif
}

#ifdef SVTHIP_SUBPEL_STAMPS
extern "C" int svthip_debug_subpel_wave_end(void* host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_subpel_wave_end), bytes < sizeof(g_subpel_wave_end) ? bytes : sizeof(g_subpel_wave_end));
}
extern "C" int svthip_debug_subpel_stamps(void* host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_subpel_stamps), bytes < sizeof(g_subpel_stamps) ? bytes : sizeof(g_subpel_stamps));
}
#endif

uint32_t subpel_planes_grid(uint32_t n_sb)
{
#ifdef SVTHIP_SUBPEL_PERSISTENT
    return xcd_grid(n_sb) < 512u ? xcd_grid(n_sb) : 512u;
#else
    return xcd_grid(n_sb);
#endif
}

size_t subpel_planes_lds_bytes(uint32_t max_sw, uint32_t max_sh)
{
    const size_t pitch = subpel_plane_pitch((int)max_sw + 69), rows = max_sh + 69;
    return 128 + 4 * ((pitch * rows + 15) & ~(size_t)15) + 32 + 4 * (512 + 209 + 3);  // + slack (the 3-dword reads run up to 11 bytes past a sample) + the PUs' starting values
}

}  // namespace svthip
