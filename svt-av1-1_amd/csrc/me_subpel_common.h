// svt-av1-1_amd/csrc/me_subpel_common.h -- arithmetic helpers and tables shared by the sub-pel refinement kernels
// (me_subpel.hip: per-PU tiles, any search area; me_subpel_planes.hip: shared half-pel planes, the default path).
// Include inside `namespace svthip { namespace {`.
#pragma once

constexpr int kMargin = 4;  // integer samples staged around the search region on every side
constexpr int kGroups16 = 4;  // 16x16 PUs refined per pass (16 lanes each)
constexpr int kGroups8 = 8;   // 8x8 PUs refined per pass (8 lanes each)

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int f4(int a, int b, int c, int d) { return clip8((-2 * a + 18 * b + 18 * c - 2 * d + 16) >> 5); }
__device__ __forceinline__ uint32_t wrap_sq(int a, int b)
{
    const int d = (a - b) & 255;
    const int e = d > 128 ? 256 - d : d;  // |int8(a - b)| with -128 -> 128 (ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c:599-608)
    return (uint32_t)(e * e);
}
// ---- four pixels at a time: packed-byte helpers -------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
// LDS pointers carry their address space: through generic pointers every tile / window read became a flat_load
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// 4 bytes at an arbitrary LDS byte address (two aligned dword reads + v_alignbyte; reads up to 7 bytes past p)
__device__ __forceinline__ uint32_t lds_u32_at(const lds_u8* p)
{
    const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(p);
    const lds_u32* q = reinterpret_cast<const lds_u32*>((uintptr_t)(a & ~3u));
    return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
}
// bytewise (a - b) mod 256
__device__ __forceinline__ uint32_t sub_u8x4(uint32_t a, uint32_t b)
{
    return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ ((a ^ ~b) & 0x80808080u);
}
// sum over 4 bytes of wrap_sq: e = |int8(s - c)| with -128 -> 128, e * e.  e * e is the square of the SIGNED byte
// difference ((-128)^2 = 128^2), so it is one signed dot product of the bytewise difference with itself.
__device__ __forceinline__ uint32_t wssd4(uint32_t s, uint32_t c, uint32_t acc)
{
    const uint32_t d = sub_u8x4(s, c);
    return (uint32_t)__builtin_amdgcn_sdot4((int)d, (int)d, (int)acc, false);
}
// wssd4 of two dwords (8 pixels) at once.  The bytewise difference mod 256 is four v_sub_u32_sdwa per dword (one per byte lane,
// writing only that byte) instead of the 6-instruction SWAR form of sub_u8x4: 11 VALU per 8 pixels instead of 14.  The two dwords'
// chains are interleaved and the block ends with a wait state: on gfx94x/gfx950 a VALU write with dst_sel needs one wait state before
// a VALU read of that register (the compiler's hazard recogniser does not look inside an asm block).
__device__ __forceinline__ uint32_t wssd8(uint32_t s0, uint32_t s1, uint32_t c0, uint32_t c1, uint32_t acc)
{
    uint32_t d0, d1;
    asm("v_sub_u32_sdwa %0, %2, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
        "v_sub_u32_sdwa %1, %3, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
        "v_sub_u32_sdwa %0, %2, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
        "v_sub_u32_sdwa %1, %3, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
        "v_sub_u32_sdwa %0, %2, %4 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2\n\t"
        "v_sub_u32_sdwa %1, %3, %5 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2\n\t"
        "v_sub_u32_sdwa %0, %2, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
        "v_sub_u32_sdwa %1, %3, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
        "s_nop 0"
        : "=&v"(d0), "=&v"(d1)
        : "v"(s0), "v"(s1), "v"(c0), "v"(c1));
    return (uint32_t)__builtin_amdgcn_sdot4((int)d1, (int)d1, __builtin_amdgcn_sdot4((int)d0, (int)d0, (int)acc, false), false);
}
// bytewise (a + b + 1) >> 1
__device__ __forceinline__ uint32_t avg_u8x4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
// sum over 4 bytes of (s - v)^2, exact: s.s + v.v - 2 s.v
__device__ __forceinline__ uint32_t ssd4(uint32_t s, uint32_t v, uint32_t acc)
{
    const uint32_t pos = __builtin_amdgcn_udot4(s, s, __builtin_amdgcn_udot4(v, v, acc, false), false);
    return pos - 2u * __builtin_amdgcn_udot4(s, v, 0u, false);
}
// {-2,18,18,-2} + 16 >> 5, clipped, on the 4 bytes of w (one output sample)
__device__ __forceinline__ uint32_t hfilt1(uint32_t w)
{
    const int v = ((int)__builtin_amdgcn_udot4(w, 0x00121200u, 16u, false) - (int)__builtin_amdgcn_udot4(w, 0x02000002u, 0u, false)) >> 5;
    uint32_t r = (uint32_t)min(max(v, 0), 255);
    // keep the clipped sample opaque: left to itself the compiler fuses "shift, clip, pack two bytes" of neighbouring samples
    // into v_ashr_pk_u8_i32, whose result did not match the C semantics here (gfx950, ROCm 7.2: wrong bytes in the tiles)
    asm volatile("" : "+v"(r));
    return r;
}
// the same filter down 4 rows for the 4 byte columns of r0..r3 (packed 16-bit lanes: even and odd columns)
__device__ __forceinline__ uint32_t vfilt4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3)
{
    const uint32_t M = 0x00ff00ffu;
    uint32_t out = 0;
#pragma unroll
    for (int odd = 0; odd < 2; odd++) {
        const uint32_t a0 = (r0 >> (8 * odd)) & M, a1 = (r1 >> (8 * odd)) & M, a2 = (r2 >> (8 * odd)) & M, a3 = (r3 >> (8 * odd)) & M;
        v2s s12 = __builtin_bit_cast(v2s, a1) + __builtin_bit_cast(v2s, a2);
        v2s s03 = __builtin_bit_cast(v2s, a0) + __builtin_bit_cast(v2s, a3);
        v2s v = (s12 * (short)18 - s03 * (short)2 + (short)16) >> (short)5;
        v = __builtin_elementwise_min(__builtin_elementwise_max(v, (v2s)(short)0), (v2s)(short)255);
        out |= __builtin_bit_cast(uint32_t, v) << (8 * odd);
    }
    return out;
}

__device__ __forceinline__ uint32_t wsum(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

template <int LPP>
__device__ __forceinline__ uint32_t gsum(uint32_t v)  // sum over the LPP consecutive lanes of a PU's lane group (LPP >= 4)
{
    // inside a 16-lane row the exchange rides on the add as a DPP modifier (quad swaps, then the mirrored half row / row:
    // any pairing of complementary partial sums works for an all-reduce); only the 16- and 32-lane steps go through the LDS crossbar
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);                       // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);                       // quad_perm [2,3,0,1]
    if constexpr (LPP >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    if constexpr (LPP >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
#pragma unroll
    for (int m = 16; m < LPP; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

#define DIR_TL 0
#define DIR_T 1
#define DIR_TR 2
#define DIR_R 3
#define DIR_BR 4
#define DIR_B 5
#define DIR_BL 6
#define DIR_L 7

// SetQuarterPelRefinementInputsOnTheFly (:3271-3323): [method][position L,R,T,B,TL,TR,BR,BL][buf1/buf2] packed as
// plane | (dx+1) << 2 | (dy+1) << 4, plane 0 = integer, 1 = b, 2 = h, 3 = j
#define QP(p, dx, dy) ((p) | (((dx) + 1) << 2) | (((dy) + 1) << 4))
__device__ const uint8_t kQuarter[4][8][2] = {
    {{QP(1, 0, 0), QP(0, 0, 0)}, {QP(0, 0, 0), QP(1, 1, 0)}, {QP(2, 0, 0), QP(0, 0, 0)}, {QP(0, 0, 0), QP(2, 0, 1)},
     {QP(1, 0, 0), QP(2, 0, 0)}, {QP(2, 0, 0), QP(1, 1, 0)}, {QP(2, 0, 1), QP(1, 1, 0)}, {QP(1, 0, 0), QP(2, 0, 1)}},
    {{QP(0, -1, 0), QP(1, 0, 0)}, {QP(1, 0, 0), QP(0, 0, 0)}, {QP(3, 0, 0), QP(1, 0, 0)}, {QP(1, 0, 0), QP(3, 0, 1)},
     {QP(2, -1, 0), QP(1, 0, 0)}, {QP(1, 0, 0), QP(2, 0, 0)}, {QP(1, 0, 0), QP(2, 0, 1)}, {QP(2, -1, 1), QP(1, 0, 0)}},
    {{QP(3, 0, 0), QP(2, 0, 0)}, {QP(2, 0, 0), QP(3, 1, 0)}, {QP(0, 0, -1), QP(2, 0, 0)}, {QP(2, 0, 0), QP(0, 0, 0)},
     {QP(1, 0, -1), QP(2, 0, 0)}, {QP(2, 0, 0), QP(1, 1, -1)}, {QP(2, 0, 0), QP(1, 1, 0)}, {QP(1, 0, 0), QP(2, 0, 0)}},
    {{QP(2, -1, 0), QP(3, 0, 0)}, {QP(3, 0, 0), QP(2, 0, 0)}, {QP(1, 0, -1), QP(3, 0, 0)}, {QP(3, 0, 0), QP(1, 0, 0)},
     {QP(2, -1, 0), QP(1, 0, -1)}, {QP(1, 0, -1), QP(2, 0, 0)}, {QP(1, 0, 0), QP(2, 0, 0)}, {QP(2, -1, 0), QP(1, 0, 0)}}};

__device__ const uint8_t kTab16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
__device__ const uint8_t kTab8[64] = {0,  1,  4,  5,  16, 17, 20, 21, 2,  3,  6,  7,  18, 19, 22, 23, 8,  9,  12, 13, 24, 25,
                                      28, 29, 10, 11, 14, 15, 26, 27, 30, 31, 32, 33, 36, 37, 48, 49, 52, 53, 34, 35, 38, 39,
                                      50, 51, 54, 55, 40, 41, 44, 45, 56, 57, 60, 61, 42, 43, 46, 47, 58, 59, 62, 63};


// prediction sample of one list at fractional position frac = (x_mv & 3) + ((y_mv & 3) << 2); (x,y) = integer position
// in search-region coordinates.  F = A(x,y), Bq = b(x+1,y), Hq = h(x,y+1), Jq = j(x+1,y+1) are the samples
// BiPredictionCompensation's buffer indices select (:5155-5158); quarter positions average two of them
// (QuarterPelCompensation :4844-4910).  Table: two (plane, dx, dy) samples per frac, averaged with rounding; the pure
// positions list the same sample twice (avg(a, a) = a), so one code path serves every lane group of a wave.
#define BS(p, dx, dy) ((p) | ((dx) << 2) | ((dy) << 3))
__device__ const uint8_t kBiFrac[16][2] = {
    {BS(0, 0, 0), BS(0, 0, 0)}, {BS(0, 0, 0), BS(1, 1, 0)}, {BS(1, 1, 0), BS(1, 1, 0)}, {BS(1, 1, 0), BS(0, 1, 0)},
    {BS(0, 0, 0), BS(2, 0, 1)}, {BS(1, 1, 0), BS(2, 0, 1)}, {BS(1, 1, 0), BS(3, 1, 1)}, {BS(1, 1, 0), BS(2, 1, 1)},
    {BS(2, 0, 1), BS(2, 0, 1)}, {BS(2, 0, 1), BS(3, 1, 1)}, {BS(3, 1, 1), BS(3, 1, 1)}, {BS(3, 1, 1), BS(2, 1, 1)},
    {BS(2, 0, 1), BS(0, 0, 1)}, {BS(2, 0, 1), BS(1, 1, 1)}, {BS(3, 1, 1), BS(1, 1, 1)}, {BS(2, 1, 1), BS(1, 1, 1)}};
#undef BS


// ---- the 124 rectangular PUs of the 209-PU mode -------------------------------------------------------------------------
// me_results / the packing loop index PUs in raster order within each shape class; the ME buffers hold each class in the order
// the full-pel stage fills it (z-order of the constituent squares; tab32x16 .. tab8x32, Codec/EbMotionEstimation.h:123-171).
// Table by raster PU index: position in the SB and ME-buffer index, derived from that buffer order.
struct PuTab {
    uint8_t me[209], px[209], py[209];
};
constexpr int z4(int col, int row) { return ((row >> 1) * 2 + (col >> 1)) * 4 + (row & 1) * 2 + (col & 1); }
constexpr PuTab make_pu_tab()
{
    PuTab t{};
    // class: base, count, w, h, columns
    const int cls[14][5] = {{0, 1, 64, 64, 1},   {1, 4, 32, 32, 2},   {5, 16, 16, 16, 4},  {21, 64, 8, 8, 8},   {85, 2, 64, 32, 1},
                            {87, 8, 32, 16, 2},  {95, 32, 16, 8, 4},  {127, 2, 32, 64, 2}, {129, 8, 16, 32, 4}, {137, 32, 8, 16, 8},
                            {169, 16, 32, 8, 2}, {185, 16, 8, 32, 8}, {201, 4, 64, 16, 1}, {205, 4, 16, 64, 4}};
    for (int c = 0; c < 14; c++)
        for (int p = 0; p < cls[c][1]; p++) {
            const int base = cls[c][0], col = p % cls[c][4], row = p / cls[c][4];
            int i = p;  // 64x64, 32x32, 64x32, 32x64, 16x32, 8x32, 64x16, 16x64: buffer order = raster
            if (base == 5) i = z4(col, row);                                              // 16x16: z-order
            if (base == 21) i = 4 * z4(col >> 1, row >> 1) + (row & 1) * 2 + (col & 1);   // 8x8: raster inside its 16x16
            if (base == 87) i = 2 * ((row >> 1) * 2 + col) + (row & 1);                   // 32x16: (quadrant, upper / lower)
            if (base == 95) i = 2 * z4(col, row >> 1) + (row & 1);                        // 16x8: (16x16 z, upper / lower)
            if (base == 137) i = 2 * z4(col >> 1, row) + (col & 1);                       // 8x16: (16x16 z, left / right)
            if (base == 169) i = 4 * ((row >> 2) * 2 + col) + (row & 3);                  // 32x8: (quadrant, row of 8)
            t.me[base + p] = (uint8_t)(base + i);
            t.px[base + p] = (uint8_t)(col * cls[c][2]);
            t.py[base + p] = (uint8_t)(row * cls[c][3]);
        }
    return t;
}
__device__ constexpr PuTab kPu = make_pu_tab();


// one me_results entry: a / b = list-0 / list-1 SAD, c = bi-pred SAD, total = number of candidates (1, 2 or 3)
__device__ __forceinline__ svthip_me_cu_result pack_result(uint32_t a, uint32_t mv0, uint32_t b, uint32_t mv1, uint32_t c, int n_lists, int total)
{
    svthip_me_cu_result o;
    o.xMvL0 = (int16_t)(mv0 & 0xffffu);
    o.yMvL0 = (int16_t)(mv0 >> 16);
    o.xMvL1 = n_lists == 2 ? (int16_t)(mv1 & 0xffffu) : 0;
    o.yMvL1 = n_lists == 2 ? (int16_t)(mv1 >> 16) : 0;
    for (int k = 0; k < 3; k++) { o.distortion[k] = 0; o.direction[k] = 0; }
    if (total == 3) {
        int o0, o1, o2;  // Sort3Elements (:5434-5463)
        if (a <= b && a <= c) { o0 = 0; if (b <= c) { o1 = 1; o2 = 2; } else { o1 = 2; o2 = 1; } }
        else if (b <= a && b <= c) { o0 = 1; if (a <= c) { o1 = 0; o2 = 2; } else { o1 = 2; o2 = 0; } }
        else if (a <= b) { o0 = 2; o1 = 0; o2 = 1; }
        else { o0 = 2; o1 = 1; o2 = 0; }
        const uint32_t v[3] = {a, b, c};
        o.distortion[0] = v[o0]; o.direction[0] = (uint8_t)o0;
        o.distortion[1] = v[o1]; o.direction[1] = (uint8_t)o1;
        o.distortion[2] = v[o2]; o.direction[2] = (uint8_t)o2;
    } else if (total == 2) {
        if (a <= b) { o.distortion[0] = a; o.direction[0] = 0; o.distortion[1] = b; o.direction[1] = 1; }
        else { o.distortion[0] = b; o.direction[0] = 1; o.distortion[1] = a; o.direction[1] = 0; }
    } else {
        o.distortion[0] = a;
        o.direction[0] = 0;
    }
    o.totalMeCandidateIndex = (uint8_t)total;
    return o;
}

