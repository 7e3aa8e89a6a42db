// Wave-wide minima of several per-lane trackers at once.
//
// Reducing N trackers one by one costs log2(64) exchange + min steps each.  The "reduce-scatter" below halves the number of live
// registers at every step instead: at the step for lane bit s, a lane keeps one tracker of each pair (the odd one if its bit s is
// set), hands the other one to its partner lane, and folds in what the partner handed over.  After log2(N) steps lane l holds
// the minimum of tracker (l & (N-1)) over all lanes of its row of 16; that is 3 VALU instructions per pair and step
// (two v_cndmask, one v_min_u32_dpp), 3 * (N - 1) in total, instead of 6..8 * N.  The four rows are combined by the caller
// (two more exchanges, or an LDS ds_min_u32 that all four rows issue to the same slot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svthip {

template <int CTRL> __device__ __forceinline__ uint32_t dpp_fetch(uint32_t v)
{
    // every DPP pattern used here (quad_perm, row_ror) has a valid source lane for every lane, so "old" is never selected
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// one halving step over NIN registers of which the first NV carry trackers (the rest are 0xffffffff and cost nothing)
template <int CTRL, int NIN, int NV>
__device__ __forceinline__ void scatter_step(const uint32_t* in, uint32_t* out, bool odd)
{
#pragma unroll
    for (int i = 0; i < NIN / 2; i++) {
        if (2 * i >= NV) {
            out[i] = 0xffffffffu;
        } else {
            const uint32_t keep = odd ? in[2 * i + 1] : in[2 * i];
            const uint32_t send = odd ? in[2 * i] : in[2 * i + 1];
            out[i] = min(keep, dpp_fetch<CTRL>(send));
        }
    }
}

// v[N], N = 8 or 16, of which v[NV..N-1] must be 0xffffffff.  Returns, in lane l, the minimum of v[l & (N-1)] over the 16 lanes
// of l's row.
template <int N, int NV = N> __device__ __forceinline__ uint32_t row_min_scatter(const uint32_t (&v)[N], int lane)
{
    static_assert(N == 8 || N == 16, "8 or 16 trackers");
    static_assert(NV >= 1 && NV <= N, "valid count");
    constexpr int NV1 = (NV + 1) / 2, NV2 = (NV1 + 1) / 2, NV3 = (NV2 + 1) / 2;
    uint32_t r1[N / 2], r2[N / 4], r3[N / 8];
    scatter_step<0xB1, N, NV>(v, r1, lane & 1);         // quad_perm [1,0,3,2]: lane ^ 1
    scatter_step<0x4E, N / 2, NV1>(r1, r2, lane & 2);   // quad_perm [2,3,0,1]: lane ^ 2
    scatter_step<0x124, N / 4, NV2>(r2, r3, lane & 4);  // row_ror:4: lane - 4 (mod 16), the same quad position with bit 2 flipped
    if constexpr (N == 16) {
        uint32_t r4[1];
        scatter_step<0x128, 2, NV3>(r3, r4, lane & 8);  // row_ror:8: lane ^ 8
        return r4[0];
    } else {
        return min(r3[0], dpp_fetch<0x128>(r3[0]));
    }
}

// The two halves of row_min_scatter<16> for callers whose trackers become final at different times (fewer registers live):
// quad_min_scatter leaves lane l with tracker (l & 3) of {a, b, c, d} reduced over its quad (the first NV of them real);
// row_min_from_quads folds four such results (trackers 0..3, 4..7, 8..11, 12..15; the first NQ real) into the row result.
template <int NV = 4> __device__ __forceinline__ uint32_t quad_min_scatter(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int lane)
{
    const uint32_t in[4] = {a, b, c, d};
    uint32_t r1[2], r2[1];
    scatter_step<0xB1, 4, NV>(in, r1, lane & 1);
    scatter_step<0x4E, 2, (NV + 1) / 2>(r1, r2, lane & 2);
    return r2[0];
}

template <int NQ = 4> __device__ __forceinline__ uint32_t row_min_from_quads(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, int lane)
{
    const uint32_t in[4] = {q0, q1, q2, q3};
    uint32_t r3[2], r4[1];
    scatter_step<0x124, 4, NQ>(in, r3, lane & 4);
    scatter_step<0x128, 2, (NQ + 1) / 2>(r3, r4, lane & 8);
    return r4[0];
}

// LDS minimum without return value at byte address addr + OFF (OFF in the instruction's offset field).  atomicMin() would do, but
// its lowering wraps every call in a "first active lane" sequence; the caller needs an s_waitcnt lgkmcnt(0) before other
// waves read the cell, the compiler's counter tracking does not see this instruction.
template <int OFF> __device__ __forceinline__ void ds_min_u32_off(uint32_t addr, uint32_t v)
{
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    asm volatile("ds_min_u32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// the same over the whole wave: every lane l ends up with the minimum of v[l & (N-1)] over all 64 lanes
template <int N, int NV = N> __device__ __forceinline__ uint32_t wave_min_scatter(const uint32_t (&v)[N], int lane)
{
    uint32_t r = row_min_scatter<N, NV>(v, lane);
    r = min(r, (uint32_t)__shfl_xor((int)r, 16));
    r = min(r, (uint32_t)__shfl_xor((int)r, 32));
    return r;
}

}  // namespace svthip
