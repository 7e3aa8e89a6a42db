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

// the same over the whole wave: every lane l ends up with the minimum of v[l & (N-1)] over all 64 lanes
template <int N, int NV = N> __device__ __forceinline__ uint32_t wave_min_scatter(const uint32_t (&v)[N], int lane)
{
    uint32_t r = row_min_scatter<N, NV>(v, lane);
    r = min(r, (uint32_t)__shfl_xor((int)r, 16));
    r = min(r, (uint32_t)__shfl_xor((int)r, 32));
    return r;
}

}  // namespace svthip
