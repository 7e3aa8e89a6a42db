// svt-av1-1_amd/csrc/me_fullpel209_impl.h -- 209-PU full-pel search (85 square + 124 rectangular PUs) of one superblock by
// one 256-thread workgroup: open_loop_me_fullpel_search_sblock / open_loop_me_get_search_point_results_block
// (Source/Lib/Codec/EbMotionEstimation.c:1556-1595, :1065-1231) with ExtSadCalculation_8x8_16x16 (:159-212),
// ExtSadCalculation_32x32_64x64 (:218-260) and ExtSadCalculation (:266-1052).
//
// The square part is the 85-PU kernel (me_fullpel_impl.h).  Rectangles are sums of the stored square SADs:
//   inside a wave's 32x32 quadrant: 16x8, 8x16 (two per 16x16), 32x8, 8x32 (four per quadrant), 32x16, 16x32 (two per quadrant)
//     -- packed-u16 adds of the 8x8 / 16x16 accumulators and the same (sad << 16 | idx) key minimum (all <= 65280);
//   across quadrants: 64x32, 32x64 (from the exchanged 32x32 sums), 64x16, 16x64 (two more exchange rounds through the same
//     16 KB buffer with the packed 32x16 / 16x32 sums) -- (raw << 14 | idx) keys, finished by wave Q for positions 4Q..4Q+3.
// PU 92 (32x16[5], bottom half of quadrant 2) follows the reference's stale-variable update (:343-347): it is overwritten
// with the current SAD whenever the SAD of 64x32[1] at that position is below the stored best.  That is a sequential
// recurrence over raster order, resolved per iteration by wave 0 from the per-position (64x32[1], 32x16[5]) pairs in LDS:
// "first later position whose 64x32[1] SAD is below the current best" repeated until none (the best strictly decreases).
// Included inside namespace svthip { namespace { namespace fp209 { ... } } }.
#pragma once


constexpr int kPitch = SVTHIP_FULLPEL_LDS_PITCH;  // bytes per window row in LDS

__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// global loads at byte alignment (one global_load_dword / _dwordx4 each)
struct __attribute__((packed, aligned(1))) unaligned_u32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) unaligned_u32x4 { uint32_t v[4]; };

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// keys for the four positions of a quad from packed u16 SADs (lo: slots 0,1  hi: slots 2,3)
__device__ __forceinline__ uint32_t track4(uint32_t best, uint64_t acc, const uint32_t* idx, uint32_t himask)
{
    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    uint32_t k0 = (lo << 16) | idx[0];
    uint32_t k1 = (lo & himask) | idx[1];
    uint32_t k2 = (hi << 16) | idx[2];
    uint32_t k3 = (hi & himask) | idx[3];
    best = min3u(best, k0, k1);
    best = min3u(best, k2, k3);
    return best;
}

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        uint32_t o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        unsigned long long o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t mv_word(int x, int y)
{
    // (uint16)(4*y) << 16 | (uint16)(4*x), Codec/EbMotionEstimation.c:1389-1391
    return ((uint32_t)(uint16_t)(y * 4) << 16) | (uint32_t)(uint16_t)(x * 4);
}


constexpr int kFp209Fixed = 16384 + 8192 + 1280;  // exchange buffer, (64x32[1], 32x16[5]) pairs, 64x64 result + best keys

// Best keys live in LDS in the order the waves produce them (one ds_min_u32 of 64 lanes per group of trackers, see
// me_wave_reduce.h), not in ME-buffer order:
//   [16 * (4Q + zz) + j]  the 16x16 block zz of quadrant Q: j = 0..3 its 8x8, 4 the 16x16, 5,6 16x8, 7,8 8x16, then the 32x8 pair
//                         (right-hand blocks) and / or the 8x32 pair (lower blocks) that end in this block
//   [256 + 8Q + j]        quadrant Q: j = 0 the 32x32, 1,2 32x16, 3,4 16x32
//   [288 + j]             across quadrants: j = 0,1 64x32, 2,3 32x64, 4..7 64x16, 8..11 16x64
constexpr int kFp209Slots = 304;
constexpr int fp209_slot_of_pu(int pu)
{
    if (pu >= 205) return 288 + 8 + (pu - 205);
    if (pu >= 201) return 288 + 4 + (pu - 201);
    if (pu >= 185) { const int t = pu - 185, Q = t >> 2, C = (t >> 1) & 1; return 16 * (4 * Q + 2 + C) + (C ? 11 : 9) + (t & 1); }
    if (pu >= 169) { const int t = pu - 169, Q = t >> 2, R = (t >> 1) & 1; return 16 * (4 * Q + 2 * R + 1) + 9 + (t & 1); }
    if (pu >= 137) { const int t = pu - 137; return 16 * (t >> 1) + 7 + (t & 1); }
    if (pu >= 129) { const int t = pu - 129; return 256 + 8 * (t >> 1) + 3 + (t & 1); }
    if (pu >= 127) return 288 + 2 + (pu - 127);
    if (pu >= 95) { const int t = pu - 95; return 16 * (t >> 1) + 5 + (t & 1); }
    if (pu >= 87) { const int t = pu - 87; return 256 + 8 * (t >> 1) + 1 + (t & 1); }
    if (pu >= 85) return 288 + (pu - 85);
    if (pu >= 21) { const int t = pu - 21; return 16 * (t >> 2) + (t & 3); }
    if (pu >= 5) return 16 * (pu - 5) + 4;
    return pu >= 1 ? 256 + 8 * (pu - 1) : 0;  // PU 0 (64x64) has its own 64-bit cell
}
struct Fp209SlotTable {
    uint16_t v[209];
};
constexpr Fp209SlotTable fp209_make_slot_table()
{
    Fp209SlotTable t{};
    for (int pu = 0; pu < 209; pu++) t.v[pu] = (uint16_t)fp209_slot_of_pu(pu);
    return t;
}
__device__ const Fp209SlotTable kFp209SlotOfPu = fp209_make_slot_table();

// d: the superblock's descriptor (6 int32: src_offset, ref_offset, x/y search origin, search width/height), any address space;
// smem: kFp209Fixed + (sh + 63) * SVTHIP_FULLPEL_LDS_PITCH bytes of workgroup LDS, 16-byte aligned.
// Results go to out_sad / out_mv [209 * sbi ...] in ME-buffer order.  Must be called by all 256 threads.
// FAST (search width a multiple of 16, the usual case): every position of every item is inside the area and idx = idx0 | i with idx0 = the
// lane's y * 128 + 16 * xg, so the keys are formed with the position's NUMBER i as an inline constant -- min over i of (sad << k | i) --
// and idx0 is OR-ed in once per tracker before the lanes are reduced (the bits of idx0, i and the SAD are disjoint, so the order of two
// keys of one lane is unchanged).  No idx registers: 16 fewer live VGPRs.  Otherwise positions outside the area carry idx = ~0.
template <bool FAST>
__device__ __forceinline__ void fullpel209_sb(const uint8_t* __restrict__ src_plane, uint32_t src_stride,
                                             const uint8_t* __restrict__ ref_plane, uint32_t ref_stride, const int32_t* d, uint32_t sbi,
                                             uint32_t* __restrict__ out_sad, uint32_t* __restrict__ out_mv, uint8_t* smem)
{
    // LDS layout: [0,16K) exchange buffer, [16K,24K) per-position 64x32[1] / 32x16[5] SADs of the current iteration,
    // [24K,24K+1280) 64x64 result and the best keys, then the window.
    uint32_t* xch = reinterpret_cast<uint32_t*>(smem);
    uint32_t* qa = reinterpret_cast<uint32_t*>(smem + 16384);          // [4 position quads][64 lanes][4] 64x32[1]
    uint32_t* qv = reinterpret_cast<uint32_t*>(smem + 16384 + 4096);   // [4 position quads][64 lanes][4] 32x16[5]
    unsigned long long* best64_lds = reinterpret_cast<unsigned long long*>(smem + 24576);
    // best (sad << k | raster idx) key of every PU (slot order above).  Per-lane trackers are reduced over the wave and merged here
    // with ds_min_u32 as soon as an iteration has produced them: kept in registers across the whole loop (as in the 85-PU kernel)
    // the 61 trackers of this mode pushed the kernel to 256 VGPRs + scratch spills, which made it 3x slower than the extra VALU work
    uint32_t* pu_key = reinterpret_cast<uint32_t*>(smem + 24576 + 16);  // [kFp209Slots]
    uint8_t* win = smem + kFp209Fixed;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int Q = __builtin_amdgcn_readfirstlane(tid >> 6);  // quadrant = wave index
    const int Qx = Q & 1, Qy = Q >> 1;

    // wave-uniform by construction; readfirstlane keeps them in SGPRs also when the descriptor is read from LDS
    const int src_off = __builtin_amdgcn_readfirstlane(d[0]);
    const int ref_off = __builtin_amdgcn_readfirstlane(d[1]);
    const int xo = __builtin_amdgcn_readfirstlane(d[2]), yo = __builtin_amdgcn_readfirstlane(d[3]);
    const int sw = __builtin_amdgcn_readfirstlane(d[4]), sh = __builtin_amdgcn_readfirstlane(d[5]);
    const int n_xg = (sw + 15) >> 4;
    const uint32_t inv_xg = (65536u + (uint32_t)n_xg - 1u) / (uint32_t)n_xg;  // wave-uniform, scalar unit

#ifdef SVTHIP_FP_STAGE_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    // ---- stage the reference window: rows 0..sh+62, bytes 0..sw+62 valid, zero beyond ----
    {
        // 16 bytes per thread and pass, read at the window's own byte alignment (global loads need no alignment on this target) and
        // written as one ds_write_b128: 6 passes for a 64x64 area instead of 24 dword passes with a second load + v_alignbyte each
        const uint8_t* base = ref_plane + ref_off;
        const int rows = sh + 63;
        const int ndw_valid = (sw + 63 + 3) >> 2;
        constexpr int q_row = kPitch >> 4;
        const int total = rows * q_row;
        for (int i = tid; i < total; i += 256) {
            const int r = i / q_row;
            const int c4 = i - r * q_row;
            const uint8_t* p = base + (size_t)r * ref_stride + 16 * c4;
            const int left = ndw_valid - 4 * c4;  // dwords of this slot that belong to the window
            uint32_t t[4] = {0u, 0u, 0u, 0u};
            if (left >= 4) {
                const unaligned_u32x4 u = *reinterpret_cast<const unaligned_u32x4*>(p);
                t[0] = u.v[0]; t[1] = u.v[1]; t[2] = u.v[2]; t[3] = u.v[3];
            } else if (left > 0) {  // the row's last dwords: nothing is read past them
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (k < left) t[k] = reinterpret_cast<const unaligned_u32*>(p + 4 * k)->v;
            }
            reinterpret_cast<uint4*>(win)[i] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        if (tid == 0) *best64_lds = ~0ull;
        for (int i = tid; i < kFp209Slots; i += 256) pu_key[i] = 0xffffffffu;
    }
    __syncthreads();
#ifdef SVTHIP_FP_STAGE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    // source pixels of this wave's quadrant (wave-uniform -> scalar loads)
    const uint32_t* src4 = reinterpret_cast<const uint32_t*>(src_plane + src_off + (size_t)(32 * Qy) * src_stride + 32 * Qx);
    const int sstride4 = src_stride >> 2;

    uint32_t best64_raw = 0xffffffffu, best64_idx = 0;
    uint32_t q5_raw = 0xffffffffu, q5_idx = 0;  // PU 32x16[5]: wave-uniform state of the recurrence (wave 0 only)

    const uint32_t himask = 0xffff0000u;
    // A group of up to 16 trackers goes through row_min_scatter (lane l: tracker l & 15 reduced over its row of 16) and then
    // one ds_min_u32 in which the four rows meet in the tracker's LDS slot: about 3 VALU instructions per tracker instead of 8,
    // and one LDS instruction per group instead of one per tracker.
    const uint32_t key_lds = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)pu_key);
    const int n_items = n_xg * sh;
    const int n_iter = (n_items + 63) >> 6;

    for (int it = 0; it < n_iter; it++) {
        // an opaque copy of the lane number, renewed every iteration: every per-lane LDS address below is formed from it INSIDE the loop
        // (one or two adds each).  Formed from `lane` they are loop-invariant, get hoisted, and at the 168-register limit of three
        // workgroups per CU the hoisted copies are what the register allocator spills and reloads around the search loop.
        uint32_t lane_v = (uint32_t)lane;
        asm volatile("" : "+v"(lane_v));
        int pg = it * 64 + lane;
        const bool lane_valid = pg < n_items;
        if (!lane_valid) pg = 0;
        const int y = (int)(((uint32_t)pg * inv_xg) >> 16);  // pg / n_xg, exact for n_xg <= 8 and pg < 1024 (the emulated division is ~20 instructions)
        const int xg = pg - y * n_xg;

        // per-position raster index; positions outside the search area get idx = ~0 so that every key
        // OR-ed with it is 0xffffffff and can never win (at least one position is always valid)
        // per-lane slot addresses of the tracker groups (byte addresses in LDS; the group offset goes into the instruction).
        // Formed inside the loop from an opaque copy of the lane number: hoisted out of the loop they would sit in VGPRs that the
        // kernel does not have (it runs at the 168-register limit of three workgroups per CU)
        const uint32_t slot16 = key_lds + 4 * (lane_v & 15);        // + 4 * (64Q + 16zz) or + 4 * 288
        const uint32_t slot8 = key_lds + 4 * (lane_v & 7) + 32 * Q;  // + 4 * 256

        // A lane past the last item repeats item 0: its keys duplicate lane 0's of the first pass and change no minimum (the 64x64 PU
        // and the 32x16[5] recurrence, which are not plain minima, check lane_valid).
        uint32_t idx[16];
        const uint32_t idx0 = (uint32_t)(y * 128 + 16 * xg);
        const uint32_t orv = FAST ? idx0 : 0u;  // what a finished tracker still has to be OR-ed with
#pragma unroll
        for (int i = 0; i < 16; i++) idx[i] = FAST ? (uint32_t)i : ((lane_valid && 16 * xg + i < sw) ? idx0 + (uint32_t)i : 0xffffffffu);
        uint32_t s16lo[4][4], s16hi[4][4];  // [zz][q] packed u16 16x16 sums
        // 32x8 / 8x32 pair a 16x8 / 8x16 of this block with the one of the block to its left / above.  Only the TOP 16x8 and the LEFT
        // 8x16 of the earlier block are kept; its bottom / right sums are its 16x16 sum (kept anyway) minus them -- packed u16 halves
        // never borrow because each part is <= the whole.  24 fewer live VGPRs for 32 more subtractions per item, which is what takes
        // the kernel's search loop out of scratch memory at the 168-register limit of three workgroups per CU.
        uint64_t hrow_top[4];     // [q] top 16x8 sums of the left 16x16 of the current row of 16x16s
        uint64_t hcol_lef[2][4];  // [C][q] left 8x16 sums of the upper row of 16x16s

        const uint8_t* wbase = win + (y + 32 * Qy) * kPitch + 16 * xg + 32 * Qx;

        // software-pipelined row steps, as in me_fullpel_impl.h: the window row and source row of step n + 1 are requested before the
        // 16 v_qsad of step n
#ifdef SVTHIP_FP209_PIPELINE
        uint4 An, Bn;
        uint32_t Sn[4];
        {
            An = *reinterpret_cast<const uint4*>(wbase);
            Bn = *reinterpret_cast<const uint4*>(wbase + 16);
#pragma unroll
            for (int h = 0; h < 4; h++) Sn[h] = src4[h];
        }
#endif
#pragma unroll
        for (int zz = 0; zz < 4; zz++) {
            const int C = zz & 1, R = zz >> 1;
            uint64_t acc[4][4];

#pragma unroll
            for (int r8 = 0; r8 < 8; r8++) {
#ifndef SVTHIP_FP209_PIPELINE  // operands loaded where they are used: the 85-PU kernel's software pipelining (next step's window row and
                                // source row requested one step ahead) costs 16 more live VGPRs, which here means spills: 764 vs 723 us per 6120 SBs
                const uint8_t* pp_ = wbase + (16 * R + 2 * r8) * kPitch + 16 * C;
                const uint4 A = *reinterpret_cast<const uint4*>(pp_), B = *reinterpret_cast<const uint4*>(pp_ + 16);
                const uint32_t* sr_ = src4 + (16 * R + 2 * r8) * sstride4 + 4 * C;
                const uint32_t S[4] = {sr_[0], sr_[1], sr_[2], sr_[3]};
#else
                const uint4 A = An, B = Bn;
                const uint32_t S[4] = {Sn[0], Sn[1], Sn[2], Sn[3]};
#endif
                asm volatile("" ::"v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w), "s"(S[0]), "s"(S[1]), "s"(S[2]), "s"(S[3]));
                __builtin_amdgcn_sched_barrier(0);
#ifdef SVTHIP_FP209_PIPELINE
                {
                    const int nstep = zz * 8 + r8 + 1;
                    if (nstep < 32) {
                        const int nzz = nstep >> 3, nr8 = nstep & 7, nC = nzz & 1, nR = nzz >> 1;
                        const uint8_t* p = wbase + (16 * nR + 2 * nr8) * kPitch + 16 * nC;
                        An = *reinterpret_cast<const uint4*>(p);
                        Bn = *reinterpret_cast<const uint4*>(p + 16);
                        const uint32_t* nsrow = src4 + (16 * nR + 2 * nr8) * sstride4 + 4 * nC;
#pragma unroll
                        for (int h = 0; h < 4; h++) Sn[h] = nsrow[h];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                // window dword pairs (W[k], W[k+1]), k = 0..6: the even ones are the loaded register pairs, the odd ones are formed
                // with one v_pk_mov_b32 each (hi of one pair, lo of the next).  Left to the compiler they cost two v_mov each
                // and, under register pressure (209-PU kernel), a round trip through scratch memory.
                const uint64_t E0 = pack64(A.x, A.y), E1 = pack64(A.z, A.w), E2 = pack64(B.x, B.y), E3 = pack64(B.z, B.w);
                uint64_t O0, O1, O2;
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O0) : "v"(E0), "v"(E1));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O1) : "v"(E1), "v"(E2));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O2) : "v"(E2), "v"(E3));
                const uint64_t PR[7] = {E0, O0, E1, O1, E2, O2, E3};
                const int krow = (r8 >> 2) * 2;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const int k = krow + (h >> 1);
                        const bool first = ((r8 & 3) == 0) && ((h & 1) == 0);  // first touch of acc[k][q]
                        acc[k][q] = __builtin_amdgcn_qsad_pk_u16_u8(PR[q + h], S[h], first ? 0ull : acc[k][q]);
                    }
            }

            // 8x8 PUs of this 16x16
            uint32_t k8[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, k16 = 0xffffffffu;
            uint32_t k16x8[2] = {0xffffffffu, 0xffffffffu}, k8x16[2] = {0xffffffffu, 0xffffffffu};
            uint32_t k32x8[2] = {0xffffffffu, 0xffffffffu}, k8x32[2] = {0xffffffffu, 0xffffffffu};
            // One position quad at a time, with a scheduling fence after each: left to itself the scheduler forms the ~200 keys of a 16x16
            // block's 13 trackers all at once for the sake of instruction-level parallelism and spills the live 16x16 sums to make room.
            if constexpr (FAST) {
                // 8x8 PUs per position CLASS, as in the 85-PU kernel (me_fullpel_impl.h): packed 16-bit minima over the lane's four quads,
                // one quad of keys (sad << 16 | class) per PU; the winner's position inside its item is resolved after the search
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t mlo = pk_min_u16(pk_min_u16((uint32_t)acc[k][0], (uint32_t)acc[k][1]), pk_min_u16((uint32_t)acc[k][2], (uint32_t)acc[k][3]));
                    const uint32_t mhi = pk_min_u16(pk_min_u16((uint32_t)(acc[k][0] >> 32), (uint32_t)(acc[k][1] >> 32)),
                                                    pk_min_u16((uint32_t)(acc[k][2] >> 32), (uint32_t)(acc[k][3] >> 32)));
                    k8[k] = track4(k8[k], pack64(mlo, mhi), &idx[0], himask);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (!FAST) {
#pragma unroll
                    for (int k = 0; k < 4; k++) k8[k] = track4(k8[k], acc[k][q], &idx[4 * q], himask);
                }
                // 16x8 (top / bottom halves) and 8x16 (left / right halves) of this 16x16: packed sums of two 8x8 (<= 16320);
                // 16x16 = top + bottom (packed u16, no carry between halves: <= 4 * 8160)
                const uint64_t top = pack64((uint32_t)acc[0][q] + (uint32_t)acc[1][q], (uint32_t)(acc[0][q] >> 32) + (uint32_t)(acc[1][q] >> 32));
                const uint64_t bot = pack64((uint32_t)acc[2][q] + (uint32_t)acc[3][q], (uint32_t)(acc[2][q] >> 32) + (uint32_t)(acc[3][q] >> 32));
                const uint64_t lef = pack64((uint32_t)acc[0][q] + (uint32_t)acc[2][q], (uint32_t)(acc[0][q] >> 32) + (uint32_t)(acc[2][q] >> 32));
                const uint64_t rig = pack64((uint32_t)acc[1][q] + (uint32_t)acc[3][q], (uint32_t)(acc[1][q] >> 32) + (uint32_t)(acc[3][q] >> 32));
                const uint32_t lo = (uint32_t)top + (uint32_t)bot, hi = (uint32_t)(top >> 32) + (uint32_t)(bot >> 32);
                k16 = track4(k16, pack64(lo, hi), &idx[4 * q], himask);
                s16lo[zz][q] = lo;
                s16hi[zz][q] = hi;
                k16x8[0] = track4(k16x8[0], top, &idx[4 * q], himask);
                k16x8[1] = track4(k16x8[1], bot, &idx[4 * q], himask);
                k8x16[0] = track4(k8x16[0], lef, &idx[4 * q], himask);
                k8x16[1] = track4(k8x16[1], rig, &idx[4 * q], himask);
                // 32x8 = two 16x8 side by side (zz pairs (0,1), (2,3)); 8x32 = two 8x16 on top of each other (pairs (0,2), (1,3)); <= 32640
                if (C == 0) hrow_top[q] = top;
                else {
                    const uint64_t a0 = hrow_top[q];
                    const uint64_t a1 = pack64(s16lo[zz - 1][q] - (uint32_t)a0, s16hi[zz - 1][q] - (uint32_t)(a0 >> 32));
                    k32x8[0] = track4(k32x8[0], pack64((uint32_t)a0 + (uint32_t)top, (uint32_t)(a0 >> 32) + (uint32_t)(top >> 32)),
                                              &idx[4 * q], himask);
                    k32x8[1] = track4(k32x8[1], pack64((uint32_t)a1 + (uint32_t)bot, (uint32_t)(a1 >> 32) + (uint32_t)(bot >> 32)),
                                              &idx[4 * q], himask);
                }
                if (R == 0) hcol_lef[C][q] = lef;
                else {
                    const uint64_t a0 = hcol_lef[C][q];
                    const uint64_t a1 = pack64(s16lo[zz - 2][q] - (uint32_t)a0, s16hi[zz - 2][q] - (uint32_t)(a0 >> 32));
                    k8x32[0] = track4(k8x32[0], pack64((uint32_t)a0 + (uint32_t)lef, (uint32_t)(a0 >> 32) + (uint32_t)(lef >> 32)),
                                              &idx[4 * q], himask);
                    k8x32[1] = track4(k8x32[1], pack64((uint32_t)a1 + (uint32_t)rig, (uint32_t)(a1 >> 32) + (uint32_t)(rig >> 32)),
                                              &idx[4 * q], himask);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint32_t c0 = quad_min_scatter<4>(k8[0] | orv, k8[1] | orv, k8[2] | orv, k8[3] | orv, lane);  // first part of the group reduction below
            // publish this 16x16's trackers into slots 16 * (4Q + zz) + j
            {
                constexpr uint32_t NONE = 0xffffffffu;
                const uint32_t c1 = quad_min_scatter<4>(k16 | orv, k16x8[0] | orv, k16x8[1] | orv, k8x16[0] | orv, lane);
                uint32_t r;
                if (C == 1 && R == 1) {
                    const uint32_t c2 = quad_min_scatter<4>(k8x16[1] | orv, k32x8[0] | orv, k32x8[1] | orv, k8x32[0] | orv, lane);
                    const uint32_t c3 = quad_min_scatter<1>(k8x32[1] | orv, NONE, NONE, NONE, lane);
                    r = row_min_from_quads<4>(c0, c1, c2, c3, lane);
                } else if (C == 1 || R == 1) {
                    const uint32_t c2 = quad_min_scatter<3>(k8x16[1] | orv, (C == 1 ? k32x8[0] : k8x32[0]) | orv, (C == 1 ? k32x8[1] : k8x32[1]) | orv, NONE, lane);
                    r = row_min_from_quads<3>(c0, c1, c2, NONE, lane);
                } else {
                    const uint32_t c2 = quad_min_scatter<1>(k8x16[1] | orv, NONE, NONE, NONE, lane);
                    r = row_min_from_quads<3>(c0, c1, c2, NONE, lane);
                }
                const uint32_t a = slot16 + 256 * Q;
                if (zz == 0) ds_min_u32_off<0>(a, r);
                else if (zz == 1) ds_min_u32_off<64>(a, r);
                else if (zz == 2) ds_min_u32_off<128>(a, r);
                else ds_min_u32_off<192>(a, r);
            }
        }
        // 32x16 (top / bottom) and 16x32 (left / right) of the quadrant: packed sums of two 16x16 (<= 65280 still fits 16 bits)
        uint32_t r32x16lo[2][4], r32x16hi[2][4], r16x32lo[2][4], r16x32hi[2][4];
        uint32_t k32x16[2] = {0xffffffffu, 0xffffffffu}, k16x32[2] = {0xffffffffu, 0xffffffffu}, k32 = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                r32x16lo[k][q] = s16lo[2 * k][q] + s16lo[2 * k + 1][q];
                r32x16hi[k][q] = s16hi[2 * k][q] + s16hi[2 * k + 1][q];
                r16x32lo[k][q] = s16lo[k][q] + s16lo[k + 2][q];
                r16x32hi[k][q] = s16hi[k][q] + s16hi[k + 2][q];
                if (!(Q == 2 && k == 1))  // 32x16[5] follows the stale-variable recurrence instead (wave-uniform branch)
                    k32x16[k] = track4(k32x16[k], pack64(r32x16lo[k][q], r32x16hi[k][q]), &idx[4 * q], himask);
                k16x32[k] = track4(k16x32[k], pack64(r16x32lo[k][q], r16x32hi[k][q]), &idx[4 * q], himask);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // 32x32 = sum of the four 16x16: pairs are added packed (<= 2*32640 fits u16), then widened
        uint32_t s32acc[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t a_lo = s16lo[0][q] + s16lo[1][q], b_lo = s16lo[2][q] + s16lo[3][q];
            const uint32_t a_hi = s16hi[0][q] + s16hi[1][q], b_hi = s16hi[2][q] + s16hi[3][q];
            s32acc[4 * q + 0] = (a_lo & 0xffffu) + (b_lo & 0xffffu);
            s32acc[4 * q + 1] = (a_lo >> 16) + (b_lo >> 16);
            s32acc[4 * q + 2] = (a_hi & 0xffffu) + (b_hi & 0xffffu);
            s32acc[4 * q + 3] = (a_hi >> 16) + (b_hi >> 16);
        }

        // 32x32 PU of this quadrant: key = raw << 14 | idx  (raw <= 130560 < 2^17)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            uint32_t k0 = (s32acc[i] << 14) | idx[i];
            uint32_t k1 = (s32acc[i + 1] << 14) | idx[i + 1];
            k32 = min3u(k32, k0, k1);
        }

        {
            // PU 92 (32x16[5], Q == 2) follows the recurrence instead: its tracker was never updated and stays 0xffffffff
            const uint32_t g[8] = {k32 | orv, k32x16[0] | orv, k32x16[1] | orv, k16x32[0] | orv, k16x32[1] | orv, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            ds_min_u32_off<4 * 256>(slot8, row_min_scatter<8, 5>(g, lane));
        }

        // Cross-quadrant PUs: wave Q finishes positions 4Q..4Q+3 of every lane's 16.  Two exchange rounds through the 16 KB buffer:
        // round B moves the packed 32x16 sums (top / bottom half of every quadrant) -- 64x16 is their sum across a quadrant pair, and the
        // 32x32 sums that 64x64 / 64x32 / 32x64 need are top + bottom of the same data, so no separate 32x32 round is needed (it was a
        // third round with two more barriers per iteration); round C moves the packed 16x32 sums for 16x64.
        const int xbase = 16 * xg + 4 * Q;
        const uint32_t ibase = (uint32_t)(y * 128 + xbase);
        uint32_t cidx[4];  // idx of positions 4Q..4Q+3 (idx[] is indexed statically, so rebuilt from Q)
#pragma unroll
        for (int j = 0; j < 4; j++) cidx[j] = FAST ? (uint32_t)j : ((lane_valid && xbase + j < sw) ? ibase + j : 0xffffffffu);
        const uint32_t corv = FAST ? ibase : 0u;
        uint32_t kc[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, kd[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        // ---- round B: 32x16 sums (packed u16, 2 PUs x 8 dwords per lane); wave 2 also publishes 32x16[5] per position
        __syncthreads();  // previous iteration's readers are done
        {
            // exchange layout [wave][uint4 index][lane][4]: 128-bit accesses of consecutive lanes are conflict-free (lane-major rows of
            // 16 dwords put every second lane on the same banks)
            uint4* dst = reinterpret_cast<uint4*>(xch + Q * 1024 + lane_v * 4);
            dst[0] = make_uint4(r32x16lo[0][0], r32x16hi[0][0], r32x16lo[0][1], r32x16hi[0][1]);
            dst[64] = make_uint4(r32x16lo[0][2], r32x16hi[0][2], r32x16lo[0][3], r32x16hi[0][3]);
            dst[128] = make_uint4(r32x16lo[1][0], r32x16hi[1][0], r32x16lo[1][1], r32x16hi[1][1]);
            dst[192] = make_uint4(r32x16lo[1][2], r32x16hi[1][2], r32x16lo[1][3], r32x16hi[1][3]);
            if (Q == 2) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<uint4*>(qv + q * 256 + lane_v * 4) =
                        make_uint4(r32x16lo[1][q] & 0xffffu, r32x16lo[1][q] >> 16, r32x16hi[1][q] & 0xffffu, r32x16hi[1][q] >> 16);
            }
        }
        __syncthreads();
        {
            // positions 4Q..4Q+3 = dwords (2Q, 2Q+1) of each PU's 8-dword run, i.e. half (Q & 1) of uint4 2R + (Q >> 1) of wave w
            uint32_t pr[4][2][4];
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int R = 0; R < 2; R++) {
                    const uint2 t = *reinterpret_cast<const uint2*>(xch + w * 1024 + (2 * R + (Q >> 1)) * 256 + lane_v * 4 + 2 * (Q & 1));
                    pr[w][R][0] = t.x & 0xffffu; pr[w][R][1] = t.x >> 16; pr[w][R][2] = t.y & 0xffffu; pr[w][R][3] = t.y >> 16;
                }
            uint32_t qbot[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t t0 = pr[0][0][j] + pr[1][0][j], t1 = pr[0][1][j] + pr[1][1][j];  // 64x16[0], 64x16[1]
                const uint32_t t2 = pr[2][0][j] + pr[3][0][j], t3 = pr[2][1][j] + pr[3][1][j];  // 64x16[2], 64x16[3]
                const uint32_t top = t0 + t1, bot = t2 + t3;                                    // 64x32[0], 64x32[1]
                const uint32_t lef = pr[0][0][j] + pr[0][1][j] + pr[2][0][j] + pr[2][1][j], all = top + bot, rig = all - lef;  // 32x64[0], [1]
                // strict '<', positions visited in raster order per lane; positions outside the area never win
                const bool better = (all < best64_raw) && (FAST ? lane_valid : cidx[j] != 0xffffffffu);
                best64_raw = better ? all : best64_raw;
                best64_idx = better ? (ibase + j) : best64_idx;
                kc[0] = min(kc[0], (top << 14) | cidx[j]);   // 64x32[0]   (<= 261120 < 2^18)
                kc[1] = min(kc[1], (bot << 14) | cidx[j]);   // 64x32[1]
                kc[2] = min(kc[2], (lef << 14) | cidx[j]);   // 32x64[0]
                kc[3] = min(kc[3], (rig << 14) | cidx[j]);   // 32x64[1]
                kd[0] = min(kd[0], (t0 << 14) | cidx[j]);    // 64x16[0] = 32x16[0] + 32x16[2]
                kd[1] = min(kd[1], (t1 << 14) | cidx[j]);    // 64x16[1] = 32x16[1] + 32x16[3]
                kd[2] = min(kd[2], (t2 << 14) | cidx[j]);    // 64x16[2] = 32x16[4] + 32x16[6]
                kd[3] = min(kd[3], (t3 << 14) | cidx[j]);    // 64x16[3] = 32x16[5] + 32x16[7]
                qbot[j] = bot;                               // 64x32[1] per position, for the 32x16[5] recurrence
            }
            *reinterpret_cast<uint4*>(qa + Q * 256 + lane_v * 4) = make_uint4(qbot[0], qbot[1], qbot[2], qbot[3]);  // [position quad][lane][4]
        }
        // ---- round C: 16x32 sums -> 16x64; meanwhile wave 0 resolves the 32x16[5] recurrence of this iteration
        __syncthreads();
        {
            uint4* dst = reinterpret_cast<uint4*>(xch + Q * 1024 + lane_v * 4);
            dst[0] = make_uint4(r16x32lo[0][0], r16x32hi[0][0], r16x32lo[0][1], r16x32hi[0][1]);
            dst[64] = make_uint4(r16x32lo[0][2], r16x32hi[0][2], r16x32lo[0][3], r16x32hi[0][3]);
            dst[128] = make_uint4(r16x32lo[1][0], r16x32hi[1][0], r16x32lo[1][1], r16x32hi[1][1]);
            dst[192] = make_uint4(r16x32lo[1][2], r16x32hi[1][2], r16x32lo[1][3], r16x32hi[1][3]);
        }
        __syncthreads();
        {
            uint32_t pc[4][2][4], ke[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int C = 0; C < 2; C++) {
                    const uint2 t = *reinterpret_cast<const uint2*>(xch + w * 1024 + (2 * C + (Q >> 1)) * 256 + lane_v * 4 + 2 * (Q & 1));
                    pc[w][C][0] = t.x & 0xffffu; pc[w][C][1] = t.x >> 16; pc[w][C][2] = t.y & 0xffffu; pc[w][C][3] = t.y >> 16;
                }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ke[0] = min(ke[0], ((pc[0][0][j] + pc[2][0][j]) << 14) | cidx[j]);    // 16x64[0] = 16x32[0] + 16x32[4]
                ke[1] = min(ke[1], ((pc[0][1][j] + pc[2][1][j]) << 14) | cidx[j]);    // 16x64[1] = 16x32[1] + 16x32[5]
                ke[2] = min(ke[2], ((pc[1][0][j] + pc[3][0][j]) << 14) | cidx[j]);  // 16x64[2] = 16x32[2] + 16x32[6]
                ke[3] = min(ke[3], ((pc[1][1][j] + pc[3][1][j]) << 14) | cidx[j]);  // 16x64[3] = 16x32[3] + 16x32[7]
            }
            // the twelve cross-quadrant trackers of this wave's positions in one group (slots 288 + j)
            const uint32_t g[16] = {kc[0] | corv, kc[1] | corv, kc[2] | corv, kc[3] | corv, kd[0] | corv, kd[1] | corv, kd[2] | corv, kd[3] | corv,
                                    ke[0] | corv, ke[1] | corv, ke[2] | corv, ke[3] | corv, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            ds_min_u32_off<4 * 288>(slot16, row_min_scatter<16, 12>(g, lane));
        }
        if (Q == 0) {
            // 32x16[5] (:343-347): in raster order, "if (sad of 64x32[1] < best) best = sad of 32x16[5]".  The items of an
            // iteration are consecutive in raster order (lane = item, 16 positions each), so the rank inside the chunk is
            // lane * 16 + position.  Every update lowers the best (32x16[5] is part of 64x32[1]), so the loop is short.
            uint32_t av[16], vv[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(qa + q * 256 + lane_v * 4), v4 = *reinterpret_cast<const uint4*>(qv + q * 256 + lane_v * 4);
                av[4 * q] = a4.x; av[4 * q + 1] = a4.y; av[4 * q + 2] = a4.z; av[4 * q + 3] = a4.w;
                vv[4 * q] = v4.x; vv[4 * q + 1] = v4.y; vv[4 * q + 2] = v4.z; vv[4 * q + 3] = v4.w;
            }
            int last = -1;  // rank of the last update inside this chunk
            for (;;) {
                uint32_t cand = 0xffffffffu;  // (rank << 16 | new best) of this lane's first qualifying position
#pragma unroll
                for (int i = 15; i >= 0; i--) {
                    const int rank = lane * 16 + i;
                    const bool ok = (FAST ? lane_valid : idx[i] != 0xffffffffu) && rank > last && av[i] < q5_raw;
                    cand = ok ? (((uint32_t)rank << 16) | vv[i]) : cand;
                }
                cand = wave_min_u32(cand);
                if (cand == 0xffffffffu) break;  // wave-uniform
                last = (int)(cand >> 16);
                q5_raw = cand & 0xffffu;
                // raster index of that position: the lane that owns it broadcasts its idx
                const int owner = last >> 4, pos = last & 15;
                q5_idx = (uint32_t)__shfl((int)idx0, owner) + (uint32_t)pos;  // the owning lane's y * 128 + 16 * xg, plus the position
            }
        }
    }

    // ---- publish: every PU's key is in LDS (64x64 as a 64-bit (raw, idx) pair, PU 92 from the recurrence state) ----
    uint32_t* osad = out_sad + (size_t)209 * sbi;
    uint32_t* omv = out_mv + (size_t)209 * sbi;
    const unsigned long long k64 = wave_min_u64(((unsigned long long)best64_raw << 32) | best64_idx);
    if (lane == 0) atomicMin(best64_lds, k64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ds_min_u32 above are invisible to the compiler's counter tracking
    __syncthreads();
    if constexpr (FAST) {
        // resolve the 8x8 winners (key = sad << 16 | y * 128 + 16 * xg + class): lane = 4 * PU + quad recomputes the SADs of positions
        // 4 quad .. 4 quad + 3 of the winning item; the first position whose SAD equals the minimum is the reference's strict-'<' winner
        const int p = lane >> 2, q = lane & 3;
        const int zz = p >> 2, k = p & 3, px = 16 * (zz & 1) + 8 * (k & 1), py = 16 * (zz >> 1) + 8 * (k >> 1);
        const uint32_t key = pu_key[16 * (4 * Q + zz) + k];
        const uint32_t s = key >> 16, id = key & 0xffffu;
        const int y = (int)(id >> 7), xb = (int)(id & 0x70u);
        const uint8_t* wp = win + (y + 32 * Qy + py) * kPitch + xb + 4 * q + 32 * Qx + px;
        const uint32_t* sp = src4 + (size_t)py * sstride4 + (px >> 2);
        uint64_t a = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(wp + 2 * r * kPitch);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            a = __builtin_amdgcn_qsad_pk_u16_u8(pack64(w0, w1), sp[(size_t)(2 * r) * sstride4], a);
            a = __builtin_amdgcn_qsad_pk_u16_u8(pack64(w1, w2), sp[(size_t)(2 * r) * sstride4 + 1], a);
        }
        const uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
        uint32_t first = (lo & 0xffffu) == s ? 0u : (lo >> 16) == s ? 1u : (hi & 0xffffu) == s ? 2u : (hi >> 16) == s ? 3u : 64u;
        first += 4u * (uint32_t)q;
        first = min(first, (uint32_t)__shfl_xor((int)first, 1));
        first = min(first, (uint32_t)__shfl_xor((int)first, 2));
        if (q == 0) {
            const int pu = 21 + 16 * Q + p;
            osad[pu] = 2u * s;
            omv[pu] = mv_word(xo + xb + (int)first, yo + y);
        }
    }
    if (tid < 209 && !(FAST && tid >= 21 && tid < 85)) {
        const int pu = tid;
        uint32_t raw, id;
        if (pu == 0) {
            const unsigned long long k = *best64_lds;
            raw = (uint32_t)(k >> 32);
            id = (uint32_t)k;
        } else {
            // 32x32 and the cross-quadrant rectangles carry raw << 14, everything else raw << 16 (idx = y * 128 + x, 14 bits)
            const bool wide = (pu >= 1 && pu <= 4) || pu == 85 || pu == 86 || pu == 127 || pu == 128 || pu >= 201;
            const uint32_t key = pu_key[kFp209SlotOfPu.v[pu]];
            raw = wide ? key >> 14 : key >> 16;
            id = key & 0x3fffu;
        }
        if (pu != 92) {
            osad[pu] = 2u * raw;
            omv[pu] = mv_word(xo + (int)(id & 127u), yo + (int)(id >> 7));
        }
    }
    if (tid == 0) {  // wave 0 holds the (wave-uniform) state of the 32x16[5] recurrence
        osad[92] = 2u * q5_raw;
        omv[92] = mv_word(xo + (int)(q5_idx & 127u), yo + (int)(q5_idx >> 7));
    }
}
