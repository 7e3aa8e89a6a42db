// svt-av1-1_amd/csrc/me_fullpel_impl.h -- 85-PU full-pel search of one superblock by one 256-thread workgroup (device code).
// Included inside namespace svthip { namespace { ... } } by me_fullpel.hip.  See me_fullpel.hip for the mapping and the reference citations.
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


// d: the superblock's descriptor (6 int32: src_offset, ref_offset, x/y search origin, search width/height), any address space;
// smem: SVTHIP_FULLPEL_LDS_FIXED + (sh + 63) * SVTHIP_FULLPEL_LDS_PITCH bytes of workgroup LDS, 16-byte aligned.
// Results go to out_sad / out_mv [85 * sbi ...].  Must be called by all 256 threads.
// CLS (search width a multiple of 16, the usual case; wave-uniform): the 8x8 PUs -- 256 of the 336 (PU, position) candidates of an item --
// are tracked per position CLASS.  The four quads of a lane's 16 positions are first reduced with packed 16-bit minima (slot c of the
// result = min over q of the SAD at position 4 q + c), then ONE quad of keys (sad << 16 | idx0 + c) goes into the running minimum: 12
// instructions per PU and item instead of 24.  The winner of a PU then names its item (idx0) and its SAD but not the position inside the
// item: items are disjoint runs of 16 raster positions, so the first minimum in raster order lies in the first item that attains the
// minimum -- which is what the key order picks -- and after the search four lanes per PU recompute that item's 16 SADs (8 v_qsad per lane,
// once per superblock) and take the first position whose SAD equals the minimum: the reference's strict-'<' rule again.
template <bool CLS>
__device__ __forceinline__ void fullpel85_sb(const uint8_t* __restrict__ src_plane, uint32_t src_stride,
                                             const uint8_t* __restrict__ ref_plane, uint32_t ref_stride, const int32_t* d, uint32_t sbi,
                                             uint32_t* __restrict__ out_sad, uint32_t* __restrict__ out_mv, uint8_t* smem)
{
    // LDS layout: [0,16K) exchange buffer for 32x32 sums, [16K,16K+16) 64x64 result, then the window.
    uint32_t* xch = reinterpret_cast<uint32_t*>(smem);
    unsigned long long* best64_lds = reinterpret_cast<unsigned long long*>(smem + 16384);
    uint8_t* win = smem + 16384 + 64;

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
    }
    __syncthreads();
#ifdef SVTHIP_FP_STAGE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    // source pixels of this wave's quadrant (wave-uniform -> scalar loads)
    const uint32_t* src4 = reinterpret_cast<const uint32_t*>(src_plane + src_off + (size_t)(32 * Qy) * src_stride + 32 * Qx);
    const int sstride4 = src_stride >> 2;

    // CLS: the source rows of the 8x8 PU this lane will resolve (lane = 4 * PU + quad), requested now, used after the search
    uint32_t rsv[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if constexpr (CLS) {
        const int p = lane >> 2, zz = p >> 2, k = p & 3, px = 16 * (zz & 1) + 8 * (k & 1), py = 16 * (zz >> 1) + 8 * (k >> 1);
        const uint32_t* sp = src4 + (size_t)py * sstride4 + (px >> 2);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            rsv[r][0] = sp[(size_t)(2 * r) * sstride4];
            rsv[r][1] = sp[(size_t)(2 * r) * sstride4 + 1];
        }
    }

    uint32_t best8[16], best16[4], best32 = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 16; i++) best8[i] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 4; i++) best16[i] = 0xffffffffu;
    uint32_t best64_raw = 0xffffffffu, best64_idx = 0;

    const uint32_t himask = 0xffff0000u;
    const int n_items = n_xg * sh;
    const int n_iter = (n_items + 63) >> 6;

    for (int it = 0; it < n_iter; it++) {
        int pg = it * 64 + lane;
        const bool lane_valid = pg < n_items;
        if (!lane_valid) pg = 0;
        const int y = (int)(((uint32_t)pg * inv_xg) >> 16);  // pg / n_xg, exact for n_xg <= 8 and pg < 1024 (the emulated division is ~20 instructions)
        const int xg = pg - y * n_xg;

        // per-position raster index; positions outside the search area get idx = ~0 so that every key
        // OR-ed with it is 0xffffffff and can never win (at least one position is always valid).  A lane past the last item
        // repeats item 0: its keys duplicate lane 0's of the first pass and change no minimum.  Areas whose width is a multiple
        // of 16 (the usual case) have no outside positions at all: one add per position instead of compare + select.
        uint32_t idx[16];
        const uint32_t idx0 = (uint32_t)(y * 128 + 16 * xg);
        if ((sw & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) idx[i] = idx0 + (uint32_t)i;
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) idx[i] = (16 * xg + i < sw) ? idx0 + (uint32_t)i : 0xffffffffu;
        }

        uint32_t s16lo[4][4], s16hi[4][4];  // [zz][q] packed u16 16x16 sums

        const uint8_t* wbase = win + (y + 32 * Qy) * kPitch + 16 * xg + 32 * Qx;

        // The 32 (16x16 sub-block, row) steps are software-pipelined: the window row (two ds_read_b128) and the source row (one
        // s_load_dwordx4) of step n + 1 are issued before the 16 v_qsad of step n, so their latency hides behind ~260 issue cycles
        // instead of being waited for at the top of every row (12 more live VGPRs; the kernel stays at three workgroups per CU).
        uint4 An, Bn;
        uint32_t Sn[4];
        {
            An = *reinterpret_cast<const uint4*>(wbase);
            Bn = *reinterpret_cast<const uint4*>(wbase + 16);
#pragma unroll
            for (int h = 0; h < 4; h++) Sn[h] = src4[h];  // uniform address, read-only -> s_load_dwordx4
        }
#pragma unroll
        for (int zz = 0; zz < 4; zz++) {
            uint64_t acc[4][4];

#pragma unroll
            for (int r8 = 0; r8 < 8; r8++) {
                const uint4 A = An, B = Bn;
                const uint32_t W[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
                const uint32_t S[4] = {Sn[0], Sn[1], Sn[2], Sn[3]};
                // this step's operands were requested one step ago: make the s_waitcnt for them land HERE, before the next requests go
                // out (scalar loads return out of order, so any later wait would be lgkmcnt(0) and cover the fresh requests too)
                asm volatile("" ::"v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w), "s"(S[0]), "s"(S[1]), "s"(S[2]), "s"(S[3]));
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int nstep = zz * 8 + r8 + 1;
                    if (nstep < 32) {
                        const int nzz = nstep >> 3, nr8 = nstep & 7, nC = nzz & 1, nR = nzz >> 1;
                        const uint8_t* p = wbase + (16 * nR + 2 * nr8) * kPitch + 16 * nC;
                        An = *reinterpret_cast<const uint4*>(p);
                        Bn = *reinterpret_cast<const uint4*>(p + 16);
                        const uint32_t* srow = src4 + (16 * nR + 2 * nr8) * sstride4 + 4 * nC;
#pragma unroll
                        for (int h = 0; h < 4; h++) Sn[h] = srow[h];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the loads above this step's arithmetic (the scheduler sinks them to their use)
                }
                const int krow = (r8 >> 2) * 2;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const int k = krow + (h >> 1);
                        const bool first = ((r8 & 3) == 0) && ((h & 1) == 0);  // first touch of acc[k][q]
                        acc[k][q] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(W[q + h], W[q + h + 1]), S[h],
                                                                    first ? 0ull : acc[k][q]);
                    }
            }

            // 8x8 PUs of this 16x16
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if constexpr (CLS) {
                    const uint32_t mlo = pk_min_u16(pk_min_u16((uint32_t)acc[k][0], (uint32_t)acc[k][1]), pk_min_u16((uint32_t)acc[k][2], (uint32_t)acc[k][3]));
                    const uint32_t mhi = pk_min_u16(pk_min_u16((uint32_t)(acc[k][0] >> 32), (uint32_t)(acc[k][1] >> 32)),
                                                    pk_min_u16((uint32_t)(acc[k][2] >> 32), (uint32_t)(acc[k][3] >> 32)));
                    best8[4 * zz + k] = track4(best8[4 * zz + k], pack64(mlo, mhi), &idx[0], himask);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) best8[4 * zz + k] = track4(best8[4 * zz + k], acc[k][q], &idx[4 * q], himask);
                }
            }

            // 16x16 = sum of the four 8x8 (packed u16, no carry between halves: <= 4*(8160+8200))
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t lo = (uint32_t)acc[0][q] + (uint32_t)acc[1][q] + (uint32_t)acc[2][q] + (uint32_t)acc[3][q];
                const uint32_t hi = (uint32_t)(acc[0][q] >> 32) + (uint32_t)(acc[1][q] >> 32) +
                                    (uint32_t)(acc[2][q] >> 32) + (uint32_t)(acc[3][q] >> 32);
                best16[zz] = track4(best16[zz], pack64(lo, hi), &idx[4 * q], himask);
                s16lo[zz][q] = lo;
                s16hi[zz][q] = hi;
            }
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
            best32 = min3u(best32, k0, k1);
        }

        // 64x64: exchange 32x32 sums between the four waves; wave Q finishes positions 4Q..4Q+3
        __syncthreads();  // previous iteration's readers are done
        {
            // [wave][position quad][lane][4]: a 128-bit access of 8 consecutive lanes covers the 32 banks once (lane-major rows of 16
            // dwords put every second lane on the same banks: 4-way conflicts on all eight accesses of the exchange)
            uint4* dst = reinterpret_cast<uint4*>(xch + Q * 1024 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; q++) dst[q * 64] = make_uint4(s32acc[4 * q], s32acc[4 * q + 1], s32acc[4 * q + 2], s32acc[4 * q + 3]);
        }
        __syncthreads();
        {
            uint4 s = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint4 v = *reinterpret_cast<const uint4*>(xch + w * 1024 + Q * 256 + lane * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const uint32_t sv[4] = {s.x, s.y, s.z, s.w};
            // idx of position 4Q+j of this lane: idx[] is indexed statically, so select by Q
            const int xbase = 16 * xg + 4 * Q;
            const uint32_t ibase = (uint32_t)(y * 128 + xbase);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // strict '<', positions visited in raster order per lane; positions outside the area never win
                const bool better = (sv[j] < best64_raw) && lane_valid && (xbase + j < sw);
                best64_raw = better ? sv[j] : best64_raw;
                best64_idx = better ? (ibase + j) : best64_idx;
            }
        }
    }

    // ---- reduce across the wave and publish ----
    uint32_t* osad = out_sad + (size_t)85 * sbi;
    uint32_t* omv = out_mv + (size_t)85 * sbi;

    // the 21 trackers of this quadrant in two reduce-scatter passes (me_wave_reduce.h): lanes 0..15 end up with the 8x8 PUs,
    // lanes 16..20 with the four 16x16 and the 32x32, and every one of those lanes stores its own PU
    const uint32_t g8 = wave_min_scatter<16>(best8, lane);
    const uint32_t top[8] = {best16[0], best16[1], best16[2], best16[3], best32, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const uint32_t g16 = wave_min_scatter<8, 5>(top, lane);
    const unsigned long long k64 = wave_min_u64(((unsigned long long)best64_raw << 32) | best64_idx);

    if constexpr (CLS) {
        // resolve the 8x8 winners: lane = 4 * PU + quad recomputes the SADs of positions 4 quad .. 4 quad + 3 of the winning item
        const int p = lane >> 2, q = lane & 3;
        const uint32_t key = (uint32_t)__shfl((int)g8, p);
        const uint32_t s = key >> 16, id = key & 0xffffu;  // id = y * 128 + 16 * xg + class
        const int y = (int)(id >> 7), xb = (int)(id & 0x70u);
        const int zz = p >> 2, k = p & 3, px = 16 * (zz & 1) + 8 * (k & 1), py = 16 * (zz >> 1) + 8 * (k >> 1);
        const uint8_t* wp = win + (y + 32 * Qy + py) * kPitch + xb + 4 * q + 32 * Qx + px;
        uint64_t a = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(wp + 2 * r * kPitch);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            a = __builtin_amdgcn_qsad_pk_u16_u8(pack64(w0, w1), rsv[r][0], a);
            a = __builtin_amdgcn_qsad_pk_u16_u8(pack64(w1, w2), rsv[r][1], a);
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
    if (lane < 21 && !(CLS && lane < 16)) {
        const uint32_t key = lane < 16 ? g8 : g16;
        const int pu = lane < 16 ? 21 + 16 * Q + lane : lane < 20 ? 5 + 4 * Q + (lane - 16) : 1 + Q;
        const uint32_t raw = lane == 20 ? key >> 14 : key >> 16, id = key & 0x3fffu;
        osad[pu] = 2u * raw;
        omv[pu] = mv_word(xo + (int)(id & 127u), yo + (int)(id >> 7));
    }
    if (lane == 0) atomicMin(best64_lds, k64);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long k = *best64_lds;
        const uint32_t raw = (uint32_t)(k >> 32), id = (uint32_t)k;
        osad[0] = 2u * raw;
        omv[0] = mv_word(xo + (int)(id & 127u), yo + (int)(id >> 7));
    }
}
