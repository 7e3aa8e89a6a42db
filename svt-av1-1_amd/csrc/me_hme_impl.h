// svt-av1-1_amd/csrc/me_hme_impl.h -- search-centre derivation of one superblock by one 256-thread workgroup (device code).
// Included inside namespace svthip { namespace { ... } } by me_hme.hip.  (A kernel fusing this chain with the full-pel
// search of the same superblock was bit-identical but slower -- both halves are VALU-bound -- and was removed.)  See me_hme.hip for the mapping and the reference citations.
#pragma once


// Unaligned 4-byte read as two ALIGNED dword loads + v_alignbyte.  A single misaligned global_load_dword is
// legal on gfx950; round 1 measured it ~10x slower in the un-staged search loop (the 64 lanes of a wave split into per-lane
// requests), while for the staging copies (stage_window_rows, the full-pel and sub-pel windows) byte-aligned dword / dwordx4
// loads turned out as fast as aligned ones and are used there.  The centre check keeps this form (no measurable difference).
// The aligned address is rebuilt from an integer; the pointer type carries the GLOBAL address space explicitly, otherwise the
// integer-to-pointer cast yields a generic pointer and every load becomes a flat_load (which also ties the loads to the LDS
// counter).  Everything these helpers read lives in the picture pool.
typedef const __attribute__((address_space(1))) uint32_t gmem_u32;
struct __attribute__((packed, aligned(1))) gmem_u32x4 { uint32_t v[4]; };  // 16 bytes at byte alignment: one global_load_dwordx4

__device__ __forceinline__ uint32_t ldu32(const uint8_t* p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    gmem_u32* q = (gmem_u32*)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u);
    const uint32_t lo = q[0];
    const uint32_t hi = sh ? q[1] : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// branch-free variant for bulk copies: always reads both aligned dwords (up to 7 bytes past p: pool slack)
__device__ __forceinline__ uint32_t ldu32_nb(const uint8_t* p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    gmem_u32* q = (gmem_u32*)(a & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(a & 3u));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// Optional per-phase time stamps (tools/hme_stamps_probe.py builds a copy of the library with -DSVTHIP_HME_STAMPS): s_memtime of lane 0 of
// every region wave at the phase boundaries of hme_center_sb, [superblock][wave][8].
// Wave priority around the latency-bound pieces (source staging, centre checks, window copies); -DSVTHIP_HME_NO_PRIO for A/B timing.
#ifdef SVTHIP_HME_NO_PRIO
#define HME_PRIO(p) do { } while (0)
#else
#define HME_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
#ifdef SVTHIP_HME_STAMPS
__device__ unsigned long long g_hme_stamps[8192 * 4 * 8];
#define HME_STAMP(i)                                                                                                          \
    do {                                                                                                                      \
        if (lane == 0 && sbi < 8192u) g_hme_stamps[((size_t)sbi * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// sub-phases of wave_sad_loop_lds, summed over all waves of the launches since load: [W == 64][staging, search loop, tail, calls]
__device__ unsigned long long g_hme_loop_phase[2 * 4];
#define HME_LOOP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define HME_LOOP_ADD(slot, t1, t0)                                                                                \
    do {                                                                                                          \
        if (lane == 0) atomicAdd(&g_hme_loop_phase[(W == 64 ? 4 : 0) + (slot)], (unsigned long long)((t1) - (t0))); \
    } while (0)
#else
#define HME_STAMP(i) do { } while (0)
#define HME_LOOP_T(var) do { } while (0)
#define HME_LOOP_ADD(slot, t1, t0) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        unsigned long long o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

// SAD of a W x H block (strides already doubled by the caller), computed by one wave.  W need not be a
// multiple of 4: the tail dword is masked on both operands.
__device__ uint32_t wave_block_sad(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride,
                                   uint32_t H, uint32_t W, int lane, const uint32_t* src_lds = nullptr)
{
    if (src_lds && W == 64 && H == 32) {
        // the 64 x 32-row source block is already in LDS ([32][16] dwords): only the reference rows come from memory
        const uint32_t c = (uint32_t)lane & 15u, r0 = (uint32_t)lane >> 4;
        const uint8_t* rp = ref + (size_t)r0 * ref_stride + 4u * c;
        uint32_t tv[8], acc4 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) tv[k] = ldu32_nb(rp + (size_t)(4 * k) * ref_stride);
#pragma unroll
        for (int k = 0; k < 8; k++) acc4 = __builtin_amdgcn_sad_u8(src_lds[(r0 + 4 * k) * 16 + c], tv[k], acc4);
        return wave_sum_u32(acc4);
    }
    if (W == 64 && (H & 3u) == 0) {
        // full-width block: lane = (row mod 4, dword column), four rows per pass, all loads of a lane in flight together;
        // src is dword aligned (SB origin multiple of 64), ref is re-aligned from aligned pairs
        const uint32_t c = (uint32_t)lane & 15u, r0 = (uint32_t)lane >> 4;
        const uint8_t* sp = src + (size_t)r0 * src_stride + 4u * c;
        const uint8_t* rp = ref + (size_t)r0 * ref_stride + 4u * c;
        uint32_t acc4 = 0;
        for (uint32_t r = 0; r < H; r += 16) {
            uint32_t sv[4], tv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool in = r + 4u * (uint32_t)k < H;  // uniform
                sv[k] = in ? *reinterpret_cast<const uint32_t*>(sp + (size_t)(4 * k) * src_stride) : 0u;
                tv[k] = in ? ldu32_nb(rp + (size_t)(4 * k) * ref_stride) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acc4 = __builtin_amdgcn_sad_u8(sv[k], tv[k], acc4);
            sp += (size_t)16 * src_stride;
            rp += (size_t)16 * ref_stride;
        }
        return wave_sum_u32(acc4);
    }
    const uint32_t ndw = (W + 3) >> 2;
    const uint32_t n = H * ndw;
    uint32_t acc = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t r = i / ndw, c = i - r * ndw;
        uint32_t s = ldu32(src + (size_t)r * src_stride + 4 * c);
        uint32_t t = ldu32(ref + (size_t)r * ref_stride + 4 * c);
        const uint32_t rem = W - 4 * c;
        if (rem < 4) {
            const uint32_t m = (1u << (8 * rem)) - 1u;
            s &= m;
            t &= m;
        }
        acc = __builtin_amdgcn_sad_u8(s, t, acc);
    }
    return wave_sum_u32(acc);
}

// Generic SadLoopKernel by one wave straight from global memory (any block shape; used for partial SBs).
__device__ void wave_sad_loop_generic(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride,
                                      uint32_t H, uint32_t W, uint32_t ref_stride_raw, int sw, int sh, int lane,
                                      uint32_t* best_sad, int* bx, int* by)
{
    const int npos = sw * sh;
    unsigned long long best = ~0ull;
    const uint32_t ndw = (W + 3) >> 2;
    const uint32_t tail = W & 3u;
    const uint32_t tailmask = tail ? ((1u << (8 * tail)) - 1u) : 0xffffffffu;
#pragma unroll 1
    for (int pos = lane; pos < npos; pos += 64) {
        const int y = pos / sw, x = pos - y * sw;
        const uint8_t* r0 = ref + (size_t)y * ref_stride_raw + x;
        uint32_t acc = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < H; r++) {
            const uint8_t* sp = src + (size_t)r * src_stride;
            const uint8_t* rp = r0 + (size_t)r * ref_stride;
#pragma unroll 2
            for (uint32_t c = 0; c < ndw; c++) {
                uint32_t s = ldu32(sp + 4 * c);
                uint32_t t = ldu32(rp + 4 * c);
                if (c == ndw - 1) {
                    s &= tailmask;
                    t &= tailmask;
                }
                acc = __builtin_amdgcn_sad_u8(s, t, acc);
            }
        }
        const unsigned long long key = ((unsigned long long)acc << 32) | (uint32_t)pos;
        best = key < best ? key : best;  // a lane visits its positions in raster order
    }
    best = wave_min_u64(best);
    // every lane holds the same minimum: hand it on in scalar registers, so that the origin arithmetic, window clipping and loop bounds
    // of the next level are scalar code instead of vector code under exec masks
    const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)best);
    *best_sad = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(best >> 32));
    *by = (int)(pos / (uint32_t)sw);
    *bx = (int)(pos - (uint32_t)(*by) * (uint32_t)sw);
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Copies `wrows` plane rows of `pitch` dwords each, starting at the (unaligned) address `base`, into LDS (row r at win + r * pitch).
// Reads up to pitch * 4 + 19 bytes per row (the pool's tail slack covers the last row of the last plane).  pitch <= 256 (a band of at
// least 15 rows has to fit the 8 KB slice, so pitch <= 136 here).
__device__ __forceinline__ void stage_window_rows(const uint8_t* base, uint32_t ref_stride_raw, int wrows, int pitch, uint32_t* win,
                                                  int lane)
{
    // A lane moves FOUR consecutive dwords of a row: one 16-byte global load at the row's own byte alignment (no alignment needed on
    // this target; round 2 used five aligned dwords + four v_alignbyte) and four LDS stores.  lpr lanes per row, rpp rows per wave
    // pass, NP passes in flight: the level-1 / level-2 windows (46 / 70 rows) are then ONE round trip to memory instead of two / three
    // -- these levels spend most of their wall time waiting for this copy (tools/hme_stamps_probe.py).
    constexpr int NP = 6;
    const int lpr = (pitch + 3) >> 2;
    const int rpp = lpr < 64 ? 64 / lpr : 1;
    const int g = (int)(((uint32_t)lane * ((1u << 16) / (uint32_t)lpr + 1u)) >> 16), q4 = 4 * (lane - g * lpr);
    const bool lane_ok = g < rpp;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(base) + 4u * (uint32_t)q4;
    for (int r0 = g; r0 < wrows; r0 += NP * rpp) {
        gmem_u32x4 w[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int r = min(r0 + u * rpp, wrows - 1);  // clamped: the loads stay unconditional and in flight together
            const __attribute__((address_space(1))) gmem_u32x4* q = (const __attribute__((address_space(1))) gmem_u32x4*)(a0 + (size_t)r * ref_stride_raw);
#pragma unroll
            for (int k = 0; k < 4; k++) w[u].v[k] = q->v[k];
        }
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int r = r0 + u * rpp;
            if (lane_ok && r < wrows) {
                uint32_t* o = win + r * pitch + q4;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (q4 + k < pitch) o[k] = w[u].v[k];
            }
        }
    }
}

// SadLoopKernel by one wave for the full-SB block shapes (W = 16 / 32 / 64 px, H rows taken every second plane
// row), LDS-staged:
//   * the W x H source block and a band of the search window are copied to this wave's LDS slice with aligned,
//     coalesced dword loads (the window is re-aligned with v_alignbyte so search column 0 sits on a dword);
//   * an item is 8 horizontally consecutive search positions of one search row; its lanes read W/4 + 2 window
//     dwords per block row and issue 2 * W/4 v_qsad_pk_u16_u8 (4 positions x 4 pixels each);
//   * when a level has fewer than 64 items (HME L1 / L2), RP lanes share an item and split its block rows;
//   * best position: 64-bit key (sad << 32 | raster index), lane-local strict min, then a wave min.
// `lds` is this wave's private slice of `lds_bytes` bytes; the search area is processed in bands of rows that fit.
// `shared_src` (may be null): the H x W source block already staged in LDS by the workgroup ([H][W / 4] dwords, shared by the
// four region waves); then the whole slice holds the window.
// KEY32: the caller guarantees sw * sh <= 4096 (the reference's level-1 / level-2 areas are 256 / 64 positions) and every W x H here has
// SAD < 2^20, so the best position is a 32-bit minimum of sad << 12 | position instead of a 64-bit one.
template <int W, bool KEY32 = false>
__device__ void wave_sad_loop_lds(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride_raw,
                                  int H, int sw, int sh, int lane, uint8_t* lds, int lds_bytes, uint32_t* best_sad,
                                  int* bx, int* by, const uint32_t* shared_src = nullptr)
{
    constexpr int WD = W / 4;                 // source dwords per block row
    constexpr int FLUSH = (W == 16) ? 16 : (W == 32 ? 8 : 4);  // rows a u16 accumulator can take: rows*W*255 < 65536
    const int own_src = shared_src ? 0 : H * WD;
    const uint32_t* srcbuf = shared_src ? shared_src : reinterpret_cast<const uint32_t*>(lds);  // [H][WD]
    const int noct = (sw + 7) >> 3;
    const int pitch = 2 * noct + WD + 1;      // window dwords per row (+1: odd pitch spreads rows over banks)
    uint32_t* win = reinterpret_cast<uint32_t*>(lds) + own_src;
    const int avail_rows = (lds_bytes / 4 - own_src) / pitch;
    int band = avail_rows - (2 * H - 2);      // search rows per band
    if (band > sh) band = sh;
    if (band < 1) {  // window row wider than the slice (cannot happen for the reference's parameter ranges)
        wave_sad_loop_generic(src, src_stride, ref, ref_stride_raw * 2, (uint32_t)H, (uint32_t)W, ref_stride_raw, sw, sh, lane,
                              best_sad, bx, by);
        return;
    }

    // source block -> LDS (rows are src_stride apart, already the doubled stride); 4 dwords per lane in flight
    for (int i0 = 0; !shared_src && i0 < H * WD; i0 += 256) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = min(i0 + k * 64 + lane, H * WD - 1);  // clamped: no branches, loads stay in flight together
            const int r = i / WD, c = i - r * WD;                // WD is a power of two: shifts
            v[k] = ldu32_nb(src + (uint32_t)r * src_stride + 4u * (uint32_t)c);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) reinterpret_cast<uint32_t*>(lds)[min(i0 + k * 64 + lane, H * WD - 1)] = v[k];
    }

    // items and row-parts
    const int items_row = noct;
    unsigned long long best = ~0ull;
    uint32_t best32 = 0xffffffffu;

    for (int y0 = 0; y0 < sh; y0 += band) {
        const int bh = min(band, sh - y0);
        const int wrows = bh + 2 * H - 2;
        HME_LOOP_T(t_a);
        HME_PRIO(3);  // the window copy is a memory round trip the search below waits for: issue it ahead of the other waves' search loops
        stage_window_rows(ref + (size_t)y0 * ref_stride_raw, ref_stride_raw, wrows, pitch, win, lane);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HME_PRIO(0);
        HME_LOOP_T(t_b);
        HME_LOOP_ADD(0, t_b, t_a);

        const int nitems = items_row * bh;
        int RP = 1, rp_shift = 0;
        while (RP < 8 && nitems * RP * 2 <= 64 && RP * 2 <= H) { RP <<= 1; rp_shift++; }
        const int ipw = 64 >> rp_shift;       // items per wave pass
        const int part = lane & (RP - 1);
        const uint32_t inv_items = (1u << 20) / (uint32_t)items_row + 1u;
        for (int it0 = 0; it0 < nitems; it0 += ipw) {
            const int item = it0 + (lane >> rp_shift);  // RP is a power of two: no per-lane integer division
            const bool valid = item < nitems;
            const int iy = valid ? (int)(((uint32_t)item * inv_items) >> 20) : 0;
            const int io = valid ? item - iy * items_row : 0;
            uint32_t sad[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sad[j] = 0;
            uint64_t acc0 = 0, acc1 = 0;
            int since = 0;
            for (int r = part; r < H; r += RP) {
                const uint32_t* wr = win + (iy + 2 * r) * pitch + 2 * io;
                const uint32_t* sr = srcbuf + r * WD;
                uint32_t d[WD + 2];
#pragma unroll
                for (int c = 0; c < WD + 2; c++) d[c] = wr[c];
#pragma unroll
                for (int c = 0; c < WD; c++) {
                    const uint32_t sv = sr[c];
                    acc0 = __builtin_amdgcn_qsad_pk_u16_u8(pack64(d[c], d[c + 1]), sv, acc0);
                    acc1 = __builtin_amdgcn_qsad_pk_u16_u8(pack64(d[c + 1], d[c + 2]), sv, acc1);
                }
                if (++since == FLUSH) {
                    since = 0;
                    sad[0] += (uint32_t)acc0 & 0xffffu; sad[1] += ((uint32_t)acc0) >> 16;
                    sad[2] += (uint32_t)(acc0 >> 32) & 0xffffu; sad[3] += (uint32_t)(acc0 >> 48);
                    sad[4] += (uint32_t)acc1 & 0xffffu; sad[5] += ((uint32_t)acc1) >> 16;
                    sad[6] += (uint32_t)(acc1 >> 32) & 0xffffu; sad[7] += (uint32_t)(acc1 >> 48);
                    acc0 = acc1 = 0;
                }
            }
            sad[0] += (uint32_t)acc0 & 0xffffu; sad[1] += ((uint32_t)acc0) >> 16;
            sad[2] += (uint32_t)(acc0 >> 32) & 0xffffu; sad[3] += (uint32_t)(acc0 >> 48);
            sad[4] += (uint32_t)acc1 & 0xffffu; sad[5] += ((uint32_t)acc1) >> 16;
            sad[6] += (uint32_t)(acc1 >> 32) & 0xffffu; sad[7] += (uint32_t)(acc1 >> 48);
            // sum the row-parts of an item (its RP lanes are consecutive)
            for (int m = 1; m < RP; m <<= 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) sad[j] += __shfl_xor(sad[j], m);
            }
            if (valid && part == 0) {
                const int ys = y0 + iy;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int xs = 8 * io + j;
                    if (xs < sw) {
                        if (KEY32) {
                            best32 = min(best32, (sad[j] << 12) | (uint32_t)(ys * sw + xs));
                        } else {
                            const unsigned long long key = ((unsigned long long)sad[j] << 32) | (uint32_t)(ys * sw + xs);
                            best = key < best ? key : best;  // raster order within the lane: strict '<' keeps the first
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        HME_LOOP_T(t_c);
        HME_LOOP_ADD(1, t_c, t_b);
    }
    HME_LOOP_T(t_d);
    if (KEY32) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) best32 = min(best32, (uint32_t)__shfl_xor((int)best32, m));
        best = ((unsigned long long)(best32 >> 12) << 32) | (best32 & 0xfffu);
    } else {
        best = wave_min_u64(best);
    }
    const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)best);  // wave-uniform: scalar from here on
    *best_sad = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(best >> 32));
    *by = (int)(pos / (uint32_t)sw);
    *bx = (int)(pos - (uint32_t)(*by) * (uint32_t)sw);
    HME_LOOP_T(t_e);
    HME_LOOP_ADD(2, t_e, t_d);
    HME_LOOP_ADD(3, 1ull, 0ull);
}

// SadLoopKernel of HME level 0 (16 x 8-row block on the 1/16 plane; ~70 % of the search-centre work), one wave.
// An item is 16 consecutive search positions of one search row = 16 bytes = one ds_read_b128 step along the window, so lane
// i of a row reads window dwords 4i .. 4i+7 with two conflict-free b128 loads per block row and issues 16 v_qsad_pk_u16_u8
// (4 source dwords x 4 position groups); the 8 block rows are fully unrolled so every LDS offset is an immediate or a
// scalar multiple of the pitch.  A 16x8 SAD is < 2^16 and the area has < 2^16 positions, so the best position is a 32-bit
// (sad << 16 | raster index) minimum, which is the reference's strict-'<' raster rule.
__device__ void wave_sad_loop_l0(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride_raw, int sw, int sh,
                                 int lane, uint8_t* lds, int lds_bytes, uint32_t* best_sad, int* bx, int* by,
                                 const uint32_t* shared_src = nullptr)
{
    constexpr int H = 8, WD = 4;
    const int own_src = shared_src ? 0 : H * WD;
    const uint32_t* srcbuf = shared_src ? shared_src : reinterpret_cast<const uint32_t*>(lds);  // [8][4], 16-byte aligned
    const int nit = (sw + 15) >> 4;                       // items per search row
    const int pitch = 4 * nit + 4;                        // window dwords per row (multiple of 4: items stay 16-byte aligned)
    uint32_t* win = reinterpret_cast<uint32_t*>(lds) + own_src;
    const int avail_rows = (lds_bytes / 4 - own_src) / pitch;
    int band = avail_rows - (2 * H - 2);
    if (band > sh) band = sh;
    if (band < 1 || sw * sh > 65536) {
        wave_sad_loop_lds<16>(src, src_stride, ref, ref_stride_raw, H, sw, sh, lane, lds, lds_bytes, best_sad, bx, by, shared_src);
        return;
    }
    if (!shared_src && lane < H * WD)
        reinterpret_cast<uint32_t*>(lds)[lane] = ldu32_nb(src + (uint32_t)(lane >> 2) * src_stride + 4u * (uint32_t)(lane & 3));
    uint32_t best = 0xffffffffu;
    const uint32_t inv_nit = (1u << 20) / (uint32_t)nit + 1u;
    for (int y0 = 0; y0 < sh; y0 += band) {
        const int bh = min(band, sh - y0);
        HME_PRIO(3);
        stage_window_rows(ref + (size_t)y0 * ref_stride_raw, ref_stride_raw, bh + 2 * H - 2, pitch, win, lane);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HME_PRIO(0);
        const int nitems = nit * bh;
        // 16 keys of one item: (sad << 16) + raster index; positions beyond the search width and lanes without an item never win
        auto track = [&](const uint64_t* acc, int iy, int io, bool valid) {
            const int xs0 = 16 * io;
            const uint32_t base = (uint32_t)((y0 + iy) * sw + xs0);
            if ((sw & 15) == 0) {
                // every item is 16 valid positions (wave-uniform case; the reference's default areas): one v_lshl_or / v_and_or per key and
                // one v_min3 per two keys instead of compare + select + add + min per position; only the padding lanes are masked, once
                uint32_t lb = 0xffffffffu;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = (uint32_t)acc[g], hi = (uint32_t)(acc[g] >> 32), ix = base + 4u * (uint32_t)g;
                    lb = min3u((lo << 16) | ix, (lo & 0xffff0000u) | (ix + 1u), lb);
                    lb = min3u((hi << 16) | (ix + 2u), (hi & 0xffff0000u) | (ix + 3u), lb);
                }
                best = min(best, valid ? lb : 0xffffffffu);
                return;
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t lo = (uint32_t)acc[g], hi = (uint32_t)(acc[g] >> 32);
                const uint32_t sad4[4] = {lo << 16, lo & 0xffff0000u, hi << 16, hi & 0xffff0000u};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int xs = xs0 + 4 * g + j;
                    const uint32_t key = (valid && xs < sw) ? (sad4[j] + base + (uint32_t)(4 * g + j)) : 0xffffffffu;
                    best = min(best, key);
                }
            }
        };
        auto row_sads = [&](const uint32_t* wr, int r, uint64_t* acc) {
            const uint4 da = *reinterpret_cast<const uint4*>(wr);
            const uint4 db = *reinterpret_cast<const uint4*>(wr + 4);
            const uint4 sv = *reinterpret_cast<const uint4*>(srcbuf + r * WD);
            const uint32_t d[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
            const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int c = 0; c < WD; c++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    acc[g] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(d[c + g], d[c + g + 1]), s4[c], acc[g]);
        };
        int it0 = 0;
        // whole passes: a lane owns an item (all 8 block rows, 128 v_qsad)
        for (; nitems - it0 >= 57; it0 += 64) {
            const int item = min(it0 + lane, nitems - 1);
            const bool valid = it0 + lane < nitems;
            const int iy = (int)(((uint32_t)item * inv_nit) >> 20), io = item - iy * nit;
            const uint32_t* w0 = win + iy * pitch + 4 * io;
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < H; r++) row_sads(w0 + 2 * r * pitch, r, acc);
            track(acc, iy, io, valid);
        }
        // the remainder: 2 / 4 / 8 lanes share an item and split its block rows, so that a handful of left-over items does not cost a
        // whole 128-v_qsad pass with most lanes idle (the reference's 100 % level-0 area is 72 items: 64 + 8; the 200 % area 288 = 4 x 64 + 32).
        // The 16-bit partial sums of an item's lanes are added as packed pairs (a 16 x 8 SAD is < 2^16).
        while (it0 < nitems) {
            const int rem = nitems - it0;
            const int rp_shift = rem > 32 ? 1 : rem > 16 ? 1 : rem > 8 ? 2 : 3;
            const int RP = 1 << rp_shift, take = min(rem, 64 >> rp_shift);
            const int slot = lane >> rp_shift, part = lane & (RP - 1);
            const int item = it0 + min(slot, take - 1);
            const int iy = (int)(((uint32_t)item * inv_nit) >> 20), io = item - iy * nit;
            const uint32_t* w0 = win + iy * pitch + 4 * io;
            uint64_t acc[4] = {0, 0, 0, 0};
            for (int r = part; r < H; r += RP) row_sads(w0 + 2 * r * pitch, r, acc);
            for (int m = 1; m < RP; m <<= 1) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = (uint32_t)acc[g] + (uint32_t)__shfl_xor((int)(uint32_t)acc[g], m);
                    const uint32_t hi = (uint32_t)(acc[g] >> 32) + (uint32_t)__shfl_xor((int)(uint32_t)(acc[g] >> 32), m);
                    acc[g] = pack64(lo, hi);
                }
            }
            track(acc, iy, io, slot < take && part == 0);
            it0 += take;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) best = min(best, (uint32_t)__shfl_xor((int)best, m));
    best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);  // wave-uniform: scalar from here on
    const uint32_t pos = best & 0xffffu;
    *best_sad = best >> 16;
    *by = (int)(pos / (uint32_t)sw);
    *bx = (int)(pos - (uint32_t)(*by) * (uint32_t)sw);
}

__device__ __forceinline__ void clamp_center(int& x, int& y, int ox, int oy, int pw, int ph)
{
    // int16 semantics of the reference hold: every intermediate fits 16 bits for pictures <= 8K
    x = (ox + x < -63) ? (-63 - ox) : x;
    x = (ox + x > pw - 1) ? (x - ((ox + x) - (pw - 1))) : x;
    y = (oy + y < -63) ? (-63 - oy) : y;
    y = (oy + y > ph - 1) ? (y - ((oy + y) - (ph - 1))) : y;
}

// four statements per axis, each re-reading what the previous wrote (statement 2 never fires), :6690-6723
__device__ __forceinline__ void clip_window(int& xo, int& yo, int& sw, int& sh, int ox, int oy, int padw, int padh,
                                            int pw, int ph)
{
    xo = (ox + xo < -padw) ? (-padw - ox) : xo;
    sw = (ox + xo < -padw) ? (sw - (-padw - (ox + xo))) : sw;
    xo = (ox + xo > pw - 1) ? (xo - ((ox + xo) - (pw - 1))) : xo;
    sw = (ox + xo + sw > pw) ? max(1, sw - ((ox + xo + sw) - pw)) : sw;
    yo = (oy + yo < -padh) ? (-padh - oy) : yo;
    sh = (oy + yo < -padh) ? (sh - (-padh - (oy + yo))) : sh;
    yo = (oy + yo > ph - 1) ? (yo - ((oy + yo) - (ph - 1))) : yo;
    sh = (oy + yo + sh > ph) ? max(1, sh - ((oy + yo + sh) - ph)) : sh;
}

__device__ __forceinline__ int s16(int v) { return (int)(int16_t)v; }

__device__ __forceinline__ int round_hme_width(int w)
{
    return (w < 8) ? 8 : ((w & 7) ? (w + (w - ((w >> 3) << 3))) : w);  // :4528 (adds the remainder, sic)
}

constexpr int kHmeLdsPerWave = 6 * 1024;  // per-wave LDS slice: source block + a band of the search window

struct HmeShared {
    unsigned long long cost[8];   // centre-check candidate costs
    int rx[3][4], ry[3][4];       // per level, per region ([w][h] flattened as w*2+h) centres
    unsigned long long rs[3][4];  // per level SADs (doubled)
    int cx, cy;
    svthip_fullpel_desc desc;     // the final descriptor, for a consumer in the same workgroup
    // source blocks of a full 64x64 SB, staged once for the four region waves: full-res rows 0,2,..,62 (64 x 32), quarter-res
    // rows 0,2,..,30 (32 x 16), sixteenth-res rows 0,2,..,14 (16 x 8)
    __attribute__((aligned(16))) uint32_t src2[32 * 16];
    __attribute__((aligned(16))) uint32_t src1[16 * 8];
    __attribute__((aligned(16))) uint32_t src0[8 * 4];
};


// Whole search-centre chain of superblock `sbi` (origin ox, oy).  Must be called by all 256 threads of the workgroup;
// `sh` and `hme_lds` (4 * kHmeLdsPerWave bytes, 16-byte aligned) are workgroup LDS.  On return (after the caller's next
// barrier) sh.desc holds the descriptor that was also written to out_desc[sbi].
__device__ __forceinline__ void hme_center_sb(const uint8_t* __restrict__ pool, const svthip_pa_picture& cur, const svthip_pa_picture& ref,
                                              const svthip_me_params& P, uint32_t list_index, int ox, int oy, uint32_t sbi,
                                              const uint32_t* __restrict__ l0_best_mv64, uint32_t l0_mv_stride,
                                              svthip_fullpel_desc* __restrict__ out_desc, int16_t* __restrict__ out_center,
                                              int16_t* __restrict__ hme_state, HmeShared& sh, uint8_t* hme_lds)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* wlds = hme_lds + wave * kHmeLdsPerWave;
    const int pw = cur.width, ph = cur.height;
    const uint32_t sb_w = (uint32_t)min(64, pw - ox), sb_h = (uint32_t)min(64, ph - oy);
    const int nw = P.number_hme_search_region_in_width, nh = P.number_hme_search_region_in_height;

    const uint8_t* cur_full = pool + cur.full_offset + (size_t)68 * cur.full_stride + 68;
    const uint8_t* ref_full = pool + ref.full_offset + (size_t)68 * ref.full_stride + 68;
    const uint8_t* src = cur_full + (size_t)oy * cur.full_stride + ox;
    HME_STAMP(0);
    HME_PRIO(3);  // until the first search loop: staging and the centre checks are a few loads and SADs on the serial path of every level

    // full 64x64 SBs: stage the three source blocks once for the whole workgroup (the four region waves search with the same
    // block at every level, and the centre checks compare the same 64 x 32-row block)
    const bool full_sb = sb_w == 64 && sb_h == 64;
    if (full_sb) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + 256 * k;  // 512 dwords of the full-res block, dword aligned (SB origin multiple of 64)
            sh.src2[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)(i >> 4) * (cur.full_stride * 2) + 4 * (i & 15));
        }
        if (tid < 128) {
            const uint8_t* s1 = pool + cur.quarter_offset + (size_t)(32 + (oy >> 1)) * cur.quarter_stride + 32 + (ox >> 1);
            sh.src1[tid] = ldu32_nb(s1 + (size_t)(tid >> 3) * (cur.quarter_stride * 2) + 4 * (tid & 7));
        } else if (tid < 160) {
            const int i = tid - 128;
            const uint8_t* s0 = pool + cur.sixteenth_offset + (size_t)(16 + (oy >> 2)) * cur.sixteenth_stride + 16 + (ox >> 2);
            sh.src0[i] = ldu32_nb(s0 + (size_t)(i >> 2) * (cur.sixteenth_stride * 2) + 4 * (i & 3));
        }
        __syncthreads();
    }
    const uint32_t* src2_lds = full_sb ? sh.src2 : nullptr;
    HME_STAMP(1);

    const bool center_path = (P.temporal_layer_index > 0) || (list_index == 0);  // :6300
    const uint32_t mv64 = (list_index == 1 && l0_best_mv64) ? l0_best_mv64[(size_t)sbi * l0_mv_stride] : 0u;
    const int dx = s16(0 - (s16((int)(mv64 & 0xffffu)) >> 2));
    const int dy = s16(0 - (s16((int)(mv64 >> 16)) >> 2));
    const int tw = P.hme_level0_total_search_area_width, th = P.hme_level0_total_search_area_height;

    if (tid == 0 && hme_state && list_index == 0) hme_state[25 * (size_t)sbi + 24] = 0;
    int xc = 0, yc = 0;
    if (center_path) {
        // ---- hme_mv_center_check (:5882-6145): candidates 0 / B / C / D (+ direct for list 1); A uses the stale
        //      zero-MV index, so its cost equals the zero cost and it can never be selected before it.
        int cxs[5] = {0, tw, 0, 0, dx}, cys[5] = {0, 0, -th, th, dy};
        const int ncand = (list_index == 1) ? 5 : 4;
        for (int c = wave; c < ncand; c += 4) {
            int x = cxs[c], y = cys[c];
            clamp_center(x, y, ox, oy, ref.width, ref.height);
            const uint32_t sad = wave_block_sad(src, cur.full_stride * 2, ref_full + (size_t)(oy + y) * ref.full_stride + ox + x,
                                                ref.full_stride * 2, sb_h >> 1, sb_w, lane, src2_lds);
            if (lane == 0) sh.cost[c] = (unsigned long long)(sad << 1) << 8;
        }
        __syncthreads();
        {
            const unsigned long long zero = sh.cost[0], b = sh.cost[1], c = sh.cost[2], d = sh.cost[3];
            const unsigned long long dir = (list_index == 1) ? sh.cost[4] : 0xFFFFFFFFFFFFFull;
            unsigned long long best = zero;
            best = b < best ? b : best;
            best = c < best ? c : best;
            best = d < best ? d : best;
            best = dir < best ? dir : best;
            if (best == zero) { xc = 0; yc = 0; }            // also covers A (same cost as zero)
            else if (best == b) { xc = tw; yc = 0; }
            else if (best == c) { xc = 0; yc = s16(0 - th); }
            else if (best == dir) { xc = dx; yc = dy; }
            else { xc = 0; yc = th; }
        }

        if (P.enable_hme_flag && sb_h == 64) {  // :6323
            const int nreg = nw * nh;
            const int16_t* st = hme_state ? hme_state + 25 * (size_t)sbi : nullptr;
            const bool carried = st && list_index == 1 && st[24];
            if (wave < nreg) {
                const int rw = wave % nw, rh = wave / nw;  // visiting order h outer, w inner
                const int k = rw * 2 + rh;                 // [w][h] slot
                int x0 = xc, y0 = yc, x1 = xc, y1 = yc, x2 = xc, y2 = yc;
                if (carried) { x0 = st[k]; y0 = st[4 + k]; x1 = st[8 + k]; y1 = st[12 + k]; x2 = st[16 + k]; y2 = st[20 + k]; }
                uint32_t sad0 = 0, sad1 = 0, sad2 = 0;
                HME_STAMP(2);
                if (P.enable_hme_level0_flag) {  // HmeLevel0 :4306-4503, 1/16 picture
                    const uint32_t mx = P.hme_level0_multiplier_x, my = P.hme_level0_multiplier_y;
                    int sw = s16((int)((P.hme_level0_search_area_in_width_array[rw] * mx) / 100));
                    int shh = s16((int)((P.hme_level0_search_area_in_height_array[rh] * my) / 100));
                    int xd = s16(xc >> 2), yd = s16(yc >> 2);
                    for (int j = rw; j > 0;) { j--; xd = s16(xd + s16((int)((P.hme_level0_search_area_in_width_array[j] * mx) / 100))); }
                    for (int j = rh; j > 0;) { j--; yd = s16(yd + s16((int)((P.hme_level0_search_area_in_height_array[j] * my) / 100))); }
                    int xo = s16(-s16((int)(((tw * mx) / 100) >> 1)) + xd);
                    int yo = s16(-s16((int)(((th * my) / 100) >> 1)) + yd);
                    const int o_x = ox >> 2, o_y = oy >> 2;
                    clip_window(xo, yo, sw, shh, o_x, o_y, 15, 15, ref.width >> 2, ref.height >> 2);
                    const uint8_t* s = pool + cur.sixteenth_offset + (size_t)(16 + o_y) * cur.sixteenth_stride + 16 + o_x;
                    const uint8_t* r = pool + ref.sixteenth_offset + (size_t)(16 + o_y + yo) * ref.sixteenth_stride + 16 + o_x + xo;
                    int bx, by;
                    if (sb_w == 64)
                        wave_sad_loop_l0(s, cur.sixteenth_stride * 2, r, ref.sixteenth_stride, sw, shh, lane, wlds, kHmeLdsPerWave, &sad0,
                                         &bx, &by, sh.src0);
                    else
                        wave_sad_loop_generic(s, cur.sixteenth_stride * 2, r, ref.sixteenth_stride * 2, (sb_h >> 2) >> 1, sb_w >> 2,
                                              ref.sixteenth_stride, sw, shh, lane, &sad0, &bx, &by);
                    x0 = s16(s16(bx + xo) * 4);
                    y0 = s16(s16(by + yo) * 4);
                }
                HME_STAMP(3);
                if (P.enable_hme_level1_flag) {  // HmeLevel1 :4505-4625, 1/4 picture
                    int sw = round_hme_width((int)(int16_t)P.hme_level1_search_area_in_width_array[rw]);
                    int shh = (int)(int16_t)P.hme_level1_search_area_in_height_array[rh];
                    int xo = s16(-(sw >> 1) + (x0 >> 1)), yo = s16(-(shh >> 1) + (y0 >> 1));
                    const int o_x = ox >> 1, o_y = oy >> 1;
                    clip_window(xo, yo, sw, shh, o_x, o_y, 31, 31, ref.width >> 1, ref.height >> 1);
                    const uint8_t* s = pool + cur.quarter_offset + (size_t)(32 + o_y) * cur.quarter_stride + 32 + o_x;
                    const uint8_t* r = pool + ref.quarter_offset + (size_t)(32 + o_y + yo) * ref.quarter_stride + 32 + o_x + xo;
                    int bx, by;
                    if (sb_w == 64 && sw * shh <= 4096)
                        wave_sad_loop_lds<32, true>(s, cur.quarter_stride * 2, r, ref.quarter_stride, 16, sw, shh, lane, wlds,
                                                    kHmeLdsPerWave, &sad1, &bx, &by, sh.src1);
                    else if (sb_w == 64)
                        wave_sad_loop_lds<32>(s, cur.quarter_stride * 2, r, ref.quarter_stride, 16, sw, shh, lane, wlds,
                                              kHmeLdsPerWave, &sad1, &bx, &by, sh.src1);
                    else
                        wave_sad_loop_generic(s, cur.quarter_stride * 2, r, ref.quarter_stride * 2, (sb_h >> 1) >> 1, sb_w >> 1,
                                              ref.quarter_stride, sw, shh, lane, &sad1, &bx, &by);
                    x1 = s16(s16(bx + xo) * 2);
                    y1 = s16(s16(by + yo) * 2);
                }
                HME_STAMP(4);
                if (P.enable_hme_level2_flag) {  // HmeLevel2 :4627-4758, full resolution
                    int sw = round_hme_width((int)(int16_t)P.hme_level2_search_area_in_width_array[rw]);
                    int shh = (int)(int16_t)P.hme_level2_search_area_in_height_array[rh];
                    int xo = s16(-(sw >> 1) + x1), yo = s16(-(shh >> 1) + y1);
                    clip_window(xo, yo, sw, shh, ox, oy, 63, 63, ref.width, ref.height);
                    const uint8_t* r = ref_full + (size_t)(oy + yo) * ref.full_stride + ox + xo;
                    int bx, by;
                    if (sb_w == 64 && sw * shh <= 4096)
                        wave_sad_loop_lds<64, true>(src, cur.full_stride * 2, r, ref.full_stride, 32, sw, shh, lane, wlds,
                                                    kHmeLdsPerWave, &sad2, &bx, &by, sh.src2);
                    else if (sb_w == 64)
                        wave_sad_loop_lds<64>(src, cur.full_stride * 2, r, ref.full_stride, 32, sw, shh, lane, wlds,
                                              kHmeLdsPerWave, &sad2, &bx, &by, sh.src2);
                    else
                        wave_sad_loop_generic(src, cur.full_stride * 2, r, ref.full_stride * 2, sb_h >> 1, sb_w, ref.full_stride, sw,
                                              shh, lane, &sad2, &bx, &by);
                    x2 = s16(bx + xo);
                    y2 = s16(by + yo);
                }
                HME_STAMP(5);
                if (lane == 0) {
                    sh.rx[0][k] = x0; sh.ry[0][k] = y0; sh.rs[0][k] = (unsigned long long)sad0 * 2;
                    sh.rx[1][k] = x1; sh.ry[1][k] = y1; sh.rs[1][k] = (unsigned long long)sad1 * 2;
                    sh.rx[2][k] = x2; sh.ry[2][k] = y2; sh.rs[2][k] = (unsigned long long)sad2 * 2;
                }
            }
            __syncthreads();
            if (tid == 0) {
                // region pick (:6510-6631): start at [0][0], then w-inner order from w = 1, strict '<'
                int lvl = -1;
                if (P.enable_hme_level0_flag && !P.enable_hme_level1_flag && !P.enable_hme_level2_flag) lvl = 0;
                if (P.enable_hme_level1_flag && !P.enable_hme_level2_flag) lvl = 1;
                if (P.enable_hme_level2_flag) lvl = 2;
                int xh = 0, yh = 0;
                if (lvl >= 0) {
                    xh = sh.rx[lvl][0]; yh = sh.ry[lvl][0];
                    unsigned long long bs = sh.rs[lvl][0];
                    int w = 1, h = 0;
                    while (h < nh) {
                        while (w < nw) {
                            const int k = w * 2 + h;
                            if (sh.rs[lvl][k] < bs) { xh = sh.rx[lvl][k]; yh = sh.ry[lvl][k]; bs = sh.rs[lvl][k]; }
                            w++;
                        }
                        w = 0;
                        h++;
                    }
                }
                if (P.enable_hme_level2_flag) {
                    const int total = nh * nw;
                    if (P.ref_poc_equal && list_index == 1 && total > 1) {
                        // bubble sort by SAD with the reference's [q / nw][q % nw] indexing, then take [0][1] (:6606-6631)
                        for (int q = 0; q < total - 1; q++)
                            for (int n = q + 1; n < total; n++) {
                                const int a = (q / nw) * 2 + (q % nw), b = (n / nw) * 2 + (n % nw);
                                if (sh.rs[2][a] > sh.rs[2][b]) {
                                    const int tx = sh.rx[2][a], ty = sh.ry[2][a];
                                    const unsigned long long ts = sh.rs[2][a];
                                    sh.rx[2][a] = sh.rx[2][b]; sh.ry[2][a] = sh.ry[2][b]; sh.rs[2][a] = sh.rs[2][b];
                                    sh.rx[2][b] = tx; sh.ry[2][b] = ty; sh.rs[2][b] = ts;
                                }
                            }
                        xh = sh.rx[2][1];
                        yh = sh.ry[2][1];
                    }
                }
                sh.cx = xh;
                sh.cy = yh;
                if (hme_state) {
                    int16_t* so = hme_state + 25 * (size_t)sbi;
                    for (int k = 0; k < 4; k++) {
                        // regions that do not exist keep the initial centre, like the reference's arrays
                        const bool live = ((k >> 1) < nw) && ((k & 1) < nh);
                        for (int l = 0; l < 3; l++) {
                            so[8 * l + k] = (int16_t)(live ? sh.rx[l][k] : (carried ? so[8 * l + k] : 0));
                            so[8 * l + 4 + k] = (int16_t)(live ? sh.ry[l][k] : (carried ? so[8 * l + 4 + k] : 0));
                        }
                    }
                    so[24] = 1;
                }
            }
            __syncthreads();
            xc = sh.cx;
            yc = sh.cy;
        }
    }

    // ---- CheckZeroZeroCenter (:5466-5552) ----
    HME_PRIO(3);
    if ((xc != 0 || yc != 0) && P.is_used_as_reference_flag) {
        clamp_center(xc, yc, ox, oy, ref.width, ref.height);
        __syncthreads();
        if (wave < 2) {
            const int x = wave ? xc : 0, y = wave ? yc : 0;
            const uint32_t sad = wave_block_sad(src, cur.full_stride * 2, ref_full + (size_t)(oy + y) * ref.full_stride + ox + x,
                                                ref.full_stride * 2, sb_h >> 1, sb_w, lane, src2_lds);
            if (lane == 0) sh.cost[6 + wave] = (unsigned long long)(sad << 1) << 8;
        }
        __syncthreads();
        const unsigned long long z = sh.cost[6], hcost = sh.cost[7];
        const unsigned long long m = z < hcost ? z : hcost;
        if (m == z) { xc = 0; yc = 0; }
    }

    HME_STAMP(6);
    if (tid == 0) {
        int sw = min((int)P.search_area_width, 127), shh = min((int)P.search_area_height, 127);
        int xo = s16(xc - (sw >> 1)), yo = s16(yc - (shh >> 1));
        clip_window(xo, yo, sw, shh, ox, oy, 63, 63, pw, ph);
        svthip_fullpel_desc d;
        d.src_offset = (int32_t)(cur.full_offset + (int64_t)(68 + oy) * cur.full_stride + 68 + ox);
        d.ref_offset = (int32_t)(ref.full_offset + (int64_t)(68 + oy + yo) * ref.full_stride + 68 + ox + xo);
        d.x_search_area_origin = xo;
        d.y_search_area_origin = yo;
        d.search_area_width = sw;
        d.search_area_height = shh;
        out_desc[sbi] = d;
        sh.desc = d;
        if (out_center) {
            out_center[2 * sbi] = (int16_t)xc;
            out_center[2 * sbi + 1] = (int16_t)yc;
        }
    }
}
