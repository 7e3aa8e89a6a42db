// svt-av1-1_amd/csrc/me_sadloop.hip
//
// SadLoopKernel for a batch of blocks, gfx950: exhaustive SAD search of one W x H block over a search_area_width x
// search_area_height grid, first minimum in raster order (strict '<').  Replaces NxMSadLoopKernel_funcPtrArray[asm_type]
// (Source/Lib/Codec/EbComputeSAD.h:183-189) = SadLoopKernel (C_DEFAULT/EbComputeSAD_C.c:73-119), SadLoopKernel_SSE4_1_INTRIN /
// _AVX2_INTRIN, the kernel behind HmeLevel0/1/2 (Codec/EbMotionEstimation.c:4306-4758) and BASELINE configs[0]
// (16x16 blocks, +-16, 856x480).  The hierarchical-ME kernel (me_hme_impl.h) has its own specialised copies for the three HME
// block shapes; this is the generic entry with the reference's full argument set (row-skipping strides included).
//
// One wave per block, four blocks per 256-thread workgroup.  The block and its reference window live in a per-wave LDS slice; a lane
// owns search positions p = lane, lane + 64, ... (raster order); per 4 block pixels: one broadcast ds_read of the source dword, an
// unaligned window dword (two aligned reads + v_alignbyte) and one v_sad_u8.  Best = min over (sad << 12 | p): the first minimum in
// raster order, exactly the reference's strict-'<' update; DPP / readlane reduction over the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace svthip {

namespace {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t lds_u32_at(const lds_u8* p)
{
    const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(p);
    const lds_u32* q = reinterpret_cast<const lds_u32*>((uintptr_t)(a & ~3u));
    return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
}

}  // namespace

__global__ void __launch_bounds__(256) sad_loop_kernel(const uint8_t* __restrict__ src, uint32_t src_stride, const uint8_t* __restrict__ ref,
                                                       uint32_t ref_stride, uint32_t ref_stride_raw, const svthip_sad_loop_desc* __restrict__ desc,
                                                       uint32_t n_blocks, int w, int h, int sw, int sh, int slice_bytes,
                                                       uint32_t* __restrict__ best_sad, int16_t* __restrict__ best_xy)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 4 + wave;
    if (b >= n_blocks) return;  // whole wave; no workgroup barrier below
    lds_u8* blk = (lds_u8*)smem + wave * slice_bytes;  // [h][w] source block, then the window
    lds_u8* win = blk + h * w;
    const int k = (int)(ref_stride / ref_stride_raw);                // rows of the plane between two block rows (1, or 2 when rows are skipped)
    const int wrows = (sh - 1) + (h - 1) * k + 1, wcols = w + sw - 1;
    const int pitch = (wcols + 3 + 4) & ~3;                          // +4: the unaligned dword read of the last columns
    const svthip_sad_loop_desc d = desc[b];
    const int w4 = w >> 2;
    for (int i = lane; i < h * w4; i += 64) {
        const int y = i / w4, x = 4 * (i - y * w4);
        const uint8_t* p = src + d.src_offset + (size_t)y * src_stride + x;
        reinterpret_cast<lds_u32*>(blk)[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    {
        const uint8_t* base = ref + d.ref_offset;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(base);
        const uint32_t shf = (uint32_t)(a0 & 3u);
        const int ndw = pitch >> 2;
        for (int i = lane; i < wrows * ndw; i += 64) {
            const int r = i / ndw, c = i - r * ndw;
            // aligned dword pair of row r (the plane's raw stride may be odd: re-derive the alignment per row)
            const uintptr_t a = a0 + (size_t)r * ref_stride_raw + 4 * c;
            const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
            reinterpret_cast<lds_u32*>(win)[i] = 4 * c < wcols + 3 ? __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(a & 3u)) : 0u;
            (void)shf;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint32_t best = 0xffffffffu;
    const int n_pos = sw * sh;
    for (int p = lane; p < n_pos; p += 64) {
        const int ys = p / sw, xs = p - ys * sw;
        uint32_t sad = 0;
        const lds_u8* wp = win + ys * pitch + xs;
        for (int y = 0; y < h; y++) {
            const lds_u32* srow = reinterpret_cast<const lds_u32*>(blk + y * w);
            const lds_u8* rrow = wp + y * k * pitch;
#pragma unroll 4
            for (int x4 = 0; x4 < w4; x4++) sad = __builtin_amdgcn_sad_u8(srow[x4], lds_u32_at(rrow + 4 * x4), sad);
        }
        const uint32_t key = (sad << 12) | (uint32_t)p;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)best, m);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        const int p = (int)(best & 0xfffu);
        best_sad[b] = best >> 12;
        best_xy[2 * b] = (int16_t)(p % sw);
        best_xy[2 * b + 1] = (int16_t)(p / sw);
    }
}

__host__ __device__ inline int sad_loop_qsad_pitch(int w, int sw)
{
    const int p = (8 * ((sw + 7) >> 3) + w + 4 + 7) & ~7;
    return (p & 8) ? p : p + 8;
}

// Fast path for block widths 4 / 8 / 16 / 32 / 64 (W4 = width / 4 dwords): same slice layout and staging, but a lane owns EIGHT
// consecutive positions of one search row and the window dwords of a block row live in registers: per source dword two
// v_qsad_pk_u16_u8 (positions 0..3 on the dword pair (j, j + 1), 4..7 on (j + 1, j + 2)) -- 16 abs-diff per instruction instead of 4.
// The packed 16-bit sums are widened every 256 / width rows (a row adds at most width / 4 * 1020 per position).
template <int W4>
__global__ void __launch_bounds__(256) sad_loop_qsad_kernel(const uint8_t* __restrict__ src, uint32_t src_stride, const uint8_t* __restrict__ ref,
                                                            uint32_t ref_stride, uint32_t ref_stride_raw,
                                                            const svthip_sad_loop_desc* __restrict__ desc, uint32_t n_blocks, int h, int sw, int sh,
                                                            int slice_bytes, uint32_t* __restrict__ best_sad, int16_t* __restrict__ best_xy)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int w = 4 * W4;
    constexpr int FL = 256 / w;  // rows between two widenings
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 4 + wave;
    if (b >= n_blocks) return;  // whole wave; no workgroup barrier below
    lds_u8* blk = (lds_u8*)smem + wave * slice_bytes;
    lds_u8* win = blk + ((h * w + 7) & ~7);  // 8-byte aligned for the paired reads
    const int k = (int)(ref_stride / ref_stride_raw);
    const int ng = (sw + 7) >> 3;                                      // 8-position groups per search row
    const int wrows = (sh - 1) + (h - 1) * k + 1, wcols = w + sw - 1;
    // every group's W4 + 2 dwords (+ 1: they are read as 8-byte pairs) exist (zero beyond the window); rows are an ODD number of 8-byte
    // units so that the ds_read_b64 of lanes on neighbouring search rows fall on different banks
    const int pitch = sad_loop_qsad_pitch(w, sw);
    const svthip_sad_loop_desc d = desc[b];
#pragma unroll 2
    for (int i = lane; i < h * W4; i += 64) {
        const int y = i / W4, x = 4 * (i - y * W4);
        const uintptr_t a = reinterpret_cast<uintptr_t>(src + d.src_offset + (size_t)y * src_stride + x);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        const uint32_t hi = (a & 3u) ? q[1] : 0u;  // never touches a dword that holds no block byte
        reinterpret_cast<lds_u32*>(blk)[i] = __builtin_amdgcn_alignbyte(hi, q[0], (uint32_t)(a & 3u));
    }
    {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(ref + d.ref_offset);
        const int ndw = pitch >> 2;
        const uint32_t inv = (1u << 20) / (uint32_t)ndw + 1u;  // i / ndw for i < 2^14 (a slice is at most 16 KB): the emulated division was
                                                               // ~20 of the ~35 vector instructions of a staging pass (round 3)
#pragma unroll 4
        for (int i = lane; i < wrows * ndw; i += 64) {
            const int r = (int)(((uint32_t)i * inv) >> 20), c = i - r * ndw;
            const uintptr_t a = a0 + (size_t)r * ref_stride_raw + 4 * c;
            const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
            reinterpret_cast<lds_u32*>(win)[i] = 4 * c < wcols + 3 ? __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(a & 3u)) : 0u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint32_t best = 0xffffffffu;
    const int n_items = ng * sh;
    const uint32_t inv_ng = (1u << 20) / (uint32_t)ng + 1u;  // it / ng for it < 4096 * 8
    for (int it = lane; it < n_items; it += 64) {
        const int ys = (int)(((uint32_t)it * inv_ng) >> 20), x0 = 8 * (it - ys * ng);
        uint32_t sum[8];
#pragma unroll
        for (int i = 0; i < 8; i++) sum[i] = 0;
        const lds_u32* wp = reinterpret_cast<const lds_u32*>(win + ys * pitch + x0);
        const int rstep = (k * pitch) >> 2;
        for (int y0 = 0; y0 < h; y0 += FL) {
            uint64_t a0 = 0, a1 = 0;
            const int y1 = min(h, y0 + FL);
            for (int y = y0; y < y1; y++) {
                const lds_u32* srow = reinterpret_cast<const lds_u32*>(blk) + y * W4;
                const lds_u32* rrow = wp + y * rstep;
                uint32_t W[W4 + 3];
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
#pragma unroll
                for (int j = 0; j < (W4 + 3) / 2; j++) {  // 8-byte aligned: x0 is a multiple of 8 and so is the pitch
                    const u32x2 t = reinterpret_cast<const lds_u32x2*>(rrow)[j];
                    W[2 * j] = t.x;
                    W[2 * j + 1] = t.y;
                }
#pragma unroll
                for (int j = 0; j < W4; j++) {
                    const uint32_t sj = srow[j];
                    a0 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)W[j + 1] << 32) | W[j], sj, a0);
                    a1 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)W[j + 2] << 32) | W[j + 1], sj, a1);
                }
            }
            sum[0] += (uint32_t)a0 & 0xffffu;
            sum[1] += ((uint32_t)a0) >> 16;
            sum[2] += (uint32_t)(a0 >> 32) & 0xffffu;
            sum[3] += (uint32_t)(a0 >> 48);
            sum[4] += (uint32_t)a1 & 0xffffu;
            sum[5] += ((uint32_t)a1) >> 16;
            sum[6] += (uint32_t)(a1 >> 32) & 0xffffu;
            sum[7] += (uint32_t)(a1 >> 48);
        }
        const uint32_t p0 = (uint32_t)(ys * sw + x0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t key = (x0 + i < sw) ? ((sum[i] << 12) | (p0 + i)) : 0xffffffffu;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)best, m);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        const int p = (int)(best & 0xfffu);
        best_sad[b] = best >> 12;
        best_xy[2 * b] = (int16_t)(p % sw);
        best_xy[2 * b + 1] = (int16_t)(p / sw);
    }
}

size_t sad_loop_qsad_slice_bytes(int w, int h, int sw, int sh, int k)
{
    const int wrows = (sh - 1) + (h - 1) * k + 1, pitch = sad_loop_qsad_pitch(w, sw);
    return ((((size_t)h * w + 7) & ~(size_t)7) + (size_t)wrows * pitch + 8 + 15) & ~(size_t)15;
}

hipError_t launch_sad_loop_qsad(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                                const svthip_sad_loop_desc* desc, uint32_t n_blocks, int w, int h, int sw, int sh, int slice_bytes,
                                uint32_t* best_sad, int16_t* best_xy, hipStream_t s)
{
    const dim3 grid((n_blocks + 3) / 4), block(256);
    const size_t lds = (size_t)slice_bytes * 4;
#define SVTHIP_SADLOOP_CASE(W4)                                                                                                             \
    case 4 * W4:                                                                                                                            \
        hipLaunchKernelGGL(sad_loop_qsad_kernel<W4>, grid, block, lds, s, src, src_stride, ref, ref_stride, ref_stride_raw, desc, n_blocks, h, sw, \
                           sh, slice_bytes, best_sad, best_xy);                                                                            \
        break;
    switch (w) {
        SVTHIP_SADLOOP_CASE(1)
        SVTHIP_SADLOOP_CASE(2)
        SVTHIP_SADLOOP_CASE(4)
        SVTHIP_SADLOOP_CASE(8)
        SVTHIP_SADLOOP_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef SVTHIP_SADLOOP_CASE
    return hipGetLastError();
}

size_t sad_loop_slice_bytes(int w, int h, int sw, int sh, int k)
{
    const int wrows = (sh - 1) + (h - 1) * k + 1, pitch = (w + sw - 1 + 3 + 4) & ~3;
    return ((size_t)h * w + (size_t)wrows * pitch + 15) & ~(size_t)15;
}

}  // namespace svthip
