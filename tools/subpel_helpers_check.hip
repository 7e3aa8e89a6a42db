#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int f4(int a, int b, int c, int d) { return clip8((-2 * a + 18 * b + 18 * c - 2 * d + 16) >> 5); }
// ---- four pixels at a time: packed-byte helpers -------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));

// 4 bytes at an arbitrary LDS byte address (two aligned dword reads + v_alignbyte; reads up to 7 bytes past p)
__device__ __forceinline__ uint32_t lds_u32_at(const uint8_t* p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(a & 3u));
}
// bytewise (a - b) mod 256
__device__ __forceinline__ uint32_t sub_u8x4(uint32_t a, uint32_t b)
{
    return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ ((a ^ ~b) & 0x80808080u);
}
// sum over 4 bytes of wrap_sq: e = |int8(s - c)| with -128 -> 128, e * e
__device__ __forceinline__ uint32_t wssd4(uint32_t s, uint32_t c, uint32_t acc)
{
    const uint32_t d = sub_u8x4(s, c);
    const uint32_t m = (d >> 7) & 0x01010101u;        // 1 in every byte whose difference is negative
    const uint32_t e = (d ^ ((m << 8) - m)) + m;      // bytewise |d|: (d ^ 0xff) + 1 <= 128 never carries
    return __builtin_amdgcn_udot4(e, e, acc, false);
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
    return (uint32_t)min(max(v, 0), 255);
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


__device__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s ^ (s >> 13); }
__global__ void chk(unsigned* bad)
{
    uint32_t st = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
    for (int it = 0; it < 2000; it++) {
        uint32_t r0 = rnd(st), r1 = rnd(st), r2 = rnd(st), r3 = rnd(st);
        if (it % 5 == 0) { r0 &= 0x03030303u; r3 = 0xffffffffu; }
        if (it % 7 == 0) { r1 = 0xffffffffu; r2 = 0xfffffffeu; r0 = 0; r3 = 0x01000000u; }
        // vfilt4
        uint32_t v = vfilt4(r0, r1, r2, r3), ref = 0;
        for (int c = 0; c < 4; c++) ref |= (uint32_t)f4((r0 >> (8 * c)) & 255, (r1 >> (8 * c)) & 255, (r2 >> (8 * c)) & 255, (r3 >> (8 * c)) & 255) << (8 * c);
        if (v != ref) { if (atomicAdd(&bad[0], 1u) < 4) printf("vfilt4 %08x %08x %08x %08x -> %08x ref %08x\n", r0, r1, r2, r3, v, ref); }
        // hfilt1
        if (hfilt1(r0) != (uint32_t)f4(r0 & 255, (r0 >> 8) & 255, (r0 >> 16) & 255, r0 >> 24)) atomicAdd(&bad[1], 1u);
        // wssd4
        uint32_t w = wssd4(r0, r1, 5u), wr = 5u;
        for (int c = 0; c < 4; c++) { int d = (((r0 >> (8 * c)) & 255) - ((r1 >> (8 * c)) & 255)) & 255; int e = d > 128 ? 256 - d : d; wr += e * e; }
        if (w != wr) atomicAdd(&bad[2], 1u);
        // avg + ssd4 + sad
        uint32_t a = avg_u8x4(r2, r3), ar = 0;
        for (int c = 0; c < 4; c++) ar |= ((((r2 >> (8 * c)) & 255) + ((r3 >> (8 * c)) & 255) + 1) >> 1) << (8 * c);
        if (a != ar) atomicAdd(&bad[3], 1u);
        uint32_t q = ssd4(r0, a, 9u), qr = 9u, sr = 3u;
        for (int c = 0; c < 4; c++) { int e = (int)((r0 >> (8 * c)) & 255) - (int)((a >> (8 * c)) & 255); qr += e * e; sr += e < 0 ? -e : e; }
        if (q != qr) atomicAdd(&bad[4], 1u);
        if (__builtin_amdgcn_sad_u8(r0, a, 3u) != sr) atomicAdd(&bad[5], 1u);
    }
}
int main()
{
    unsigned* d; hipMalloc(&d, 32); hipMemset(d, 0, 32);
    hipLaunchKernelGGL(chk, dim3(64), dim3(256), 0, 0, d);
    unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("mismatches: vfilt4 %u hfilt1 %u wssd4 %u avg %u ssd4 %u sad %u\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
