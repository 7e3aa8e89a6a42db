// tools/ubench_ldskey.hip -- can the LDS pipe form the (sad << 16 | idx) keys of the full-pel search instead of the VALU?
// A ds_read_u16_d16_hi into a register whose low half holds idx produces the key without a VALU instruction.  Three loop bodies with the
// full-pel kernel's instruction ratio (512 v_qsad : 336 keys : 168 v_min3 per item = 32 : 21 : 10.5):
//   0  32 v_qsad + 21 v_lshl_or / v_and_or + 10 v_min3          (today's kernel)
//   1  32 v_qsad + 6 ds_write_b64 + 21 ds_read_u16_d16_hi + 10 v_min3, reads issued before the v_qsad, waited for after them
//   2  32 v_qsad + 10 v_min3                                   (lower bound: keys for free)
//   3  as 1, plus one v_or_b32_e32 per key (what is really needed: the D16 load clears the half that should keep idx)
//   4  as 0 with every key formed by TWO VOP2 instructions (v_lshlrev_b32 / v_and_b32, then v_or_b32) instead of one VOP3
//   5  as 4 with every v_min3_u32 replaced by two v_min_u32 (VOP2)
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldskey.hip -o gpurun_out/ubench_ldskey
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kIter = 512;

template <int OFF> __device__ __forceinline__ void rd16hi(uint32_t& k, uint32_t addr)
{
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(k) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void wr64(uint32_t addr, uint64_t v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// halves of the 12 dwords written one iteration ago: slot I / 4 (2048 bytes apart), half I % 4 of its 8-byte cell
template <int... I> __device__ __forceinline__ void read_keys(uint32_t (&key)[21], uint32_t rd, std::integer_sequence<int, I...>)
{
    (rd16hi<(I / 4) * 2048 + (I % 4) * 2>(key[I], rd), ...);
}
template <int... I> __device__ __forceinline__ void write_accs(const uint64_t (&acc)[8], uint32_t wr, std::integer_sequence<int, I...>)
{
    (wr64<I * 2048>(wr, acc[I]), ...);
}

template <int MODE>
__global__ void __launch_bounds__(256, 4) body(uint32_t* out, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int t = threadIdx.x;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = t; i < 256 * 16; i += 256) l32[i] = i * 2654435761u + seed;
    __syncthreads();
    // per-thread slice: 2 buffers x 6 x 8 bytes, laid out [buffer][slot][thread] so that a wave's access is 64 consecutive 8-byte cells
    const uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)smem) + 16384 + 8 * t;
    uint64_t win[8], acc[8];
    uint32_t key[21], best[10];
#pragma unroll
    for (int i = 0; i < 8; i++) { win[i] = ((uint64_t)l32[(t + 64 * i) & 4095] << 32) | l32[(3 * t + i) & 4095]; acc[i] = 0; }
#pragma unroll
    for (int i = 0; i < 21; i++) key[i] = (uint32_t)(t * 16 + i);
#pragma unroll
    for (int i = 0; i < 10; i++) best[i] = 0xffffffffu;
    uint32_t s = seed * 77u + 13u;
    uint32_t idxv[4] = {(uint32_t)t, (uint32_t)t + 1u, (uint32_t)t + 2u, (uint32_t)t + 3u};
    for (int it = 0; it < kIter; it++) {
        const uint32_t rd = base + ((it & 1) ? 6 * 2048 : 0), wr = base + ((it & 1) ? 0 : 6 * 2048);
        if (MODE == 1 || MODE == 3) {
            read_keys(key, rd, std::make_integer_sequence<int, 21>{});
        }
        (void)rd;
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(win[i], s + u, u ? acc[i] : 0ull);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 21; i++) {
                const uint32_t a = (uint32_t)(acc[i & 7] >> ((i & 8) ? 32 : 0));
                if (i & 1) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(key[i]) : "v"(a), "v"(0xffff0000u));
                else asm volatile("v_lshl_or_b32 %0, %1, 16, %0" : "+v"(key[i]) : "v"(a));
            }
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < 21; i++) {
                const uint32_t a = (uint32_t)(acc[i & 7] >> ((i & 8) ? 32 : 0));
                uint32_t tmp;
                if (i & 1) asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(tmp) : "s"(0xffff0000u), "v"(a));
                else asm volatile("v_lshlrev_b32_e32 %0, 16, %1" : "=v"(tmp) : "v"(a));
                asm volatile("v_or_b32_e32 %0, %1, %2" : "=v"(key[i]) : "v"(tmp), "v"(idxv[i & 3]));
            }
        }
        if (MODE == 1 || MODE == 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            write_accs(acc, wr, std::make_integer_sequence<int, 6>{});
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 21; i++) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(key[i]) : "v"(idxv[i & 3]));
        }
        (void)wr;
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 10; i++) {
                asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(best[i]) : "v"(key[2 * i]));
                asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(best[i]) : "v"(key[2 * i + 1]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 10; i++) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(best[i]) : "v"(key[2 * i]), "v"(key[2 * i + 1]));
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(acc[i]));
        }
        s += 3;
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) r ^= best[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
    out[blockIdx.x * 256 + t] = r;
}

// does ds_read_u16_d16_hi keep the low half of its destination on this part (SRAM-ECC parts may clear it)?
__global__ void d16_check(uint32_t* out)
{
    __shared__ uint32_t cell[64];
    cell[threadIdx.x] = 0xabcd0000u | threadIdx.x;  // low half: lane number
    __syncthreads();
    uint32_t k = 0x00001234u;
    const uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)&cell[threadIdx.x]);
    asm volatile("ds_read_u16_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(k) : "v"(addr));
    out[threadIdx.x] = k;
}

template <int MODE> static float run(uint32_t* d_out, int blocks)
{
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    const size_t lds = 16384 + 2 * 6 * 2048;  // 40 KB like the full-pel kernel: four workgroups per CU
    HIPCHECK(hipFuncSetAttribute((const void*)body<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    body<MODE><<<blocks, 256, lds>>>(d_out, 1);
    HIPCHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        HIPCHECK(hipEventRecord(e0));
        body<MODE><<<blocks, 256, lds>>>(d_out, 2 + rep);
        HIPCHECK(hipEventRecord(e1));
        HIPCHECK(hipEventSynchronize(e1));
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int blocks = 256 * 4 * 4;  // four rounds of four workgroups per CU
    uint32_t* d_out;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    uint32_t h[64];
    d16_check<<<1, 64>>>(d_out);
    HIPCHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("ds_read_u16_d16_hi into 0x00001234 from a cell holding lane number 5: 0x%08x (%s)\n", h[5],
           h[5] == 0x00051234u ? "low half preserved" : "low half NOT preserved");
    const float t0 = run<0>(d_out, blocks), t1 = run<1>(d_out, blocks), t2 = run<2>(d_out, blocks), t3 = run<3>(d_out, blocks), t4 = run<4>(d_out, blocks),
                t5 = run<5>(d_out, blocks);
    // per wave and loop iteration, in SIMD cycles at 2.4 GHz: a SIMD runs 4 resident waves x 4 rounds
    const double f = 2.4e6 / (16.0 * kIter);
    printf("per loop iteration of one wave (SIMD cycles at 2.4 GHz; 32 v_qsad alone = 517):\n");
    printf("  keys on the VALU (21 v_lshl_or / v_and_or + 10 v_min3): %7.0f   (%.3f ms)\n", t0 * f, t0);
    printf("  keys through LDS (6 ds_write_b64 + 21 ds_read_u16_d16_hi): %7.0f   (%.3f ms)\n", t1 * f, t1);
    printf("  no key formation (10 v_min3 only):                      %7.0f   (%.3f ms)\n", t2 * f, t2);
    printf("  keys through LDS + one v_or_b32_e32 per key:            %7.0f   (%.3f ms)\n", t3 * f, t3);
    printf("  keys by two VOP2 each (shift / and, then or):           %7.0f   (%.3f ms)\n", t4 * f, t4);
    printf("  ... and two v_min_u32 instead of each v_min3_u32:       %7.0f   (%.3f ms)\n", t5 * f, t5);
    return 0;
}
