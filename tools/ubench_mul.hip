// tools/ubench_mul.hip -- issue-rate micro-benchmark for the multiply / 64-bit instructions the transform kernels are
// built from (gfx950).  Same method as tools/ubench_valu.hip: ITER x 32 instructions per lane on 8 independent chains,
// timed with s_memtime; reported as SIMD cycles per wave-instruction at 1/2/4 waves per SIMD.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ITER 256
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) bench(uint32_t* out, uint64_t* cyc, uint32_t seed)
{
    const int t = threadIdx.x;
    uint32_t a[8], b[8];
    uint64_t p[8];
    double d[8], e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (t * 2654435761u + seed + i * 977u) | 1u;
        b[i] = (t * 40503u + i * 131u + seed) & 0x3fff;
        p[i] = ((uint64_t)a[i] << 20) ^ b[i];
        d[i] = (double)(int32_t)a[i];
        e[i] = 1.0 + (double)i * 1e-9;
    }
    uint32_t s = seed * 77u + 13u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 1) { asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 2) { asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 3) { asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 4) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(a[i]), "v"(b[i]) : "vcc"); }
                else if (OP == 5) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(a[i]), "v"(b[i]) : "vcc"); }
                else if (OP == 6) { asm volatile("v_ashrrev_i64 %0, 13, %0" : "+v"(p[i])); }
                else if (OP == 7) { asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 8) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i])); }
                else if (OP == 9) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i])); }
                else if (OP == 10) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i])); }
                else if (OP == 11) { asm volatile("v_floor_f64 %0, %0" : "+v"(d[i])); }
                else if (OP == 12) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i])); }
                else if (OP == 13) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i])); }
                else if (OP == 14) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 15) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(d[i])); }
                else if (OP == 16) { asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 17) { asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 18) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc"); }
                else if (OP == 19) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p[i]) : "v"(d[i])); }
                else if (OP == 20) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 21) { asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 22) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 23) { asm volatile("v_ashrrev_i32 %0, 13, %0" : "+v"(a[i])); }
                else if (OP == 24) { asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b[i])); }
                else if (OP == 25) { asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i])); }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)p[i] ^ (uint32_t)(p[i] >> 32) ^ (uint32_t)__double_as_longlong(d[i]);
    out[blockIdx.x * blockDim.x + t] = r;
    if ((t & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (t >> 6)] = t1 - t0;
}

struct Entry { const char* name; void (*fn)(uint32_t*, uint64_t*, uint32_t); };

int main()
{
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int nblk = prop.multiProcessorCount;
    uint32_t* out; uint64_t* cyc;
    HIPCHECK(hipMalloc(&out, (size_t)nblk * 1024 * 4));
    HIPCHECK(hipMalloc(&cyc, (size_t)nblk * 16 * 8));
    Entry tab[] = {
        {"v_mul_lo_u32", bench<0>}, {"v_mul_hi_i32", bench<1>}, {"v_mul_i32_i24", bench<2>}, {"v_mad_i32_i24", bench<3>},
        {"v_mad_i64_i32", bench<4>}, {"v_mad_u64_u32", bench<5>}, {"v_ashrrev_i64", bench<6>}, {"v_alignbit_b32", bench<7>},
        {"v_fma_f64", bench<8>}, {"v_mul_f64", bench<9>}, {"v_add_f64", bench<10>}, {"v_floor_f64", bench<11>},
        {"v_cvt_i32_f64", bench<12>}, {"v_cvt_f64_i32", bench<13>}, {"v_fma_f32", bench<14>}, {"v_pk_fma_f32", bench<15>},
        {"v_pk_mul_lo_u16", bench<16>}, {"v_dot2_i32_i16", bench<17>}, {"v_add_co_u32", bench<18>}, {"v_lshl_add_u64", bench<19>},
        {"v_mul_u32_u24", bench<20>}, {"v_mul_hi_i32_i24", bench<21>}, {"v_sub_u32", bench<22>}, {"v_ashrrev_i32", bench<23>},
        {"v_ldexp_f64", bench<24>}, {"v_rndne_f64", bench<25>},
    };
    const int ninstr = ITER * 32;
    printf("%-22s %10s %10s %10s   (SIMD cycles per wave-instruction at 1/2/4 waves per SIMD)\n", "instr", "1w", "2w", "4w");
    for (auto& e : tab) {
        double res[3];
        const int wps[3] = {1, 2, 4};
        for (int k = 0; k < 3; k++) {
            const int threads = 256 * wps[k];
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(e.fn, dim3(nblk), dim3(threads), 0, 0, out, cyc, 12345u + rep);
                HIPCHECK(hipDeviceSynchronize());
            }
            std::vector<uint64_t> h((size_t)nblk * threads / 64);
            HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            res[k] = (double)h[h.size() / 2] / ninstr / wps[k];
        }
        printf("%-22s %10.2f %10.2f %10.2f\n", e.name, res[0], res[1], res[2]);
    }
    return 0;
}
