// tools/ubench_occupancy.hip -- per-SIMD VALU throughput of the packed-SAD instruction mix as a
// function of waves per SIMD, with 256-thread workgroups (the ME kernels' shape).  Waves are
// attributed to their SIMD via HW_REG_HW_ID; throughput per SIMD = instructions issued on it /
// (last end - first start).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

#define ITER 512
#ifndef WPS_MAX
#define WPS_MAX 8
#endif
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Rec { uint64_t t0, t1; uint32_t hwid, xcc; };

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, Rec* rec, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    const int t = threadIdx.x;
    lds[t] = t * 2654435761u + seed;
    __syncthreads();
    uint32_t a[8], b[8];
    uint64_t q[8], p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = lds[(t + i * 7) & 255]; b[i] = lds[(t * 3 + i) & 255]; q[i] = ((uint64_t)a[i] << 32) | b[i]; p[i] = 0; }
    uint32_t s = seed * 77u + 13u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) { p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]); }
            else if (OP == 1) { a[i] = __builtin_amdgcn_sad_u8(a[i], b[i], a[i]); }
            else if (OP == 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
            else if (OP == 3) { asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
            else if (OP == 4) { /* kernel-like mix: 4 qsad + 3 VOP3 + 1 add */
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]);
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], a[i], p[i]);
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], b[i], p[i]);
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[(i + 1) & 7], s, p[i]);
                asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b[i]));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(a[i]), "v"(s));
                asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ b[i] ^ (uint32_t)p[i] ^ (uint32_t)(p[i] >> 32);
    out[blockIdx.x * 256 + t] = r;
    if ((t & 63) == 0) {
        uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        rec[blockIdx.x * 4 + (t >> 6)] = Rec{t0, t1, hw, xcc};
    }
}

template <int OP>
void run(const char* name, int instr_per_iter, int ncu, uint32_t* out, Rec* rec)
{
    printf("%-28s", name);
    for (int wps = 1; wps <= WPS_MAX; wps++) {
        // wps blocks of 256 threads per CU: force with dynamic LDS = floor(160K / wps) (minus a little)
        size_t lds = (160 * 1024) / wps - 256;
        if (wps == 1) lds = 160 * 1024 - 1024;
        lds = (lds / 1024) * 1024;
        if (lds > 65536) HIPCHECK(hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int blocks = ncu * wps;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, out, rec, 99u + rep);
            HIPCHECK(hipDeviceSynchronize());
        }
        std::vector<Rec> h(blocks * 4);
        HIPCHECK(hipMemcpy(h.data(), rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
        // group by (xcc, se, cu, simd)
        std::map<uint64_t, std::vector<Rec>> g;
        for (auto& r : h) {
            uint32_t simd = (r.hwid >> 4) & 3, cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
            uint64_t key = ((uint64_t)(r.xcc & 15) << 32) | (se << 16) | (sh << 12) | (cu << 4) | simd;
            g[key].push_back(r);
        }
        std::vector<double> cyc_per_instr;
        double avg_waves = 0;
        for (auto& kv : g) {
            uint64_t lo = ~0ull, hi = 0;
            for (auto& r : kv.second) { lo = std::min(lo, r.t0); hi = std::max(hi, r.t1); }
            double n = (double)kv.second.size() * ITER * 8 * instr_per_iter;
            cyc_per_instr.push_back((double)(hi - lo) / n);
            avg_waves += kv.second.size();
        }
        std::sort(cyc_per_instr.begin(), cyc_per_instr.end());
        printf(" %dw:%5.2f(%.1f)", wps, cyc_per_instr[cyc_per_instr.size() / 2], avg_waves / g.size());
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    uint32_t* out; Rec* rec;
    HIPCHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4));
    HIPCHECK(hipMalloc(&rec, (size_t)ncu * 8 * 4 * sizeof(Rec)));
    printf("SIMD cycles per wave-instruction (median over SIMDs) vs 256-thread blocks per CU; (avg waves seen per SIMD)\n");
    run<0>("v_qsad_pk_u16_u8", 1, ncu, out, rec);
    run<1>("v_sad_u8", 1, ncu, out, rec);
    run<2>("v_add_u32", 1, ncu, out, rec);
    run<3>("v_min3_u32", 1, ncu, out, rec);
    run<4>("mix 4qsad+3vop3+1add (x8)", 8, ncu, out, rec);

    return 0;
}
