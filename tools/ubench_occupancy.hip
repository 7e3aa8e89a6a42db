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
#define WPS_MAX 3
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
            else if (OP == 10) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 11) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 12) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 13) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 14) { asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 15) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 16) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 17) { asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 18) { asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 19) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 20) { asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 21) { asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 22) { asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 23) { asm volatile("v_min_f16 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 24) { asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 25) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 26) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 27) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 28) { asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 29) { asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 30) { asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 31) { asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 32) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 33) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 34) { asm volatile("v_pk_add_f32 %3, %3, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 35) { asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 36) { asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 37) { asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 38) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 39) { asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 40) { asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 41) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 42) { asm volatile("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 43) { asm volatile("v_pk_mov_b32 %3, %3, %3 op_sel:[1,0]" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 44) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 45) { asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 46) { asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 47) { asm volatile("v_pk_lshlrev_b16 %0, 3, %0 op_sel_hi:[0,1]" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
            else if (OP == 48) { asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(s), "v"(p[i]) : "vcc"); }
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
    run<10>("v_and_b32", 1, ncu, out, rec);
    run<11>("v_or_b32", 1, ncu, out, rec);
    run<12>("v_xor_b32", 1, ncu, out, rec);
    run<13>("v_lshlrev_b32", 1, ncu, out, rec);
    run<14>("v_lshrrev_b32", 1, ncu, out, rec);
    run<15>("v_sub_u32", 1, ncu, out, rec);
    run<16>("v_min_u32", 1, ncu, out, rec);
    run<17>("v_max_u32", 1, ncu, out, rec);
    run<18>("v_min_i32", 1, ncu, out, rec);
    run<19>("v_min_f32", 1, ncu, out, rec);
    run<20>("v_min3_f32", 1, ncu, out, rec);
    run<21>("v_min_u16", 1, ncu, out, rec);
    run<22>("v_add_u16", 1, ncu, out, rec);
    run<23>("v_min_f16", 1, ncu, out, rec);
    run<24>("v_pk_min_f16", 1, ncu, out, rec);
    run<25>("v_cndmask_b32", 1, ncu, out, rec);
    run<26>("v_cmp_lt_u32", 1, ncu, out, rec);
    run<27>("v_mul_u32_u24", 1, ncu, out, rec);
    run<28>("v_bfi_b32", 1, ncu, out, rec);
    run<29>("v_alignbit_b32", 1, ncu, out, rec);
    run<30>("v_add_u32_sdwa", 1, ncu, out, rec);
    run<31>("v_min_u16_sdwa_hi", 1, ncu, out, rec);
    run<32>("v_add_f32", 1, ncu, out, rec);
    run<33>("v_fma_f32", 1, ncu, out, rec);
    run<34>("v_pk_add_f32", 1, ncu, out, rec);
    run<35>("v_max3_u32", 1, ncu, out, rec);
    run<36>("v_med3_u32", 1, ncu, out, rec);
    run<37>("v_or3_b32", 1, ncu, out, rec);
    run<38>("v_lshl_add_u32", 1, ncu, out, rec);
    run<39>("v_add_lshl_u32", 1, ncu, out, rec);
    run<40>("v_xad_u32", 1, ncu, out, rec);
    run<41>("v_mov_b32_dpp", 1, ncu, out, rec);
    run<42>("v_min_u32_dpp", 1, ncu, out, rec);
    run<43>("v_pk_mov_b32", 1, ncu, out, rec);
    run<44>("v_perm_b32", 1, ncu, out, rec);
    run<45>("v_sad_hi_u8", 1, ncu, out, rec);
    run<46>("v_pk_min_u16", 1, ncu, out, rec);
    run<47>("v_pk_lshlrev_b16", 1, ncu, out, rec);
    run<48>("v_pk_mad_u16", 1, ncu, out, rec);

    return 0;
}
