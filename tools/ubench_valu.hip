// tools/ubench_valu.hip -- issue-rate micro-benchmark for the integer VALU / LDS instructions the ME
// kernels are built from (gfx950).  Not part of the product; results are recorded in DESIGN.md and
// profiles/.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
//
// For every instruction: a loop of ITER x 32 independent-chain instructions per lane (8 chains x 4),
// cycles measured per wave with s_memtime, at 1, 2 and 4 waves per SIMD.  Reported figure:
// SIMD cycles per wave-instruction = (cycles * waves_per_simd) / n_instr  (lower bound on issue cost).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>

#define ITER 256

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) bench(uint32_t* out, uint64_t* cyc, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a[8], b[8];
    uint64_t q[8], p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = lds[(t + i * 64) & 4095];
        b[i] = lds[(t * 3 + i) & 4095];
        q[i] = ((uint64_t)a[i] << 32) | b[i];
        p[i] = 0;
    }
    uint32_t s = seed * 77u + 13u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_amdgcn_sad_u8(a[i], b[i], a[i]);
                else if (OP == 1) a[i] = __builtin_amdgcn_sad_hi_u8(b[i], s, a[i]);
                else if (OP == 2) p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]);
                else if (OP == 3) p[i] = __builtin_amdgcn_mqsad_pk_u16_u8(q[i], s, p[i]);
                else if (OP == 4) a[i] = __builtin_amdgcn_sad_u16(a[i], b[i], a[i]);
                else if (OP == 5) a[i] = __builtin_amdgcn_perm(a[i], b[i], 0x06050403u);
                else if (OP == 6) a[i] = __builtin_amdgcn_alignbyte(a[i], b[i], 1);
                else if (OP == 7) { asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 8) { asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 9) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 10) { asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 11) { asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 12) { asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 13) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 14) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 15) { asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(q[i])); }
                else if (OP == 16) { asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(p[i]) : "v"(q[i])); }
                else if (OP == 17) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 18) { asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 19) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i])); }
                else if (OP == 20) { asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(s) : "vcc"); }
                else if (OP == 21) { asm volatile("v_msad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 22) { asm volatile("v_bfe_u32 %0, %1, 16, 16" : "=v"(a[i]) : "v"(b[i])); }
                else if (OP == 23) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 24) { asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(p[i]) : "v"(q[i])); }
                else if (OP == 25) { asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 26) { asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 27) { asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 28) { asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 29) { asm volatile("v_sad_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 30) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(s)); }
                else if (OP == 31) { /* mixed: 1 qsad + 4 VOP3 ops (5 instr) */
                    p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]);
                    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
                    asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(b[i]) : "v"(a[i]));
                    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
                    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(a[i]), "v"(s)); }
                else if (OP == 32) { /* mixed: 1 qsad + 4 v_add_u32 */
                    p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]);
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(a[i]));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
                else if (OP == 33) { /* qsad with SGPR src1 */
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "s"(s)); }
                else if (OP == 34) { /* (sad << 16 | idx) key in one SDWA move: the low half of the destination (idx) is kept */
                    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 35) { asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 36) { asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); }
                else if (OP == 37) { /* mixed: 1 qsad + 2 SDWA keys + 1 min3 + 1 SDWA key + ... (x5) */
                    p[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], s, p[i]);
                    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a[i]) : "v"(b[i]));
                    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(b[i]) : "v"(a[i]));
                    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
                    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(b[i]) : "v"(a[i])); }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)p[i] ^ (uint32_t)(p[i] >> 32);
    out[blockIdx.x * blockDim.x + t] = r;
    if ((t & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (t >> 6)] = t1 - t0;
}

// LDS read-rate benches: each lane reads 32 values per iteration with the given width.
template <int W>
__global__ void __launch_bounds__(1024) bench_lds(uint32_t* out, uint64_t* cyc, uint32_t seed, int stride_dw)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[16384];
    const int t = threadIdx.x;
    for (int i = t; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc = 0;
    const int lane = t & 63;
    int base = (lane * stride_dw) & 8191;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            int off = (base + u * 256 + (it & 7) * 4) & 16380;
            if (W == 1) { acc ^= lds[off]; }
            else if (W == 2) { uint2 v = *reinterpret_cast<const uint2*>(&lds[off & ~1]); acc ^= v.x ^ v.y; }
            else { uint4 v = *reinterpret_cast<const uint4*>(&lds[off & ~3]); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + t] = acc;
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
    HIPCHECK(hipMalloc(&out, (size_t)nblk * 8 * 1024 * 4));
    HIPCHECK(hipMalloc(&cyc, (size_t)nblk * 8 * 16 * 8));
    Entry tab[] = {
        {"v_sad_u8", bench<0>}, {"v_sad_hi_u8", bench<1>}, {"v_qsad_pk_u16_u8", bench<2>},
        {"v_mqsad_pk_u16_u8", bench<3>}, {"v_sad_u16", bench<4>}, {"v_perm_b32", bench<5>},
        {"v_alignbyte_b32", bench<6>}, {"v_min3_u32", bench<7>}, {"v_lshl_or_b32", bench<8>},
        {"v_and_or_b32", bench<9>}, {"v_pk_add_u16", bench<10>}, {"v_pk_min_u16", bench<11>},
        {"v_pk_mad_u16", bench<12>}, {"v_add_u32", bench<13>}, {"v_add3_u32", bench<14>},
        {"v_mov_b64", bench<15>}, {"v_pk_mov_b32", bench<16>}, {"v_min_u32", bench<17>},
        {"v_add_u32_dpp", bench<18>}, {"v_mov_b32", bench<19>}, {"v_cmp+v_cndmask", bench<20>},
        {"v_msad_u8", bench<21>}, {"v_bfe_u32", bench<22>}, {"v_mad_u32_u24", bench<23>},
        {"v_lshlrev_b64", bench<24>}, {"v_min_u16", bench<25>}, {"v_pk_sub_u16", bench<26>},
        {"v_pk_max_u16", bench<27>}, {"v_pk_sub_i16_clamp", bench<28>}, {"v_sad_u32", bench<29>},
        {"v_dot4_u32_u8", bench<30>}, {"mix qsad+4xVOP3 (x5)", bench<31>}, {"mix qsad+4xadd (x5)", bench<32>},
        {"v_qsad sgpr src1", bench<33>}, {"v_mov_b32_sdwa key lo", bench<34>}, {"v_mov_b32_sdwa key hi", bench<35>},
        {"v_or_b32_e32", bench<36>}, {"mix qsad+3sdwa+min3 (x5)", bench<37>},
    };
    const int ninstr = ITER * 32;
    if (getenv("UB_BLOCKS")) {
        // occupancy built from k independent 256-thread workgroups per CU (the ME kernels' shape) instead of one large workgroup
        printf("%-22s   per-wave cycles per instruction (median) at k = 1..6 workgroups of 256 threads per CU; (SIMD rate = value / k)\n", "instr");
        for (auto& e : tab) {
            if (!(strstr(e.name, "sad") || strstr(e.name, "mix") || strstr(e.name, "v_add_u32") || strstr(e.name, "v_min3") || strstr(e.name, "sdwa") || strstr(e.name, "v_or") ||
                  strstr(e.name, "v_lshl_or"))) continue;
            printf("%-22s", e.name);
            for (int k = 1; k <= 6; k++) {
                const int blocks = nblk * k;
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u + rep);
                    HIPCHECK(hipDeviceSynchronize());
                }
                std::vector<uint64_t> h((size_t)blocks * 4);
                HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                int extra = (strcmp(e.name, "v_cmp+v_cndmask") == 0) ? 2 : (strstr(e.name, "(x5)") ? 5 : 1);
                const double med = (double)h[h.size() / 2] / (ninstr * extra), lo = (double)h[h.size() / 10] / (ninstr * extra),
                             hi = (double)h[h.size() * 9 / 10] / (ninstr * extra);
                printf("  k%d %6.2f [%5.2f %5.2f] (%5.2f)", k, med, lo, hi, med / k);
            }
            printf("\n");
        }
        return 0;
    }
    printf("%-22s %10s %10s %10s   (SIMD cycles per wave-instruction at 1/2/4 waves per SIMD)\n", "instr", "1w", "2w", "4w");
    for (auto& e : tab) {
        double res[6];
        int wps[6] = {1, 2, 4, 3, 6, 8};
        for (int k = 0; k < 6; k++) {
            if (wps[k] * 256 > 1024) {  // more than one block per CU
            }
            int threads = 256 * wps[k];
            int blocks = nblk;
            if (threads > 1024) { threads /= 2; blocks *= 2; }
            if (wps[k] == 3) { threads = 768; }
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, cyc, 12345u + rep);
                HIPCHECK(hipDeviceSynchronize());
            }
            std::vector<uint64_t> h((size_t)blocks * threads / 64);
            HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            double med = (double)h[h.size() / 2];
            int extra = (strcmp(e.name, "v_cmp+v_cndmask") == 0) ? 2 : (strstr(e.name, "(x5)") ? 5 : 1);
            res[k] = med * 1.0 / (ninstr * extra) ;  // cycles per instruction as seen by one wave
        }
        // per-SIMD issue cost = per-wave cycles / waves sharing the SIMD
        printf("%-22s %10.2f %10.2f %10.2f   3w/6w/8w: %.2f %.2f %.2f\n", e.name, res[0] / 1, res[1] / 2, res[2] / 4,
               res[3] / 3, res[4] / 6, res[5] / 8);
    }
    // LDS
    struct L { const char* name; void (*fn)(uint32_t*, uint64_t*, uint32_t, int); int bytes; int stride; };
    L lt[] = {{"ds_read_b32 s1", bench_lds<1>, 4, 1}, {"ds_read_b64 s2", bench_lds<2>, 8, 2}, {"ds_read_b128 s4", bench_lds<4>, 16, 4},
              {"ds_read_b128 s48+4", bench_lds<4>, 16, 52}};
    for (auto& e : lt) {
        for (int w : {1, 2, 4}) {
            int threads = 256 * w;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(e.fn, dim3(nblk), dim3(threads), 0, 0, out, cyc, 777u, e.stride);
                HIPCHECK(hipDeviceSynchronize());
            }
            std::vector<uint64_t> h((size_t)nblk * threads / 64);
            HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            double med = (double)h[h.size() / 2];
            double bytes = (double)ITER * 16 * e.bytes * 64 * (4 * w);  // per CU
            printf("%-22s waves/SIMD %d: %.1f B/clk/CU (%.2f cyc per wave-instr per wave)\n", e.name, w, bytes / med, med / (ITER * 16));
        }
    }
    return 0;
}
