// tools/mfma_dct_experiment.hip -- NOT PRODUCT CODE, NOT BIT-EXACT.  The non-parity experiment of SURVEY 7 step 7:
// the 32x32 forward 2-D DCT of a batch of residual blocks as two matrix products on the matrix cores,
//     Y = round(s * M X M^T),   M[k][n] = cos((2n + 1) k pi / 64)  (row 0: 1 / sqrt 2),
// to be compared against the bit-exact butterfly kernel (tq_fwd_txfm.hip), which reproduces the reference's per-stage integer
// rounding (half_btf, Codec/EbTransforms.c:1292).  A dense contraction cannot round after every butterfly stage, so its integers differ
// from the reference's; tools/mfma_dct_probe.py reports how often and by how much, and what it would buy.
//
// One wave per block, v_mfma_f32_32x32x16_f16, f32 accumulation.  Operands are f16 pairs (hi + lo) where one f16 is not exact:
// the residual (|x| <= 255) is exact in f16; M = Mhi + Mlo (22 bits);  U = X M^T from 2 x 2 MFMAs;  U = Uhi + Ulo;
// Y = M U from (Mhi Uhi + Mlo Uhi + Mhi Ulo) = 3 x 2 MFMAs: 10 MFMAs per 1024 coefficients.  U never leaves the registers: the
// accumulator tile of the first product is the B operand of the second one with the k order permuted accordingly
// (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

__global__ void __launch_bounds__(256) mfma_dct32_kernel(const int16_t* __restrict__ in, int32_t* __restrict__ out, uint32_t n_tu,
                                                         const float* __restrict__ M, float scale)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;

    // stage-1 B operand: B[k = x][n = v] = M[v][x], natural k order: element j of k-step t = M[r][16 t + 8 h + j]
    // stage-2 A operand: A[u][k-slot] = M[u][y], y = 16 s + 8 (j >> 2) + 4 h + (j & 3)   (the accumulator tile's row order)
    half8 b1hi[2], b1lo[2], a2hi[2], a2lo[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            _Float16 hi, lo;
            split_f16(M[r * 32 + 16 * t + 8 * h + j], hi, lo);
            b1hi[t][j] = hi;
            b1lo[t][j] = lo;
            split_f16(M[r * 32 + 16 * t + 8 * (j >> 2) + 4 * h + (j & 3)], hi, lo);
            a2hi[t][j] = hi;
            a2lo[t][j] = lo;
        }

    for (uint32_t tu = wave; tu < n_tu; tu += n_waves) {
        // A operand of stage 1: X[y = r][x = 16 t + 8 h + j], exact in f16
        const int16_t* x = in + (size_t)tu * 1024 + r * 32 + 8 * h;
        half8 a1[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + 16 * t);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 8; j++) a1[t][j] = (_Float16)(int16_t)(w[j >> 1] >> (16 * (j & 1)));
        }
        floatx16 U = {};
#pragma unroll
        for (int t = 0; t < 2; t++) {
            U = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b1hi[t], U, 0, 0, 0);
            U = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b1lo[t], U, 0, 0, 0);
        }
        // U[y][v]: column v on the lane, rows in the registers -> B operand of Y = M U
        floatx16 Y = {};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            half8 uhi, ulo;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                _Float16 hi, lo;
                split_f16(U[8 * s + j], hi, lo);
                uhi[j] = hi;
                ulo[j] = lo;
            }
            Y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2hi[s], uhi, Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2lo[s], uhi, Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2hi[s], ulo, Y, 0, 0, 0);
        }
        int32_t* o = out + (size_t)tu * 1024 + r;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int u = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            o[u * 32] = (int32_t)__builtin_rintf(Y[reg] * scale);
        }
    }
}

extern "C" int mfma_dct32_run(const int16_t* d_in, int32_t* d_out, uint32_t n_tu, const float* d_M, float scale, void* stream)
{
    if (!n_tu) return 0;
    uint32_t blocks = (n_tu + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;  // persistent waves: the constant fragments are built once per wave
    hipLaunchKernelGGL(mfma_dct32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n_tu, d_M, scale);
    return (int)hipGetLastError();
}
