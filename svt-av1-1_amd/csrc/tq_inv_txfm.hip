// svt-av1-1_amd/csrc/tq_inv_txfm.hip
//
// Batched inverse 2-D transform + reconstruction for AV1 transform units, gfx950.  Replaces av1_inv_txfm2d_add_{WxH}_c
// (Source/Lib/Codec/EbTransforms.c:7714-7900) = inv_txfm2d_add_c (:7617-7700) configured by av1_get_inv_txfm_cfg
// (:7590-7616), reached from Av1InvTransformRecon / Av1InvTransformRecon8bit (:8344-8399) through highbd_inv_txfm_add
// (:8252-8320); 1-D networks av1_idct{4..64}_new (:4902-7000), av1_iadst{4,8,16}_new (:5560-6100), av1_iidentity*_c.
//
// Mapping (mirror of tq_fwd_txfm.hip).  A wave owns G = 64 / min(W, H) TUs at a time (the pass over the longer dimension takes
// max / min rounds of 64 lanes):
//   row pass:    lane = (tu, row); loads its min(W,32) dequantised coefficients (64-point dimensions are stored packed
//                32 wide / 32 high, the rest is zero -- :7736-7760 -- so rows >= 32 are skipped and the upper inputs are
//                compile-time zeros), 1/sqrt(2) pre-scaling for 2:1 rectangles, clamp to bd+8 bits, row network in
//                registers, round shift, write to the wave's LDS tile;
//   column pass: lane = (tu, column); reads its column from LDS, clamp to max(bd+6,16) bits, column network, round shift
//                by 4, then prediction + residual -> clip, in place on the 8- or 16-bit reconstruction plane (a row of a
//                TU is contiguous across lanes).
// Every add/sub layer of the inverse networks clamps to the pass's stage range (av1_gen_inv_stage_range, :4841-4893).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace svthip {

namespace {

#include "tq_txfm_common.h"

#include "tq_inv_networks.h"

template <int WL, int HL, typename PIX>
__global__ void __launch_bounds__(256) inv_txfm2d_add_kernel(const int32_t* __restrict__ coeff, const svthip_itxfm_desc* __restrict__ desc,
                                                             uint32_t n_tu, int bd, PIX* __restrict__ recon)
{
    constexpr int W = 1 << WL, H = 1 << HL, WI = WL - 2, HI = HL - 2;
    constexpr int MIND = W < H ? W : H, G = 64 / MIND, P = W + 1;  // a wave owns 64 / min(W, H) TUs: see tq_encode_tu.hip
    constexpr int ROUNDS_COL = G * W / 64, ROUNDS_ROW = G * H / 64;
    constexpr int WIN = W > 32 ? 32 : W, HIN = H > 32 ? 32 : H;
    constexpr int SH0 = kInvShift0[WI][HI];
    constexpr bool RECT2 = (WL - HL == 1) || (HL - WL == 1);
    constexpr bool PACKED_RECON = sizeof(PIX) == 1 && W >= 16;  // 8-bit planes, rows of at least 16 pixels
    extern __shared__ int32_t lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* tile = lds_all + wave * (G * H * P);
    const Clamp cl_in = {-(1 << (bd + 7)), (1 << (bd + 7)) - 1};  // bd + 8 bits: row input and row stage range
    const Clamp cl_col = {-(1 << 15), (1 << 15) - 1};             // max(bd + 6, 16) = 16 bits for bd 8 and 10
    const int32_t res_max = (1 << (7 + bd)) - 1 + (914 << (bd - 7));
    const int32_t pix_max = (1 << bd) - 1;
    const uint32_t groups = (n_tu + G - 1) / G;
    for (uint32_t grp = blockIdx.x * 4 + wave; grp < groups; grp += gridDim.x * 4) {
        // ---- row pass ----
#pragma unroll 1
        for (int round = 0; round < ROUNDS_ROW; round++) {
            const int t = round * 64 + lane, g = t / H, r = t % H;
            const uint32_t tu = grp * G + g;
            if (tu < n_tu && r < HIN) {
                const svthip_itxfm_desc d = desc[tu];
                const int kr = kHtx[d.tx_type & 15];
                const int32_t* in = coeff + d.coeff_offset + r * WIN;
                int32_t x[W], y[W];
#pragma unroll
                for (int c = 0; c < WIN; c += 4) {
                    const int4 v = *reinterpret_cast<const int4*>(in + c);
                    x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
                }
#pragma unroll
                for (int c = 0; c < WIN; c++) {
                    if constexpr (RECT2) x[c] = mulrs<12>(x[c], 2896);
                    x[c] = cl_in(x[c]);
                }
#pragma unroll
                for (int c = WIN; c < W; c++) x[c] = 0;
                itxfm1d<W, WIN>(kr, x, y, cl_in);
                int32_t* row = tile + g * (H * P) + r * P;
#pragma unroll
                for (int c = 0; c < W; c++) {
                    if constexpr (SH0 > 0) row[c] = rs<(SH0 > 0 ? SH0 : 1)>((int64_t)y[c]);
                    else row[c] = y[c];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- column pass ----
#pragma unroll 1
        for (int round = 0; round < ROUNDS_COL; round++) {
            const int t = round * 64 + lane, g = t / W, c = t % W;
            const uint32_t tu = grp * G + g;
            if (tu < n_tu) {
                const svthip_itxfm_desc d = desc[tu];
                const int kc = kVtx[d.tx_type & 15], kr = kHtx[d.tx_type & 15];
                const int32_t* col = tile + g * (H * P) + (kr == 2 ? W - 1 - c : c);
                int32_t x[H], y[H];
#pragma unroll
                for (int r = 0; r < HIN; r++) x[r] = cl_col(col[r * P]);
#pragma unroll
                for (int r = HIN; r < H; r++) x[r] = 0;
                itxfm1d<H, HIN>(kc, x, y, cl_col);
                if constexpr (PACKED_RECON) {
                    // residual column back to the tile (same lane, same column); reconstruction follows four pixels per lane
                    int32_t* ocol = tile + g * (H * P) + c;
#pragma unroll
                    for (int r = 0; r < H; r++) {
                        const int32_t t = rs<4>((int64_t)y[r]);
                        ocol[flip_row<H>(r, kc) * P] = min(max(t, -res_max - 1), res_max);
                    }
                } else {
                    PIX* out = recon + d.recon_offset + c;
                    const int stride = d.recon_stride;
#pragma unroll
                    for (int r = 0; r < H; r++) {
                        int32_t t = rs<4>((int64_t)y[r]);
                        t = min(max(t, -res_max - 1), res_max);
                        PIX* p = out + flip_row<H>(r, kc) * stride;
                        const int32_t v = (int32_t)*p + t;
                        *p = (PIX)min(max(v, 0), pix_max);
                    }
                }
            }
        }
        if constexpr (PACKED_RECON) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // prediction + residual -> clip, 4 horizontally consecutive 8-bit pixels per lane (one dword load / store instead of
            // four byte loads / stores per lane); TUs whose rows are not 4-byte aligned take the byte path
            if constexpr (W * H / 4 < 64) {
                // TUs smaller than one pass of the wave (16x4: 16 dwords): all TUs of the group in one flattened loop
                constexpr int NQ = W * H / 4;
#pragma unroll 2
                for (int i = lane; i < G * NQ; i += 64) {
                    const int g = i / NQ, q = i - g * NQ;
                    const uint32_t tu = grp * G + g;
                    if (tu >= n_tu) continue;
                    const svthip_itxfm_desc d = desc[tu];
                    uint8_t* base = reinterpret_cast<uint8_t*>(recon) + d.recon_offset;
                    const int stride = d.recon_stride;
                    const int r = (4 * q) / W, c = (4 * q) % W;
                    const int32_t* t4 = tile + g * (H * P) + r * P + c;
                    uint8_t* p8 = base + r * stride + c;
                    if ((((uintptr_t)base | (uintptr_t)stride) & 3u) == 0) {
                        uint32_t* p = reinterpret_cast<uint32_t*>(p8);
                        const uint32_t pv = *p;
                        uint32_t o = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) o |= (uint32_t)min(max((int32_t)((pv >> (8 * k)) & 255u) + t4[k], 0), 255) << (8 * k);
                        *p = o;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) p8[k] = (uint8_t)min(max((int32_t)p8[k] + t4[k], 0), 255);
                    }
                }
            } else {
                for (int g = 0; g < G; g++) {
                    const uint32_t tu = grp * G + g;  // uniform
                    if (tu >= n_tu) break;
                    const svthip_itxfm_desc d = desc[tu];
                    uint8_t* base = reinterpret_cast<uint8_t*>(recon) + d.recon_offset;
                    const int stride = d.recon_stride;
                    const int32_t* tg = tile + g * (H * P);
                    if ((((uintptr_t)base | (uintptr_t)stride) & 3u) == 0) {
#pragma unroll 2
                        for (int q = lane; q < W * H / 4; q += 64) {
                            const int r = (4 * q) / W, c = (4 * q) % W;
                            uint32_t* p = reinterpret_cast<uint32_t*>(base + r * stride + c);
                            const uint32_t pv = *p;
                            const int32_t* t4 = tg + r * P + c;
                            uint32_t o = 0;
#pragma unroll
                            for (int k = 0; k < 4; k++) o |= (uint32_t)min(max((int32_t)((pv >> (8 * k)) & 255u) + t4[k], 0), 255) << (8 * k);
                            *p = o;
                        }
                    } else {
                        for (int q = lane; q < W * H; q += 64) {
                            const int r = q / W, c = q % W;
                            uint8_t* p = base + r * stride + c;
                            *p = (uint8_t)min(max((int32_t)*p + tg[r * P + c], 0), 255);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int WL, int HL, typename PIX>
hipError_t launch_one(const int32_t* coeff, const svthip_itxfm_desc* desc, uint32_t n_tu, int bd, PIX* recon, hipStream_t s)
{
    constexpr int W = 1 << WL, H = 1 << HL, MIND = W < H ? W : H, G = 64 / MIND;
    constexpr size_t lds = (size_t)4 * G * H * (W + 1) * sizeof(int32_t);
    const uint32_t groups = (n_tu + G - 1) / G;
    uint32_t blocks = (groups + 3) / 4;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    if (lds > 64 * 1024) {
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_txfm2d_add_kernel<WL, HL, PIX>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((inv_txfm2d_add_kernel<WL, HL, PIX>), dim3(blocks), dim3(256), lds, s, coeff, desc, n_tu, bd, recon);
    return hipGetLastError();
}

template <typename PIX>
hipError_t launch_sized(const int32_t* coeff, const svthip_itxfm_desc* desc, uint32_t n_tu, int w, int h, int bd, PIX* recon,
                        hipStream_t s)
{
    const int key = clog2(w) * 8 + clog2(h);
#define CASE(WL, HL) case (WL) * 8 + (HL): return launch_one<WL, HL, PIX>(coeff, desc, n_tu, bd, recon, s)
    switch (key) {
        CASE(2, 2); CASE(3, 3); CASE(4, 4); CASE(5, 5); CASE(6, 6);
        CASE(2, 3); CASE(3, 2); CASE(3, 4); CASE(4, 3); CASE(4, 5); CASE(5, 4); CASE(5, 6); CASE(6, 5);
        CASE(2, 4); CASE(4, 2); CASE(3, 5); CASE(5, 3); CASE(4, 6); CASE(6, 4);
        default: return hipErrorInvalidValue;
    }
#undef CASE
}

}  // namespace

hipError_t launch_inv_txfm2d_add(const int32_t* coeff, const svthip_itxfm_desc* desc, uint32_t n_tu, int w, int h, int bd,
                                 void* recon, int recon_16bit, hipStream_t s)
{
    if (recon_16bit) return launch_sized<uint16_t>(coeff, desc, n_tu, w, h, bd, static_cast<uint16_t*>(recon), s);
    return launch_sized<uint8_t>(coeff, desc, n_tu, w, h, bd, static_cast<uint8_t*>(recon), s);
}

}  // namespace svthip
