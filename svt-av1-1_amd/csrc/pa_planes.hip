// svt-av1-1_amd/csrc/pa_planes.hip
//
// Picture-analysis producers of the ME inputs and the border padding of reconstructed reference pictures, gfx950.
// Replaces (paths under Source/Lib/Codec of the reference):
//   generate_padding / generate_padding16_bit      EbMcp.c:173-215, :220-262   (horizontal then vertical edge replication)
//   PadPictureToMultipleOfLcuDimensions            EbPictureAnalysisProcess.c:4866-4880
//   Decimation2D + DecimateInputPicture            EbPictureAnalysisProcess.c:100-125, :4885-4936
//   PadRefAndSetFlags (the padding half)           EbEncDecProcess.c:1135-1204
//
// The reference pads in two dependent passes (rows first, then whole padded rows up and down) and decimates into a
// plane that is then padded again.  All of that is a pure gather: every output byte is the input sample at clamped
// coordinates, out[Y][X] = in[clamp(Y - pad, 0, h-1) * step][clamp(X - pad, 0, w-1) * step], so one launch writes the
// borders of the full-resolution plane and the complete 1/4 and 1/16 planes of a batch of pictures with no pass
// ordering.  HBM-bound byte work: a thread owns 4 consecutive output bytes (one coalesced dword store); interior dwords of
// the decimated planes read 2 or 4 source dwords and pick bytes with v_perm_b32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace svthip {

namespace {

// one output dword (4 samples of `SB` bytes... here SB = 1) of a plane that is a clamped, point-sampled view of `src`
__device__ __forceinline__ uint32_t gather4_u8(const uint8_t* __restrict__ src, uint32_t src_stride, int w, int h, int step_log2, int pad,
                                               int X, int Y)
{
    const int y = min(max(Y - pad, 0), h - 1);
    const uint8_t* row = src + (size_t)(y << step_log2) * src_stride;
    const int x0 = X - pad;
    if (x0 >= 0 && x0 + 3 < w) {  // all four samples inside the picture: vector loads
        const int sx = x0 << step_log2;
        if (step_log2 == 1 && (sx & 3) == 0) {
            const uint2 v = *reinterpret_cast<const uint2*>(row + sx);
            return __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);  // bytes 0,2 of v.x then 0,2 of v.y
        }
        if (step_log2 == 2 && (sx & 3) == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + sx);
            return (v.x & 0xffu) | ((v.y & 0xffu) << 8) | ((v.z & 0xffu) << 16) | (v.w << 24);
        }
    }
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = min(max(x0 + k, 0), w - 1);
        o |= (uint32_t)row[x << step_log2] << (8 * k);
    }
    return o;
}

}  // namespace

// grid: (ceil(max plane dwords / 256), 3 planes, n pictures).  Plane 0 = full resolution (borders only), 1 = quarter, 2 = sixteenth.
__global__ void __launch_bounds__(256) pa_derive_planes_kernel(uint8_t* __restrict__ pool, PaJobTable jobs, int do_quarter, int do_sixteenth)
{
    const svthip_pa_picture P = jobs.pic[blockIdx.z];
    const int plane = blockIdx.y;
    if ((plane == 1 && !do_quarter) || (plane == 2 && !do_sixteenth)) return;
    const int step_log2 = plane;                       // 1, 2, 4 -> log2
    const int pad = plane == 0 ? 68 : (plane == 1 ? 32 : 16);  // EbEncHandle.c:1006-1030
    const int w = P.width >> step_log2, h = P.height >> step_log2;
    const uint32_t stride = plane == 0 ? P.full_stride : (plane == 1 ? P.quarter_stride : P.sixteenth_stride);
    uint8_t* dst = pool + (plane == 0 ? P.full_offset : (plane == 1 ? P.quarter_offset : P.sixteenth_offset));
    const uint8_t* src = pool + P.full_offset + (size_t)68 * P.full_stride + 68;  // picture sample (0,0)
    const int tw = w + 2 * pad, row_dw = (tw + 3) >> 2, rows = h + 2 * pad;
    const int total = row_dw * rows;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int Y = i / row_dw, X = 4 * (i - Y * row_dw);
        // interior of the full plane is already there (68 and the width are multiples of 4: a dword never straddles the edge)
        if (plane == 0 && Y >= pad && Y < pad + h && X >= pad && X + 3 < pad + w) continue;
        const uint32_t v = gather4_u8(src, P.full_stride, w, h, step_log2, pad, X, Y);
        uint8_t* o = dst + (size_t)Y * stride + X;
        if (X + 3 < tw && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            *reinterpret_cast<uint32_t*>(o) = v;
        } else {  // 1/16 planes of widths that are not multiples of 16 have rows of 4k+2 bytes
            for (int k = 0; k < 4 && X + k < tw; k++) o[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// generate_padding (sample_bytes = 1) / generate_padding16_bit (sample_bytes = 2) of one plane in place: the interior
// (width x height samples at (pad_w, pad_h)) is read, everything else of the (width + 2 pad_w) x (height + 2 pad_h) area is written.
template <typename T>
__global__ void __launch_bounds__(256) pad_plane_kernel(T* __restrict__ plane, uint32_t stride, int width, int height, int pad_w, int pad_h)
{
    // only the border is visited: 2 pad_h rows of `stride` samples (the reference's vertical pass copies whole rows of `stride`
    // bytes, spare samples behind the padded width included) and 2 pad_w columns of the `height` middle rows
    const int tw = width + 2 * pad_w;
    const int n_tb = 2 * pad_h * (int)stride, total = n_tb + height * 2 * pad_w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int X, Y;
        if (i < n_tb) {
            const int r = i / (int)stride;
            X = i - r * (int)stride;
            Y = r < pad_h ? r : height + r;
        } else {
            const int j = i - n_tb, r = j / (2 * pad_w), c = j - r * 2 * pad_w;
            Y = pad_h + r;
            X = c < pad_w ? c : width + c;
        }
        const int y = min(max(Y - pad_h, 0), height - 1);
        const int xs = X < tw ? pad_w + min(max(X - pad_w, 0), width - 1) : X;  // spare samples are copied as they are
        plane[(size_t)Y * stride + X] = plane[(size_t)(y + pad_h) * stride + xs];
    }
}

hipError_t launch_pad_plane(void* plane, uint32_t stride, int width, int height, int pad_w, int pad_h, int sample_bytes, hipStream_t s)
{
    const int total = 2 * pad_h * (int)stride + height * 2 * pad_w;
    if (total <= 0) return hipSuccess;
    const int blocks = min((total + 255) / 256, 4096);
    if (sample_bytes == 1)
        hipLaunchKernelGGL(pad_plane_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, static_cast<uint8_t*>(plane), stride, width, height, pad_w, pad_h);
    else
        hipLaunchKernelGGL(pad_plane_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, static_cast<uint16_t*>(plane), stride, width, height, pad_w, pad_h);
    return hipGetLastError();
}

// svthip_me_cu_result (24 B) -> the reference's MeCuResults_t layout (40 B, Codec/EbMotionEstimationLcuResults.h:56-76)
__global__ void __launch_bounds__(256) me_results_ref_layout_kernel(const svthip_me_cu_result* __restrict__ in, uint32_t n,
                                                                     svthip_me_cu_result_ref* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const svthip_me_cu_result r = in[i];
    svthip_me_cu_result_ref o;
    o.xMvL0 = r.xMvL0; o.yMvL0 = r.yMvL0; o.xMvL1 = r.xMvL1; o.yMvL1 = r.yMvL1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o.distortionDirection[k].distortion = r.distortion[k];
        o.distortionDirection[k].direction = r.direction[k];
    }
    o.totalMeCandidateIndex = r.totalMeCandidateIndex;
#pragma unroll
    for (int k = 0; k < 7; k++) o.pad_[k] = 0;
    out[i] = o;
}

}  // namespace svthip
