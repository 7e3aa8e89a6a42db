// svt-av1-1_amd/csrc/svthip_abi.hip -- C-ABI glue of libsvtav1_hip.so (include/svtav1_hip.h).
//
// Host side of the drop-in boundary: context/stream ownership, argument validation, kernel launches.
// There is deliberately no CPU fallback here: a missing device or a failed launch is an error.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "../../include/svtav1_hip.h"
#include "me_kernels.h"

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, const char* a = "", int b = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(SVTHIP_ERR_DEVICE, "%s failed at line %d", hipGetErrorString(e_), __LINE__); \
    } while (0)

}  // namespace

struct svthip_ctx {
    int device;
    hipStream_t stream;
    // grow-only device scratch: slots 0-4 host-pointer full-pel form, 5 per-list ME arrays, 6 bi-pred SADs, 7 stored predictions,
    // 8-15 host-pointer picture / TU forms
    void* scratch[16];
    size_t scratch_bytes[16];
    // the stream the context-owned scratch was last used on, and an event to order a different stream behind it
    hipStream_t scratch_stream;
    hipEvent_t scratch_event;
    // kernel-selection overrides (svthip_set_option): per context, never read from the environment
    int32_t opt[SVTHIP_OPT_COUNT];
    // geometry the SB-origin table in slot 9 was last built for (host-pointer picture forms)
    uint32_t sb_table_w, sb_table_h;
};

namespace {

// One-time, process-wide, per device: every kernel that takes dynamic LDS gets its limit raised to what the largest legal launch needs
// (160 KB minus the kernel's static LDS).  hipFuncSetAttribute is per-FUNCTION state, so it must not be cached per context: a second
// context with a smaller search area would lower the limit under a first one's launches (round-1 defect).
std::once_flag g_attr_once[16];
hipError_t g_attr_status[16];

void set_kernel_attrs(int device)
{
    const void* kernels[] = {reinterpret_cast<const void*>(svthip::fullpel85_kernel),  reinterpret_cast<const void*>(svthip::fullpel209_kernel),
                             reinterpret_cast<const void*>(svthip::bipred_pack_kernel), reinterpret_cast<const void*>(svthip::bipred_nsq_pack_kernel),
                             reinterpret_cast<const void*>(svthip::subpel_planes_kernel), svthip::convolve_compound_kernel_ptr(0),
                             svthip::convolve_compound_kernel_ptr(1),                     svthip::convolve_compound_kernel_ptr(2),
                             svthip::convolve_compound_kernel_ptr(3)};
    hipError_t st = hipSuccess;
    for (const void* k : kernels) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, k);
        if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
        if (e != hipSuccess) st = e;
    }
    g_attr_status[device] = st;
}

// entry prologue of every call: the context's device becomes current for the calling thread
int32_t enter(svthip_ctx* c)
{
    if (!c) return fail(SVTHIP_ERR_BAD_PARAMETER, "null context%s", "");
    HIP_TRY(hipSetDevice(c->device));
    return SVTHIP_OK;
}
#define ENTER(ctx)                         \
    do {                                   \
        int32_t rc_ = enter(ctx);          \
        if (rc_) return rc_;               \
    } while (0)

// Context-owned scratch is about to be used by work enqueued on `s`: if the previous user was a different stream, order `s` behind it
// (a caller may hand any stream to a `_dev` entry; two in-flight calls of one context on two streams then serialise instead of racing).
int32_t scratch_on_stream(svthip_ctx* c, hipStream_t s)
{
    if (c->scratch_stream && c->scratch_stream != s) {
        HIP_TRY(hipEventRecord(c->scratch_event, c->scratch_stream));
        HIP_TRY(hipStreamWaitEvent(s, c->scratch_event, 0));
    }
    c->scratch_stream = s;
    return SVTHIP_OK;
}

// Growing a slot is stream-ordered: the old buffer is released with hipFreeAsync behind the last stream that used the context's scratch
// (every earlier user is ordered before that stream, see scratch_on_stream) and the new one comes from hipMallocAsync on the same
// stream, so one context's growth never synchronises the device under the other contexts' work (hipFree would: round-2 finding).
// svthip_reserve pre-sizes the slots so that steady-state calls never get here.
int32_t ensure_scratch(svthip_ctx* c, int slot, size_t bytes)
{
    if (c->scratch_bytes[slot] >= bytes) return SVTHIP_OK;
    hipStream_t os = c->scratch_stream ? c->scratch_stream : c->stream;
    if (c->scratch[slot] && hipFreeAsync(c->scratch[slot], os) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipFree(c->scratch[slot]));
    }
    c->scratch[slot] = nullptr;
    c->scratch_bytes[slot] = 0;
    size_t want = bytes + bytes / 4 + 4096;
    if (hipMallocAsync(&c->scratch[slot], want, os) != hipSuccess) {
        (void)hipGetLastError();
        c->scratch[slot] = nullptr;
        return fail(SVTHIP_ERR_INSUFFICIENT_RESOURCES, "hipMallocAsync of %s scratch failed (slot %d)", "device", slot);
    }
    c->scratch_bytes[slot] = want;
    c->scratch_stream = os;
    return SVTHIP_OK;
}

// scratch sizes of the whole-picture ME chain (slot 5: descriptors, per-list arrays, HME state; slot 7: stored predictions)
size_t me_chain_bytes(size_t n, uint32_t n_pu)
{
    const size_t desc_b = sizeof(svthip_fullpel_desc) * n, arr_b = sizeof(uint32_t) * n_pu * n;
    const size_t state_b = ((sizeof(int16_t) * SVTHIP_HME_STATE_INT16 * n) + 15) & ~(size_t)15;
    return 2 * desc_b + 4 * arr_b + state_b + 64;
}
size_t me_pred_bytes(size_t n, uint32_t n_pu) { return 2 * (size_t)(n_pu == 209 ? 14 : 4) * 4096 * n; }

// device pool of the host-pointer picture forms: per picture the padded full plane (stride = width + 136), the 1/4 and the 1/16 plane
struct HostPoolLayout {
    uint32_t fs, qs, ss;
    size_t fb, qb, sb, per;
};
HostPoolLayout host_pool_layout(uint32_t w, uint32_t h)
{
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    HostPoolLayout L;
    L.fs = w + 136; L.qs = (w >> 1) + 64; L.ss = (w >> 2) + 32;
    L.fb = al((size_t)L.fs * (h + 136)); L.qb = al((size_t)L.qs * ((h >> 1) + 64)); L.sb = al((size_t)L.ss * ((h >> 2) + 32));
    L.per = L.fb + L.qb + L.sb;
    return L;
}

// raster SB origins of a w x h picture in slot 9, rebuilt only when the geometry changes (the upload is from pageable memory, so it is
// followed by a stream synchronisation; steady-state calls skip both)
int32_t ensure_sb_table(svthip_ctx* c, uint32_t w, uint32_t h, hipStream_t s)
{
    const uint32_t nx = (w + 63) / 64, ny = (h + 63) / 64, n_sb = nx * ny;
    const bool grew = c->scratch_bytes[9] < sizeof(svthip_sb_origin) * n_sb;
    int32_t rc;
    if ((rc = ensure_scratch(c, 9, sizeof(svthip_sb_origin) * n_sb))) return rc;
    if (!grew && c->sb_table_w == w && c->sb_table_h == h) return SVTHIP_OK;
    svthip_sb_origin* sbs = new (std::nothrow) svthip_sb_origin[n_sb];
    if (!sbs) return fail(SVTHIP_ERR_INSUFFICIENT_RESOURCES, "out of host memory%s", "");
    for (uint32_t y = 0; y < ny; y++)
        for (uint32_t x = 0; x < nx; x++) sbs[y * nx + x] = svthip_sb_origin{(uint16_t)(x * 64), (uint16_t)(y * 64)};
    hipError_t e = hipMemcpyAsync(c->scratch[9], sbs, sizeof(svthip_sb_origin) * n_sb, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    delete[] sbs;
    HIP_TRY(e);
    c->sb_table_w = w;
    c->sb_table_h = h;
    return SVTHIP_OK;
}

int32_t launch_fullpel(svthip_ctx* ctx, const uint8_t* d_src, uint32_t src_stride, const uint8_t* d_ref, uint32_t ref_stride,
                       const svthip_fullpel_desc* d_desc, uint32_t n_sb, uint32_t max_sw, uint32_t max_sh,
                       uint32_t* d_sad, uint32_t* d_mv, hipStream_t s)
{
    ENTER(ctx);
    if (n_sb == 0) return SVTHIP_OK;
    if (!d_src || !d_ref || !d_desc || !d_sad || !d_mv) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (max_sw < 1 || max_sw > 127 || max_sh < 1 || max_sh > 127)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "search area must be 1..127 (%s%d)", "got ", (int)(max_sw > max_sh ? max_sw : max_sh));
    if ((src_stride & 3u) || (ref_stride & 3u) || (reinterpret_cast<uintptr_t>(d_src) & 3u))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "plane strides and the source plane base must be multiples of 4%s", "");
    const size_t lds = svthip::fullpel_lds_bytes(max_sh);
    hipLaunchKernelGGL(svthip::fullpel85_kernel, dim3(svthip::xcd_grid(n_sb)), dim3(256), lds, s, d_src, src_stride, d_ref, ref_stride,
                       reinterpret_cast<const int32_t*>(d_desc), n_sb, d_sad, d_mv);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

}  // namespace

extern "C" {

const char* svthip_last_error(void) { return g_err; }

int32_t svthip_create(int32_t device, svthip_ctx** out_ctx)
{
    if (!out_ctx) return fail(SVTHIP_ERR_BAD_PARAMETER, "out_ctx is null%s", "");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SVTHIP_ERR_DEVICE, "no HIP device available%s (this library has no CPU fallback)", "");
    if (device < 0 || device >= n) return fail(SVTHIP_ERR_BAD_PARAMETER, "device index out of range%s (%d)", "", device);
    if (device >= 16) return fail(SVTHIP_ERR_BAD_PARAMETER, "device index above 15 is not supported%s (%d)", "", device);
    HIP_TRY(hipSetDevice(device));
    std::call_once(g_attr_once[device], set_kernel_attrs, device);
    if (g_attr_status[device] != hipSuccess)
        return fail(SVTHIP_ERR_DEVICE, "raising the kernels' dynamic LDS limit failed: %s", hipGetErrorString(g_attr_status[device]));
    svthip_ctx* c = new (std::nothrow) svthip_ctx();
    if (!c) return fail(SVTHIP_ERR_INSUFFICIENT_RESOURCES, "out of host memory%s", "");
    memset(c, 0, sizeof(*c));
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(SVTHIP_ERR_DEVICE, "hipStreamCreate failed%s", "");
    }
    if (hipEventCreateWithFlags(&c->scratch_event, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(SVTHIP_ERR_DEVICE, "hipEventCreate failed%s", "");
    }
    *out_ctx = c;
    return SVTHIP_OK;
}

void svthip_destroy(svthip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 16; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    (void)hipEventDestroy(ctx->scratch_event);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* svthip_stream(svthip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int32_t svthip_synchronize(svthip_ctx* ctx)
{
    ENTER(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVTHIP_OK;
}

int32_t svthip_set_option(svthip_ctx* ctx, int32_t option, int32_t value)
{
    if (!ctx) return fail(SVTHIP_ERR_BAD_PARAMETER, "null context%s", "");
    if (option < 0 || option >= SVTHIP_OPT_COUNT) return fail(SVTHIP_ERR_BAD_PARAMETER, "unknown option%s %d", "", (int)option);
    ctx->opt[option] = value;
    return SVTHIP_OK;
}

int32_t svthip_reserve(svthip_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_pu, uint32_t n_jobs, int32_t host_forms)
{
    ENTER(ctx);
    if ((width & 7) || (height & 7) || !width || !height || width > 16384 || height > 16384)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be non-zero multiples of 8%s (width %d)", "", (int)width);
    if (n_pu != 85 && n_pu != 209) return fail(SVTHIP_ERR_BAD_PARAMETER, "n_pu must be 85 or 209%s (got %d)", "", (int)n_pu);
    if (n_jobs == 0) n_jobs = 1;
    const size_t n_sb = (size_t)((width + 63) / 64) * ((height + 63) / 64), n = n_sb * n_jobs;
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 5, me_chain_bytes(n, n_pu)))) return rc;
    if ((rc = ensure_scratch(ctx, 6, sizeof(uint32_t) * 85 * n))) return rc;
    if ((rc = ensure_scratch(ctx, 7, me_pred_bytes(n, n_pu)))) return rc;
    if (host_forms) {
        const HostPoolLayout L = host_pool_layout(width, height);
        if ((rc = ensure_scratch(ctx, 8, L.per * 3 + 256))) return rc;
        if ((rc = ensure_scratch(ctx, 9, sizeof(svthip_sb_origin) * n_sb))) return rc;
        if ((rc = ensure_scratch(ctx, 10, sizeof(svthip_me_cu_result) * n_sb * n_pu))) return rc;
        if ((rc = ensure_scratch(ctx, 11, sizeof(svthip_me_cu_result_ref) * n_sb * n_pu))) return rc;
    }
    return SVTHIP_OK;
}

int32_t svthip_me_fullpel_search_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride,
                                     const uint8_t* d_ref_plane, uint32_t ref_stride, const svthip_fullpel_desc* d_desc,
                                     uint32_t n_sb, uint32_t max_search_area_width, uint32_t max_search_area_height,
                                     uint32_t* d_best_sad, uint32_t* d_best_mv, void* stream)
{
    ENTER(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return launch_fullpel(ctx, d_src_plane, src_stride, d_ref_plane, ref_stride, d_desc, n_sb, max_search_area_width,
                          max_search_area_height, d_best_sad, d_best_mv, s);
}

static int32_t subpel_refine_common(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref_plane,
                                     uint32_t ref_stride, const svthip_fullpel_desc* d_desc, uint32_t n_sb, uint32_t max_search_area_width,
                                     uint32_t max_search_area_height, int32_t disable_8x8_refinement, int n_pu, uint32_t* d_best_sad,
                                     uint32_t* d_best_mv, void* stream, uint32_t* d_pred = nullptr,
                                     int32_t method = SVTHIP_FRACTIONAL_SSD_SEARCH)
{
    ENTER(ctx);
    if (method != SVTHIP_FRACTIONAL_SUB_SAD_SEARCH && method != SVTHIP_FRACTIONAL_FULL_SAD_SEARCH && method != SVTHIP_FRACTIONAL_SSD_SEARCH)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "fractional_search_method must be 0 (SUB_SAD), 1 (FULL_SAD) or 2 (SSD)%s, got %d", "", (int)method);
    if (n_sb == 0) return SVTHIP_OK;
    if (!d_src_plane || !d_ref_plane || !d_desc || !d_best_sad || !d_best_mv)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (max_search_area_width < 1 || max_search_area_width > 127 || max_search_area_height < 1 || max_search_area_height > 127)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "search area must be 1..127%s", "");
    if ((src_stride & 3u) || (ref_stride & 3u) || (reinterpret_cast<uintptr_t>(d_src_plane) & 3u))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "plane strides and the source plane base must be multiples of 4%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // the half-pel planes of the whole (bounded) search region interpolated once per (SB, list) in LDS, all PUs in one launch; the planes
    // of the largest legal area (127 x 127) take 152.6 KB, so every legal call fits
    const size_t lds_planes = svthip::subpel_planes_lds_bytes(max_search_area_width, max_search_area_height);
    if (lds_planes > 160 * 1024 - 512) return fail(SVTHIP_ERR_BAD_PARAMETER, "search area too large for the LDS planes%s", "");
    hipLaunchKernelGGL(svthip::subpel_planes_kernel, dim3(svthip::subpel_planes_grid(n_sb)), dim3(n_pu == 209 ? 448 : 512), lds_planes, s, d_src_plane, src_stride,
                       d_ref_plane, ref_stride, reinterpret_cast<const int32_t*>(d_desc), n_sb, (int)(disable_8x8_refinement != 0), n_pu,
                       d_best_sad, d_best_mv, d_pred, (int)method);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_me_subpel_refine_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref_plane,
                                    uint32_t ref_stride, const svthip_fullpel_desc* d_desc, uint32_t n_sb,
                                    uint32_t max_search_area_width, uint32_t max_search_area_height,
                                    int32_t disable_8x8_refinement, uint32_t* d_best_sad, uint32_t* d_best_mv, void* stream)
{
    return subpel_refine_common(ctx, d_src_plane, src_stride, d_ref_plane, ref_stride, d_desc, n_sb, max_search_area_width,
                                max_search_area_height, disable_8x8_refinement, 85, d_best_sad, d_best_mv, stream);
}

int32_t svthip_me_subpel_refine209_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref_plane,
                                       uint32_t ref_stride, const svthip_fullpel_desc* d_desc, uint32_t n_sb,
                                       uint32_t max_search_area_width, uint32_t max_search_area_height,
                                       int32_t disable_8x8_refinement, uint32_t* d_best_sad, uint32_t* d_best_mv, void* stream)
{
    return subpel_refine_common(ctx, d_src_plane, src_stride, d_ref_plane, ref_stride, d_desc, n_sb, max_search_area_width,
                                max_search_area_height, disable_8x8_refinement, 209, d_best_sad, d_best_mv, stream);
}

int32_t svthip_me_subpel_search_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref_plane,
                                    uint32_t ref_stride, const svthip_fullpel_desc* d_desc, uint32_t n_sb, uint32_t max_search_area_width,
                                    uint32_t max_search_area_height, int32_t disable_8x8_refinement, int32_t all_pu,
                                    int32_t fractional_search_method, uint32_t* d_best_sad, uint32_t* d_best_mv, void* stream)
{
    return subpel_refine_common(ctx, d_src_plane, src_stride, d_ref_plane, ref_stride, d_desc, n_sb, max_search_area_width,
                                max_search_area_height, disable_8x8_refinement, all_pu ? 209 : 85, d_best_sad, d_best_mv, stream, nullptr,
                                fractional_search_method);
}

static int32_t bipred_pack_common(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref0_plane,
                                  uint32_t ref0_stride, const svthip_fullpel_desc* d_desc0, const uint8_t* d_ref1_plane,
                                  uint32_t ref1_stride, const svthip_fullpel_desc* d_desc1, uint32_t n_sb,
                                  uint32_t max_search_area_width, uint32_t max_search_area_height, const uint32_t* d_sad0,
                                  const uint32_t* d_mv0, const uint32_t* d_sad1, const uint32_t* d_mv1, uint32_t n_lists,
                                  int32_t bipred_8x8, int n_pu, svthip_me_cu_result* d_out, void* stream)
{
    ENTER(ctx);
    if (n_sb == 0) return SVTHIP_OK;
    if (n_lists < 1 || n_lists > 2) return fail(SVTHIP_ERR_BAD_PARAMETER, "n_lists must be 1 or 2%s", "");
    if (!d_sad0 || !d_mv0 || !d_out) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    size_t lds = 0, lds_nsq = 0;
    int win_bytes = 0;
    if (n_lists == 2) {
        if (!d_src_plane || !d_ref0_plane || !d_ref1_plane || !d_desc0 || !d_desc1 || !d_sad1 || !d_mv1)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
        if (max_search_area_width < 1 || max_search_area_width > 127 || max_search_area_height < 1 || max_search_area_height > 127)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "search area must be 1..127%s", "");
        if ((src_stride & 3u) || (ref0_stride & 3u) || (ref1_stride & 3u) || (reinterpret_cast<uintptr_t>(d_src_plane) & 3u))
            return fail(SVTHIP_ERR_BAD_PARAMETER, "plane strides and the source plane base must be multiples of 4%s", "");
        lds = svthip::bipred_lds_bytes(max_search_area_width, max_search_area_height);
        lds_nsq = svthip::bipred_nsq_lds_bytes(max_search_area_width, max_search_area_height);
        win_bytes = (int)svthip::subpel_window_bytes(max_search_area_width, max_search_area_height);
        if (lds > 160 * 1024 || (n_pu == 209 && lds_nsq > 160 * 1024))
            return fail(SVTHIP_ERR_BAD_PARAMETER, "search area too large for the LDS windows%s", "");
    }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (n_pu == 85) {
        hipLaunchKernelGGL(svthip::bipred_pack_kernel, dim3(n_sb), dim3(256), lds, s, d_src_plane, src_stride, d_ref0_plane, ref0_stride,
                           reinterpret_cast<const int32_t*>(d_desc0), d_ref1_plane, ref1_stride, reinterpret_cast<const int32_t*>(d_desc1),
                           d_sad0, d_mv0, d_sad1, d_mv1, (int)n_lists, (int)bipred_8x8, win_bytes, 85, (uint32_t*)nullptr, d_out);
        HIP_TRY(hipGetLastError());
        return SVTHIP_OK;
    }
    // 209-PU mode: the squares' bi-pred SADs go through scratch slot 6 ([n_sb][85]) to the kernel that packs all 209 PUs
    uint32_t* bisad_sq = nullptr;
    if (n_lists == 2) {
        int32_t rc;
        if ((rc = ensure_scratch(ctx, 6, sizeof(uint32_t) * 85 * (size_t)n_sb))) return rc;
        if ((rc = scratch_on_stream(ctx, s))) return rc;
        bisad_sq = static_cast<uint32_t*>(ctx->scratch[6]);
        hipLaunchKernelGGL(svthip::bipred_pack_kernel, dim3(n_sb), dim3(256), lds, s, d_src_plane, src_stride, d_ref0_plane, ref0_stride,
                           reinterpret_cast<const int32_t*>(d_desc0), d_ref1_plane, ref1_stride, reinterpret_cast<const int32_t*>(d_desc1),
                           d_sad0, d_mv0, d_sad1, d_mv1, 2, 1, win_bytes, 209, bisad_sq, (svthip_me_cu_result*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(svthip::bipred_nsq_pack_kernel, dim3(n_sb), dim3(320), lds_nsq, s, d_src_plane, src_stride, d_ref0_plane, ref0_stride,
                       reinterpret_cast<const int32_t*>(d_desc0), d_ref1_plane, ref1_stride, reinterpret_cast<const int32_t*>(d_desc1),
                       d_sad0, d_mv0, d_sad1, d_mv1, (int)n_lists, win_bytes, (const uint32_t*)bisad_sq, d_out);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_me_bipred_pack_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref0_plane,
                                  uint32_t ref0_stride, const svthip_fullpel_desc* d_desc0, const uint8_t* d_ref1_plane,
                                  uint32_t ref1_stride, const svthip_fullpel_desc* d_desc1, uint32_t n_sb,
                                  uint32_t max_search_area_width, uint32_t max_search_area_height, const uint32_t* d_sad0,
                                  const uint32_t* d_mv0, const uint32_t* d_sad1, const uint32_t* d_mv1, uint32_t n_lists,
                                  int32_t bipred_8x8, svthip_me_cu_result* d_out, void* stream)
{
    return bipred_pack_common(ctx, d_src_plane, src_stride, d_ref0_plane, ref0_stride, d_desc0, d_ref1_plane, ref1_stride, d_desc1, n_sb,
                              max_search_area_width, max_search_area_height, d_sad0, d_mv0, d_sad1, d_mv1, n_lists, bipred_8x8, 85, d_out,
                              stream);
}

int32_t svthip_me_bipred_pack209_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref0_plane,
                                     uint32_t ref0_stride, const svthip_fullpel_desc* d_desc0, const uint8_t* d_ref1_plane,
                                     uint32_t ref1_stride, const svthip_fullpel_desc* d_desc1, uint32_t n_sb,
                                     uint32_t max_search_area_width, uint32_t max_search_area_height, const uint32_t* d_sad0,
                                     const uint32_t* d_mv0, const uint32_t* d_sad1, const uint32_t* d_mv1, uint32_t n_lists,
                                     svthip_me_cu_result* d_out, void* stream)
{
    return bipred_pack_common(ctx, d_src_plane, src_stride, d_ref0_plane, ref0_stride, d_desc0, d_ref1_plane, ref1_stride, d_desc1, n_sb,
                              max_search_area_width, max_search_area_height, d_sad0, d_mv0, d_sad1, d_mv1, n_lists, 1, 209, d_out, stream);
}

int32_t svthip_quantize_b_batch_dev(svthip_ctx* ctx, const int32_t* d_coeff, const svthip_quant_desc* d_desc, uint32_t n_tu,
                                    const int16_t* d_qparams, const int16_t* d_iscan, int32_t* d_qcoeff, int32_t* d_dqcoeff,
                                    uint16_t* d_eob, void* stream)
{
    ENTER(ctx);
    if (n_tu == 0) return SVTHIP_OK;
    if (!d_coeff || !d_desc || !d_qparams || !d_iscan || !d_qcoeff || !d_dqcoeff || !d_eob)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if ((reinterpret_cast<uintptr_t>(d_coeff) | reinterpret_cast<uintptr_t>(d_qcoeff) | reinterpret_cast<uintptr_t>(d_dqcoeff)) & 15u)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "coefficient pools must be 16-byte aligned%s", "");
    if (reinterpret_cast<uintptr_t>(d_iscan) & 7u) return fail(SVTHIP_ERR_BAD_PARAMETER, "iscan pool must be 8-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const uint32_t waves = n_tu < 8192u ? n_tu : 8192u;  // grid-stride beyond 2048 workgroups
    hipLaunchKernelGGL(svthip::quantize_b_batch_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, d_coeff, d_desc, n_tu, d_qparams,
                       d_iscan, d_qcoeff, d_dqcoeff, d_eob);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_me_fullpel_search209_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride, const uint8_t* d_ref_plane,
                                        uint32_t ref_stride, const svthip_fullpel_desc* d_desc, uint32_t n_sb,
                                        uint32_t max_search_area_width, uint32_t max_search_area_height, uint32_t* d_best_sad,
                                        uint32_t* d_best_mv, void* stream)
{
    ENTER(ctx);
    if (n_sb == 0) return SVTHIP_OK;
    if (!d_src_plane || !d_ref_plane || !d_desc || !d_best_sad || !d_best_mv) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (max_search_area_width < 1 || max_search_area_width > 127 || max_search_area_height < 1 || max_search_area_height > 127)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "search area must be 1..127 (%s%d)", "got ",
                    (int)(max_search_area_width > max_search_area_height ? max_search_area_width : max_search_area_height));
    if ((src_stride & 3u) || (ref_stride & 3u) || (reinterpret_cast<uintptr_t>(d_src_plane) & 3u))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "plane strides and the source plane base must be multiples of 4%s", "");
    const size_t lds = svthip::fullpel209_lds_bytes(max_search_area_height);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(svthip::fullpel209_kernel, dim3(svthip::xcd_grid(n_sb)), dim3(256), lds, s, d_src_plane, src_stride, d_ref_plane, ref_stride,
                       reinterpret_cast<const int32_t*>(d_desc), n_sb, d_best_sad, d_best_mv);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_fwd_txfm2d_batch_dev(svthip_ctx* ctx, const int16_t* d_residual, const svthip_txfm_desc* d_desc, uint32_t n_tu,
                                    uint32_t tx_width, uint32_t tx_height, uint32_t bit_depth, int32_t* d_coeff, void* stream)
{
    ENTER(ctx);
    if (!svthip::fwd_txfm2d_size_valid((int)tx_width, (int)tx_height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "unsupported transform size%s (width %d)", "", (int)tx_width);
    if (bit_depth != 8 && bit_depth != 10) return fail(SVTHIP_ERR_BAD_PARAMETER, "bit_depth must be 8 or 10%s (got %d)", "", (int)bit_depth);
    if (n_tu == 0) return SVTHIP_OK;
    if (!d_residual || !d_desc || !d_coeff) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (reinterpret_cast<uintptr_t>(d_coeff) & 15u) return fail(SVTHIP_ERR_BAD_PARAMETER, "coefficient pool must be 16-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(svthip::launch_fwd_txfm2d(d_residual, d_desc, n_tu, (int)tx_width, (int)tx_height, d_coeff, s));
    return SVTHIP_OK;
}

int32_t svthip_inv_txfm2d_add_batch_dev(svthip_ctx* ctx, const int32_t* d_coeff, const svthip_itxfm_desc* d_desc, uint32_t n_tu,
                                        uint32_t tx_width, uint32_t tx_height, uint32_t bit_depth, uint32_t recon_16bit,
                                        void* d_recon, void* stream)
{
    ENTER(ctx);
    if (!svthip::fwd_txfm2d_size_valid((int)tx_width, (int)tx_height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "unsupported transform size%s (width %d)", "", (int)tx_width);
    if (bit_depth != 8 && bit_depth != 10) return fail(SVTHIP_ERR_BAD_PARAMETER, "bit_depth must be 8 or 10%s (got %d)", "", (int)bit_depth);
    if (bit_depth == 10 && !recon_16bit) return fail(SVTHIP_ERR_BAD_PARAMETER, "10-bit reconstruction needs a 16-bit plane%s", "");
    if (n_tu == 0) return SVTHIP_OK;
    if (!d_coeff || !d_desc || !d_recon) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (reinterpret_cast<uintptr_t>(d_coeff) & 15u) return fail(SVTHIP_ERR_BAD_PARAMETER, "coefficient pool must be 16-byte aligned%s", "");
    if (recon_16bit && (reinterpret_cast<uintptr_t>(d_recon) & 1u)) return fail(SVTHIP_ERR_BAD_PARAMETER, "16-bit plane must be 2-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(svthip::launch_inv_txfm2d_add(d_coeff, d_desc, n_tu, (int)tx_width, (int)tx_height, (int)bit_depth, d_recon,
                                          recon_16bit ? 1 : 0, s));
    return SVTHIP_OK;
}

static int32_t encode_tu_common(svthip_ctx* ctx, const void* d_src, const void* d_pred, void* d_recon, int planes_16bit,
                                const svthip_tu_desc* d_desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height,
                                const int16_t* d_qparams, const int16_t* d_iscan, int32_t* d_coeff, int32_t* d_qcoeff,
                                int32_t* d_dqcoeff, uint16_t* d_eob, uint64_t* d_three_quad_energy, uint64_t* d_distortion, void* stream)
{
    ENTER(ctx);
    if (!svthip::fwd_txfm2d_size_valid((int)tx_width, (int)tx_height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "unsupported transform size%s (width %d)", "", (int)tx_width);
    if (n_tu == 0) return SVTHIP_OK;
    if (!d_src || !d_pred || !d_recon || !d_desc || !d_qparams || !d_iscan || !d_qcoeff || !d_eob)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if ((reinterpret_cast<uintptr_t>(d_coeff) | reinterpret_cast<uintptr_t>(d_qcoeff) | reinterpret_cast<uintptr_t>(d_dqcoeff)) & 15u)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "coefficient pools must be 16-byte aligned%s", "");
    if (reinterpret_cast<uintptr_t>(d_iscan) & 7u) return fail(SVTHIP_ERR_BAD_PARAMETER, "iscan pool must be 8-byte aligned%s", "");
    if ((reinterpret_cast<uintptr_t>(d_three_quad_energy) | reinterpret_cast<uintptr_t>(d_distortion)) & 7u)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "energy / distortion outputs must be 8-byte aligned%s", "");
    if (planes_16bit && ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_pred) | reinterpret_cast<uintptr_t>(d_recon)) & 1u))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "16-bit planes must be 2-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(svthip::launch_encode_tu(d_src, d_pred, d_recon, planes_16bit, d_desc, n_tu, (int)tx_width, (int)tx_height, d_qparams, d_iscan,
                                     d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_three_quad_energy, d_distortion,
                                     (uint32_t)ctx->opt[SVTHIP_OPT_TQ_MAX_WORKGROUPS], s));
    return SVTHIP_OK;
}

int32_t svthip_encode_tu_batch_dev(svthip_ctx* ctx, const uint8_t* d_src, const uint8_t* d_pred, uint8_t* d_recon,
                                   const svthip_tu_desc* d_desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height,
                                   const int16_t* d_qparams, const int16_t* d_iscan, int32_t* d_coeff, int32_t* d_qcoeff,
                                   int32_t* d_dqcoeff, uint16_t* d_eob, uint64_t* d_three_quad_energy, uint64_t* d_distortion,
                                   void* stream)
{
    return encode_tu_common(ctx, d_src, d_pred, d_recon, 0, d_desc, n_tu, tx_width, tx_height, d_qparams, d_iscan, d_coeff, d_qcoeff,
                            d_dqcoeff, d_eob, d_three_quad_energy, d_distortion, stream);
}

int32_t svthip_encode_tu16_batch_dev(svthip_ctx* ctx, const uint16_t* d_src, const uint16_t* d_pred, uint16_t* d_recon,
                                     const svthip_tu_desc* d_desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height,
                                     const int16_t* d_qparams, const int16_t* d_iscan, int32_t* d_coeff, int32_t* d_qcoeff,
                                     int32_t* d_dqcoeff, uint16_t* d_eob, uint64_t* d_three_quad_energy, uint64_t* d_distortion,
                                     void* stream)
{
    return encode_tu_common(ctx, d_src, d_pred, d_recon, 1, d_desc, n_tu, tx_width, tx_height, d_qparams, d_iscan, d_coeff, d_qcoeff,
                            d_dqcoeff, d_eob, d_three_quad_energy, d_distortion, stream);
}

int32_t svthip_me_hme_search_center_batch_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                              const svthip_pa_picture* ref, uint32_t n_jobs, const svthip_me_params* params,
                                              uint32_t list_index, const svthip_sb_origin* d_sb, uint32_t n_sb,
                                              const uint32_t* d_l0_best_mv64, uint32_t l0_mv_stride, svthip_fullpel_desc* d_desc,
                                              int16_t* d_center, int16_t* d_hme_state, void* stream)
{
    ENTER(ctx);
    if (n_sb == 0 || n_jobs == 0) return SVTHIP_OK;
    if (!d_pool || !cur || !ref || !params || !d_sb || !d_desc) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (list_index > 1) return fail(SVTHIP_ERR_BAD_PARAMETER, "list_index must be 0 or 1%s", "");
    if (list_index == 1 && !d_l0_best_mv64 && params->temporal_layer_index > 0)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "list 1 needs the list-0 64x64 MVs (hme_mv_center_check direct candidate)%s", "");
    const svthip_me_params& P = *params;
    if (P.number_hme_search_region_in_width < 1 || P.number_hme_search_region_in_width > 2 ||
        P.number_hme_search_region_in_height < 1 || P.number_hme_search_region_in_height > 2)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "HME search regions must be 1..2 per axis%s", "");
    for (uint32_t j = 0; j < n_jobs; j++) {
        const svthip_pa_picture *c = cur + j, *r = ref + j;
        if ((c->width & 7) || (c->height & 7) || c->width != r->width || c->height != r->height || c->width != cur->width ||
            c->height != cur->height)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be equal multiples of 8%s (job %d)", "", (int)j);
        if ((c->full_stride & 3u) || (r->full_stride & 3u) || (c->full_offset & 3))
            return fail(SVTHIP_ERR_BAD_PARAMETER, "full-resolution strides / current-plane offset must be multiples of 4%s (job %d)", "",
                        (int)j);
        const int64_t max_off = (c->full_offset > r->full_offset ? c->full_offset : r->full_offset) +
                                (int64_t)(c->height + 136) * (c->full_stride > r->full_stride ? c->full_stride : r->full_stride);
        if (max_off > 0x7fffffffLL) return fail(SVTHIP_ERR_BAD_PARAMETER, "picture pool offsets must fit 31 bits%s (job %d)", "", (int)j);
    }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const uint32_t mvs = l0_mv_stride ? l0_mv_stride : 1u;
    for (uint32_t j0 = 0; j0 < n_jobs; j0 += SVTHIP_HME_MAX_JOBS) {
        const uint32_t nj = (n_jobs - j0 < SVTHIP_HME_MAX_JOBS) ? n_jobs - j0 : SVTHIP_HME_MAX_JOBS;
        svthip::HmeJobTable jt;
        memset(&jt, 0, sizeof(jt));
        for (uint32_t j = 0; j < nj; j++) {
            jt.cur[j] = cur[j0 + j];
            jt.ref[j] = ref[j0 + j];
        }
        const size_t base = (size_t)j0 * n_sb;
        const uint32_t* mv64 = d_l0_best_mv64 ? d_l0_best_mv64 + base * mvs : nullptr;
        int16_t* cen = d_center ? d_center + 2 * base : nullptr;
        int16_t* st = d_hme_state ? d_hme_state + SVTHIP_HME_STATE_INT16 * base : nullptr;
        hipLaunchKernelGGL(svthip::hme_center_kernel, dim3(svthip::xcd_grid(n_sb * nj)), dim3(256), 0, s, d_pool, jt, P, list_index, d_sb, n_sb, nj, mv64,
                           mvs, d_desc + base, cen, st);
        HIP_TRY(hipGetLastError());
    }
    return SVTHIP_OK;
}

int32_t svthip_me_hme_search_center_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                        const svthip_pa_picture* ref, const svthip_me_params* params, uint32_t list_index,
                                        const svthip_sb_origin* d_sb, uint32_t n_sb, const uint32_t* d_l0_best_mv64,
                                        uint32_t l0_mv_stride, svthip_fullpel_desc* d_desc, int16_t* d_center,
                                        int16_t* d_hme_state, void* stream)
{
    return svthip_me_hme_search_center_batch_dev(ctx, d_pool, cur, ref, 1, params, list_index, d_sb, n_sb, d_l0_best_mv64, l0_mv_stride,
                                                 d_desc, d_center, d_hme_state, stream);
}

static int32_t motion_estimate_batch_common(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                            const svthip_pa_picture* ref0, const svthip_pa_picture* ref1, uint32_t n_jobs,
                                            const svthip_me_params* params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                            const svthip_sb_origin* d_sb, uint32_t n_sb, uint32_t n_pu, svthip_me_cu_result* d_out,
                                            uint32_t* d_list_sad, uint32_t* d_list_mv, void* stream)
{
    ENTER(ctx);
    if (n_sb == 0 || n_jobs == 0) return SVTHIP_OK;
    if (!d_pool || !cur || !ref0 || !params || !d_sb || !d_out) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    for (uint32_t j = 0; j < n_jobs; j++)  // the per-SB kernels take one stride per plane role
        if (cur[j].full_stride != cur[0].full_stride || ref0[j].full_stride != ref0[0].full_stride ||
            (ref1 && ref1[j].full_stride != ref1[0].full_stride))
            return fail(SVTHIP_ERR_BAD_PARAMETER, "all pictures of a batch must share their full-resolution strides%s (job %d)", "", (int)j);
    const uint32_t n_lists = ref1 ? 2u : 1u;
    const size_t n = (size_t)n_jobs * n_sb;
    // scratch: slot 5 holds  desc[2][n] | sad[2][n][n_pu] | mv[2][n][n_pu] | hme_state[n][25]
    const size_t desc_b = sizeof(svthip_fullpel_desc) * n, arr_b = sizeof(uint32_t) * n_pu * n;
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 5, me_chain_bytes(n, n_pu)))) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->scratch[5]);
    svthip_fullpel_desc* desc[2] = {reinterpret_cast<svthip_fullpel_desc*>(base), reinterpret_cast<svthip_fullpel_desc*>(base + desc_b)};
    uint32_t* sad[2] = {reinterpret_cast<uint32_t*>(base + 2 * desc_b), reinterpret_cast<uint32_t*>(base + 2 * desc_b + arr_b)};
    uint32_t* mv[2] = {reinterpret_cast<uint32_t*>(base + 2 * desc_b + 2 * arr_b), reinterpret_cast<uint32_t*>(base + 2 * desc_b + 3 * arr_b)};
    int16_t* state = reinterpret_cast<int16_t*>(base + 2 * desc_b + 4 * arr_b);
    if (d_list_sad && d_list_mv) {  // caller wants the per-list arrays: write them in place
        sad[0] = d_list_sad; sad[1] = d_list_sad + n_pu * n;
        mv[0] = d_list_mv; mv[1] = d_list_mv + n_pu * n;
    }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if ((rc = scratch_on_stream(ctx, s))) return rc;  // slots 5 / 7 are about to be used by work on `s`
    const uint32_t sw = params->search_area_width < 127 ? params->search_area_width : 127;
    const uint32_t sh = params->search_area_height < 127 ? params->search_area_height : 127;
    const svthip_pa_picture* refs[2] = {ref0, ref1};
    // B pictures with sub-pel on: the sub-pel kernels also store each PU's prediction at its refined MV (scratch slot 7,
    // [2 lists][n][slots][4096 bytes]) and the bi-prediction stage averages the stored blocks instead of interpolating again
    const size_t pred_b = (size_t)(n_pu == 209 ? 14 : 4) * 4096 * n;
    uint8_t* pred[2] = {nullptr, nullptr};
    if (n_lists == 2 && use_subpel_flag) {
        if ((rc = ensure_scratch(ctx, 7, me_pred_bytes(n, n_pu)))) return rc;
        pred[0] = static_cast<uint8_t*>(ctx->scratch[7]);
        pred[1] = pred[0] + pred_b;
    }
    // seven launches whatever the number of pictures: per list search centres -> full-pel -> sub-pel, then bi-prediction + packing
    for (uint32_t l = 0; l < n_lists; l++) {
        if ((rc = svthip_me_hme_search_center_batch_dev(ctx, d_pool, cur, refs[l], n_jobs, params, l, d_sb, n_sb, l ? mv[0] : nullptr, n_pu,
                                                        desc[l], nullptr, state, s)))
            return rc;
        rc = n_pu == 209 ? svthip_me_fullpel_search209_dev(ctx, d_pool, cur->full_stride, d_pool, refs[l]->full_stride, desc[l], (uint32_t)n, sw, sh,
                                                           sad[l], mv[l], s)
                         : launch_fullpel(ctx, d_pool, cur->full_stride, d_pool, refs[l]->full_stride, desc[l], (uint32_t)n, sw, sh, sad[l], mv[l], s);
        if (rc) return rc;
        if (use_subpel_flag &&
            (rc = subpel_refine_common(ctx, d_pool, cur->full_stride, d_pool, refs[l]->full_stride, desc[l], (uint32_t)n, sw, sh,
                                       cu8x8_mode == 1, (int)n_pu, sad[l], mv[l], s, reinterpret_cast<uint32_t*>(pred[l]))))
            return rc;
    }
    if (pred[0]) {
        hipLaunchKernelGGL(svthip::bipred_stored_pack_kernel, dim3((uint32_t)n), dim3(256), 0, s, d_pool, cur->full_stride,
                           reinterpret_cast<const int32_t*>(desc[0]), (const uint8_t*)pred[0], (const uint8_t*)pred[1],
                           (const uint32_t*)sad[0], (const uint32_t*)mv[0], (const uint32_t*)sad[1], (const uint32_t*)mv[1], (int)n_pu,
                           (int)(cu8x8_mode == 0), d_out);
        HIP_TRY(hipGetLastError());
        return SVTHIP_OK;
    }
    return bipred_pack_common(ctx, d_pool, cur->full_stride, d_pool, ref0->full_stride, desc[0], n_lists == 2 ? d_pool : nullptr,
                              n_lists == 2 ? ref1->full_stride : 0, n_lists == 2 ? desc[1] : nullptr, (uint32_t)n, sw, sh, sad[0], mv[0],
                              n_lists == 2 ? sad[1] : nullptr, n_lists == 2 ? mv[1] : nullptr, n_lists, cu8x8_mode == 0, (int)n_pu, d_out, s);
}

int32_t svthip_motion_estimate_batch_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                         const svthip_pa_picture* ref0, const svthip_pa_picture* ref1, uint32_t n_jobs,
                                         const svthip_me_params* params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                         const svthip_sb_origin* d_sb, uint32_t n_sb, svthip_me_cu_result* d_out,
                                         uint32_t* d_list_sad, uint32_t* d_list_mv, void* stream)
{
    return motion_estimate_batch_common(ctx, d_pool, cur, ref0, ref1, n_jobs, params, use_subpel_flag, cu8x8_mode, d_sb, n_sb, 85, d_out,
                                        d_list_sad, d_list_mv, stream);
}

int32_t svthip_motion_estimate209_batch_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                            const svthip_pa_picture* ref0, const svthip_pa_picture* ref1, uint32_t n_jobs,
                                            const svthip_me_params* params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                            const svthip_sb_origin* d_sb, uint32_t n_sb, svthip_me_cu_result* d_out,
                                            uint32_t* d_list_sad, uint32_t* d_list_mv, void* stream)
{
    return motion_estimate_batch_common(ctx, d_pool, cur, ref0, ref1, n_jobs, params, use_subpel_flag, cu8x8_mode, d_sb, n_sb, 209, d_out,
                                        d_list_sad, d_list_mv, stream);
}


int32_t svthip_motion_estimate_picture_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur,
                                           const svthip_pa_picture* ref0, const svthip_pa_picture* ref1,
                                           const svthip_me_params* params, int32_t use_subpel_flag, int32_t cu8x8_mode,
                                           const svthip_sb_origin* d_sb, uint32_t n_sb, svthip_me_cu_result* d_out,
                                           uint32_t* d_list_sad, uint32_t* d_list_mv, void* stream)
{
    return svthip_motion_estimate_batch_dev(ctx, d_pool, cur, ref0, ref1, 1, params, use_subpel_flag, cu8x8_mode, d_sb, n_sb, d_out,
                                            d_list_sad, d_list_mv, stream);
}

int32_t svthip_me_fullpel_search(svthip_ctx* ctx, const uint8_t* src_plane, size_t src_plane_bytes, uint32_t src_stride,
                                 const uint8_t* ref_plane, size_t ref_plane_bytes, uint32_t ref_stride,
                                 const svthip_fullpel_desc* desc, uint32_t n_sb, uint32_t* best_sad, uint32_t* best_mv)
{
    ENTER(ctx);
    if (n_sb == 0) return SVTHIP_OK;
    if (!src_plane || !ref_plane || !desc || !best_sad || !best_mv)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    uint32_t max_sw = 1, max_sh = 1;
    for (uint32_t i = 0; i < n_sb; i++) {
        const svthip_fullpel_desc& d = desc[i];
        if (d.search_area_width < 1 || d.search_area_width > 127 || d.search_area_height < 1 || d.search_area_height > 127)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: search area must be 1..127", "", (int)i);
        if (d.src_offset < 0 || (d.src_offset & 3) || (size_t)d.src_offset + 63u * src_stride + 64u > src_plane_bytes)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: source block outside the plane or not 4-byte aligned", "", (int)i);
        const size_t ref_end = (size_t)d.ref_offset + (size_t)(d.search_area_height + 62) * ref_stride + d.search_area_width + 63;
        if (d.ref_offset < 0 || ref_end > ref_plane_bytes)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: search window outside the reference plane", "", (int)i);
        if ((uint32_t)d.search_area_width > max_sw) max_sw = d.search_area_width;
        if ((uint32_t)d.search_area_height > max_sh) max_sh = d.search_area_height;
    }
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 0, src_plane_bytes + 16))) return rc;
    if ((rc = ensure_scratch(ctx, 1, ref_plane_bytes + 16))) return rc;
    if ((rc = ensure_scratch(ctx, 2, sizeof(svthip_fullpel_desc) * n_sb))) return rc;
    if ((rc = ensure_scratch(ctx, 3, sizeof(uint32_t) * 85 * n_sb))) return rc;
    if ((rc = ensure_scratch(ctx, 4, sizeof(uint32_t) * 85 * n_sb))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = scratch_on_stream(ctx, s))) return rc;
    // copies from / to the caller's buffers are in flight inside `queued`: every exit synchronises the stream first
    auto queued = [&]() -> int32_t {
        HIP_TRY(hipMemcpyAsync(ctx->scratch[0], src_plane, src_plane_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(ctx->scratch[1], ref_plane, ref_plane_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(ctx->scratch[2], desc, sizeof(svthip_fullpel_desc) * n_sb, hipMemcpyHostToDevice, s));
        int32_t r = launch_fullpel(ctx, (const uint8_t*)ctx->scratch[0], src_stride, (const uint8_t*)ctx->scratch[1], ref_stride,
                                   (const svthip_fullpel_desc*)ctx->scratch[2], n_sb, max_sw, max_sh, (uint32_t*)ctx->scratch[3],
                                   (uint32_t*)ctx->scratch[4], s);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(best_sad, ctx->scratch[3], sizeof(uint32_t) * 85 * n_sb, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(best_mv, ctx->scratch[4], sizeof(uint32_t) * 85 * n_sb, hipMemcpyDeviceToHost, s));
        return SVTHIP_OK;
    };
    rc = queued();
    const hipError_t sync_e = hipStreamSynchronize(s);
    if (rc) return rc;
    HIP_TRY(sync_e);
    return SVTHIP_OK;
}

int32_t svthip_pa_derive_planes_dev(svthip_ctx* ctx, uint8_t* d_pool, const svthip_pa_picture* pics, uint32_t n_pics, int32_t want_quarter,
                                    int32_t want_sixteenth, void* stream)
{
    ENTER(ctx);
    if (n_pics == 0) return SVTHIP_OK;
    if (!d_pool || !pics) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    uint32_t max_dw = 0;
    for (uint32_t j = 0; j < n_pics; j++) {
        const svthip_pa_picture& p = pics[j];
        if ((p.width & 7) || (p.height & 7) || p.width == 0 || p.height == 0)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be non-zero multiples of 8%s (picture %d)", "", (int)j);
        if (p.full_stride < (uint32_t)p.width + 136u || (p.full_stride & 3u) || (p.full_offset & 3))
            return fail(SVTHIP_ERR_BAD_PARAMETER, "full-resolution stride must be a multiple of 4 and >= width + 136%s (picture %d)", "", (int)j);
        if (want_quarter && p.quarter_stride < (uint32_t)(p.width >> 1) + 64u)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "quarter stride must be >= width/2 + 64%s (picture %d)", "", (int)j);
        if (want_sixteenth && p.sixteenth_stride < (uint32_t)(p.width >> 2) + 32u)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "sixteenth stride must be >= width/4 + 32%s (picture %d)", "", (int)j);
        const uint32_t dw = ((uint32_t)p.width + 136u + 3u) / 4u * ((uint32_t)p.height + 136u);
        if (dw > max_dw) max_dw = dw;
    }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    for (uint32_t j0 = 0; j0 < n_pics; j0 += SVTHIP_HME_MAX_JOBS) {
        const uint32_t nj = (n_pics - j0 < SVTHIP_HME_MAX_JOBS) ? n_pics - j0 : SVTHIP_HME_MAX_JOBS;
        svthip::PaJobTable jt;
        memset(&jt, 0, sizeof(jt));
        for (uint32_t j = 0; j < nj; j++) jt.pic[j] = pics[j0 + j];
        const uint32_t bx = (max_dw + 255u) / 256u;
        hipLaunchKernelGGL(svthip::pa_derive_planes_kernel, dim3(bx < 1024u ? bx : 1024u, 3, nj), dim3(256), 0, s, d_pool, jt, (int)want_quarter,
                           (int)want_sixteenth);
        HIP_TRY(hipGetLastError());
    }
    return SVTHIP_OK;
}

int32_t svthip_pad_plane_dev(svthip_ctx* ctx, void* d_plane, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_width,
                             uint32_t pad_height, uint32_t sample_bytes, void* stream)
{
    ENTER(ctx);
    if (!d_plane) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (sample_bytes != 1 && sample_bytes != 2) return fail(SVTHIP_ERR_BAD_PARAMETER, "sample_bytes must be 1 or 2%s (got %d)", "", (int)sample_bytes);
    if (width == 0 || height == 0 || stride < width + 2 * pad_width || width > 16384 || height > 16384 || pad_width > 1024 || pad_height > 1024)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "bad plane geometry%s (stride %d)", "", (int)stride);
    if (sample_bytes == 2 && (reinterpret_cast<uintptr_t>(d_plane) & 1u)) return fail(SVTHIP_ERR_BAD_PARAMETER, "16-bit plane must be 2-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(svthip::launch_pad_plane(d_plane, stride, (int)width, (int)height, (int)pad_width, (int)pad_height, (int)sample_bytes, s));
    return SVTHIP_OK;
}

int32_t svthip_sad_loop_batch_dev(svthip_ctx* ctx, const uint8_t* d_src, uint32_t src_stride, const uint8_t* d_ref, uint32_t ref_stride,
                                  uint32_t ref_stride_raw, const svthip_sad_loop_desc* d_desc, uint32_t n_blocks, uint32_t width, uint32_t height,
                                  uint32_t search_area_width, uint32_t search_area_height, uint32_t* d_best_sad, int16_t* d_best_xy, void* stream)
{
    ENTER(ctx);
    if (width < 4 || width > 64 || (width & 3u) || height < 1 || height > 64)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "block must be 4..64 wide (multiple of 4) and 1..64 high%s (width %d)", "", (int)width);
    if (!search_area_width || !search_area_height || search_area_width * search_area_height > 4096u)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "search area must hold 1..4096 positions%s (width %d)", "", (int)search_area_width);
    if (!ref_stride_raw || (ref_stride != ref_stride_raw && ref_stride != 2 * ref_stride_raw))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "ref_stride must be ref_stride_raw or twice it%s (got %d)", "", (int)ref_stride);
    if (n_blocks == 0) return SVTHIP_OK;
    if (!d_src || !d_ref || !d_desc || !d_best_sad || !d_best_xy) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if ((width == 4 || width == 8 || width == 16 || width == 32 || width == 64) && !ctx->opt[SVTHIP_OPT_SADLOOP_GENERIC]) {
        // packed-SAD kernel (8 / 12 / 16 positions per lane, several blocks per workgroup); falls through to the generic one when its
        // slightly wider window rows do not fit
        const size_t qs = svthip::sad_loop_qsad_lds_bytes((int)width, (int)height, (int)search_area_width, (int)search_area_height,
                                                          (int)(ref_stride / ref_stride_raw));
        if (qs <= 64 * 1024) {
            HIP_TRY(svthip::launch_sad_loop_qsad(d_src, src_stride, d_ref, ref_stride, ref_stride_raw, d_desc, n_blocks, (int)width, (int)height,
                                                 (int)search_area_width, (int)search_area_height, d_best_sad, d_best_xy, s));
            return SVTHIP_OK;
        }
    }
    const size_t slice = svthip::sad_loop_slice_bytes((int)width, (int)height, (int)search_area_width, (int)search_area_height,
                                                      (int)(ref_stride / ref_stride_raw));
    if (slice * 4 > 64 * 1024) return fail(SVTHIP_ERR_BAD_PARAMETER, "block + search window too large for the LDS slice%s (%d bytes)", "", (int)slice);
    hipLaunchKernelGGL(svthip::sad_loop_kernel, dim3((n_blocks + 3) / 4), dim3(256), slice * 4, s, d_src, src_stride, d_ref, ref_stride, ref_stride_raw,
                       d_desc, n_blocks, (int)width, (int)height, (int)search_area_width, (int)search_area_height, (int)slice, d_best_sad, d_best_xy);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_av1_convolve_sr_batch_dev(svthip_ctx* ctx, const uint8_t* d_src, uint32_t src_stride, uint8_t* d_dst, uint32_t dst_stride,
                                         const svthip_convolve_desc* d_desc, uint32_t n_blocks, uint32_t width, uint32_t height, void* stream)
{
    ENTER(ctx);
    if (!svthip::convolve_size_valid((int)width, (int)height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "not an AV1 block size%s (width %d)", "", (int)width);
    if (n_blocks == 0) return SVTHIP_OK;
    if (!d_src || !d_dst || !d_desc) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (reinterpret_cast<uintptr_t>(d_desc) & 15u) return fail(SVTHIP_ERR_BAD_PARAMETER, "descriptor array must be 16-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (svthip::convolve_mfma_size_valid((int)width, (int)height) && !ctx->opt[SVTHIP_OPT_CONVOLVE_VALU]) {
        // sides that are multiples of 32: both passes as exact i8 matrix products on the matrix cores (ip_convolve_mfma.hip)
        HIP_TRY(svthip::launch_av1_convolve_sr_mfma(d_src, src_stride, d_dst, dst_stride, d_desc, n_blocks, (int)width, (int)height, s));
        return SVTHIP_OK;
    }
    HIP_TRY(svthip::launch_av1_convolve_sr(d_src, src_stride, d_dst, dst_stride, d_desc, n_blocks, (int)width, (int)height, s));
    return SVTHIP_OK;
}

int32_t svthip_me_results_to_ref_layout_dev(svthip_ctx* ctx, const svthip_me_cu_result* d_in, uint32_t n, svthip_me_cu_result_ref* d_out,
                                            void* stream)
{
    ENTER(ctx);
    if (n == 0) return SVTHIP_OK;
    if (!d_in || !d_out) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(svthip::me_results_ref_layout_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_in, n, d_out);
    HIP_TRY(hipGetLastError());
    return SVTHIP_OK;
}

int32_t svthip_av1_convolve_compound_batch_dev(svthip_ctx* ctx, const uint8_t* d_src0, uint32_t src0_stride, const uint8_t* d_src1, uint32_t src1_stride,
                                               uint8_t* d_dst, uint32_t dst_stride, const svthip_convolve_compound_desc* d_desc, uint32_t n_blocks,
                                               uint32_t width, uint32_t height, void* stream)
{
    ENTER(ctx);
    if (!svthip::convolve_size_valid((int)width, (int)height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "not an AV1 block size%s (width %d)", "", (int)width);
    if (n_blocks == 0) return SVTHIP_OK;
    if (!d_src0 || !d_src1 || !d_dst || !d_desc) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (reinterpret_cast<uintptr_t>(d_desc) & 15u) return fail(SVTHIP_ERR_BAD_PARAMETER, "descriptor array must be 16-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (svthip::convolve_mfma_size_valid((int)width, (int)height) && !ctx->opt[SVTHIP_OPT_CONVOLVE_VALU]) {
        HIP_TRY(svthip::launch_av1_convolve_compound_mfma(d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, n_blocks, (int)width,
                                                          (int)height, s));
        return SVTHIP_OK;
    }
    HIP_TRY(svthip::launch_av1_convolve_compound(d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_desc, n_blocks, (int)width, (int)height, s));
    return SVTHIP_OK;
}

int32_t svthip_av1_highbd_convolve_batch_dev(svthip_ctx* ctx, const uint16_t* d_src0, uint32_t src0_stride, const uint16_t* d_src1, uint32_t src1_stride,
                                             uint16_t* d_dst, uint32_t dst_stride, const void* d_desc, int32_t compound, uint32_t n_blocks, uint32_t width,
                                             uint32_t height, uint32_t bit_depth, void* stream)
{
    ENTER(ctx);
    if (!svthip::convolve_size_valid((int)width, (int)height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "not an AV1 block size%s (width %d)", "", (int)width);
    if (bit_depth != 10) return fail(SVTHIP_ERR_BAD_PARAMETER, "bit_depth must be 10%s (got %d)", "", (int)bit_depth);
    if (n_blocks == 0) return SVTHIP_OK;
    if (!d_src0 || (compound && !d_src1) || !d_dst || !d_desc) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if ((reinterpret_cast<uintptr_t>(d_desc) & 15u) || (reinterpret_cast<uintptr_t>(d_src0) & 1u) || (reinterpret_cast<uintptr_t>(d_src1) & 1u) ||
        (reinterpret_cast<uintptr_t>(d_dst) & 1u))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "descriptor array must be 16-byte aligned, planes 2-byte aligned%s", "");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(svthip::launch_av1_highbd_convolve(d_src0, src0_stride, compound ? d_src1 : d_src0, compound ? src1_stride : src0_stride, d_dst, dst_stride,
                                               d_desc, compound != 0, n_blocks, (int)width, (int)height, (int)bit_depth, s));
    return SVTHIP_OK;
}

int32_t svthip_open_loop_intra_search_batch_dev(svthip_ctx* ctx, const uint8_t* d_pool, const svthip_pa_picture* cur, uint32_t n_jobs,
                                                const svthip_ois_params* params, const svthip_sb_origin* d_sb, uint32_t n_sb,
                                                const svthip_me_cu_result* d_me, uint32_t me_pu_stride, uint32_t* d_cand, uint8_t* d_total,
                                                void* stream)
{
    ENTER(ctx);
    if (n_jobs == 0 || n_sb == 0) return SVTHIP_OK;
    if (!d_pool || !cur || !params || !d_sb || !d_cand || !d_total) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (params->temporal_layer_index > 5) return fail(SVTHIP_ERR_BAD_PARAMETER, "temporal_layer_index must be 0..5%s (got %d)", "", (int)params->temporal_layer_index);
    const bool general = !params->slice_is_intra && !(params->temporal_layer_index == 0 && !params->input_resolution_4k) &&
                         !params->limit_ois_to_dc_mode_flag;
    if (general && (!d_me || (me_pu_stride != 85 && me_pu_stride != 209)))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "this picture's branch reads the ME distortions: d_me with me_pu_stride 85 or 209 is required%s (stride %d)", "", (int)me_pu_stride);
    for (uint32_t j = 0; j < n_jobs; j++) {
        const svthip_pa_picture& p = cur[j];
        if ((p.width & 7) || (p.height & 7) || p.width == 0 || p.height == 0 || p.width != cur[0].width || p.height != cur[0].height)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be equal non-zero multiples of 8%s (picture %d)", "", (int)j);
        if (p.full_stride < (uint32_t)p.width + 136u || (p.full_stride & 3u) || (p.full_offset & 3) || p.full_offset < 0)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "full-resolution stride must be a multiple of 4 and >= width + 136, offset a multiple of 4%s (picture %d)", "", (int)j);
    }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    for (uint32_t j0 = 0; j0 < n_jobs; j0 += SVTHIP_HME_MAX_JOBS) {
        const uint32_t nj = (n_jobs - j0 < SVTHIP_HME_MAX_JOBS) ? n_jobs - j0 : SVTHIP_HME_MAX_JOBS;
        svthip::PaJobTable jt;
        memset(&jt, 0, sizeof(jt));
        for (uint32_t j = 0; j < nj; j++) jt.pic[j] = cur[j0 + j];
        const size_t first = (size_t)j0 * n_sb;
        hipLaunchKernelGGL(svthip::ois_kernel, dim3(svthip::xcd_grid(n_sb * nj)), dim3(256), 0, s, d_pool, jt, *params, d_sb, n_sb, nj,
                           d_me ? d_me + first * me_pu_stride : nullptr, me_pu_stride, d_cand + first * 85 * 18, d_total + first * 85);
        HIP_TRY(hipGetLastError());
    }
    return SVTHIP_OK;
}

int32_t svthip_motion_estimate_picture(svthip_ctx* ctx, const svthip_host_picture* cur, const svthip_host_picture* ref0,
                                       const svthip_host_picture* ref1, const svthip_me_params* params, int32_t use_subpel_flag,
                                       int32_t cu8x8_mode, uint32_t n_pu, void* const* me_results)
{
    ENTER(ctx);
    if (!cur || !ref0 || !params || !me_results) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    if (n_pu != 85 && n_pu != 209) return fail(SVTHIP_ERR_BAD_PARAMETER, "n_pu must be 85 or 209%s (got %d)", "", (int)n_pu);
    const svthip_host_picture* hp[3] = {cur, ref0, ref1};
    const int n_pic = ref1 ? 3 : 2;
    const uint32_t w = cur->width, h = cur->height;
    if ((w & 7) || (h & 7) || !w || !h) return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be non-zero multiples of 8%s", "");
    for (int i = 0; i < n_pic; i++) {
        if (!hp[i]->buffer_y || hp[i]->width != w || hp[i]->height != h || hp[i]->origin_x != 68 || hp[i]->origin_y != 68 ||
            hp[i]->stride_y < w + 136u)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "picture %s%d: needs a luma plane of the same size with origin (68,68) and stride >= width + 136", "", i);
    }
    const uint32_t n_sb = ((w + 63) / 64) * ((h + 63) / 64);
    for (uint32_t i = 0; i < n_sb; i++)  // every caller row is checked BEFORE any work is queued: no transfer ever outlives a failed call
        if (!me_results[i]) return fail(SVTHIP_ERR_BAD_PARAMETER, "me_results[%s%d] is null", "", (int)i);
    const HostPoolLayout L = host_pool_layout(w, h);
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 8, L.per * n_pic + 256))) return rc;
    if ((rc = ensure_scratch(ctx, 10, sizeof(svthip_me_cu_result) * (size_t)n_sb * n_pu))) return rc;
    if ((rc = ensure_scratch(ctx, 11, sizeof(svthip_me_cu_result_ref) * (size_t)n_sb * n_pu))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = scratch_on_stream(ctx, s))) return rc;
    if ((rc = ensure_sb_table(ctx, w, h, s))) return rc;
    // from here on copies from / to the caller's buffers are in flight: every exit synchronises the stream first
    auto queued = [&]() -> int32_t {
        uint8_t* pool = static_cast<uint8_t*>(ctx->scratch[8]);
        svthip_pa_picture pd[3];
        for (int i = 0; i < n_pic; i++) {
            pd[i].full_offset = (int64_t)(L.per * i);
            pd[i].quarter_offset = (int64_t)(L.per * i + L.fb);
            pd[i].sixteenth_offset = (int64_t)(L.per * i + L.fb + L.qb);
            pd[i].full_stride = L.fs; pd[i].quarter_stride = L.qs; pd[i].sixteenth_stride = L.ss;
            pd[i].width = (uint16_t)w; pd[i].height = (uint16_t)h;
            // the picture rows only (borders and decimated planes are derived on the device, bit-identically to Picture Analysis)
            HIP_TRY(hipMemcpy2DAsync(pool + L.per * i + (size_t)68 * L.fs + 68, L.fs, hp[i]->buffer_y + (size_t)68 * hp[i]->stride_y + 68,
                                     hp[i]->stride_y, w, h, hipMemcpyHostToDevice, s));
        }
        int32_t r;
        if ((r = svthip_pa_derive_planes_dev(ctx, pool, pd, (uint32_t)n_pic, params->enable_hme_level1_flag, params->enable_hme_level0_flag, s))) return r;
        svthip_me_cu_result* d_res = static_cast<svthip_me_cu_result*>(ctx->scratch[10]);
        const svthip_sb_origin* d_sb = static_cast<const svthip_sb_origin*>(ctx->scratch[9]);
        r = n_pu == 209 ? svthip_motion_estimate209_batch_dev(ctx, pool, &pd[0], &pd[1], ref1 ? &pd[2] : nullptr, 1, params, use_subpel_flag, cu8x8_mode,
                                                              d_sb, n_sb, d_res, nullptr, nullptr, s)
                        : svthip_motion_estimate_batch_dev(ctx, pool, &pd[0], &pd[1], ref1 ? &pd[2] : nullptr, 1, params, use_subpel_flag, cu8x8_mode, d_sb,
                                                           n_sb, d_res, nullptr, nullptr, s);
        if (r) return r;
        svthip_me_cu_result_ref* d_ref = static_cast<svthip_me_cu_result_ref*>(ctx->scratch[11]);
        if ((r = svthip_me_results_to_ref_layout_dev(ctx, d_res, n_sb * n_pu, d_ref, s))) return r;
        // rows of one allocation (the reference's EB_MALLOC'd me_results rows usually are not) leave in one copy
        bool contiguous = true;
        for (uint32_t i = 1; i < n_sb && contiguous; i++)
            contiguous = static_cast<uint8_t*>(me_results[i]) == static_cast<uint8_t*>(me_results[0]) + sizeof(svthip_me_cu_result_ref) * (size_t)n_pu * i;
        if (contiguous) {
            HIP_TRY(hipMemcpyAsync(me_results[0], d_ref, sizeof(svthip_me_cu_result_ref) * (size_t)n_pu * n_sb, hipMemcpyDeviceToHost, s));
        } else {
            for (uint32_t i = 0; i < n_sb; i++)
                HIP_TRY(hipMemcpyAsync(me_results[i], d_ref + (size_t)i * n_pu, sizeof(svthip_me_cu_result_ref) * n_pu, hipMemcpyDeviceToHost, s));
        }
        return SVTHIP_OK;
    };
    rc = queued();
    const hipError_t sync_e = hipStreamSynchronize(s);
    if (rc) return rc;
    HIP_TRY(sync_e);
    return SVTHIP_OK;
}

int32_t svthip_open_loop_intra_search_picture(svthip_ctx* ctx, const svthip_host_picture* cur, const svthip_ois_params* params,
                                              const void* const* me_results, uint32_t n_pu, uint32_t* cand, uint8_t* total)
{
    ENTER(ctx);
    if (!cur || !params || !cand || !total || !cur->buffer_y) return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer argument%s", "");
    const uint32_t w = cur->width, h = cur->height;
    if ((w & 7) || (h & 7) || !w || !h) return fail(SVTHIP_ERR_BAD_PARAMETER, "picture dimensions must be non-zero multiples of 8%s", "");
    if (cur->stride_y < w + cur->origin_x) return fail(SVTHIP_ERR_BAD_PARAMETER, "stride smaller than origin_x + width%s", "");
    const bool general = !params->slice_is_intra && !(params->temporal_layer_index == 0 && !params->input_resolution_4k) &&
                         !params->limit_ois_to_dc_mode_flag;
    if (general && (!me_results || (n_pu != 85 && n_pu != 209)))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "this picture's branch reads me_results (n_pu 85 or 209)%s (n_pu %d)", "", (int)n_pu);
    const uint32_t fs = w + 136, n_sb = ((w + 63) / 64) * ((h + 63) / 64);
    const size_t cand_bytes = (size_t)n_sb * 85 * 18 * 4, total_bytes = (size_t)n_sb * 85;
    if (general)
        for (uint32_t i = 0; i < n_sb; i++)  // checked before any work is queued
            if (!me_results[i]) return fail(SVTHIP_ERR_BAD_PARAMETER, "me_results[%s%d] is null", "", (int)i);
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 8, (size_t)fs * (h + 136) + 256))) return rc;
    if ((rc = ensure_scratch(ctx, 10, sizeof(svthip_me_cu_result) * (size_t)n_sb * 85))) return rc;
    if ((rc = ensure_scratch(ctx, 11, cand_bytes + total_bytes))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = scratch_on_stream(ctx, s))) return rc;
    if ((rc = ensure_sb_table(ctx, w, h, s))) return rc;
    svthip_me_cu_result* rows = nullptr;
    if (general) {
        // pinned staging for the ME distortions: the copy may still be reading it when this function queues the kernel
        if (hipHostMalloc(reinterpret_cast<void**>(&rows), sizeof(svthip_me_cu_result) * (size_t)n_sb * 85, hipHostMallocDefault) != hipSuccess)
            return fail(SVTHIP_ERR_INSUFFICIENT_RESOURCES, "out of pinned host memory%s", "");
        memset(rows, 0, sizeof(svthip_me_cu_result) * (size_t)n_sb * 85);
        for (uint32_t i = 0; i < n_sb; i++) {
            const svthip_me_cu_result_ref* r = static_cast<const svthip_me_cu_result_ref*>(me_results[i]);
            for (uint32_t cu = 0; cu < 85; cu++) rows[(size_t)i * 85 + cu].distortion[0] = r[cu].distortionDirection[0].distortion;
        }
    }
    // from here on copies from / to the caller's buffers are in flight: every exit synchronises the stream first
    auto queued = [&]() -> int32_t {
        uint8_t* pool = static_cast<uint8_t*>(ctx->scratch[8]);
        // only the picture interior is ever read by the search (samples outside the picture count as 128): any origin is accepted
        HIP_TRY(hipMemcpy2DAsync(pool + (size_t)68 * fs + 68, fs, cur->buffer_y + (size_t)cur->origin_y * cur->stride_y + cur->origin_x, cur->stride_y, w,
                                 h, hipMemcpyHostToDevice, s));
        if (general) HIP_TRY(hipMemcpyAsync(ctx->scratch[10], rows, sizeof(svthip_me_cu_result) * (size_t)n_sb * 85, hipMemcpyHostToDevice, s));
        svthip_pa_picture pd;
        memset(&pd, 0, sizeof(pd));
        pd.full_stride = fs;
        pd.width = (uint16_t)w;
        pd.height = (uint16_t)h;
        uint32_t* d_cand = static_cast<uint32_t*>(ctx->scratch[11]);
        uint8_t* d_total = static_cast<uint8_t*>(ctx->scratch[11]) + cand_bytes;
        int32_t r;
        if ((r = svthip_open_loop_intra_search_batch_dev(ctx, pool, &pd, 1, params, static_cast<const svthip_sb_origin*>(ctx->scratch[9]), n_sb,
                                                         general ? static_cast<const svthip_me_cu_result*>(ctx->scratch[10]) : nullptr, 85, d_cand, d_total, s)))
            return r;
        HIP_TRY(hipMemcpyAsync(cand, d_cand, cand_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(total, d_total, total_bytes, hipMemcpyDeviceToHost, s));
        return SVTHIP_OK;
    };
    rc = queued();
    const hipError_t sync_e = hipStreamSynchronize(s);
    if (rows) (void)hipHostFree(rows);
    if (rc) return rc;
    HIP_TRY(sync_e);
    return SVTHIP_OK;
}

int32_t svthip_encode_tu_batch(svthip_ctx* ctx, const void* src, const void* pred, void* recon, size_t plane_samples, int32_t planes_16bit,
                               const svthip_tu_desc* desc, uint32_t n_tu, uint32_t tx_width, uint32_t tx_height, const int16_t* qparams,
                               uint32_t n_qparam_rows, const int16_t* iscan, uint32_t n_iscan, size_t coeff_samples, int32_t* coeff,
                               int32_t* qcoeff, int32_t* dqcoeff, uint16_t* eob, uint64_t* three_quad_energy, uint64_t* distortion)
{
    ENTER(ctx);
    if (n_tu == 0) return SVTHIP_OK;
    if (!src || !pred || !recon || !desc || !qparams || !iscan || !qcoeff || !eob || !plane_samples || !coeff_samples || !n_qparam_rows || !n_iscan)
        return fail(SVTHIP_ERR_BAD_PARAMETER, "null pointer / empty buffer argument%s", "");
    if (!svthip::fwd_txfm2d_size_valid((int)tx_width, (int)tx_height))
        return fail(SVTHIP_ERR_BAD_PARAMETER, "unsupported transform size%s (width %d)", "", (int)tx_width);
    const size_t es = planes_16bit ? 2 : 1, pb = plane_samples * es;
    const uint32_t win = tx_width > 32 ? 32 : tx_width, hin = tx_height > 32 ? 32 : tx_height;
    for (uint32_t i = 0; i < n_tu; i++) {  // the kernel trusts its descriptors: check them against the buffers the caller declared
        const svthip_tu_desc& d = desc[i];
        const size_t last = (size_t)(tx_height - 1);
        if ((size_t)d.src_offset + last * d.src_stride + tx_width > plane_samples || (size_t)d.pred_offset + last * d.pred_stride + tx_width > plane_samples ||
            (size_t)d.recon_offset + last * d.recon_stride + tx_width > plane_samples)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: block outside the planes", "", (int)i);
        if ((d.coeff_offset & 3u) || (size_t)d.coeff_offset + (size_t)win * hin > coeff_samples)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: coefficient block outside the pools or not 4-aligned", "", (int)i);
        if ((d.iscan_offset & 3u) || (size_t)d.iscan_offset + (size_t)win * hin > n_iscan || d.qparam_index >= n_qparam_rows)
            return fail(SVTHIP_ERR_BAD_PARAMETER, "desc[%s%d]: scan table / quantiser row out of range", "", (int)i);
    }
    const bool in_place = recon == pred;
    const size_t cb = coeff_samples * sizeof(int32_t);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // slot 12: src | pred | recon planes; 13: desc | qparams | iscan; 14: coeff | qcoeff | dqcoeff; 15: eob | energy | dist
    int32_t rc;
    if ((rc = ensure_scratch(ctx, 12, al(pb) * 3))) return rc;
    if ((rc = ensure_scratch(ctx, 13, al(sizeof(svthip_tu_desc) * n_tu) + al(20 * (size_t)n_qparam_rows) + al(2 * (size_t)n_iscan)))) return rc;
    if ((rc = ensure_scratch(ctx, 14, al(cb) * 3))) return rc;
    if ((rc = ensure_scratch(ctx, 15, al(2 * (size_t)n_tu) + al(8 * (size_t)n_tu) + al(16 * (size_t)n_tu)))) return rc;
    hipStream_t s = ctx->stream;
    if ((rc = scratch_on_stream(ctx, s))) return rc;
    uint8_t* p12 = static_cast<uint8_t*>(ctx->scratch[12]);
    uint8_t *d_src = p12, *d_pred = p12 + al(pb), *d_recon = in_place ? d_pred : p12 + 2 * al(pb);
    uint8_t* p13 = static_cast<uint8_t*>(ctx->scratch[13]);
    svthip_tu_desc* d_desc = reinterpret_cast<svthip_tu_desc*>(p13);
    int16_t* d_qp = reinterpret_cast<int16_t*>(p13 + al(sizeof(svthip_tu_desc) * n_tu));
    int16_t* d_iscan = reinterpret_cast<int16_t*>(p13 + al(sizeof(svthip_tu_desc) * n_tu) + al(20 * (size_t)n_qparam_rows));
    uint8_t* p14 = static_cast<uint8_t*>(ctx->scratch[14]);
    int32_t *d_coeff = coeff ? reinterpret_cast<int32_t*>(p14) : nullptr, *d_q = reinterpret_cast<int32_t*>(p14 + al(cb)),
            *d_dq = dqcoeff ? reinterpret_cast<int32_t*>(p14 + 2 * al(cb)) : nullptr;
    uint8_t* p15 = static_cast<uint8_t*>(ctx->scratch[15]);
    uint16_t* d_eob = reinterpret_cast<uint16_t*>(p15);
    uint64_t* d_en = three_quad_energy ? reinterpret_cast<uint64_t*>(p15 + al(2 * (size_t)n_tu)) : nullptr;
    uint64_t* d_dist = distortion ? reinterpret_cast<uint64_t*>(p15 + al(2 * (size_t)n_tu) + al(8 * (size_t)n_tu)) : nullptr;
    // copies from / to the caller's buffers are in flight inside `queued`: every exit synchronises the stream first
    auto queued = [&]() -> int32_t {
        HIP_TRY(hipMemcpyAsync(d_src, src, pb, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_pred, pred, pb, hipMemcpyHostToDevice, s));
        if (!in_place) HIP_TRY(hipMemcpyAsync(d_recon, recon, pb, hipMemcpyHostToDevice, s));  // samples outside the TUs keep the caller's values
        HIP_TRY(hipMemcpyAsync(d_desc, desc, sizeof(svthip_tu_desc) * n_tu, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_qp, qparams, 20 * (size_t)n_qparam_rows, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_iscan, iscan, 2 * (size_t)n_iscan, hipMemcpyHostToDevice, s));
        // pool words no TU covers come back as the caller left them
        if (coeff) HIP_TRY(hipMemcpyAsync(d_coeff, coeff, cb, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_q, qcoeff, cb, hipMemcpyHostToDevice, s));
        if (dqcoeff) HIP_TRY(hipMemcpyAsync(d_dq, dqcoeff, cb, hipMemcpyHostToDevice, s));
        HIP_TRY(svthip::launch_encode_tu(d_src, d_pred, d_recon, planes_16bit ? 1 : 0, d_desc, n_tu, (int)tx_width, (int)tx_height, d_qp, d_iscan, d_coeff,
                                         d_q, d_dq, d_eob, d_en, d_dist, (uint32_t)ctx->opt[SVTHIP_OPT_TQ_MAX_WORKGROUPS], s));
        HIP_TRY(hipMemcpyAsync(recon, d_recon, pb, hipMemcpyDeviceToHost, s));
        if (coeff) HIP_TRY(hipMemcpyAsync(coeff, d_coeff, cb, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(qcoeff, d_q, cb, hipMemcpyDeviceToHost, s));
        if (dqcoeff) HIP_TRY(hipMemcpyAsync(dqcoeff, d_dq, cb, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(eob, d_eob, 2 * (size_t)n_tu, hipMemcpyDeviceToHost, s));
        if (three_quad_energy) HIP_TRY(hipMemcpyAsync(three_quad_energy, d_en, 8 * (size_t)n_tu, hipMemcpyDeviceToHost, s));
        if (distortion) HIP_TRY(hipMemcpyAsync(distortion, d_dist, 16 * (size_t)n_tu, hipMemcpyDeviceToHost, s));
        return SVTHIP_OK;
    };
    rc = queued();
    const hipError_t sync_e = hipStreamSynchronize(s);
    if (rc) return rc;
    HIP_TRY(sync_e);
    return SVTHIP_OK;
}

int32_t svthip_me_fullpel_search_time_dev(svthip_ctx* ctx, const uint8_t* d_src_plane, uint32_t src_stride,
                                          const uint8_t* d_ref_plane, uint32_t ref_stride,
                                          const svthip_fullpel_desc* d_desc, uint32_t n_sb, uint32_t max_search_area_width,
                                          uint32_t max_search_area_height, uint32_t* d_best_sad, uint32_t* d_best_mv,
                                          uint32_t iters, float* avg_ms)
{
    ENTER(ctx);
    if (!avg_ms || iters == 0) return fail(SVTHIP_ERR_BAD_PARAMETER, "bad timing arguments%s", "");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s = ctx->stream;
    int32_t rc = SVTHIP_OK;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    for (uint32_t i = 0; e == hipSuccess && i < iters && rc == SVTHIP_OK; i++)
        rc = launch_fullpel(ctx, d_src_plane, src_stride, d_ref_plane, ref_stride, d_desc, n_sb, max_search_area_width,
                            max_search_area_height, d_best_sad, d_best_mv, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    HIP_TRY(e);
    *avg_ms = ms / (float)iters;
    return SVTHIP_OK;
}

}  // extern "C"
