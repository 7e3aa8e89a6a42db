// svt-av1-1_amd/csrc/svthip_batcher.hip -- host-side gather / scatter for the transform / quantisation callers (SURVEY 8f-2;
// include/svtav1_hip.h "Batching layer").  Host C++ only: it builds svthip_tu_desc arrays grouped by transform size, launches the
// fused chain (svthip_encode_tu[16]_batch_dev) once per size present and scatters eob / energy / distortion back to the handles the
// caller got from _add.  No kernels of its own, no CPU arithmetic path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/svtav1_hip.h"

namespace {

const uint8_t kTxW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
const uint8_t kTxH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

struct Cand {
    uint8_t tx_size, tx_type;
    uint32_t slot;  // index inside its size group
    uint32_t coeff_offset;
};

}  // namespace

struct svthip_tu_batcher {
    svthip_ctx* ctx;
    uint32_t max_cand;
    size_t max_coeff, max_recon;
    // bound per picture
    const void *d_src, *d_pred;
    void* d_recon;
    int planes_16bit;
    const int16_t *d_qparams, *d_iscan;
    bool bound;
    // candidates since begin
    std::vector<Cand> cands;
    // candidates bucketed as they are added: [transform size][reconstruct into scratch?][transform type] -- a flush only concatenates.
    // Ordering a launch by transform type matters: a wave of the fused kernel owns 64 / min(W, H) consecutive TUs and its lanes branch on
    // their TU's 1-D transform kinds, so a wave of mixed types executes the DCT AND the ADST network in every one of its four passes;
    // sorted, almost every wave runs one network (TUs are independent: the order changes nothing but the time).  4-point dimensions stay
    // in the caller's order (bucket 0): their networks are a handful of instructions and neighbouring TUs share cache lines.
    std::vector<svthip_tu_desc> group[19][2][16];
    std::vector<uint32_t> group_handle[19][2][16];
    uint32_t group_count[19];  // candidates of a size (slot numbering)
    size_t coeff_used, recon_used;
    size_t flushed;  // candidates already launched
    // device pools
    int32_t *d_q, *d_dq;
    uint8_t* d_recon_scratch;
    svthip_tu_desc* d_desc;
    // per-candidate outputs, ONE device block and ONE pinned mirror so that a flush is one upload, the launches, one download:
    // [dist 16 B x max_cand][energy 8 B x max_cand][eob 2 B x max_cand]
    uint8_t *d_out, *h_out;
    uint64_t *d_dist, *d_energy, *h_dist, *h_energy;
    uint16_t *d_eob, *h_eob;
    svthip_tu_desc* h_desc;  // pinned: descriptors in launch order
    std::vector<uint32_t> launch_handle;
    std::vector<svthip_tu_result> results;
};

extern "C" {

int32_t svthip_tu_batcher_create(svthip_ctx* ctx, uint32_t max_candidates, uint32_t max_coeff_samples, svthip_tu_batcher** out)
{
    if (!ctx || !out || !max_candidates || !max_coeff_samples) return SVTHIP_ERR_BAD_PARAMETER;
    *out = nullptr;
    svthip_tu_batcher* b = new (std::nothrow) svthip_tu_batcher();
    if (!b) return SVTHIP_ERR_INSUFFICIENT_RESOURCES;
    b->ctx = ctx;
    b->max_cand = max_candidates;
    b->max_coeff = max_coeff_samples;
    b->max_recon = (size_t)max_coeff_samples * 4;  // a scratch tile is W x H samples; 64-point sizes keep 1/4 of them as coefficients
    b->bound = false;
    b->coeff_used = b->recon_used = b->flushed = 0;
    b->d_q = b->d_dq = nullptr;
    b->d_recon_scratch = nullptr;
    b->d_desc = b->h_desc = nullptr;
    b->d_out = b->h_out = nullptr;
    if (svthip_synchronize(ctx) != SVTHIP_OK) { delete b; return SVTHIP_ERR_DEVICE; }  // makes the context's device current
    const size_t nc = max_candidates, out_bytes = 26 * nc + 64;
    bool ok = hipMalloc(reinterpret_cast<void**>(&b->d_q), sizeof(int32_t) * b->max_coeff) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&b->d_dq), sizeof(int32_t) * b->max_coeff) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&b->d_recon_scratch), 2 * b->max_recon + 64) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&b->d_desc), sizeof(svthip_tu_desc) * nc) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&b->d_out), out_bytes) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&b->h_out), out_bytes) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&b->h_desc), sizeof(svthip_tu_desc) * nc) == hipSuccess;
    if (!ok) {
        svthip_tu_batcher_destroy(b);
        return SVTHIP_ERR_INSUFFICIENT_RESOURCES;
    }
    b->d_dist = reinterpret_cast<uint64_t*>(b->d_out);
    b->d_energy = reinterpret_cast<uint64_t*>(b->d_out + 16 * nc);
    b->d_eob = reinterpret_cast<uint16_t*>(b->d_out + 24 * nc);
    b->h_dist = reinterpret_cast<uint64_t*>(b->h_out);
    b->h_energy = reinterpret_cast<uint64_t*>(b->h_out + 16 * nc);
    b->h_eob = reinterpret_cast<uint16_t*>(b->h_out + 24 * nc);
    b->launch_handle.reserve(max_candidates);
    b->cands.reserve(max_candidates);
    b->results.reserve(max_candidates);
    *out = b;
    return SVTHIP_OK;
}

void svthip_tu_batcher_destroy(svthip_tu_batcher* b)
{
    if (!b) return;
    (void)svthip_synchronize(b->ctx);
    (void)hipFree(b->d_q);
    (void)hipFree(b->d_dq);
    (void)hipFree(b->d_recon_scratch);
    (void)hipFree(b->d_desc);
    (void)hipFree(b->d_out);
    if (b->h_out) (void)hipHostFree(b->h_out);
    if (b->h_desc) (void)hipHostFree(b->h_desc);
    delete b;
}

int32_t svthip_tu_batcher_begin(svthip_tu_batcher* b, const void* d_src, const void* d_pred, void* d_recon, int32_t planes_16bit,
                                const int16_t* d_qparams, const int16_t* d_iscan)
{
    if (!b || !d_src || !d_pred || !d_qparams || !d_iscan) return SVTHIP_ERR_BAD_PARAMETER;
    b->d_src = d_src;
    b->d_pred = d_pred;
    b->d_recon = d_recon;
    b->planes_16bit = planes_16bit ? 1 : 0;
    b->d_qparams = d_qparams;
    b->d_iscan = d_iscan;
    b->bound = true;
    b->cands.clear();
    b->results.clear();
    for (int i = 0; i < 19; i++) {
        b->group_count[i] = 0;
        for (int sc = 0; sc < 2; sc++)
            for (int t = 0; t < 16; t++) {
                b->group[i][sc][t].clear();
                b->group_handle[i][sc][t].clear();
            }
    }
    b->coeff_used = b->recon_used = b->flushed = 0;
    return SVTHIP_OK;
}

int32_t svthip_tu_batcher_add(svthip_tu_batcher* b, uint32_t tx_size, uint32_t tx_type, uint32_t src_offset, uint32_t src_stride, uint32_t pred_offset,
                              uint32_t pred_stride, uint32_t recon_offset, uint32_t recon_stride, uint32_t qparam_index, uint32_t iscan_offset,
                              uint32_t* out_handle)
{
    if (!b || !b->bound || !out_handle || tx_size >= 19 || tx_type >= 16) return SVTHIP_ERR_BAD_PARAMETER;
    if (b->cands.size() >= b->max_cand) return SVTHIP_ERR_INSUFFICIENT_RESOURCES;
    const uint32_t w = kTxW[tx_size], h = kTxH[tx_size];
    const uint32_t n = (w > 32 ? 32 : w) * (h > 32 ? 32 : h);
    if (b->coeff_used + n > b->max_coeff) return SVTHIP_ERR_INSUFFICIENT_RESOURCES;
    if (src_stride > 0xffff || pred_stride > 0xffff || qparam_index > 0xffff || (iscan_offset & 3u)) return SVTHIP_ERR_BAD_PARAMETER;
    svthip_tu_desc d;
    int scratch = 0;
    memset(&d, 0, sizeof(d));
    d.src_offset = src_offset;
    d.pred_offset = pred_offset;
    d.src_stride = (uint16_t)src_stride;
    d.pred_stride = (uint16_t)pred_stride;
    if (recon_offset == SVTHIP_TU_RECON_SCRATCH) {
        if (b->recon_used + (size_t)w * h > b->max_recon) return SVTHIP_ERR_INSUFFICIENT_RESOURCES;
        // the flush marks scratch tiles with the top bit of recon_stride's companion: a scratch tile has stride = width
        d.recon_offset = (uint32_t)b->recon_used;
        d.recon_stride = (uint16_t)w;
        scratch = 1;  // reconstruct into the scratch pool (a launch has ONE reconstruction plane)
        b->recon_used += (size_t)w * h;
    } else {
        if (!b->d_recon || recon_stride > 0xffff) return SVTHIP_ERR_BAD_PARAMETER;
        d.recon_offset = recon_offset;
        d.recon_stride = (uint16_t)recon_stride;
    }
    d.coeff_offset = (uint32_t)b->coeff_used;
    d.iscan_offset = iscan_offset;
    d.qparam_index = (uint16_t)qparam_index;
    d.tx_type = (uint8_t)tx_type;
    Cand c = {(uint8_t)tx_size, (uint8_t)tx_type, b->group_count[tx_size]++, d.coeff_offset};
    *out_handle = (uint32_t)b->cands.size();
    const int bucket = (w >= 8 && h >= 8) ? (int)tx_type : 0;
    b->group_handle[tx_size][scratch][bucket].push_back(*out_handle);
    b->group[tx_size][scratch][bucket].push_back(d);
    b->cands.push_back(c);
    b->coeff_used += n;
    return SVTHIP_OK;
}

int32_t svthip_tu_batcher_flush(svthip_tu_batcher* b)
{
    if (!b || !b->bound) return SVTHIP_ERR_BAD_PARAMETER;
    if (b->flushed == b->cands.size()) return SVTHIP_OK;
    if (b->flushed != 0) return SVTHIP_ERR_BAD_PARAMETER;  // one flush per _begin: handles index the whole batch
    hipStream_t s = static_cast<hipStream_t>(svthip_stream(b->ctx));
    b->results.resize(b->cands.size());
    // Two launches per size at most: candidates reconstructing into the caller's plane and candidates reconstructing into scratch
    // (a launch has ONE reconstruction plane).  All descriptors go to the pinned array in launch order and up in ONE copy; the launches
    // follow on the same stream with no host synchronisation in between; the three output arrays come back in ONE copy.
    struct Launch { int ts, scratch; uint32_t base, n; };
    Launch launches[38];
    int n_launch = 0;
    uint32_t base = 0;
    b->launch_handle.clear();
    for (int ts = 0; ts < 19; ts++)
        for (int scratch = 0; scratch < 2; scratch++) {
            const uint32_t first = base;
            for (int t = 0; t < 16; t++) {
                const std::vector<svthip_tu_desc>& g = b->group[ts][scratch][t];
                if (g.empty()) continue;
                memcpy(b->h_desc + base, g.data(), sizeof(svthip_tu_desc) * g.size());
                b->launch_handle.insert(b->launch_handle.end(), b->group_handle[ts][scratch][t].begin(), b->group_handle[ts][scratch][t].end());
                base += (uint32_t)g.size();
            }
            if (base != first) launches[n_launch++] = Launch{ts, scratch, first, base - first};
        }
    const size_t total = base;
    if (hipMemcpyAsync(b->d_desc, b->h_desc, sizeof(svthip_tu_desc) * total, hipMemcpyHostToDevice, s) != hipSuccess) return SVTHIP_ERR_DEVICE;
    for (int k = 0; k < n_launch; k++) {
        const Launch& L = launches[k];
        void* recon = L.scratch ? static_cast<void*>(b->d_recon_scratch) : b->d_recon;
        int32_t rc = b->planes_16bit
                         ? svthip_encode_tu16_batch_dev(b->ctx, static_cast<const uint16_t*>(b->d_src), static_cast<const uint16_t*>(b->d_pred),
                                                        static_cast<uint16_t*>(recon), b->d_desc + L.base, L.n, kTxW[L.ts], kTxH[L.ts], b->d_qparams,
                                                        b->d_iscan, nullptr, b->d_q, b->d_dq, b->d_eob + L.base, b->d_energy + L.base,
                                                        b->d_dist + 2 * (size_t)L.base, s)
                         : svthip_encode_tu_batch_dev(b->ctx, static_cast<const uint8_t*>(b->d_src), static_cast<const uint8_t*>(b->d_pred),
                                                      static_cast<uint8_t*>(recon), b->d_desc + L.base, L.n, kTxW[L.ts], kTxH[L.ts], b->d_qparams,
                                                      b->d_iscan, nullptr, b->d_q, b->d_dq, b->d_eob + L.base, b->d_energy + L.base,
                                                      b->d_dist + 2 * (size_t)L.base, s);
        if (rc) {
            (void)hipStreamSynchronize(s);  // earlier launches of this flush still read the pinned descriptors' device copy
            return rc;
        }
    }
    for (size_t i = 0; i < total; i++) b->results[b->launch_handle[i]].coeff_offset = (uint32_t)i;  // position in the output arrays for now
    if (hipMemcpyAsync(b->h_out, b->d_out, 26 * (size_t)b->max_cand, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return SVTHIP_ERR_DEVICE;
    for (size_t hnd = 0; hnd < b->cands.size(); hnd++) {
        svthip_tu_result& r = b->results[hnd];
        const uint32_t pos = r.coeff_offset;
        r.eob = b->h_eob[pos];
        r.three_quad_energy = b->h_energy[pos];
        r.distortion[0] = b->h_dist[2 * pos];
        r.distortion[1] = b->h_dist[2 * pos + 1];
        r.coeff_offset = b->cands[hnd].coeff_offset;
        r.tx_size = b->cands[hnd].tx_size;
        r.tx_type = b->cands[hnd].tx_type;
    }
    b->flushed = b->cands.size();
    return SVTHIP_OK;
}

int32_t svthip_tu_batcher_result(const svthip_tu_batcher* b, uint32_t handle, svthip_tu_result* out)
{
    if (!b || !out || handle >= b->flushed) return SVTHIP_ERR_BAD_PARAMETER;
    *out = b->results[handle];
    return SVTHIP_OK;
}

int32_t svthip_tu_batcher_read_coeffs(svthip_tu_batcher* b, uint32_t handle, int32_t* qcoeff, int32_t* dqcoeff)
{
    if (!b || handle >= b->flushed) return SVTHIP_ERR_BAD_PARAMETER;
    const Cand& c = b->cands[handle];
    const uint32_t w = kTxW[c.tx_size], h = kTxH[c.tx_size];
    const size_t bytes = sizeof(int32_t) * (w > 32 ? 32 : w) * (h > 32 ? 32 : h);
    hipStream_t s = static_cast<hipStream_t>(svthip_stream(b->ctx));
    if (qcoeff && hipMemcpyAsync(qcoeff, b->d_q + c.coeff_offset, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return SVTHIP_ERR_DEVICE;
    if (dqcoeff && hipMemcpyAsync(dqcoeff, b->d_dq + c.coeff_offset, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return SVTHIP_ERR_DEVICE;
    if (hipStreamSynchronize(s) != hipSuccess) return SVTHIP_ERR_DEVICE;
    return SVTHIP_OK;
}

int32_t svthip_tu_batcher_pools(const svthip_tu_batcher* b, const int32_t** d_qcoeff, const int32_t** d_dqcoeff, const void** d_recon_scratch)
{
    if (!b) return SVTHIP_ERR_BAD_PARAMETER;
    if (d_qcoeff) *d_qcoeff = b->d_q;
    if (d_dqcoeff) *d_dqcoeff = b->d_dq;
    if (d_recon_scratch) *d_recon_scratch = b->d_recon_scratch;
    return SVTHIP_OK;
}

}  // extern "C"
