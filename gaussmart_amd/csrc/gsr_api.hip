// Host side of the C ABI declared in include/gsr.h: argument validation (mirrors the Python
// operator's checks), buffer carving, kernel sequencing, error strings and the opt-in profiler.
#include "gsr_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
void gsr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gsr_last_error(void) { return g_err; }
extern "C" int32_t gsr_abi_version(void) { return GSR_ABI_VERSION; }

// ------------------------------------------------------------------------------- profiler
namespace {
struct KernelProf {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipStream_t, hipEvent_t>> open;   // one open bracket per stream (two models / two streams per device)
    double total_ms = 0.0;
    int launches = 0;
};
std::mutex g_prof_mu;
uint32_t g_prof_mask = 0;   // bit k enables timing of kernel id k
KernelProf g_prof[GSR_K_COUNT];
const char* const g_kernel_names[GSR_K_COUNT] = {
    "preprocess_fwd", "sort_hist", "sort_scatter", "scan", "emit_instances",
    "finalize_bins", "render_fwd", "render_bwd", "preprocess_bwd", "knn", "loss_fwd", "loss_bwd", "regularizer_fwd", "regularizer_bwd", "adam"};
constexpr size_t kMaxPending = 1 << 16;

void prof_drain(KernelProf& k) {
    for (auto& pr : k.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess &&
            hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            k.total_ms += ms;
            k.launches += 1;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    k.pending.clear();
}
}  // namespace

bool gsr_profile_on() { return g_prof_mask != 0; }
void gsr_profile_begin(int kernel, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!((g_prof_mask >> kernel) & 1u)) return;
    KernelProf& k = g_prof[kernel];
    if (k.pending.size() >= kMaxPending) return;
    for (auto& o : k.open) if (o.first == s) return;      // (a bracket of this kernel is already open on this stream)
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    k.open.emplace_back(s, e);
}
void gsr_profile_end(int kernel, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    KernelProf& k = g_prof[kernel];
    for (size_t i = 0; i < k.open.size(); ++i) {
        if (k.open[i].first != s) continue;
        hipEvent_t b = k.open[i].second, e;
        k.open.erase(k.open.begin() + (long)i);
        if (hipEventCreate(&e) != hipSuccess) { (void)hipEventDestroy(b); return; }
        (void)hipEventRecord(e, s);
        k.pending.emplace_back(b, e);
        return;
    }
}
extern "C" void gsr_profile_enable(int32_t mask) { g_prof_mask = (uint32_t)mask; }
extern "C" void gsr_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& k : g_prof) { prof_drain(k); k.total_ms = 0.0; k.launches = 0; }
}
extern "C" int32_t gsr_profile_read(const char* kernel, double* total_ms, int32_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < GSR_K_COUNT; ++i) {
        if (strcmp(kernel, g_kernel_names[i]) == 0) {
            prof_drain(g_prof[i]);
            if (total_ms) *total_ms = g_prof[i].total_ms;
            if (launches) *launches = g_prof[i].launches;
            return GSR_OK;
        }
    }
    gsr_set_error("unknown kernel name '%s'", kernel);
    return GSR_E_INVALID;
}

// ------------------------------------------------------------------------------- helpers
static int validate(const GsrView* v, const GsrGaussians* g) {
    if (!v || !g) { gsr_set_error("null view / gaussians"); return GSR_E_INVALID; }
    if (v->width <= 0 || v->height <= 0) { gsr_set_error("bad image size %dx%d", v->width, v->height); return GSR_E_INVALID; }
    if (g->count < 0) { gsr_set_error("negative Gaussian count"); return GSR_E_INVALID; }
    // 3 channels (the reference's RGB) or a wide per-pixel payload: precomputed features, C = 4, 8, ... 64
    if (v->channels != 3) {
        const bool wide_ok = v->channels >= 4 && v->channels <= GSR_MAX_CHANNELS && (v->channels % 4) == 0;
        if (!wide_ok || g->shs != nullptr || g->colors_precomp == nullptr) {
            gsr_set_error("channels=%d: 3 channels, or 4..%d (multiple of 4) with colors_precomp, are supported",
                          v->channels, GSR_MAX_CHANNELS);
            return GSR_E_UNSUPPORTED;
        }
        if ((reinterpret_cast<uintptr_t>(g->colors_precomp) & 15u) != 0) {
            gsr_set_error("wide colors_precomp must be 16-byte aligned");
            return GSR_E_INVALID;
        }
    }
    if ((v->flags & (uint32_t)GSR_FLAG_COLOR_ONLY) && (v->channels != 3 || (v->flags & (uint32_t)GSR_FLAG_FORWARD_ONLY))) {
        gsr_set_error("GSR_FLAG_COLOR_ONLY is for the 3-channel training forward (not with GSR_FLAG_FORWARD_ONLY)");
        return GSR_E_INVALID;
    }
    const bool color_cached = (v->flags & (uint32_t)GSR_FLAG_COLOR_CACHED) != 0;
    if (color_cached) {   // SH coefficients AND their cached colour for this view (gsr_adam_sh_factored_next)
        if (!g->shs || !g->colors_precomp || v->channels != 3 || v->sh_coeffs > 16) {
            gsr_set_error("GSR_FLAG_COLOR_CACHED needs shs (<= 16 coefficients), the colour cache in colors_precomp and 3 channels");
            return GSR_E_INVALID;
        }
    } else if ((g->shs == nullptr) == (g->colors_precomp == nullptr)) {
        gsr_set_error("Please provide excatly one of either SHs or precomputed colors!");
        return GSR_E_INVALID;
    }
    const bool sr = g->scales != nullptr && g->rotations != nullptr;
    const bool any_sr = g->scales != nullptr || g->rotations != nullptr;
    if ((!sr && g->transmat_precomp == nullptr) || (any_sr && g->transmat_precomp != nullptr)) {
        gsr_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return GSR_E_INVALID;
    }
    if (g->shs) {
        if (v->sh_degree < 0 || v->sh_degree > 3) { gsr_set_error("sh_degree %d not in 0..3", v->sh_degree); return GSR_E_UNSUPPORTED; }
        if (v->sh_coeffs < (v->sh_degree + 1) * (v->sh_degree + 1)) {
            gsr_set_error("sh_coeffs=%d too small for degree %d", v->sh_coeffs, v->sh_degree);
            return GSR_E_INVALID;
        }
    }
    if (g->shs_rest && (!g->shs || v->sh_coeffs < 2)) { gsr_set_error("shs_rest given without shs (dc) or with fewer than 2 coefficients"); return GSR_E_INVALID; }
    if ((v->flags & (uint32_t)GSR_FLAG_RAW_PARAMS) && g->transmat_precomp) { gsr_set_error("raw parameters and transmat_precomp are mutually exclusive"); return GSR_E_INVALID; }
    if (g->count > 0 && (!g->means3D || !g->opacities)) { gsr_set_error("means3D / opacities missing"); return GSR_E_INVALID; }
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos) { gsr_set_error("view pointers missing"); return GSR_E_INVALID; }
    return GSR_OK;
}

// Pinned host memory per calling thread, for the one device -> host read of gsr_forward (the partial sums of the
// instance count, one 64-bit word per workgroup of the geometry pass): allocated on first use, grown when a call needs
// more, kept otherwise (a forward is synchronous with respect to this read, so one buffer per thread suffices).
static unsigned long long* pinned_words(size_t words) {
    thread_local unsigned long long* buf = nullptr;
    thread_local size_t cap = 0;
    if (words > cap) {
        if (buf) (void)hipHostFree(buf);
        buf = nullptr; cap = 0;
        size_t want = words < 256 ? 256 : words + words / 4;
        void* p = nullptr;
        if (hipHostMalloc(&p, want * 8, hipHostMallocDefault) != hipSuccess) return nullptr;
        buf = static_cast<unsigned long long*>(p); cap = want;
    }
    return buf;
}
static uint32_t* pinned_counter() { return reinterpret_cast<uint32_t*>(pinned_words(1)); }

// above this many bytes of worst-case gradient rows the backward reads the exact row count back (one host wait)
static size_t exact_rows_threshold() {
    const char* e = getenv("GSR_EXACT_ROWS_BYTES");      // read per call: tests flip it inside one process
    return e && *e ? (size_t)strtoull(e, nullptr, 10) : (size_t(8) << 30);
}

// two timing-less events per calling thread -- [0]: end of the geometry pass (the host's wait for the instance count and the
// colour stream's wait), [1]: end of a colour pass on a second stream -- created on first use and kept; re-recording an
// event does not disturb waits that were enqueued on its previous record
static hipEvent_t* color_events() {
    thread_local hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i)
        if (!ev[i] && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { ev[i] = nullptr; return nullptr; }
    return ev;
}

// up to this many workgroups of the geometry pass the host adds their count partials itself (8 M Gaussians);
// GSR_COUNT_FUSED_MAX_BLOCKS (read per call) lets a test take the other branch at a small size
static int fused_count_max_blocks() {
    const char* e = getenv("GSR_COUNT_FUSED_MAX_BLOCKS");
    return e && *e ? atoi(e) : 32768;
}

static inline int bits_for(uint32_t n_values) {   // bits needed to represent 0..n_values-1
    int b = 0;
    while ((1ull << b) < n_values) ++b;
    return b;
}

template <typename T>
static inline T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
template <typename T>
static inline const T* at(const void* base, size_t off) { return reinterpret_cast<const T*>(static_cast<const char*>(base) + off); }

// ------------------------------------------------------------------------------- forward
extern "C" int32_t gsr_forward(const GsrView* view, const GsrGaussians* g, GsrForwardOut* out,
                               gsr_alloc_fn alloc, void* ctx, gsr_stream_t stream_) {
    int rc = validate(view, g);
    if (rc != GSR_OK) return rc;
    if (!out || !alloc || !out->out_color || !out->out_allmap || (g->count > 0 && !out->radii)) {
        gsr_set_error("forward outputs / allocator missing");
        return GSR_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int N = g->count, W = view->width, H = view->height;
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    const int n_tiles = gx * gy;
    const int64_t P = (int64_t)W * H;

    const bool fwd_only = (view->flags & (uint32_t)GSR_FLAG_FORWARD_ONLY) != 0 && view->channels == 3;
    // (the colour pass leaves d(rgb)/d(dir) in GEOM only for a backward that has no colour cache to take it from)
    const bool geom_jac = !fwd_only && (view->flags & (uint32_t)GSR_FLAG_COLOR_CACHED) == 0;
    const GsrGeomLayout GL(N, geom_jac);
    const GsrImageLayout IL(P);
    void* geom = alloc(ctx, GSR_BUF_GEOM, GL.total);
    void* image = fwd_only ? nullptr : alloc(ctx, GSR_BUF_IMAGE, IL.total);
    if (!geom || (!image && !fwd_only)) { gsr_set_error("allocator returned NULL (geom/image)"); return GSR_E_ALLOC; }
    out->geom = geom; out->image = image; out->binning = nullptr; out->num_rendered = 0;

    float* splat = at<float>(geom, GL.splat);
    uint32_t* clamped = at<uint32_t>(geom, GL.clamped);
    uint32_t* tiles_touched = at<uint32_t>(geom, GL.tiles_touched);
    uint2* tile_rect = at<uint2>(geom, GL.tile_rect);
    uint32_t* depth_key = at<uint32_t>(geom, GL.depth_key);
    float* color_jac = geom_jac ? at<float>(geom, GL.color_jac) : nullptr;   // (only a backward without a colour cache reads it)

    // N-sized scratch: depth-sort double buffers, sort + scan workspaces
    const size_t nb = gsr_align(size_t(N > 0 ? N : 1) * 4);
    const size_t sort_ws_n = gsr_sort_ws_bytes(N);
    const size_t scan_ws_n = gsr_scan_workspace_bytes(N);
    const size_t scratch_bytes = 4 * nb + sort_ws_n + scan_ws_n;
    char* scratch = static_cast<char*>(alloc(ctx, GSR_BUF_SCRATCH, scratch_bytes));
    if (!scratch) { gsr_set_error("allocator returned NULL (scratch)"); return GSR_E_ALLOC; }
    uint32_t* keys_sorted = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(scratch + nb);
    uint32_t* vals_tmp = reinterpret_cast<uint32_t*>(scratch + 2 * nb);
    uint32_t* order = at<uint32_t>(geom, GL.order);     // kept for the backward's row reduction
    uint32_t* offs = at<uint32_t>(geom, GL.offs);
    // after the depth sort its double buffers are free: they hold the rank-ordered tile rects and counts
    uint2* rank_rect = reinterpret_cast<uint2*>(scratch + 2 * nb);   // 2 nb: vals_tmp + the 4th block
    void* sort_ws = scratch + 4 * nb;
    void* scan_ws = static_cast<char*>(sort_ws) + sort_ws_n;

    const bool defer_color = (view->flags & (uint32_t)GSR_FLAG_DEFER_COLOR) != 0;
    // colour pass on a second stream (GSR_BUF_COLOR_STREAM): whatever happens below, the call's stream ends up ordered
    // after it, so the caller's stream-ordered buffer reuse stays safe
    struct ColorJoin {
        hipStream_t main; hipEvent_t done; bool armed;
        ~ColorJoin() { if (armed) (void)hipStreamWaitEvent(main, done, 0); }
    } color_join{s, nullptr, false};
    uint32_t D = 0;
    if (N > 0) {
        // The instance count D = sum of tiles_touched sizes the next buffers, so the host has to see it.  It does not
        // depend on the depth order: the geometry pass leaves one partial sum per workgroup straight in pinned host
        // memory (device-mapped: no copy launch in the stream) and the depth sort, the rank gather and the scan are enqueued
        // BEHIND the event that marks its end -- the host round trip (wait, add the partials up, allocate, launch) is over
        // long before they are, so the stream never drains.  (Rounds 1-2 read the scan's total back with nothing but the
        // colour pass behind it: with the colour pass on another stream the main stream idled for the round trip.)  The
        // same launch clears the depth sort's supergroup table.  Above 8 M Gaussians a small counting kernel produces
        // <= 256 partials instead (the host would add up tens of thousands).
        const int pre_blocks = gsr_preprocess_fwd_blocks(N);
        const bool fused_count = pre_blocks <= fused_count_max_blocks();
        int n_partial = fused_count ? pre_blocks : GSR_COUNT_PARTIALS;
        unsigned long long* d_host = pinned_words((size_t)n_partial);
        if (!d_host) { gsr_set_error("hipHostMalloc failed (instance-count read-back buffer)"); return GSR_E_HIP; }
        void* d_partial = nullptr;
        GSR_HIP_CHECK(hipHostGetDevicePointer(&d_partial, d_host, 0));
        uint32_t* zero_n = nullptr; size_t zero_n_words = 0;
        gsr_sort_zero_region(sort_ws, N, &zero_n, &zero_n_words);
        rc = gsr_launch_preprocess_fwd(*view, *g, splat, clamped, tiles_touched, tile_rect, depth_key, out->radii,
                                       fused_count ? static_cast<unsigned long long*>(d_partial) : nullptr,
                                       fused_count ? zero_n : nullptr, fused_count ? zero_n_words : 0, s);
        if (rc != GSR_OK) return rc;
        if (!fused_count) {
            rc = gsr_launch_count_partials(tiles_touched, N, static_cast<unsigned long long*>(d_partial), &n_partial,
                                           zero_n, zero_n_words, s);
            if (rc != GSR_OK) return rc;
        }
        // ONE event behind the geometry pass serves both its readers -- the host (instance count) and, with
        // GSR_FLAG_DEFER_COLOR, the stream the colour pass runs on: every record on the call's stream is a barrier packet
        // worth ~6 us of its timeline (two back-to-back records showed as a 14 us hole behind K1 in the kernel trace)
        hipEvent_t* evs = color_events();
        if (!evs) { gsr_set_error("hipEventCreate failed (forward events)"); return GSR_E_HIP; }
        hipEvent_t ev = evs[0];
        hipStream_t cs = defer_color ? static_cast<hipStream_t>(alloc(ctx, GSR_BUF_COLOR_STREAM, 0)) : nullptr;
        hipError_t e2 = hipEventRecord(ev, s);
        if (e2 == hipSuccess && cs && cs != s) {
            GSR_HIP_CHECK(hipStreamWaitEvent(cs, ev, 0));
            rc = gsr_launch_preprocess_color(*view, *g, splat, clamped, out->radii, color_jac, cs);
            if (rc != GSR_OK) return rc;
            GSR_HIP_CHECK(hipEventRecord(evs[1], cs));
            color_join.done = evs[1]; color_join.armed = true;
        }
        if (e2 == hipSuccess && !defer_color)
            rc = gsr_launch_preprocess_color(*view, *g, splat, clamped, out->radii, color_jac, s);
        // depth order of the Gaussians (stable; culled ones carry key 0xFFFFFFFF and no tiles)
        if (e2 == hipSuccess && rc == GSR_OK)
            rc = gsr_radix_sort_pairs(depth_key, nullptr, keys_sorted, order, keys_tmp, vals_tmp, N, 0, 32, sort_ws, s,
                                      nullptr, nullptr, nullptr, /*table_zeroed=*/true);
        if (e2 == hipSuccess && rc == GSR_OK)
            rc = gsr_launch_rank_gather_scan(N, order, tile_rect, rank_rect, scan_ws, s);
        hipError_t e3 = e2 == hipSuccess ? hipEventSynchronize(ev) : e2;   // the one host wait of the forward
        GSR_HIP_CHECK(e3);
        if (rc != GSR_OK) return rc;
        uint64_t total = 0;
        for (int i = 0; i < n_partial; ++i) total += d_host[i];
        if (total > 0x7FFFFFF0ull) { gsr_set_error("instance count %llu overflows", (unsigned long long)total); return GSR_E_UNSUPPORTED; }
        D = (uint32_t)total;
    }
    if (D > 0x7FFFFFF0u) { gsr_set_error("instance count %u overflows", D); return GSR_E_UNSUPPORTED; }
    out->num_rendered = (int32_t)D;

    const GsrBinLayout BL(D, n_tiles, !fwd_only);
    void* binning = alloc(ctx, GSR_BUF_BINNING, BL.total);
    if (!binning) { gsr_set_error("allocator returned NULL (binning)"); return GSR_E_ALLOC; }
    out->binning = binning;
    uint32_t* point_list = at<uint32_t>(binning, BL.point_list);
    uint32_t* inst_row = at<uint32_t>(binning, BL.inst_row);
    uint32_t* ranges = at<uint32_t>(binning, BL.ranges);

    if (D > 0) {
        const size_t db = gsr_align(size_t(D) * 4);
        const size_t sort_ws_d = gsr_sort_ws_bytes(D);
        char* sc2 = static_cast<char*>(alloc(ctx, GSR_BUF_SCRATCH2, 8 * db + sort_ws_d));
        if (!sc2) { gsr_set_error("allocator returned NULL (scratch2)"); return GSR_E_ALLOC; }
        uint32_t* tile_keys = reinterpret_cast<uint32_t*>(sc2);
        uint32_t* emit_gid = reinterpret_cast<uint32_t*>(sc2 + 2 * db);
        uint32_t* tile_sorted = reinterpret_cast<uint32_t*>(sc2 + 3 * db);
        uint32_t* perm = inst_row;   // the sorted values ARE the emission indices of the list entries: sort straight into place
        uint32_t* tk_tmp = reinterpret_cast<uint32_t*>(sc2 + 5 * db);
        uint32_t* tv_tmp = reinterpret_cast<uint32_t*>(sc2 + 6 * db);
        uint32_t* tg_tmp = reinterpret_cast<uint32_t*>(sc2 + 7 * db);
        void* sort_ws2 = sc2 + 8 * db;
        uint32_t* zero_d = nullptr; size_t zero_d_words = 0;
        gsr_sort_zero_region(sort_ws2, D, &zero_d, &zero_d_words);
        rc = gsr_launch_emit(N, gx, gy, order, scan_ws, offs, rank_rect, tile_keys, emit_gid, ranges, size_t(n_tiles) * 2,
                             zero_d, zero_d_words, s);
        if (rc != GSR_OK) return rc;
        // values = emission indices 0..D-1: the sort generates them itself (vals_in = NULL); the Gaussian ids travel as a
        // second value array straight into point_list (no gather through the emission index afterwards)
        rc = gsr_radix_sort_pairs(tile_keys, nullptr, tile_sorted, perm, tk_tmp, tv_tmp, D, 0,
                                  bits_for((uint32_t)n_tiles), sort_ws2, s, emit_gid, point_list, tg_tmp, /*table_zeroed=*/true);
        if (rc != GSR_OK) return rc;
        rc = gsr_launch_finalize_bins((int)D, n_tiles, tile_sorted, ranges,
                                      fwd_only ? nullptr : at<uint8_t>(binning, BL.slot_cnt), s);
        if (rc != GSR_OK) return rc;
    } else {
        GSR_HIP_CHECK(hipMemsetAsync(ranges, 0, size_t(n_tiles) * 8, s));
        GSR_HIP_CHECK(hipMemsetAsync(offs, 0, size_t(N + 1) * 4, s));     // (the emission, which writes them, did not run)
    }

    if (color_join.armed) {       // the colour pass has been running on the second stream: join it here
        color_join.armed = false;
        GSR_HIP_CHECK(hipStreamWaitEvent(s, color_join.done, 0));
    } else if (defer_color && N > 0) {   // binning did not need the colours: announce the pass, then enqueue it
        if (!alloc(ctx, GSR_BUF_SYNC_SH, 0)) { gsr_set_error("allocator refused GSR_BUF_SYNC_SH"); return GSR_E_ALLOC; }
        rc = gsr_launch_preprocess_color(*view, *g, splat, clamped, out->radii, color_jac, s);
        if (rc != GSR_OK) return rc;
    }
    return gsr_launch_render_fwd(*view, ranges, splat, fwd_only ? nullptr : at<float>(image, IL.final_T),
                                 fwd_only ? nullptr : at<uint32_t>(image, IL.n_contrib), out->out_color, out->out_allmap,
                                 at<uint8_t>(binning, BL.touch), at<uint32_t>(binning, BL.covered), inst_row,
                                 fwd_only ? nullptr : at<uint8_t>(binning, BL.slot_cnt),
                                 view->channels == 3 ? nullptr : g->colors_precomp,
                                 point_list, s);
}

// ------------------------------------------------------------------------------- backward
extern "C" int32_t gsr_backward(const GsrView* view, const GsrGaussians* g, int32_t num_rendered,
                                const int32_t* radii, const void* geom, const void* binning,
                                const void* image, const float* dL_dcolor, const float* dL_dallmap,
                                GsrGrads* grads, gsr_alloc_fn alloc, void* ctx, gsr_stream_t stream_) {
    return gsr_backward_with_job(view, g, num_rendered, radii, geom, binning, image, dL_dcolor, dL_dallmap, grads, nullptr,
                                 alloc, ctx, stream_);
}

extern "C" int32_t gsr_row_scan_job(const void* binning, int32_t num_rendered, int32_t width, int32_t height,
                                    GsrRowScanJob* job) {
    if (!binning || !job || num_rendered < 0 || width <= 0 || height <= 0) { gsr_set_error("bad row_scan_job arguments"); return GSR_E_INVALID; }
    const int gx = (width + GSR_TILE - 1) / GSR_TILE, gy = (height + GSR_TILE - 1) / GSR_TILE;
    const GsrBinLayout BL(num_rendered, gx * gy);
    job->counts = at<uint8_t>(binning, BL.slot_cnt);
    job->slot_off = const_cast<uint32_t*>(at<uint32_t>(binning, BL.slot_off));
    job->workspace = const_cast<char*>(at<char>(binning, BL.row_scan_ws));
    job->n = num_rendered;
    job->stage = 0;
    return GSR_OK;
}

extern "C" int32_t gsr_backward_with_job(const GsrView* view, const GsrGaussians* g, int32_t num_rendered,
                                         const int32_t* radii, const void* geom, const void* binning,
                                         const void* image, const float* dL_dcolor, const float* dL_dallmap,
                                         GsrGrads* grads, const GsrRowScanJob* job, gsr_alloc_fn alloc, void* ctx,
                                         gsr_stream_t stream_) {
    int rc = validate(view, g);
    if (rc != GSR_OK) return rc;
    // factored SH gradient: dL_dcolors [N,3] receives the masked colour gradient, the SH arrays are not written
    const bool factored = (view->flags & (uint32_t)GSR_FLAG_FACTORED_SH_GRAD) != 0 && g->shs != nullptr;
    if (!grads || !alloc || !geom || !binning || !image || !dL_dcolor || !dL_dallmap || num_rendered < 0 ||
        (g->count > 0 && (!radii || !grads->dL_dmeans3D || !grads->dL_dmeans2D || !grads->dL_dopacity)) ||
        (g->shs && !factored && !grads->dL_dshs) || (g->shs_rest && !factored && !grads->dL_dshs_rest) ||
        (factored && !grads->dL_dcolors) ||
        (g->colors_precomp && !(view->flags & (uint32_t)GSR_FLAG_COLOR_CACHED) && !grads->dL_dcolors) ||
        (g->scales && (!grads->dL_dscales || !grads->dL_drotations)) ||
        (g->transmat_precomp && !grads->dL_dtransmat)) {
        gsr_set_error("backward inputs / gradient outputs missing");
        return GSR_E_INVALID;
    }
    if (view->flags & (uint32_t)GSR_FLAG_FORWARD_ONLY) {
        gsr_set_error("gsr_backward called for a GSR_FLAG_FORWARD_ONLY forward: nothing was kept for it");
        return GSR_E_INVALID;
    }
    if ((view->flags & (uint32_t)GSR_FLAG_COLOR_ONLY) && !(view->flags & (uint32_t)GSR_FLAG_NO_SURFACE_GRAD)) {
        gsr_set_error("gsr_backward of a GSR_FLAG_COLOR_ONLY forward needs GSR_FLAG_NO_SURFACE_GRAD: the forward kept no "
                      "state for gradients of allmap");
        return GSR_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int N = g->count, W = view->width, H = view->height;
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    const GsrGeomLayout GL(N, (view->flags & (uint32_t)GSR_FLAG_COLOR_CACHED) == 0);   // as the forward laid it out
    const GsrBinLayout BL(num_rendered, gx * gy);
    const GsrImageLayout IL((int64_t)W * H);

    // Gradient rows: one 72-byte row (64 + 8 in two arrays) per (instance, 4x4 pixel block the forward blended it into), DENSE in
    // (emission index, quad, block) order.  Their number R <= 16 D is only known on the device (scan below).
    //   * normally the row buffer is sized for the bound and only its first R rows are ever touched: no host wait;
    //   * when the bound is huge (> GSR_EXACT_ROWS_BYTES, default 8 GiB: tens of millions of instances) R is read
    //     back first -- one host wait in the backward -- and the buffer is sized exactly (R is ~2.6 D in practice).
    const size_t n_inst = size_t(num_rendered > 0 ? num_rendered : 1);
    if (n_inst * GSR_SUBROWS > 0xFFFFFFFFull) {   // slot indices are u32
        gsr_set_error("%zu instances: gradient-row slots would overflow 32 bits", n_inst);
        return GSR_E_UNSUPPORTED;
    }
    const bool wide = view->channels != 3;
    const size_t row_bytes_each = size_t(GSR_GROW_MAIN + (wide ? 0 : GSR_GROW_XY)) * 4 + (wide ? size_t(view->channels) * 4 : 0);
    const size_t sums_bytes = gsr_align(size_t(N > 0 ? N : 1) * GSR_GROW_FLOATS * 4);
    const size_t begin_bytes = gsr_align(size_t(N + 1) * 4);        // first gradient row of every depth rank
    char* scratch = static_cast<char*>(alloc(ctx, GSR_BUF_SCRATCH, sums_bytes + begin_bytes));
    if (!scratch) { gsr_set_error("allocator returned NULL (backward scratch)"); return GSR_E_ALLOC; }
    // (row counts, their scan and its workspace live in BINNING: the forward leaves the counts, and the scan may already
    // have run as a side job of kernels between the forward and this call)
    uint8_t* slot_cnt = const_cast<uint8_t*>(at<uint8_t>(binning, BL.slot_cnt));
    uint32_t* slot_off = const_cast<uint32_t*>(at<uint32_t>(binning, BL.slot_off));
    void* scan_ws = const_cast<char*>(at<char>(binning, BL.row_scan_ws));
    float* row_sums = reinterpret_cast<float*>(scratch);
    uint32_t* row_begin = reinterpret_cast<uint32_t*>(scratch + sums_bytes);
    const bool scanned = job && job->stage == 2 && job->slot_off == slot_off && job->counts == slot_cnt &&
                         job->n == (int64_t)num_rendered;

    size_t n_rows = n_inst * GSR_SUBROWS;       // the bound
    const uint32_t* touch = at<uint32_t>(binning, BL.touch);
    if (num_rendered > 0) {
        // (the row counts per instance were left in BINNING by the forward: render_fwd.hip, last wave of every tile)
        if (!scanned) {
            rc = gsr_exclusive_scan_u8(slot_cnt, slot_off, num_rendered, scan_ws, s);
            if (rc != GSR_OK) return rc;
        }
        if (n_rows * row_bytes_each > exact_rows_threshold()) {
            uint32_t* r_host = pinned_counter();
            if (!r_host) { gsr_set_error("hipHostMalloc failed (row-count read-back buffer)"); return GSR_E_HIP; }
            GSR_HIP_CHECK(hipMemcpyAsync(r_host, slot_off + num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            GSR_HIP_CHECK(hipStreamSynchronize(s));
            n_rows = size_t(*r_host > 0 ? *r_host : 1);
        }
    } else {
        GSR_HIP_CHECK(hipMemsetAsync(slot_off, 0, 8, s));
        GSR_HIP_CHECK(hipMemsetAsync(row_begin, 0, size_t(N + 1) * 4, s));   // no instance, no rows
    }
    const size_t main_bytes = gsr_align(n_rows * GSR_GROW_MAIN * 4);
    const size_t rows_bytes = main_bytes + (wide ? 0 : gsr_align(n_rows * GSR_GROW_XY * 4));   // (wide: dxy rides in the row)
    const size_t feat_bytes = wide ? gsr_align(n_rows * size_t(view->channels) * 4) : 0;   // feature rows
    char* rows_mem = static_cast<char*>(alloc(ctx, GSR_BUF_SCRATCH2, rows_bytes + feat_bytes));
    if (!rows_mem) { gsr_set_error("allocator returned NULL (gradient rows)"); return GSR_E_ALLOC; }
    float* grad_rows = reinterpret_cast<float*>(rows_mem);
    float* grad_xy = wide ? nullptr : reinterpret_cast<float*>(rows_mem + main_bytes);
    float* feat_rows = wide ? reinterpret_cast<float*>(rows_mem + rows_bytes) : nullptr;

    if (num_rendered > 0) {
        rc = gsr_launch_render_bwd(*view, at<uint32_t>(binning, BL.ranges), at<uint32_t>(binning, BL.covered),
                                   at<uint32_t>(binning, BL.inst_row),
                                   at<float>(geom, GL.splat), touch, slot_off, at<float>(image, IL.final_T),
                                   at<uint32_t>(image, IL.n_contrib), dL_dcolor, dL_dallmap, grad_rows, grad_xy,
                                   N, at<uint32_t>(geom, GL.offs), row_begin,
                                   wide ? g->colors_precomp : nullptr, at<uint32_t>(binning, BL.point_list), feat_rows, n_rows, s);
        if (rc != GSR_OK) return rc;
    }
    GsrGrads o = *grads;
    if (!g->shs) o.dL_dshs = nullptr;
    if (!g->shs_rest) o.dL_dshs_rest = nullptr;
    if ((!g->colors_precomp || (view->flags & (uint32_t)GSR_FLAG_COLOR_CACHED)) && !factored) o.dL_dcolors = nullptr;
    if (factored) { o.dL_dshs = nullptr; o.dL_dshs_rest = nullptr; }
    if (!g->scales) { o.dL_dscales = nullptr; o.dL_drotations = nullptr; }
    if (!g->transmat_precomp) o.dL_dtransmat = nullptr;
    rc = gsr_launch_reduce_rows(N, at<uint32_t>(geom, GL.order), row_begin, grad_rows, grad_xy, row_sums, s);
    if (rc != GSR_OK) return rc;
    if (wide) {   // dL_dcolors [N,C] comes from the feature sub-rows, not from the 3 RGB columns of the row sums
        if (num_rendered > 0) {
            rc = gsr_launch_reduce_feat_rows(N, view->channels, at<uint32_t>(geom, GL.order), row_begin, feat_rows,
                                             o.dL_dcolors, s);
            if (rc != GSR_OK) return rc;
        } else if (N > 0) {
            GSR_HIP_CHECK(hipMemsetAsync(o.dL_dcolors, 0, size_t(N) * view->channels * 4, s));
        }
        o.dL_dcolors = nullptr;
    }
    const bool color_cached = (view->flags & (uint32_t)GSR_FLAG_COLOR_CACHED) != 0;
    return gsr_launch_preprocess_bwd(*view, *g, radii, at<float>(geom, GL.splat), at<uint32_t>(geom, GL.clamped),
                                     row_sums, color_cached ? g->colors_precomp + 4 * (size_t)N : at<float>(geom, GL.color_jac), o, s);
}

// ------------------------------------------------------------------------------- introspection
extern "C" int32_t gsr_buffer_field(int32_t which, const char* name, int32_t N, int32_t D, int32_t W,
                                    int32_t H, size_t* offset, size_t* bytes) {
    if (!name || !offset || !bytes) { gsr_set_error("null argument"); return GSR_E_INVALID; }
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    const int64_t P = (int64_t)W * H;
    if (which == GSR_BUF_GEOM) {
        const GsrGeomLayout L(N);
        if (!strcmp(name, "splat")) { *offset = L.splat; *bytes = size_t(N) * GSR_SPLAT_FLOATS * 4; return GSR_OK; }
        if (!strcmp(name, "clamped")) { *offset = L.clamped; *bytes = size_t(N) * 4; return GSR_OK; }
        if (!strcmp(name, "tiles_touched")) { *offset = L.tiles_touched; *bytes = size_t(N) * 4; return GSR_OK; }
        if (!strcmp(name, "depth_key")) { *offset = L.depth_key; *bytes = size_t(N) * 4; return GSR_OK; }
        if (!strcmp(name, "order")) { *offset = L.order; *bytes = size_t(N) * 4; return GSR_OK; }
        if (!strcmp(name, "offs")) { *offset = L.offs; *bytes = size_t(N + 1) * 4; return GSR_OK; }
    } else if (which == GSR_BUF_BINNING) {
        const GsrBinLayout L(D, gx * gy);
        if (!strcmp(name, "point_list")) { *offset = L.point_list; *bytes = size_t(D) * 4; return GSR_OK; }
        if (!strcmp(name, "inst_row")) { *offset = L.inst_row; *bytes = size_t(D) * 4; return GSR_OK; }
        if (!strcmp(name, "ranges")) { *offset = L.ranges; *bytes = size_t(gx) * gy * 8; return GSR_OK; }
        if (!strcmp(name, "covered")) { *offset = L.covered; *bytes = size_t(gx) * gy * 16; return GSR_OK; }
        if (!strcmp(name, "touch")) { *offset = L.touch; *bytes = size_t(D) * 4; return GSR_OK; }
        if (!strcmp(name, "row_count")) { *offset = L.slot_cnt; *bytes = size_t(D); return GSR_OK; }
    } else if (which == GSR_BUF_IMAGE) {
        const GsrImageLayout L(P);
        if (!strcmp(name, "final_T")) { *offset = L.final_T; *bytes = size_t(P) * 12; return GSR_OK; }
        if (!strcmp(name, "n_contrib")) { *offset = L.n_contrib; *bytes = size_t(P) * 8; return GSR_OK; }
    }
    gsr_set_error("unknown buffer field %d/'%s'", which, name);
    return GSR_E_INVALID;
}

// ------------------------------------------------------------------------------- sort entry point
extern "C" size_t gsr_sort_workspace_bytes(int32_t n) {
    return 2 * gsr_align(size_t(n > 0 ? n : 1) * 4) + gsr_sort_ws_bytes(n);
}
extern "C" int32_t gsr_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                      uint32_t* vals_out, int32_t n, int32_t begin_bit, int32_t end_bit,
                                      void* ws, size_t ws_bytes, gsr_stream_t stream) {
    if (n < 0 || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) { gsr_set_error("bad sort arguments"); return GSR_E_INVALID; }
    if (n == 0) return GSR_OK;
    if (!keys_in || !keys_out || !vals_out || !ws || ws_bytes < gsr_sort_workspace_bytes(n)) {
        gsr_set_error("sort buffers / workspace too small");
        return GSR_E_INVALID;
    }
    const size_t nb = gsr_align(size_t(n) * 4);
    char* w = static_cast<char*>(ws);
    return gsr_radix_sort_pairs(keys_in, vals_in, keys_out, vals_out, reinterpret_cast<uint32_t*>(w),
                                reinterpret_cast<uint32_t*>(w + nb), n, begin_bit, end_bit, w + 2 * nb,
                                static_cast<hipStream_t>(stream));
}
