// context.hip -- library context, error reporting, device memory helpers, HIP-event profiling.
#include "common.h"

static thread_local char g_err[1024] = "";

void sf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *sf_last_error(void) { return g_err; }
#ifndef SF_BUILD_ID
#define SF_BUILD_ID "unknown"
#endif
extern "C" const char *sf_version(void) { return "shotfpfh-gfx950 0.3 build " SF_BUILD_ID; }

extern "C" int sf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" sf_ctx *sf_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        sf_set_error("no HIP device available (%s); libshotfpfh has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        sf_set_error("device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    SF_HIP_NULL(hipSetDevice(device));
    sf_ctx *ctx = new sf_ctx();
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking);
    // the side stream carries small, latency-bound kernels that are meant to disappear under a large one on the main
    // stream (the frame eigen-solves under K7): at equal priority the large kernel's waves crowd them out and the small
    // kernel ends AFTER it, delaying whatever joins both -- so the side stream gets the higher priority
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->streams[1], hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->mark_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        sf_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->streams[0];
    void *flag = nullptr;
    if (hipHostMalloc(&flag, 64, hipHostMallocMapped) != hipSuccess) {
        sf_set_error("hipHostMalloc failed");
        delete ctx;
        return nullptr;
    }
    memset(flag, 0, 64);
    ctx->dev_flag = (volatile int *)flag;
    return ctx;
}

int sf_ctx_check_flag(sf_ctx *ctx)
{
    if (!ctx->dev_flag || !*ctx->dev_flag) return SF_OK;
    const int bits = *ctx->dev_flag;
    *ctx->dev_flag = 0;
    sf_set_error("an index array passed to the library holds values out of range:%s%s (the offending elements were skipped)",
                 bits & SF_FLAG_ROWS_GATHER ? " sf_rows_gather's row selection" : "",
                 bits & SF_FLAG_VOXEL_ORDER ? " sf_voxels_select's visiting order" : "");
    return SF_ERR_ARG;
}

extern "C" void sf_destroy(sf_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->streams[0]);
    (void)hipStreamSynchronize(ctx->streams[1]);
    ctx->stream = ctx->streams[0];
    sf_comm_destroy(ctx);
    for (auto &kv : ctx->prof)
        for (auto &p : kv.second.pending) {
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
    for (hipEvent_t ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->shot_coef) (void)hipFree(ctx->shot_coef);
    if (ctx->dev_flag) (void)hipHostFree((void *)ctx->dev_flag);
    sf_pool_trim(ctx);
    for (auto &kv : ctx->pool_size) (void)hipFree(kv.first); // blocks still held by live handles
    (void)hipEventDestroy(ctx->join_event);
    (void)hipStreamDestroy(ctx->streams[0]);
    (void)hipStreamDestroy(ctx->streams[1]);
    delete ctx;
}

extern "C" int sf_sync(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    SF_HIP(hipStreamSynchronize(ctx->streams[0]));
    SF_HIP(hipStreamSynchronize(ctx->streams[1]));
    return sf_ctx_check_flag(ctx);
}

// Two streams per context let independent stages of the path (the FPFH chain K6 -> K7 and the SHOT chain
// K4 -> K5, both consumers of the same neighbour lists) run side by side.  sf_fork: subsequent calls go to the
// side stream, ordered after everything issued so far on the main one.  sf_join: back to the main stream,
// ordered after the side stream.  Only calls that neither allocate nor free device memory may be issued while
// forked (the caching allocator is ordered on the main stream).
extern "C" int sf_fork(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (ctx->stream != ctx->streams[0]) { sf_set_error("sf_fork: already forked"); return SF_ERR_STATE; }
    SF_HIP(hipEventRecord(ctx->join_event, ctx->streams[0]));
    SF_HIP(hipStreamWaitEvent(ctx->streams[1], ctx->join_event, 0));
    ctx->stream = ctx->streams[1];
    return SF_OK;
}

extern "C" int sf_switch(sf_ctx *ctx, int side)
{
    if (!ctx || side < 0 || side > 1) { sf_set_error("sf_switch: bad argument"); return SF_ERR_ARG; }
    ctx->stream = ctx->streams[side];
    return SF_OK;
}

extern "C" int sf_join(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    SF_HIP(hipEventRecord(ctx->join_event, ctx->streams[1]));
    SF_HIP(hipStreamWaitEvent(ctx->streams[0], ctx->join_event, 0));
    ctx->stream = ctx->streams[0];
    return SF_OK;
}

// A finer dependency than sf_join: sf_mark remembers the point reached on the CURRENT stream, sf_wait_mark makes the
// current stream wait for that point only -- not for what was issued on the marked stream afterwards.  (The sharded pass:
// the boundary keypoints of K7 wait for the row exchange on the side stream, not for the eigen-solves queued behind it.)
extern "C" int sf_mark(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    SF_HIP(hipEventRecord(ctx->mark_event, ctx->stream));
    return SF_OK;
}

extern "C" int sf_wait_mark(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    SF_HIP(hipStreamWaitEvent(ctx->stream, ctx->mark_event, 0));
    return SF_OK;
}

extern "C" void *sf_stream(sf_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---- a step as a HIP graph ------------------------------------------------------------------------------------------------
// A repeated step of the path -- same cloud, same radius, same buffers -- issues the same launches with the same arguments
// (round 5: no read-back left in it, every launch is planned from the previous search's record).  Small steps are then bound
// by the host: a rank's eighth of the 1M-point cloud is 0.5 ms of GPU work behind some thirty library calls and a dozen
// launches.  sf_graph_begin puts the context's main stream into capture (the side stream joins through the fork / join
// events), sf_graph_end turns what was issued into an executable graph, sf_graph_launch replays it with ONE call.  While
// capturing, nothing may synchronise with the device (a call that would -- a first search of a range, a table whose block
// mask the host does not know -- fails with the runtime's error and the capture is abandoned by sf_graph_end); pool blocks
// released during the capture stay with the graph, which writes into them at every replay, until sf_graph_free.
std::atomic<unsigned long long> g_sf_sync_count{0};
extern "C" unsigned long long sf_sync_count(void) { return g_sf_sync_count.load(std::memory_order_relaxed); }

extern "C" int sf_graph_begin(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (ctx->capture) { sf_set_error("sf_graph_begin: already capturing"); return SF_ERR_STATE; }
    if (ctx->stream != ctx->streams[0]) { sf_set_error("sf_graph_begin: forked (sf_join first)"); return SF_ERR_STATE; }
    if (ctx->profiling) { sf_set_error("sf_graph_begin: launch timers are on (sf_profile(0) first)"); return SF_ERR_STATE; }
    SF_HIP(hipSetDevice(ctx->device));
    SF_HIP(hipStreamSynchronize(ctx->streams[0]));
    SF_HIP(hipStreamSynchronize(ctx->streams[1]));
    // (thread-local mode, not relaxed -- advisor, round 5: a synchronous hipMemcpy, hipMalloc / hipFree or event wait issued by this
    // thread while capturing then FAILS and invalidates the capture -- sf_graph_end reports it and the caller runs the step eagerly --
    // instead of executing at once against buffers the captured kernels have not written yet and being baked into the graph as a
    // host-side decision.  The eager step that precedes every capture has warmed the pool: a captured step allocates nothing.)
    SF_HIP(hipStreamBeginCapture(ctx->streams[0], hipStreamCaptureModeThreadLocal));
    ctx->capture = new sf_graph();
    return SF_OK;
}

extern "C" sf_graph *sf_graph_end(sf_ctx *ctx)
{
    if (!ctx || !ctx->capture) { sf_set_error("sf_graph_end: not capturing"); return nullptr; }
    sf_graph *g = ctx->capture;
    ctx->capture = nullptr;
    ctx->stream = ctx->streams[0];
    hipError_t e = hipStreamEndCapture(ctx->streams[0], &g->graph);
    if (e == hipSuccess && g->graph) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess || !g->exec) {
        sf_set_error("sf_graph_end: the step could not be captured (%s)", hipGetErrorString(e));
        (void)hipGetLastError();
        sf_graph_free(ctx, g);
        return nullptr;
    }
    return g;
}

extern "C" int sf_graph_launch(sf_ctx *ctx, sf_graph *g)
{
    if (!ctx || !g || !g->exec) { sf_set_error("sf_graph_launch: bad argument"); return SF_ERR_ARG; }
    if (ctx->capture) { sf_set_error("sf_graph_launch: capturing"); return SF_ERR_STATE; }
    SF_HIP(hipGraphLaunch(g->exec, ctx->streams[0]));
    return SF_OK;
}

extern "C" void sf_graph_free(sf_ctx *ctx, sf_graph *g)
{
    if (!g) return;
    if (ctx) { (void)hipStreamSynchronize(ctx->streams[0]); (void)hipStreamSynchronize(ctx->streams[1]); }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    for (void *p : g->held) { if (ctx) sf_pool_release(ctx, p); else (void)hipFree(p); }
    delete g;
}

extern "C" void *sf_dev_alloc(sf_ctx *ctx, size_t bytes)
{
    if (!ctx) { sf_set_error("null ctx"); return nullptr; }
    void *p = nullptr;
    SF_HIP_NULL(hipSetDevice(ctx->device));
    SF_HIP_NULL(hipMalloc(&p, bytes ? bytes : 8));
    return p;
}

extern "C" int sf_dev_free(sf_ctx *ctx, void *p)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (!p) return SF_OK;
    SF_HIP(hipStreamSynchronize(ctx->stream));
    SF_HIP(hipFree(p));
    return SF_OK;
}

// Page-locked host memory for callers that want their results at PCIe speed: a device-to-host copy into pageable
// memory is staged by the runtime through its own pinned buffers (~10-25 GB/s with first-touch page faults on a
// fresh array); into memory from sf_host_alloc it is one DMA (~55 GB/s).  Allocation pins pages and is slow, so
// callers keep and reuse these blocks (shot_fpfh_amd.engine does: NumPy outputs above 32 MiB live in them).
extern "C" void *sf_host_alloc(sf_ctx *ctx, size_t bytes)
{
    if (!ctx) { sf_set_error("null ctx"); return nullptr; }
    void *p = nullptr;
    SF_HIP_NULL(hipSetDevice(ctx->device));
    SF_HIP_NULL(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault));
    return p;
}

extern "C" int sf_host_free(sf_ctx *ctx, void *p)
{
    (void)ctx; // (valid after the context is gone: outputs may outlive their engine)
    if (!p) return SF_OK;
    SF_HIP(hipHostFree(p));
    return SF_OK;
}

extern "C" int sf_h2d(sf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (!bytes) return SF_OK;
    SF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_d2h(sf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (!bytes) return SF_OK;
    SF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return sf_ctx_check_flag(ctx);
}

extern "C" int sf_d2d(sf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (!bytes) return SF_OK;
    SF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream)); // stream-ordered, no sync
    return SF_OK;
}

int sf_ctx_scratch(sf_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            SF_HIP(hipStreamSynchronize(ctx->stream));
            SF_HIP(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        SF_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return SF_OK;
}

int sf_ctx_pinned(sf_ctx *ctx, void **out)
{
    if (!ctx->pinned) SF_HIP(hipHostMalloc(&ctx->pinned, SF_PINNED_BYTES, hipHostMallocDefault));
    *out = ctx->pinned;
    return SF_OK;
}

int sf_pool_alloc(sf_ctx *ctx, size_t bytes, void **out)
{
    if (bytes < 256) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->pool_free.lower_bound(bytes);
    if (it != ctx->pool_free.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
        *out = it->second;
        ctx->pool_cached -= it->first;
        ctx->pool_free.erase(it);
        return SF_OK;
    }
    void *p = nullptr;
    // (while a step is being captured -- thread-local mode -- the allocation, and only it, runs with the thread's capture mode
    // relaxed: a fresh block has no pending work, so allocating it eagerly is safe; every other eager call stays forbidden)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (ctx->capture) (void)hipThreadExchangeStreamCaptureMode(&mode);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && !ctx->capture) { // give cached blocks back and retry once
        sf_pool_trim(ctx);
        e = hipMalloc(&p, bytes);
    }
    if (ctx->capture) (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (e != hipSuccess) {
        sf_set_error("out of device memory allocating %zu bytes: %s", bytes, hipGetErrorString(e));
        return SF_ERR_NOMEM;
    }
    ctx->pool_size[p] = bytes;
    *out = p;
    return SF_OK;
}

void sf_pool_release(sf_ctx *ctx, void *p)
{
    if (!p) return;
    auto it = ctx->pool_size.find(p);
    if (it == ctx->pool_size.end()) { (void)hipFree(p); return; }
    if (ctx->capture) { ctx->capture->held.push_back(p); return; } // (a captured launch refers to it: the graph keeps it)
    ctx->pool_free.emplace(it->second, p);
    ctx->pool_cached += it->second;
    if (ctx->pool_cached > ((size_t)96 << 30)) sf_pool_trim(ctx);
}

void sf_pool_trim(sf_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool_free) {
        (void)hipFree(kv.second);
        ctx->pool_size.erase(kv.second);
    }
    ctx->pool_free.clear();
    ctx->pool_cached = 0;
}

hipEvent_t sf_ctx_event(sf_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

static void prof_collect(sf_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->streams[0]);
    (void)hipStreamSynchronize(ctx->streams[1]);
    for (auto &kv : ctx->prof) {
        for (auto &p : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) kv.second.total_ms += ms;
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

extern "C" int sf_profile_enable(sf_ctx *ctx, int on)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    if (!on) prof_collect(ctx);
    ctx->profiling = on != 0;
    return SF_OK;
}

extern "C" int sf_profile_only(sf_ctx *ctx, const char *name)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    ctx->prof_only = name ? name : "";
    return SF_OK;
}

extern "C" int sf_profile_reset(sf_ctx *ctx)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    prof_collect(ctx);
    ctx->prof.clear();
    return SF_OK;
}

extern "C" int64_t sf_profile_report(sf_ctx *ctx, char *buf, int64_t cap)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    prof_collect(ctx);
    std::string s;
    char line[256];
    for (auto &kv : ctx->prof) {
        snprintf(line, sizeof(line), "%s %lld %.6f\n", kv.first.c_str(), (long long)kv.second.launches,
                 kv.second.total_ms);
        s += line;
    }
    if (buf && cap > 0) {
        size_t ncopy = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
        memcpy(buf, s.data(), ncopy);
        buf[ncopy] = 0;
    }
    return (int64_t)s.size() + 1;
}
