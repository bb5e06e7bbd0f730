// ks_copy.hip — host <-> device copies of the boundary (ks_sketch_batch, ks_kmer_positions, *_copy_to_host, ks_dev_upload /
// ks_dev_download) that overlap their two halves.
//
// A pageable host buffer cannot be DMA'd directly: the runtime bounces it through a small pinned buffer, serially, at a
// fraction of the link rate (round 1: 3.5 GB of sketches came back at ~17 GB/s).  Here:
//   * a host buffer that is already pinned (ks_host_alloc, hipHostMalloc, hipHostRegister) is copied in ONE asynchronous
//     DMA at link rate — callers that want the sketches back should hand in pinned arrays;
//   * a pageable DESTINATION goes through two pinned staging buffers of the context: while the DMA engine moves chunk i, a
//     small pool of host threads copies chunk i-1 out of the other one — the memcpy, not the DMA, is the slower half (first
//     touch of fresh destination pages), hence several threads: 53 GB/s instead of the runtime's ~17 GB/s for 3.5 GB of
//     sketches.  Pageable SOURCES are left to the runtime, which uploads warm pages near link rate by itself.
// Everything is ordered on the context's stream; the calls return when the data has arrived.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "ks_common.h"

#define KC_CHUNK ((size_t)32 << 20) // bytes per staging buffer
#define KC_THREADS 8

struct ks_copy_engine {
    u8 *stage[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    // a tiny fork-join pool: run(f) calls f(t) for t in [0, KC_THREADS) on the workers and returns when all are done
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    u64 generation = 0;
    int remaining = 0;
    bool quit = false;

    void worker(int t) {
        u64 seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> l(m);
                cv_go.wait(l, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                f = job;
            }
            f(t);
            {
                std::lock_guard<std::mutex> l(m);
                if (--remaining == 0) cv_done.notify_all();
            }
        }
    }
    void run(const std::function<void(int)> &f) {
        std::unique_lock<std::mutex> l(m);
        job = f;
        remaining = KC_THREADS;
        generation++;
        cv_go.notify_all();
        cv_done.wait(l, [&] { return remaining == 0; });
    }
    static void split_copy(ks_copy_engine *e, void *dst, const void *src, size_t n) {
        if (n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
        e->run([=](int t) {
            const size_t lo = n * (size_t)t / KC_THREADS, hi = n * (size_t)(t + 1) / KC_THREADS;
            memcpy((u8 *)dst + lo, (const u8 *)src + lo, hi - lo);
        });
    }
};

static ks_copy_engine *engine(ks_ctx *ctx) {
    if (ctx->copy) return ctx->copy;
    ks_copy_engine *e = new ks_copy_engine();
    bool ok = true;
    for (int i = 0; i < 2 && ok; i++)
        ok = hipHostMalloc((void **)&e->stage[i], KC_CHUNK) == hipSuccess && hipEventCreateWithFlags(&e->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < 2; i++) {
            if (e->stage[i]) (void)hipHostFree(e->stage[i]);
            if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
        }
        delete e;
        return nullptr;
    }
    for (int t = 0; t < KC_THREADS; t++) e->workers.emplace_back(&ks_copy_engine::worker, e, t);
    ctx->copy = e;
    return e;
}

void ks_copy_engine_destroy(ks_ctx *ctx) {
    ks_copy_engine *e = ctx->copy;
    if (!e) return;
    {
        std::lock_guard<std::mutex> l(e->m);
        e->quit = true;
        e->cv_go.notify_all();
    }
    for (auto &t : e->workers) t.join();
    for (int i = 0; i < 2; i++) {
        (void)hipHostFree(e->stage[i]);
        (void)hipEventDestroy(e->ev[i]);
    }
    delete e;
    ctx->copy = nullptr;
}

static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; } // plain malloc'ed memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}

// Enqueues (and, for a pageable source, completes) the upload.  The caller synchronises the stream before it lets go of src.
int ks_copy_h2d(ks_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return KS_OK;
    // Measured (MI355X box, 310 MB of residues): the runtime's own pageable upload already runs near link rate (the source
    // pages are warm), and staging it through host threads is slower — unlike the download direction, where the runtime
    // manages ~17 GB/s and the staged path 53 GB/s.  So uploads are staged only on request.
    ks_copy_engine *e = (bytes >= ((size_t)4 << 20) && ks_dbg(ctx, KS_DBG_STAGED_H2D) && !host_is_pinned(src)) ? engine(ctx) : nullptr;
    if (!e) { // the default: one copy
        KS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return KS_OK;
    }
    int i = 0;
    for (size_t off = 0; off < bytes; off += KC_CHUNK, i ^= 1) {
        const size_t n = bytes - off < KC_CHUNK ? bytes - off : KC_CHUNK;
        if (off >= 2 * KC_CHUNK) KS_HIP(ctx, hipEventSynchronize(e->ev[i])); // the DMA that last read this staging buffer is done
        ks_copy_engine::split_copy(e, e->stage[i], (const u8 *)src + off, n);
        KS_HIP(ctx, hipMemcpyAsync((u8 *)dst + off, e->stage[i], n, hipMemcpyHostToDevice, ctx->stream));
        KS_HIP(ctx, hipEventRecord(e->ev[i], ctx->stream));
    }
    return KS_OK;
}

// Returns when dst holds the data.
int ks_copy_d2h(ks_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return KS_OK;
    ks_copy_engine *e = (bytes >= ((size_t)4 << 20) && !host_is_pinned(dst) && !ks_dbg(ctx, KS_DBG_PLAIN_COPIES)) ? engine(ctx) : nullptr;
    if (!e) {
        KS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return KS_OK;
    }
    const size_t n_chunks = (bytes + KC_CHUNK - 1) / KC_CHUNK;
    auto len_of = [&](size_t c) { return bytes - c * KC_CHUNK < KC_CHUNK ? bytes - c * KC_CHUNK : KC_CHUNK; };
    for (size_t c = 0; c <= n_chunks; c++) {
        if (c < n_chunks) { // DMA of chunk c into its staging buffer ...
            KS_HIP(ctx, hipMemcpyAsync(e->stage[c & 1], (const u8 *)src + c * KC_CHUNK, len_of(c), hipMemcpyDeviceToHost, ctx->stream));
            KS_HIP(ctx, hipEventRecord(e->ev[c & 1], ctx->stream));
        }
        if (c > 0) { // ... while the host threads move chunk c - 1 out of the other one
            KS_HIP(ctx, hipEventSynchronize(e->ev[(c - 1) & 1]));
            ks_copy_engine::split_copy(e, (u8 *)dst + (c - 1) * KC_CHUNK, e->stage[(c - 1) & 1], len_of(c - 1));
        }
    }
    return KS_OK;
}

extern "C" int ks_host_alloc(ks_ctx *ctx, uint64_t bytes, void **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !out) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(out, bytes ? bytes : 64) != hipSuccess) { (void)hipGetLastError(); return ks_fail(ctx, KS_ERR_OOM, "hipHostMalloc(%llu) failed", (unsigned long long)bytes); }
    return KS_OK;
    });
}
extern "C" int ks_host_free(ks_ctx *ctx, void *ptr) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (ptr) KS_HIP(ctx, hipHostFree(ptr));
    return KS_OK;
    });
}
