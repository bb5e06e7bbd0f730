// ks_ctx.hip — context, device memory pool, HIP-event timing, host-side helpers and the C-ABI
// entry points that are pure plumbing.  No CPU fallback lives here: every compute entry point
// ends in a HIP kernel launch (ks_sketch.hip / ks_search.hip).
#include <cstdarg>
#include <algorithm>

#include "ks_common.h"

// ---------------------------------------------------------------------------------------------
int ks_fail(ks_ctx *ctx, int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        try { ctx->err = buf; } catch (...) { } // (the status code still says what happened)
    }
    return status;
}

extern "C" uint32_t ks_abi_version(void) { return KS_ABI_VERSION; }

extern "C" const char *ks_status_string(int s) {
    switch (s) {
    case KS_OK: return "ok";
    case KS_ERR_INVALID_MOLTYPE: return "invalid moltype";
    case KS_ERR_INVALID_KSIZE: return "invalid k-mer size";
    case KS_ERR_INVALID_RESIDUE: return "invalid amino acid";
    case KS_ERR_INVALID_ARG: return "invalid argument";
    case KS_ERR_OOM: return "out of device memory";
    case KS_ERR_HIP: return "HIP runtime error";
    case KS_ERR_NO_DEVICE: return "no HIP device";
    case KS_ERR_CAPACITY: return "device list capacity exceeded";
    case KS_ERR_INVALID_SCALED: return "invalid scaled";
    default: return "unknown status";
    }
}

// get_hash_function_from_moltype, src/rust/encoding.rs:17-27
extern "C" int ks_moltype_from_string(const char *name, uint32_t *out) {
    return ks_guard(nullptr, [&]() -> int {
    if (!name || !out) return KS_ERR_INVALID_ARG;
    if (!strcmp(name, "protein") || !strcmp(name, "raw")) { *out = KS_PROTEIN; return KS_OK; }
    if (!strcmp(name, "hp")) { *out = KS_HP; return KS_OK; }
    if (!strcmp(name, "dayhoff")) { *out = KS_DAYHOFF; return KS_OK; }
    return KS_ERR_INVALID_MOLTYPE;
    });
}

// sourmash max_hash_for_scaled: (u64::MAX as f64 / scaled as f64) as u64, saturating
extern "C" uint64_t ks_max_hash(uint32_t scaled) {
    if (scaled == 0) return 0;
    if (scaled == 1) return UINT64_MAX;
    double v = 18446744073709551616.0 / (double)scaled;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

int ks_check_params(ks_ctx *ctx, const ks_params *p) {
    if (!p) return ks_fail(ctx, KS_ERR_INVALID_ARG, "params is NULL");
    if (p->moltype > KS_HP)
        return ks_fail(ctx, KS_ERR_INVALID_MOLTYPE,
                       "Invalid moltype: %u, only 'protein', 'hp', or 'dayhoff' are supported", p->moltype);
    if (p->ksize < 1 || p->ksize > KS_MAX_KSIZE)
        return ks_fail(ctx, KS_ERR_INVALID_KSIZE, "Invalid k-mer size: %u", p->ksize);
    if (p->scaled < 1) return ks_fail(ctx, KS_ERR_INVALID_SCALED, "Invalid scaled: %u", p->scaled);
    return KS_OK;
}

// ---- encode LUTs: byte -> encoded byte of upper-cased residue --------------------------------
// sourmash aa_to_dayhoff / aa_to_hp (selected at src/rust/encoding.rs:43-53); sourmash upper-cases
// the sequence inside add_protein, so the LUT folds to_ascii_uppercase in.
static void build_luts(u8 *lut) {
    for (int m = 0; m < 3; m++)
        for (int b = 0; b < 256; b++) {
            u8 c = (b >= 'a' && b <= 'z') ? (u8)(b - 32) : (u8)b;
            u8 e = c;
            if (m == KS_DAYHOFF) {
                switch (c) {
                case 'C': e = 'a'; break;
                case 'A': case 'G': case 'P': case 'S': case 'T': e = 'b'; break;
                case 'D': case 'E': case 'N': case 'Q': e = 'c'; break;
                case 'H': case 'K': case 'R': e = 'd'; break;
                case 'I': case 'L': case 'M': case 'V': e = 'e'; break;
                case 'F': case 'W': case 'Y': e = 'f'; break;
                default: e = 'X';
                }
            } else if (m == KS_HP) {
                switch (c) {
                case 'A': case 'F': case 'G': case 'I': case 'L': case 'M': case 'P': case 'V':
                case 'W': case 'Y': e = 'h'; break;
                case 'N': case 'C': case 'S': case 'T': case 'D': case 'E': case 'R': case 'H':
                case 'K': case 'Q': e = 'p'; break;
                default: e = 'X';
                }
            }
            lut[m * 256 + b] = e;
        }
}

// ---- diagnostic knobs: read once (see ks_common.h) ------------------------------------------
void ks_debug_load(ks_debug *d) {
    static const char *const names[KS_DBG_COUNT] = {
#define KS_DBG_NAME(n) "KS_DEBUG_" #n,
        KS_DBG_LIST(KS_DBG_NAME)
#undef KS_DBG_NAME
    };
    for (int i = 0; i < KS_DBG_COUNT; i++) {
        const char *v = getenv(names[i]);
        d->set[i] = v != nullptr;
        d->val[i][0] = 0;
        if (v) { strncpy(d->val[i], v, sizeof d->val[i] - 1); d->val[i][sizeof d->val[i] - 1] = 0; }
    }
}
static void debug_apply(ks_ctx *ctx) {
    ks_debug_load(&ctx->dbg);
    const char *cap = ks_dbg(ctx, KS_DBG_POOL_CAP);
    ctx->pool_cap = cap ? strtoull(cap, nullptr, 10) : 0;
}
extern "C" int ks_ctx_reload_debug_env(ks_ctx *ctx) {
    if (!ctx) return KS_ERR_INVALID_ARG;
    debug_apply(ctx);
    return KS_OK;
}

// ---- exception guard of the extern "C" boundary (see ks_common.h) ---------------------------
int ks_guard_fail(ks_ctx *ctx, int status, const char *what) noexcept {
    if (ctx) {
        try {
            ctx->err.assign(status == KS_ERR_OOM ? "" : "internal error: ");
            ctx->err.append(what ? what : "?");
        } catch (...) { // (no memory even for the message: the status code still says what happened)
        }
    }
    return status;
}
void ks_guard_enter(ks_ctx *ctx) {
    const char *t = ks_dbg(ctx, KS_DBG_THROW);
    if (!t) return;
    if (!strcmp(t, "bad_alloc")) throw std::bad_alloc();
    if (!strcmp(t, "runtime")) throw std::runtime_error("KS_DEBUG_THROW");
    if (!strcmp(t, "other")) throw 42;
}

// Self-test of the guard, callable without a device (tests/test_abi.py): runs a body that throws the named exception
// ("bad_alloc", "runtime", "other"; anything else: throws nothing) through ks_guard and returns the status that came out.
extern "C" int ks_debug_guard_selftest(const char *what) {
    return ks_guard(nullptr, [&]() -> int {
        if (what && !strcmp(what, "bad_alloc")) { std::vector<char> v; v.reserve((size_t)-1 / 2); } // (length_error / bad_alloc: a real failed reservation)
        if (what && !strcmp(what, "runtime")) throw std::runtime_error("selftest");
        if (what && !strcmp(what, "other")) throw 42;
        return KS_OK;
    });
}

// ---------------------------------------------------------------------------------------------
extern "C" int ks_ctx_create(int device, void *hip_stream, ks_ctx **out) {
    if (!out) return KS_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return KS_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return KS_ERR_INVALID_ARG;
    ks_ctx *ctx = new (std::nothrow) ks_ctx();
    if (!ctx) return KS_ERR_OOM;
    ctx->device = device;
    try { ctx->err.reserve(640); ctx->pool.reserve(256); } catch (...) { delete ctx; return KS_ERR_OOM; }
    debug_apply(ctx);
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return KS_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return KS_ERR_HIP; }
        ctx->own_stream = true;
    }
    // (coherent, explicitly: the host SPINS on words the device writes while its kernel is still in the queue — with
    // HIP_HOST_COHERENT=0 in the environment the default flags would give non-coherent memory, visible only after a synchronisation)
    if (hipHostMalloc((void **)&ctx->h_pin, KS_PIN_WORDS * sizeof(u64), hipHostMallocCoherent) != hipSuccess) { delete ctx; return KS_ERR_HIP; }
    if (hipHostMalloc((void **)&ctx->h_flag, 64, hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); ctx->h_flag = nullptr; } // (ks_stream_wait falls back to the API)
    else *ctx->h_flag = 0;
    if (hipMalloc((void **)&ctx->d_lut, 3 * 256) != hipSuccess) { delete ctx; return KS_ERR_OOM; }
    u8 lut[768];
    build_luts(lut);
    if (hipMemcpy(ctx->d_lut, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return KS_ERR_HIP; }
    *out = ctx;
    return KS_OK;
}

extern "C" void ks_ctx_destroy(ks_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ks_copy_engine_destroy(ctx);
    for (auto &t : ctx->t_pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto &t : ctx->t_free) { (void)hipEventDestroy(t.first); (void)hipEventDestroy(t.second); }
    for (auto &b : ctx->pool) (void)hipFree(b.ptr);
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    if (ctx->scan_ring) (void)hipFree(ctx->scan_ring);
    if (ctx->scan_ticket) (void)hipFree(ctx->scan_ticket);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *ks_last_error(const ks_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }
extern "C" void *ks_ctx_stream(const ks_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int ks_ctx_synchronize(ks_ctx *ctx) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}

extern "C" int ks_ctx_pool_stats(const ks_ctx *ctx, uint64_t *n_blocks, uint64_t *bytes_held, uint64_t *bytes_in_use,
                                 uint64_t *n_mallocs) {
    return ks_guard((ks_ctx *)ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    u64 held = 0, used = 0;
    for (auto &b : ctx->pool) { held += b.size; if (b.in_use) used += b.size; }
    if (n_blocks) *n_blocks = ctx->pool.size();
    if (bytes_held) *bytes_held = held;
    if (bytes_in_use) *bytes_in_use = used;
    if (n_mallocs) *n_mallocs = ctx->pool_mallocs;
    return KS_OK;
    });
}

extern "C" int ks_ctx_sketch_stats(const ks_ctx *ctx, uint64_t out[4]) {
    return ks_guard((ks_ctx *)ctx, [&]() -> int {
    if (!ctx || !out) return KS_ERR_INVALID_ARG;
    out[0] = ctx->sketch_ticket_fallbacks; out[1] = ctx->sketch_use_ticket ? 1 : 0;
    out[2] = ctx->sketch_compact_fallbacks; out[3] = ctx->sketch_cap_fallbacks;
    return KS_OK;
    });
}

extern "C" int ks_ctx_search_stats(const ks_ctx *ctx, uint64_t out[2]) {
    return ks_guard((ks_ctx *)ctx, [&]() -> int {
    if (!ctx || !out) return KS_ERR_INVALID_ARG;
    out[0] = ctx->join_retries; out[1] = ctx->rows_ticket_fallbacks;
    return KS_OK;
    });
}

extern "C" int ks_ctx_fused_stats(const ks_ctx *ctx, uint64_t out[2]) {
    return ks_guard((ks_ctx *)ctx, [&]() -> int {
    if (!ctx || !out) return KS_ERR_INVALID_ARG;
    out[0] = ctx->fused_deferred; out[1] = ctx->fused_redos;
    return KS_OK;
    });
}

extern "C" int ks_dev_malloc(ks_ctx *ctx, uint64_t bytes, void **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !out) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(out, bytes ? bytes : 256) != hipSuccess) { (void)hipGetLastError(); return ks_fail(ctx, KS_ERR_OOM, "hipMalloc(%llu) failed", (unsigned long long)bytes); }
    return KS_OK;
    });
}
extern "C" int ks_dev_free(ks_ctx *ctx, void *ptr) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KS_HIP(ctx, hipFree(ptr));
    return KS_OK;
    });
}
extern "C" int ks_dev_upload(ks_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || (bytes && (!dst || !src))) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_copy_h2d(ctx, dst, src, (size_t)bytes));
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}
extern "C" int ks_dev_download(ks_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || (bytes && (!dst || !src))) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_copy_d2h(ctx, dst, src, (size_t)bytes));
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}

// ---- ks_stream_wait -------------------------------------------------------------------------
__global__ void k_host_stamp(unsigned long long *flag, unsigned long long seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The words the host is waiting for, written to its pinned block by the SAME one-workgroup kernel that stamps: a device ->
// host copy of a few words is a dispatch of its own on this runtime (~4.5 us of queue each), and a step reads back three to
// five such blocks.  Up to KS_FETCH_MAX segments; a segment is `rows` rows of `row_words` 32-bit words, `src_stride` words apart
// in device memory, packed densely at dst.
struct ks_fetch_args { ks_fetch_seg seg[KS_FETCH_MAX]; int n; };
__global__ __launch_bounds__(256) void k_host_report(ks_fetch_args F, unsigned long long *flag, unsigned long long seq) {
    for (int s = 0; s < F.n; s++) {
        const ks_fetch_seg g = F.seg[s];
        const u32 total = g.rows * g.row_words;
        for (u32 i = threadIdx.x; i < total; i += blockDim.x) {
            const u32 r = i / g.row_words, c = i - r * g.row_words;
            const u32 v = __hip_atomic_load((const u32 *)g.src + (size_t)r * g.src_stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int stream_wait_poll(ks_ctx *ctx, unsigned long long seq);

int ks_stream_wait_fetch(ks_ctx *ctx, const ks_fetch_seg *segs, int n) {
    if (n > KS_FETCH_MAX) return ks_fail(ctx, KS_ERR_INVALID_ARG, "too many fetch segments");
    if (!ctx->h_flag || ks_dbg(ctx, KS_DBG_SYNC_API)) {
        for (int i = 0; i < n; i++) {
            const ks_fetch_seg &g = segs[i];
            if (g.rows == 1 || g.src_stride == g.row_words)
                KS_HIP(ctx, hipMemcpyAsync(g.dst, g.src, (size_t)g.rows * g.row_words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
            else
                KS_HIP(ctx, hipMemcpy2DAsync(g.dst, (size_t)g.row_words * sizeof(u32), g.src, (size_t)g.src_stride * sizeof(u32),
                                            (size_t)g.row_words * sizeof(u32), g.rows, hipMemcpyDeviceToHost, ctx->stream));
        }
        KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return KS_OK;
    }
    ks_fetch_args F;
    memset(&F, 0, sizeof F);
    for (int i = 0; i < n; i++) F.seg[i] = segs[i];
    F.n = n;
    const unsigned long long seq = ++ctx->wait_seq;
    hipLaunchKernelGGL(k_host_report, dim3(1), dim3(256), 0, ctx->stream, F, ctx->h_flag, seq);
    if (hipGetLastError() != hipSuccess) return ks_fail(ctx, KS_ERR_HIP, "k_host_report launch failed");
    return stream_wait_poll(ctx, seq);
}

int ks_stream_wait(ks_ctx *ctx) {
    if (!ctx->h_flag || ks_dbg(ctx, KS_DBG_SYNC_API)) {
        KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return KS_OK;
    }
    const unsigned long long seq = ++ctx->wait_seq;
    hipLaunchKernelGGL(k_host_stamp, dim3(1), dim3(1), 0, ctx->stream, ctx->h_flag, seq);
    if (hipGetLastError() != hipSuccess) { KS_HIP(ctx, hipStreamSynchronize(ctx->stream)); return KS_OK; }
    return stream_wait_poll(ctx, seq);
}

static int stream_wait_poll(ks_ctx *ctx, unsigned long long seq) {
    // (the stamp arrives behind every copy and kernel queued before it; a stream that has failed never stamps: the
    // runtime is asked now and then, and after a while it is simply left to the blocking call)
    for (unsigned spins = 1;; spins++) {
        if (__atomic_load_n(ctx->h_flag, __ATOMIC_ACQUIRE) == seq) return KS_OK;
        if ((spins & 0xfffu) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) { // everything has run: the stamp is in host memory by now
                if (__atomic_load_n(ctx->h_flag, __ATOMIC_ACQUIRE) == seq) return KS_OK;
                KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
                return KS_OK;
            }
            if (q != hipErrorNotReady) return ks_fail(ctx, KS_ERR_HIP, "stream failed: %s", hipGetErrorString(q));
            if (spins > (1u << 26)) { KS_HIP(ctx, hipStreamSynchronize(ctx->stream)); return KS_OK; } // (seconds: a long queue)
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// ---- pool ---------------------------------------------------------------------------------
// Size classes, exact fit.  A request is rounded up to its class (four classes per octave: 1.25 / 1.5 / 1.75 / 2 x 2^e, at most
// 25 % over the request, 256-byte steps below 4 KB) and only a free block of exactly that class serves it.  So the number of
// blocks a class ever holds is the largest number of them alive at once: a caller that keeps the last two or three results
// alive while the next step runs (views of a hit list, a pipelined exchange) reaches its steady state after as many steps
// and never calls hipMalloc — which synchronises the device — again.  (Round 3's rule let a request take any free block up
// to twice its size: small requests took the big blocks of results whose release was deferred, the big requests then found
// nothing and the pool kept growing — 5 hipMalloc calls inside a 20-step timed region of the configs[4] bench line.)
static size_t pool_class(size_t bytes) {
    if (bytes <= 4096) return (bytes + 255) & ~(size_t)255;
    const int e = 63 - __builtin_clzll((unsigned long long)(bytes - 1)); // bytes in (2^e, 2^(e+1)]
    const size_t step = (size_t)1 << (e - 2);
    return (bytes + step - 1) & ~(step - 1);
}

void *ks_pool_alloc(ks_ctx *ctx, size_t bytes) {
    if (bytes == 0) bytes = 256;
    const size_t want = pool_class(bytes);
    for (size_t i = 0; i < ctx->pool.size(); i++) {
        auto &b = ctx->pool[i];
        if (!b.in_use && b.size == want) { b.in_use = true; return b.ptr; }
    }
    if (ctx->pool_cap) { // KS_DEBUG_POOL_CAP (tests): behave like a device that is full
        size_t held = 0;
        for (auto &b : ctx->pool) held += b.size;
        if (held + want > ctx->pool_cap) {
            ks_fail(ctx, KS_ERR_OOM, "device pool cap: %zu bytes held, %zu wanted, cap %llu", held, want, (unsigned long long)ctx->pool_cap);
            return nullptr;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        // the device is full: any free block that is large enough will do (smallest first) ...
        (void)hipGetLastError();
        int best = -1;
        for (size_t i = 0; i < ctx->pool.size(); i++) {
            auto &b = ctx->pool[i];
            if (!b.in_use && b.size >= want && (best < 0 || b.size < ctx->pool[best].size)) best = (int)i;
        }
        if (best >= 0) { ctx->pool[best].in_use = true; return ctx->pool[best].ptr; }
        ks_pool_trim(ctx); // ... else give the cached blocks back and retry once
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ks_fail(ctx, KS_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return nullptr;
    }
    ctx->pool.push_back({p, want, true});
    ctx->pool_mallocs++;
    return p;
}

void ks_pool_free(ks_ctx *ctx, void *ptr) {
    if (!ptr || !ctx) return;
    for (auto &b : ctx->pool)
        if (b.ptr == ptr) { b.in_use = false; return; }
}

void ks_pool_trim(ks_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    std::vector<ks_pool_block> keep;
    for (auto &b : ctx->pool) {
        if (b.in_use) keep.push_back(b);
        else (void)hipFree(b.ptr);
    }
    ctx->pool.swap(keep);
}

// ---- timing ---------------------------------------------------------------------------------
static int timer_name_id(ks_ctx *ctx, const char *name) {
    for (size_t i = 0; i < ctx->t_names.size(); i++)
        if (ctx->t_names[i] == name) return (int)i;
    ctx->t_names.push_back(name);
    ctx->t_launches.push_back(0);
    ctx->t_ms.push_back(0.0);
    return (int)ctx->t_names.size() - 1;
}

static void timer_resolve(ks_ctx *ctx) {
    if (ctx->t_pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->t_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            ctx->t_ms[t.name_id] += ms;
            ctx->t_launches[t.name_id] += 1;
        }
        ctx->t_free.push_back({t.a, t.b});
    }
    ctx->t_pending.clear();
}

// mode 2 brackets only the kernel the roofline is quoted for: every bracketed launch costs two event records, and the
// kernel trace shows 10-28 us of idle queue around each of them (profiles/README.md) — a per-step tax a timed region
// should not pay for kernels whose durations the untimed full pass (mode 1) gives as well
static bool timer_is_major(const char *name) { return !strcmp(name, "sketch_tiles"); }

void ks_timer_begin(ks_ctx *ctx, const char *name) {
    ctx->t_open = false;
    if (!ctx->timing) return;
    if (ctx->timing == 2 && !timer_is_major(name)) return;
    if (ctx->t_pending.size() >= 4096) timer_resolve(ctx);
    ks_timer_slot s;
    if (!ctx->t_free.empty()) {
        s.a = ctx->t_free.back().first; s.b = ctx->t_free.back().second;
        ctx->t_free.pop_back();
    } else {
        (void)hipEventCreate(&s.a);
        (void)hipEventCreate(&s.b);
    }
    s.name_id = timer_name_id(ctx, name);
    (void)hipEventRecord(s.a, ctx->stream);
    ctx->t_pending.push_back(s);
    ctx->t_open = true;
}

void ks_timer_end(ks_ctx *ctx) {
    if (!ctx->t_open || ctx->t_pending.empty()) return;
    (void)hipEventRecord(ctx->t_pending.back().b, ctx->stream);
    ctx->t_open = false;
}

extern "C" int ks_timing_enable(ks_ctx *ctx, int enable) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!enable) timer_resolve(ctx);
    ctx->timing = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return KS_OK;
    });
}

extern "C" int ks_timing_reset(ks_ctx *ctx) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    timer_resolve(ctx);
    for (auto &v : ctx->t_ms) v = 0.0;
    for (auto &v : ctx->t_launches) v = 0;
    return KS_OK;
    });
}

extern "C" int ks_timing_get(ks_ctx *ctx, ks_kernel_time *rows, uint32_t cap, uint32_t *n) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !n) return KS_ERR_INVALID_ARG;
    timer_resolve(ctx);
    *n = (uint32_t)ctx->t_names.size();
    for (uint32_t i = 0; i < *n && i < cap && rows; i++) {
        memset(rows[i].name, 0, sizeof rows[i].name);
        strncpy(rows[i].name, ctx->t_names[i].c_str(), sizeof rows[i].name - 1);
        rows[i].launches = ctx->t_launches[i];
        rows[i].total_ms = ctx->t_ms[i];
    }
    return KS_OK;
    });
}

// ---- device ceilings for bench.py: u64 multiply rate, device copy rate ----
__global__ __launch_bounds__(256) void k_bench_u64_mul(u64 *out, u32 iters) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 a = t * 0x87c37b91114253d5ULL + 1, b = t * 0x4cf5ad432745937fULL + 3, c = t + 5, d = ~t; // four independent chains hide the multiplier latency
    for (u32 i = 0; i < iters; i++) {
        a = a * 0x87c37b91114253d5ULL + i; b = b * 0x4cf5ad432745937fULL + i; c = c * 0x87c37b91114253d5ULL + i; d = d * 0x4cf5ad432745937fULL + i; // MurmurHash3 c1, c2
    }
    out[t] = a ^ b ^ c ^ d;
}

extern "C" int ks_bench_device_rates(ks_ctx *ctx, double *gmul_per_s, double *copy_gb_per_s, double *nominal_gb_per_s) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    KS_HIP(ctx, hipEventCreate(&e0));
    KS_HIP(ctx, hipEventCreate(&e1));
    int st = KS_OK;
    float ms = 0;
    if (gmul_per_s) {
        const u32 blocks = 256 * 16, iters = 8192;
        u64 *out = nullptr;
        st = ks_alloc(ctx, &out, (size_t)blocks * 256);
        if (st == KS_OK) {
            hipLaunchKernelGGL(k_bench_u64_mul, dim3(blocks), dim3(256), 0, ctx->stream, out, 64u); // warm-up
            (void)hipEventRecord(e0, ctx->stream);
            hipLaunchKernelGGL(k_bench_u64_mul, dim3(blocks), dim3(256), 0, ctx->stream, out, iters);
            (void)hipEventRecord(e1, ctx->stream);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                st = ks_fail(ctx, KS_ERR_HIP, "u64-mul micro-benchmark failed");
            else
                *gmul_per_s = (double)blocks * 256 * iters * 4 / (ms * 1e-3) / 1e9;
            ks_pool_free(ctx, out);
        }
    }
    if (st == KS_OK && copy_gb_per_s) {
        const size_t bytes = (size_t)2 << 30;
        u8 *a = nullptr, *b = nullptr;
        st = ks_alloc(ctx, &a, bytes);
        if (st == KS_OK) st = ks_alloc(ctx, &b, bytes);
        if (st == KS_OK) {
            (void)hipMemsetAsync(a, 1, bytes, ctx->stream);
            (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, ctx->stream);
            (void)hipEventRecord(e0, ctx->stream);
            for (int i = 0; i < 4; i++) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, ctx->stream);
            (void)hipEventRecord(e1, ctx->stream);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                st = ks_fail(ctx, KS_ERR_HIP, "copy micro-benchmark failed");
            else
                *copy_gb_per_s = 4.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
        }
        ks_pool_free(ctx, a); ks_pool_free(ctx, b);
    }
    if (st == KS_OK && nominal_gb_per_s) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "hipGetDeviceProperties failed");
        else *nominal_gb_per_s = 2.0 * (double)prop.memoryClockRate * 1e3 * ((double)prop.memoryBusWidth / 8.0) / 1e9;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return st;
    });
}

// ---- random-gather rates (SURVEY §7's "small-alphabet table path" for hp: a rolling 24-bit window index into a table of
// precomputed murmur hashes, 2^24 x 8 B = 134 MB, or into a 2^24-bit keep-bitmap, 2 MB).  The probe answers whether a
// gather can replace the ten 64-bit multiplies of the hash: it cannot.
__global__ __launch_bounds__(256) void k_bench_gather(const u64 *table, u64 mask, u32 iters, u64 *out) {
    u64 x = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ULL + 1, acc = 0;
    for (u32 i = 0; i < iters; i++) { // four independent gathers per round: latency hidden by ILP + occupancy
        u64 a = x * 0xbf58476d1ce4e5b9ULL, b = a ^ (a >> 29), c = b * 0x94d049bb133111ebULL, d = c ^ (c >> 31);
        acc += table[a >> 40 & mask] + table[b >> 13 & mask] + table[c >> 37 & mask] + table[d >> 7 & mask];
        x = d + i;
    }
    out[(u64)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// gathers_per_s[0]: random 8-byte gathers from a 134 MB table; [1]: from a 2 MB table (fits one XCD's L2)
extern "C" int ks_bench_gather_rates(ks_ctx *ctx, double *gathers_per_s) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !gathers_per_s) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    KS_HIP(ctx, hipEventCreate(&e0));
    KS_HIP(ctx, hipEventCreate(&e1));
    const u32 blocks = 256 * 24, iters = 256;
    u64 *table = nullptr, *out = nullptr;
    int st = ks_alloc(ctx, &table, (size_t)1 << 24);
    if (st == KS_OK) st = ks_alloc(ctx, &out, (size_t)blocks * 256);
    if (st == KS_OK) {
        (void)hipMemsetAsync(table, 1, ((size_t)1 << 24) * sizeof(u64), ctx->stream);
        const u64 masks[2] = {(1ULL << 24) - 1, (1ULL << 18) - 1};
        for (int v = 0; v < 2 && st == KS_OK; v++) {
            float ms = 0;
            hipLaunchKernelGGL(k_bench_gather, dim3(blocks), dim3(256), 0, ctx->stream, (const u64 *)table, masks[v], 8u, out); // warm-up
            (void)hipEventRecord(e0, ctx->stream);
            hipLaunchKernelGGL(k_bench_gather, dim3(blocks), dim3(256), 0, ctx->stream, (const u64 *)table, masks[v], iters, out);
            (void)hipEventRecord(e1, ctx->stream);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                st = ks_fail(ctx, KS_ERR_HIP, "gather micro-benchmark failed");
            else
                gathers_per_s[v] = (double)blocks * 256 * iters * 4 / (ms * 1e-3);
        }
    }
    ks_pool_free(ctx, table); ks_pool_free(ctx, out);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return st;
    });
}

// ---- host-side pre-step: AminoAcidAmbiguity::validate_and_resolve, src/rust/aminoacid.rs:74-105 ----
static inline u64 splitmix64(u64 *s) {
    u64 z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

extern "C" int ks_validate_and_resolve(const uint8_t *seq, uint64_t len, int upper, uint64_t rng_seed,
                                       uint8_t *out, uint64_t *out_len, ks_residue_error *err) {
    return ks_guard(nullptr, [&]() -> int {
    if ((!seq && len) || !out || !out_len) return KS_ERR_INVALID_ARG;
    // class LUT: 0 invalid, 1 plain valid (20 standard + X U O), 2 stop, 3/4/5 = B/Z/J.  Called from many packer threads
    // at once (ks_ingest.cpp, ks_host.cpp): the table is a function-local static built by its initialiser (C++11
    // guarantees one thread runs it and the others wait), and never written afterwards.
    struct cls_table {
        u8 v[256];
        cls_table() {
            memset(v, 0, sizeof v);
            for (const char *p = "ACDEFGHIKLMNPQRSTVWYXUO"; *p; p++) v[(u8)*p] = 1; // aminoacid.rs:8-14
            v[(u8)'*'] = 2;
            v[(u8)'B'] = 3; v[(u8)'Z'] = 4; v[(u8)'J'] = 5; // aminoacid.rs:32-36
        }
    };
    static const cls_table cls_tab;
    const u8 *cls = cls_tab.v;
    u64 n = 0, rng = rng_seed, bits = 0;
    int nbits = 0;
    for (u64 i = 0; i < len; i++) {
        u8 c = seq[i];
        if (upper && c >= 'a' && c <= 'z') c = (u8)(c - 32); // index.rs:1000
        u8 k = cls[c];
        if (k == 2) { out[n++] = c; break; }                  // aminoacid.rs:79-83
        if (k == 0) {                                         // aminoacid.rs:85-87
            if (err) { err->seq_index = 0; err->position = (u32)(n + 1); err->residue = c; }
            *out_len = n;
            return KS_ERR_INVALID_RESIDUE;
        }
        if (k >= 3) {
            if (nbits == 0) { bits = splitmix64(&rng); nbits = 64; }
            int pick = (int)(bits & 1); bits >>= 1; nbits--;
            static const char *cand[3] = {"DN", "EQ", "IL"};
            c = (u8)cand[k - 3][pick];
        }
        out[n++] = c;
    }
    *out_len = n;
    return KS_OK;
    });
}
