// ks_common.h — internal definitions shared by the HIP translation units of libkmerseek_amd.
// gfx950 (MI355X) only: wave = 64 lanes, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/kmerseek_amd.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define KS_WAVE 64

// ---- device memory pool: grow-only, size classes with exact-fit reuse (ks_ctx.hip), so steady-state batches never hipMalloc ----
struct ks_pool_block {
    void *ptr;
    size_t size;
    bool in_use;
};

struct ks_timer_slot {
    hipEvent_t a, b;
    int name_id;
};

struct ks_copy_engine; // ks_copy.hip: pinned staging + host copy threads for pageable host buffers

// Diagnostic knobs (the KS_DEBUG_* environment variables: they force the rarely taken paths in the tests; results never
// depend on them).  They are read ONCE, into the context, when it is created — never on the per-call path, so a stray
// variable cannot switch kernels under a running service — and again only on request (ks_ctx_reload_debug_env: the tests
// switch paths on one context).
#define KS_DBG_LIST(X)                                                                                                   \
    X(STAGED_H2D) X(PLAIN_COPIES) X(PAIRS_LSD) X(MSD_LDS_CAP) X(SCAN_3PASS) X(INDEX_LSD) X(JOIN_FP) X(FP_COARSEN)        \
    X(PAIR_LIMIT) X(PBITS_MAX) X(UNPACKED_PAIRS) X(ONE_CURSOR) X(JOIN_SEGS) X(JOIN_SEG_CAP) X(JOIN_SPARSE) X(UNFUSED_ROWS) \
    X(NO_ROWS_HINT) X(ROWS_TICKET) X(FORCE_ROWS_TICKET_RETRY) X(FORCE_TICKET_RETRY) X(NO_PLAN) X(NO_COMPACT) X(SPAN)      \
    X(NO_PACK) X(PLAN_SYNC) X(TILE_R) X(OUT_CAP) X(POOL_CAP) X(THROW) X(QCAP) X(LOOKBACK_SKIP) X(SYNC_API) X(POSTINGS12) X(POSTINGS10) X(NO_DEFER) X(BUCKET) X(JOIN_SPLIT) X(SUBSHIFT)
enum ks_dbg_id {
#define KS_DBG_ENUM(n) KS_DBG_##n,
    KS_DBG_LIST(KS_DBG_ENUM)
#undef KS_DBG_ENUM
    KS_DBG_COUNT
};
struct ks_debug {
    bool set[KS_DBG_COUNT];
    char val[KS_DBG_COUNT][32];
};
void ks_debug_load(ks_debug *d); // ks_ctx.hip

struct ks_ctx {
    int device = 0;
    ks_copy_engine *copy = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cus = 256;
    ks_debug dbg;            // KS_DEBUG_* as they were when the context was created
    u64 pool_cap = 0;        // KS_DEBUG_POOL_CAP: the pool refuses to hold more device bytes than this (0 = no cap)
    std::string err;
    std::vector<ks_pool_block> pool;
    u64 pool_mallocs = 0; // hipMalloc calls made by the pool (0 in steady state)
    // timing
    int timing = 0; // 0 off, 1 every launch, 2 only the kernels that carry the bytes (lower event overhead)
    std::vector<std::string> t_names;
    std::vector<u64> t_launches;
    std::vector<double> t_ms;
    std::vector<ks_timer_slot> t_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> t_free;
    bool t_open = false; // the last ks_timer_begin recorded a start event
    // small pinned host scratch for counters read back from the device
    u64 *h_pin = nullptr; // KS_PIN_WORDS x u64; [KS_PIN_SKETCH, +32): the control block of a sketch whose read-back is pending
    // matched posting pairs of recent searches (+ slack): sizes the next search's match list so the join runs once
    u64 pair_cap_hint = 0;
    // hit rows of recent searches (+ slack): sizes the next search's row arrays so that the row count can be read with the
    // final synchronisation instead of a round trip of its own
    u64 rows_hint = 0;
    // the sketch tiles take their ids from blockIdx.x (dispatch order) until a look-back ever gives up on this context;
    // from then on from an atomic ticket (guaranteed order, one more memory round trip per tile)
    bool sketch_use_ticket = false;
    // how often a sketch batch had to be repeated: look-back gave up (-> ticket ids from then on), a compacting tile
    // overflowed its LDS lists (-> plain tiles for that batch), bounded outputs too small (-> window-count sized)
    u64 sketch_ticket_fallbacks = 0, sketch_compact_fallbacks = 0, sketch_cap_fallbacks = 0;
    bool rows_use_ticket = false; u64 rows_ticket_fallbacks = 0; // k_pair_rows_fused: dispatch-order tile ids until a look-back gives up
    u64 fused_deferred = 0, fused_redos = 0; // ks_sketch_search_device: calls that folded the sketch wait into the first wait of the search / were repeated plainly
    u64 join_retries = 0; // searches whose match list outgrew a segment and ran the join twice
    // single-launch scans (ks_prims.hip): status ring + ticket counter in device memory, never reset: every entry is
    // tagged with the global tile number that wrote it
    unsigned long long *scan_ring = nullptr;
    u32 *scan_ticket = nullptr;
    u32 scan_ticket_base = 0; // value the device counter will have when the next scan starts
    // encode LUTs (3 x 256 bytes) in device memory
    u8 *d_lut = nullptr;
    // ks_stream_wait: a pinned host word the stream's last kernel stamps, polled by the host (see ks_ctx.hip)
    unsigned long long *h_flag = nullptr;
    unsigned long long wait_seq = 0;
};

// The host waits for everything queued on the context's stream (what hipStreamSynchronize does) by polling a pinned word
// that a one-thread kernel at the end of the queue stamps: the wake-up of a blocking synchronisation costs 10 - 25 us of idle
// queue on this runtime, and a sketch + search step waits three times.  KS_DEBUG_SYNC_API = the runtime's call instead.
int ks_stream_wait(ks_ctx *ctx);
// ... and the same with the words the host wants to read afterwards: the stamping kernel writes them to pinned memory itself
// (instead of one device -> host copy dispatch per block before it).  dst must lie in pinned host memory (ctx->h_pin).
#define KS_FETCH_MAX 4
struct ks_fetch_seg {
    const void *src; // device
    u32 *dst;        // pinned host
    u32 rows, row_words, src_stride; // 32-bit words
};
static inline ks_fetch_seg ks_fetch_words(const void *src, void *dst, u32 n_words32) { return ks_fetch_seg{src, (u32 *)dst, 1u, n_words32, n_words32}; }
int ks_stream_wait_fetch(ks_ctx *ctx, const ks_fetch_seg *segs, int n);

// returns nullptr and sets ctx->err on failure
void *ks_pool_alloc(ks_ctx *ctx, size_t bytes);
void ks_pool_free(ks_ctx *ctx, void *ptr);
void ks_pool_trim(ks_ctx *ctx);

int ks_fail(ks_ctx *ctx, int status, const char *fmt, ...);
// value of a diagnostic knob as the context holds it, or nullptr (drop-in for getenv("KS_DEBUG_" #name))
static inline const char *ks_dbg(const ks_ctx *ctx, int id) { return (ctx && ctx->dbg.set[id]) ? ctx->dbg.val[id] : nullptr; }

// Every extern "C" entry point that can allocate host memory (new, std::vector / std::string growth in the context, the
// pool's block list, timers) runs its body inside this guard: a C++ exception never unwinds into the caller's frames
// (Rust, ctypes: that is std::terminate).  bad_alloc -> KS_ERR_OOM, anything else -> KS_ERR_HIP "internal error".
// Reference convention: every failure is a value (src/rust/errors.rs:8-24).
int ks_guard_fail(ks_ctx *ctx, int status, const char *what) noexcept;
void ks_guard_enter(ks_ctx *ctx); // KS_DEBUG_THROW (tests): throws the named exception from inside the guard
template <typename F>
static inline int ks_guard(ks_ctx *ctx, F &&body) noexcept {
    try {
        ks_guard_enter(ctx);
        return body();
    } catch (const std::bad_alloc &) {
        return ks_guard_fail(ctx, KS_ERR_OOM, "out of host memory (std::bad_alloc)");
    } catch (const std::exception &e) {
        return ks_guard_fail(ctx, KS_ERR_HIP, e.what());
    } catch (...) {
        return ks_guard_fail(ctx, KS_ERR_HIP, "unknown exception");
    }
}

void ks_timer_begin(ks_ctx *ctx, const char *name);
void ks_timer_end(ks_ctx *ctx);

#define KS_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ks_fail(ctx, KS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                                    \
    } while (0)

#define KS_TRY(expr)                        \
    do {                                    \
        int s__ = (expr);                   \
        if (s__ != KS_OK) return s__;       \
    } while (0)

// Launch on the context's stream, bracketed by HIP events when timing is enabled.
#define KS_LAUNCH(ctx, name, kernel, grid, block, ...)                                   \
    do {                                                                                 \
        ks_timer_begin(ctx, name);                                                       \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__); \
        ks_timer_end(ctx);                                                               \
        KS_HIP(ctx, hipGetLastError());                                                  \
    } while (0)

template <typename T>
static inline int ks_alloc(ks_ctx *ctx, T **out, size_t count) {
    void *p = ks_pool_alloc(ctx, (count ? count : 1) * sizeof(T));
    if (!p) return KS_ERR_OOM;
    *out = (T *)p;
    return KS_OK;
}

// ---- opaque objects ----
struct ks_sketches {
    ks_ctx *ctx;
    ks_params params;
    u32 n_seqs;
    u64 n_hashes;
    u64 n_windows;
    // CSR with slots: d_offsets[s] is where sequence s's SLOT starts — as long as the sequence's kept hashes, repeats included,
    // so that a tile knows its place as soon as it has hashed (ks_sketch.hip: the look-back runs on kept counts) — and the
    // sequence's sketch (ascending distinct hashes + abundances) is the first d_counts[s] entries of the slot.  gapped == false
    // (no sequence repeats a k-mer: every synthetic batch, most real ones at large k; or after ks_sketches_make_dense): slots and
    // runs coincide and d_offsets is the plain CSR of the C ABI.  Everything that reads the arrays as a plain CSR calls
    // ks_sketches_make_dense first (one gather pass, only when gapped); the fused postings of a query batch never do.
    u64 *d_offsets; // n_seqs + 1
    u64 *d_hashes;  // n_slots
    u32 *d_abunds;  // n_slots
    u32 *d_counts;  // n_seqs distinct hashes per sequence, or NULL (sketches from host arrays, unions: dense by construction)
    u64 n_slots;    // d_offsets[n_seqs]
    bool gapped;
    // optional, made by ks_sketch_queries_device: postings (hash, seq) already partitioned on hash bits
    // low 8 bits of the join prefix (ks_join_prefix) into part_regions fixed-capacity regions (region r holds part_len[r] records
    // starting at r * part_cap) — the first partition pass of a search against an index that joins on part_pbits bits
    u64 *part_keys;
    u32 *part_vals;
    u32 *part_len;  // device, [part_regions]
    u64 part_cap;
    u32 part_regions;
    int part_pbits;
    u32 part_K;     // ks_join_prefix multiplier the regions were cut with
    // each region is split into 2^part_sub_shift sub-regions, one per XCD of the sketch launch (segment r << shift | x
    // at (r << shift | x) * part_cap, part_len likewise): slices that are neighbours in memory were written through ONE L2
    u32 part_sub_shift;
    // part_s != 0: 10-byte postings.  Inside region r the 8 hash bits [part_s, part_s + 8) ARE r (scaled = 1: the join
    // prefix is a bit field of the hash), so the key column carries the low 8 bits of the sequence id there and part_vals is
    // a u16 column with the rest (sequence ids < 2^24): 10 instead of 12 bytes through the sketch write, the bucket scatter
    // (in and out) and the join's read.
    u32 part_s;
    // pending != 0 (ks_sketch_search_device): the launches are queued, the copy of the control block to
    // h_pin + KS_PIN_SKETCH too, but nobody has waited yet — n_hashes / n_windows are upper bounds until
    // ks_sketch_finish_pending has run behind a wait on the context's stream (the search's first wait)
    // the sketch call's control block (device): statistics | posting cursors (part_len points INTO it) | tile status words —
    // one allocation, one memset per call; freed with the object
    u64 *ctl_block;
    int pending;
    u64 *pend_stats; // the statistics words of ctl_block, read back by the caller's next wait
    u64 pend_out_cap;
    u32 pend_max_seq_len;
    int pend_planned;
};
#define KS_PIN_WORDS 256
#define KS_PIN_SKETCH 128

// one index posting as the join fetches it for a candidate match: one 16-byte load
struct __attribute__((aligned(16))) ks_post {
    u64 key;
    u32 tid;
    u32 abund;
};

// per join bucket: first key (the fingerprints count from it) and the multiplier of the in-LDS directory slot
// (floor(2^32 * JN_DIR / (largest fingerprint of the bucket + 1)), ks_search.hip)
struct __attribute__((aligned(16))) ks_bmeta {
    u64 base;
    u32 slot_mul;
    u32 pad;
};

struct ks_index {
    ks_ctx *ctx;
    ks_params params;
    u32 n_targets;
    u64 n_postings;
    u64 *d_keys;   // sorted hashes      } columns of the sort: what the join of a small / medium index reads;
    u32 *d_tids;   // target id          } released once d_fp / d_post are written when the index is big (fp_layout)
    u32 *d_abunds; // target abundance   }
    u32 *d_fp;     // per posting, in hash order: min((key - first key of its join bucket) >> (32 - pbits), 2^32 - 1) — the 4 bytes
                   // per posting the join streams; monotone inside a bucket
    ks_post *d_post; // per posting, in hash order: (key, tid, abundance)
    ks_bmeta *d_bmeta; // per join bucket (2^pbits entries)
    bool fp_layout;    // true: d_fp / d_post / d_bmeta are what the join reads; false: the sorted columns
    int fp_shift;      // 32 - pbits (KS_DEBUG_FP_COARSEN adds to it: coarser fingerprints, more false candidates — tests)
    u32 max_abund; // largest of them: how many low bits of a packed match record the abundance needs
    u64 *d_dir;    // join-bucket directory: d_dir[b] = first posting whose join prefix is >= b (2^pbits + 1 entries)
    int pbits;     // join prefix bits, a function of n_postings and the layout alone (ks_join_pbits)
};

struct ks_hits {
    ks_ctx *ctx;
    u64 n_hits;
    u64 n_pair_instances;
    int partition_path; // how the query postings reached their join buckets (see ks_hits_partition_path)
    int bucket_posting_bytes; // bytes per query posting inside the join buckets (12, 10 or 9; 0: no bucket scatter ran)
    u32 *d_qid, *d_tid, *d_isect;
    u64 *d_nw;
    // != NULL: d_nw and d_isect point INTO this block (with the row pass's status words: one allocation, one memset)
    u64 *d_block;
};

struct ks_kmerpos {
    ks_ctx *ctx;
    u64 n;
    u32 *d_seq, *d_start;
    u64 *d_hash;
};

// ---- device-wide primitives (ks_scan.hip, ks_sort.hip) ----
// exclusive scan of n u32 values into u64 (out[n] = total is also written: out has n+1 entries)
int ks_scan_u32_to_u64(ks_ctx *ctx, const u32 *in, u64 *out, u64 n);
// exclusive scan u32 -> u32 in place (n < 2^32 total); optionally writes the total to d_total
// The one-launch scans cannot report a look-back that gave up (bounded spin) by themselves: callers enqueue
// ks_scan_status_fetch before a stream synchronisation they do anyway and call ks_scan_status_check after it.
int ks_scan_status_fetch(ks_ctx *ctx);
int ks_scan_status_check(ks_ctx *ctx);
int ks_scan_u32_inplace(ks_ctx *ctx, u32 *data, u64 n, u32 *d_total);

// Stable LSD radix passes over (key u64, value V) records, one 8-bit digit at each listed shift.
// keys_in / vals_in are only read (they may be the caller's own data, or one of the scratch pairs);
// passes ping-pong between the scratch pairs (ka, va) and (kb, vb); *keys_out / *vals_out point at the
// result.  `tag` picks the kernel instantiation name ("radix_scatter.<tag>" in the timing table).
enum { KS_SORT_INDEX = 0, KS_SORT_QPART = 1, KS_SORT_PAIRS = 2 };
// optional segmented input of the FIRST pass: `regions` fixed-capacity regions, region r = len[r] records at r * cap
struct ks_rs_segments {
    const u32 *len; // device
    u64 cap;
    u32 regions;    // number of segments
    u32 sub_shift;  // partition digit of segment s = s >> sub_shift (sub-regions of one region share it)
};
int ks_radix_sort_u32(ks_ctx *ctx, int tag, const u64 *keys_in, const u32 *vals_in, u64 *ka, u32 *va, u64 *kb, u32 *vb,
                      u64 n, const int *shifts, int n_shifts, u64 **keys_out, u32 **vals_out,
                      const ks_rs_segments *seg = nullptr, u32 pfxK = 0);
// last partition pass of a search without a histogram: segmented input -> 2^pbits buckets of capacity bcap;
// bcur[bucket] ends as the bucket's record count, status[1] != 0 if some bucket overflowed
// keys only (the match list with the abundance packed under the ids)
int ks_radix_sort_keys(ks_ctx *ctx, int tag, const u64 *keys_in, u64 *ka, u64 *kb, u64 n, const int *shifts, int n_shifts,
                       u64 **keys_out);
// storage slot of join bucket (high digit d, region r): region-major, so the 2^(P-8) runs a scatter tile writes lie in
// one window of n_hi * bcap records (a handful of pages) instead of 256 * bcap records apart (one page each: 5 M
// UTCL1 translation misses per launch with the digit-major order, 0.02 M with this one — same run time, though)
#define KS_BSLOT(d, r, n_hi) ((r) * (n_hi) + (d))
int ks_bucket_scatter_u32(ks_ctx *ctx, const u64 *keys_in, const u32 *vals_in, const ks_rs_segments *seg, int shift,
                          u32 pfxK, u64 *bkeys, u32 *bvals, u32 *bcur, u32 bcap, unsigned long long *status, u32 n_hi, int vfmt);
// index build in three passes (two partition passes + in-LDS bucket sort); *overflowed = 1: use the LSD sort instead
int ks_index_sort_partitioned(ks_ctx *ctx, const u64 *keys_in, const u64 *vals_in, u64 n, u64 max_hash, u64 *okeys, u32 *otids,
                              u32 *oabunds, u32 *d_max_abund, int *overflowed);
// Match-list sort in three moves whatever the key width (ks_msd.hip): two exact MSD partition levels of 8 bits + in-LDS sort of the
// 65,536 buckets.  Sorts `ka` in place on key bits [lo_bit, lo_bit + nbits); kb = scratch.  *done = 0: not applicable (small list /
// narrow key / KS_DEBUG_PAIRS_LSD), the caller takes the LSD passes.
// segs != NULL: the list lies in ka in segs->n segments seg_cap records apart, count[s] of them filled (the join's segmented
// pair list): the first partition level reads it as it lies — no copy that makes it dense first
#define KS_MSD_MAX_SEGS 64
struct ks_msd_segs {
    u32 n;
    u32 tile_start[KS_MSD_MAX_SEGS + 1]; // first level-1 tile of every segment (the last entry: all tiles)
    u32 count[KS_MSD_MAX_SEGS];
    u64 seg_cap;
};
int ks_sort_pairs_msd(ks_ctx *ctx, u64 *ka, u64 *kb, u64 n, int lo_bit, int nbits, int *done, const ks_msd_segs *segs = nullptr);
int ks_radix_sort_u64(ks_ctx *ctx, int tag, const u64 *keys_in, const u64 *vals_in, u64 *ka, u64 *va, u64 *kb, u64 *vb,
                      u64 n, const int *shifts, int n_shifts, u64 **keys_out, u64 **vals_out);

// ---- pipelines (ks_sketch.hip, ks_search.hip) ----
// part_pbits > 0: also emit postings partitioned for a join on the top part_pbits hash bits
// allow_defer: the first attempt may return with ks_sketches::pending set (no wait at the end), see there
int ks_sketch_device_impl(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res,
                          u32 max_seq_len, const ks_params *p, int part_pbits, int part_fmt10, int allow_defer, ks_sketches **out);
// After a wait on the stream: the exact counts of a pending sketch.  *redo != 0: the launch has to be repeated the plain way
// (1 compacting tile overflowed, 2 bounded outputs too small, 3 look-back gave up, 4 postings dropped) — the caller frees S.
int ks_sketch_finish_pending(ks_sketches *S, int *redo);
// the fetch segment (ks_stream_wait_fetch) that brings a pending sketch's control block to h_pin + KS_PIN_SKETCH
ks_fetch_seg ks_sketch_pending_seg(const ks_sketches *S);
// the fetch segment of the one-launch scans' give-up flag (see ks_scan_status_check); false: nothing to fetch
bool ks_scan_status_seg(ks_ctx *ctx, ks_fetch_seg *out);
// bits of hash prefix the join against an index of n_postings uses (buckets of ~3k index postings, <= 16)
int ks_join_pbits(const ks_ctx *ctx, u64 n_postings, u64 per_bucket);
// multiplier of ks_join_prefix (ks_device.h) for a join on pbits prefix bits of hashes kept below max_hash
u32 ks_join_prefix_mul(int pbits, u64 max_hash);
int ks_kmerpos_tiles_launch(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, const ks_params *p, u32 *d_seq,
                            u32 *d_start, u64 *d_hash, u64 *n_out);
int ks_kmerpos_device_impl(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res,
                           const ks_params *p, ks_kmerpos **out);
int ks_index_build_impl(ks_ctx *ctx, const ks_sketches *t, ks_index **out);
int ks_search_impl(ks_ctx *ctx, const ks_index *ix, const ks_sketches *q, ks_hits **out, int *sketch_redo = nullptr);
int ks_union_impl(ks_ctx *ctx, const ks_sketches *in, ks_sketches **out);

int ks_check_params(ks_ctx *ctx, const ks_params *p);
// gapped slots -> plain CSR (see ks_sketches); no-op for dense sketches.  Enqueued on ctx's stream.
int ks_sketches_make_dense(ks_ctx *ctx, ks_sketches *s);

// ---- boundary copies (ks_copy.hip): one DMA for pinned host memory, double-buffered pinned staging + copy threads otherwise
int ks_copy_h2d(ks_ctx *ctx, void *dst_device, const void *src_host, size_t bytes); // enqueued; complete for pageable sources
int ks_copy_d2h(ks_ctx *ctx, void *dst_host, const void *src_device, size_t bytes); // returns when dst holds the data
void ks_copy_engine_destroy(ks_ctx *ctx);
