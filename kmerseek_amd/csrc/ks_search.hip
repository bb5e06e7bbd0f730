// ks_search.hip — inverted-index build and many-vs-many sketch search on gfx950.
//
// Replaces branchwater manysearch as called by do_manysearch(threshold=0, output_all=False)
// (src/python/kmerseek/search.py:125-141): for every (query, target) pair with >= 1 shared hash,
// intersect = |mins_q ∩ mins_t| and n_weighted = Σ target abundance over the shared hashes.
// The reference does |Q|x|T| pairwise sorted merges; here both sides become postings sorted by hash
// and are joined once (identical pair results, O(N log N) instead of O(|Q||T|(|q|+|t|))).
//
// Data layout in HBM (SoA, coalesced):
//   index  : keys u64[N_T] ascending, tids u32[N_T], abunds u32[N_T]
//   queries: postings (hash u64, qid u32) radix-PARTITIONED on the top P <= 16 hash bits only — the join
//            needs locality, not order: each query posting binary-searches its full 64-bit hash in the
//            LDS-staged index keys of its bucket.
//   matches: (qid<<32|tid, abund) appended through a wave-aggregated atomic cursor, radix-sorted on the
//            live id bits, then run-length reduced to COO sorted by (qid, tid).
#include "ks_device.h"

// one wave per sequence: value[j] = f(seq id) for every posting j of the sequence
__global__ __launch_bounds__(256) void k_fill_index_vals(const u64 *csr, const u32 *abunds, u32 n_seqs, u64 *vals) {
    const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seqs) return;
    const u64 b = csr[s], e = csr[s + 1];
    for (u64 j = b + (threadIdx.x & 63); j < e; j += 64) vals[j] = ((u64)abunds[j] << 32) | s;
}

__global__ __launch_bounds__(256) void k_fill_query_vals(const u64 *csr, u32 n_seqs, u32 *vals) {
    const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seqs) return;
    const u64 b = csr[s], e = csr[s + 1];
    for (u64 j = b + (threadIdx.x & 63); j < e; j += 64) vals[j] = s;
}

__global__ __launch_bounds__(256) void k_split_vals(const u64 *vals, u64 n, u32 *tids, u32 *abunds, u32 *max_abund) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 a = 0;
    if (i < n) {
        u64 v = vals[i];
        tids[i] = (u32)v;
        a = (u32)(v >> 32);
        abunds[i] = a;
    }
    for (int d = 32; d > 0; d >>= 1) {
        const u32 o = __shfl_down(a, d, 64);
        a = o > a ? o : a;
    }
    if ((threadIdx.x & 63) == 0 && a > *max_abund) atomicMax(max_abund, a); // (plain read first: almost every wave's maximum is already covered)
}

int ks_join_pbits(const ks_ctx *ctx, u64 n_postings, u64 per_bucket);
__global__ __launch_bounds__(256) void k_bucket_dir(const u64 *keys, u64 n, int pbits, u32 pfxK, u64 *dir);
__global__ __launch_bounds__(256) void k_index_finish(const u64 *keys, const u32 *tids, const u32 *abunds, const u64 *dir, u64 n, int pbits,
                                                      u32 pfxK, int fp_shift, u32 *fp, ks_post *post);
__global__ __launch_bounds__(256) void k_index_bmeta(const u64 *keys, const u64 *dir, u32 n_buckets, int fp_shift, ks_bmeta *bmeta);

#define JN_FP_PBITS 15 // indexes with this many join-prefix bits or more (> 50M postings) use the fingerprint layout (see the join)

int ks_index_build_impl(ks_ctx *ctx, const ks_sketches *t, ks_index **out) {
    if (!t || !out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_sketches_make_dense(ctx, const_cast<ks_sketches *>(t))); // (the sort reads the hashes as one dense array)
    ks_index *ix = new ks_index();
    memset(ix, 0, sizeof *ix);
    ix->ctx = ctx;
    ix->params = t->params;
    ix->n_targets = t->n_seqs;
    ix->n_postings = t->n_hashes;
    const u64 n = t->n_hashes;
    u64 *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    u32 *d_max = nullptr;
    int st = KS_OK;
#define IX_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
#define IX_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
    IX_CHECK(ks_alloc(ctx, &v0, (size_t)n));
    IX_CHECK(ks_alloc(ctx, &ix->d_tids, (size_t)n));
    IX_CHECK(ks_alloc(ctx, &ix->d_abunds, (size_t)n));
    IX_CHECK(ks_alloc(ctx, &d_max, 1));
    IX_HIP(hipMemsetAsync(d_max, 0, sizeof(u32), ctx->stream));
    if (n > 0) {
        // values = (abund << 32 | tid); keys are read straight from the sketches
        ks_timer_begin(ctx, "fill_index_vals");
        hipLaunchKernelGGL(k_fill_index_vals, dim3((t->n_seqs + 3) / 4), dim3(256), 0, ctx->stream, (const u64 *)t->d_offsets,
                           (const u32 *)t->d_abunds, t->n_seqs, v0);
        ks_timer_end(ctx);
        IX_HIP(hipGetLastError());
        // three passes: partition on the sort prefix twice, sort the buckets in LDS (ks_prims.hip)
        int overflowed = 0;
        IX_CHECK(ks_alloc(ctx, &k0, (size_t)n));
        if (!ks_dbg(ctx, KS_DBG_INDEX_LSD))
            IX_CHECK(ks_index_sort_partitioned(ctx, t->d_hashes, v0, n, ks_max_hash(t->params.scaled), k0, ix->d_tids, ix->d_abunds, d_max,
                                               &overflowed));
        else
            overflowed = 1;
        if (!overflowed) {
            ix->d_keys = k0; k0 = nullptr;
        } else {
            // skewed hashes (a fixed capacity did not hold): the always-correct 8-pass LSD sort
            IX_CHECK(ks_alloc(ctx, &k1, (size_t)n));
            IX_CHECK(ks_alloc(ctx, &v1, (size_t)n));
            const int shifts[8] = {0, 8, 16, 24, 32, 40, 48, 56};
            u64 *ks = nullptr, *vs = nullptr;
            // v0 holds the input values, so the first pass must land in (k1, v1): pass it as the "a" pair
            IX_CHECK(ks_radix_sort_u64(ctx, KS_SORT_INDEX, t->d_hashes, v0, k1, v1, k0, v0, n, shifts, 8, &ks, &vs));
            IX_HIP(hipMemsetAsync(d_max, 0, sizeof(u32), ctx->stream));
            ks_timer_begin(ctx, "split_vals");
            hipLaunchKernelGGL(k_split_vals, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u64 *)vs, n, ix->d_tids, ix->d_abunds, d_max);
            ks_timer_end(ctx);
            IX_HIP(hipGetLastError());
            if (ks != k0 && ks != k1) { // (n == 1: nothing to sort, the keys are still the sketch's own array — found by the fuzz campaign)
                IX_HIP(hipMemcpyAsync(k0, ks, (size_t)n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
                ks = k0;
            }
            if (ks == k0) { ix->d_keys = k0; k0 = nullptr; } else { ix->d_keys = k1; k1 = nullptr; }
        }
        IX_HIP(hipMemcpyAsync(ctx->h_pin, d_max, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        IX_HIP(hipStreamSynchronize(ctx->stream));
        ix->max_abund = *(u32 *)ctx->h_pin;
    } else {
        IX_CHECK(ks_alloc(ctx, &k0, 1));
        ix->d_keys = k0; k0 = nullptr;
    }
    {
        // The layout of the join first (it sizes the buckets).  Indexes are joined on 4-byte fingerprints (streamed; a directory
        // over them in LDS instead of a binary search) + 16-byte postings (fetched per candidate match).  Medium and small hp
        // indexes keep the sorted columns (k_join_buckets_keys): with a two-letter alphabet most queries match, several times
        // (200k x 200k hp k = 24: 49 M matches from 11 M query hashes), and the confirmation fetches then cost more than the
        // searches save (0.885 vs 0.845 ms there; protein / dayhoff: 100k x 100k k = 10 join 0.262 -> 0.194 ms, configs[1]
        // 35 -> 29 us, configs[2] 56 -> 46 us).  KS_DEBUG_JOIN_FP = 1 / 0 forces either (big indexes: always fingerprints).
        const bool big = (n >> (JN_FP_PBITS - 1)) > 3072; // >= 2^JN_FP_PBITS buckets of ~3k postings (> 50 M postings)
        ix->fp_layout = big || t->params.moltype != KS_HP;
        if (const char *f = ks_dbg(ctx, KS_DBG_JOIN_FP)) ix->fp_layout = atoi(f) != 0 || big;
        // the join-bucket directory belongs to the index (its prefix width depends on the posting count alone): buckets of ~3k
        // postings, one LDS stage.  (Buckets of ~768 for the key-column join: 200k x 200k hp — 49 M matches, a kernel that is the
        // emission of its records — 0.79 -> 0.47 ms, but 50k x 50k hp k = 32 — 0.4 M matches — 0.135 -> 0.415 ms: how many
        // workgroups a bucket deserves depends on the matches, which the search knows and the index does not: see `split` there.)
        ix->pbits = ks_join_pbits(ctx, n, 3072);
        const u32 nb = 1u << ix->pbits;
        const u32 K = ks_join_prefix_mul(ix->pbits, ks_max_hash(t->params.scaled));
        IX_CHECK(ks_alloc(ctx, &ix->d_dir, (size_t)nb + 1));
        ks_timer_begin(ctx, "bucket_dir");
        hipLaunchKernelGGL(k_bucket_dir, dim3((nb + 256) / 256), dim3(256), 0, ctx->stream, (const u64 *)ix->d_keys, n, ix->pbits, K, ix->d_dir);
        ks_timer_end(ctx);
        IX_HIP(hipGetLastError());
        if (ix->fp_layout) {
        IX_CHECK(ks_alloc(ctx, &ix->d_fp, (size_t)(n ? n : 1)));
        IX_CHECK(ks_alloc(ctx, &ix->d_post, (size_t)(n ? n : 1)));
        ix->fp_shift = 32 - ix->pbits;
        if (ks_dbg(ctx, KS_DBG_FP_COARSEN)) {
            ix->fp_shift += atoi(ks_dbg(ctx, KS_DBG_FP_COARSEN));
            if (ix->fp_shift > 63) ix->fp_shift = 63;
        }
        IX_CHECK(ks_alloc(ctx, &ix->d_bmeta, (size_t)nb));
        hipLaunchKernelGGL(k_index_bmeta, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, (const u64 *)ix->d_keys, (const u64 *)ix->d_dir,
                           nb, ix->fp_shift, ix->d_bmeta);
        IX_HIP(hipGetLastError());
        if (n > 0) {
            ks_timer_begin(ctx, "index_finish");
            hipLaunchKernelGGL(k_index_finish, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u64 *)ix->d_keys,
                               (const u32 *)ix->d_tids, (const u32 *)ix->d_abunds, (const u64 *)ix->d_dir, n, ix->pbits, K, ix->fp_shift, ix->d_fp, ix->d_post);
            ks_timer_end(ctx);
            IX_HIP(hipGetLastError());
        }
        }
    }
    IX_CHECK(ks_scan_status_fetch(ctx));
    IX_HIP(hipStreamSynchronize(ctx->stream));
    IX_CHECK(ks_scan_status_check(ctx));
    if (ix->fp_layout) { // the sorted columns were the input of the finish pass only
        ks_pool_free(ctx, ix->d_keys); ks_pool_free(ctx, ix->d_tids); ks_pool_free(ctx, ix->d_abunds);
        ix->d_keys = nullptr; ix->d_tids = nullptr; ix->d_abunds = nullptr;
    }
done:
    ks_pool_free(ctx, k0); ks_pool_free(ctx, k1); ks_pool_free(ctx, v0); ks_pool_free(ctx, v1); ks_pool_free(ctx, d_max);
    if (st != KS_OK) { (void)hipStreamSynchronize(ctx->stream); ks_index_free(ix); return st; }
    *out = ix;
    return KS_OK;
#undef IX_CHECK
#undef IX_HIP
}

// ---------------------------------------------------------------------------------------------
// join
//
// Murmur output is uniform, so the top P bits of the hash cut both posting lists into 2^P buckets of
// near-equal size.  Query postings are only PARTITIONED on those bits (ceil(P/8) radix passes instead
// of a 64-bit sort); the index is fully sorted once at build time.  One workgroup joins one bucket:
// the bucket's index keys are staged in LDS (coalesced 8-B loads), every query posting of the bucket
// binary-searches them there, and each match is appended to the pair list through an atomic cursor.
// Buckets larger than the LDS stage (heavy duplicate hashes) are walked in chunks.
// ---------------------------------------------------------------------------------------------
#ifndef JN_THREADS
#define JN_THREADS 512
#endif
#ifndef JN_E
#define JN_E 10
#endif
//                     JN_E query postings per thread per round (10: one round covers the ~4.5k queries of a bucket of the
//                     1M-vs-1M workload with fewer registers than 12; 9 needs a second round for half of the buckets)
// most records one match list may hold (32-bit offsets in the sort and the reduce); a debug override makes the
// slicing path testable on small inputs
#define KS_PAIR_LIMIT (ks_dbg(ctx, KS_DBG_PAIR_LIMIT) ? strtoull(ks_dbg(ctx, KS_DBG_PAIR_LIMIT), nullptr, 10) : 0xfffffff0ULL)
#ifndef JN_WLIST
#define JN_WLIST 128
#endif
#ifndef JN_FILLU
#define JN_FILLU 10 // index fingerprints per thread and staging round (fingerprint join): 5,120 per round — a bucket of ~4.5k in ONE round trip
#endif
#ifndef JK_FILLU
#define JK_FILLU 6  // index keys (64 bits) per thread and staging round (key-column join)
#endif
#ifndef JN_CAP
#define JN_CAP 6144
#endif
//                          index keys staged per chunk: 48 KiB of LDS -> 3 workgroups (24 waves) per CU

static int bits_for_value(u32 v) { // bits needed to represent the value v itself (>= 1)
    int b = 1;
    while (b < 32 && (v >> b)) b++;
    return b;
}

static int bits_for(u32 n) { // bits needed to represent ids 0..n-1
    int b = 0;
    while (b < 32 && (n > (1u << b))) b++;
    return b < 1 ? 1 : b;
}

int ks_join_pbits(const ks_ctx *ctx, u64 n_postings, u64 per) { // buckets of ~`per` index postings; the query side is partitioned on the same bits
    // (8 bits by the sketch kernel + up to 9 by the bucket scatter; KS_DEBUG_PBITS_MAX is a tuning aid.  An index keeps the
    // width it was built with: ks_index::pbits)
    int cap = 16;
    if (const char *f = ks_dbg(ctx, KS_DBG_PBITS_MAX)) { const int v = atoi(f); cap = v < 1 ? 1 : (v > 17 ? 17 : v); }
    int pbits = 0;
    if (const char *f = ks_dbg(ctx, KS_DBG_BUCKET)) { const long v = atol(f); if (v >= 64) per = (u64)v; } // (tuning aid)
    while (pbits < cap && (n_postings >> pbits) > per) pbits++;
    return pbits;
}

u32 ks_join_prefix_mul(int pbits, u64 max_hash) {
    const u64 den = (max_hash >> 32) + 1ULL;
    const u64 K = pbits >= 32 ? 0xffffffffULL : ((1ULL << (pbits + 32)) / den);
    return K > 0xffffffffULL ? 0xffffffffu : (u32)K; // smaller only coarsens (still < 2^pbits, still monotone)
}

// query-side bucket bounds when the sketch kernel's regions ARE the buckets (pbits <= 8): no partition pass at all
__global__ __launch_bounds__(256) void k_region_dir(const u32 *len, u64 cap, u32 n_regions, u64 *lo, u64 *hi, u32 n_hi) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_regions) return;
    // n_hi != 0: b is a join prefix (high digit << 8 | region) and its postings sit in storage slot KS_BSLOT
    const u32 slot = n_hi ? KS_BSLOT(b >> 8, b & 255u, n_hi) : b;
    // (a cursor keeps counting past the capacity when a region / bucket overflows — the host then repartitions — but the join
    // that is already queued behind this kernel must not walk into the neighbour, or past the end of the array)
    const u64 n = len[slot];
    lo[b] = (u64)slot * cap;
    hi[b] = (u64)slot * cap + (n < cap ? n : cap);
}

// dir[b] = first posting whose join prefix is >= b, for b in [0, 2^pbits]; keys are ordered on the prefix
__global__ __launch_bounds__(256) void k_bucket_dir(const u64 *keys, u64 n, int pbits, u32 pfxK, u64 *dir) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nb = 1u << pbits;
    if (b > nb) return;
    if (b == nb) { dir[b] = n; return; }
    u64 lo = 0, hi = n;
    while (lo < hi) {
        u64 mid = lo + ((hi - lo) >> 1);
        u32 pre = pbits ? ks_join_prefix(keys[mid], pfxK) : 0;
        if (pre < b) lo = mid + 1; else hi = mid;
    }
    dir[b] = lo;
}

// ---- the join of small and medium indexes (fewer than 2^JN_FP_PBITS buckets): full 64-bit keys staged in LDS, one cursor.
// With few buckets there is little to gain from a narrower stream, and searches that match most of their queries
// (all-vs-all) are better off without the confirmation fetches of the fingerprint kernel below.
#ifndef JN_CAP_KEYS
#define JN_CAP_KEYS 6144 // 48 KiB of LDS -> 3 workgroups (24 waves) per CU
#endif
#ifndef JK_WLIST
#define JK_WLIST JN_WLIST // pairs a wave lists per round before it falls back to per-lane emission
#endif
// number of staged keys < h.  Branchless halving on the ACTUAL bucket size: the trip count ceil(log2 n) is uniform
// across the workgroup, and the probe positions are multiples of n/2, n/4, ... rather than of powers of two —
// power-of-two probe strides put every lane of a wave on ONE LDS bank (measured: 86 % of the LDS cycles of this
// kernel were bank-conflict cycles with the 4096/2048/... ladder).
KS_DEV u32 jn_lower_bound_keys(const u64 *lk, u32 n, u64 h) {
    u32 base = 0, len = n;
    while (len > 1) { // uniform: n is the same for every lane
        const u32 half = len >> 1;
        base = (lk[base + half - 1] < h) ? base + half : base;
        len -= half;
    }
    return base + (lk[base] < h ? 1u : 0u);
}

// cursor[0] = matches appended so far (keeps counting past `cap` so the host can size a retry).
// Per round of JN_THREADS*JN_E query postings: search once, keep (position, run length) in registers,
// reserve the round's slice of the pair list with ONE device-wide atomic, then write from registers.
__global__ __launch_bounds__(JN_THREADS) void k_join_buckets_keys(const u64 *qkeys, const u32 *qids, const u64 *ikeys,
                                                             const u32 *itids, const u32 *iabunds, const u64 *q_lo,
                                                             const u64 *q_hi, const u64 *dir_t, u64 *pair_keys,
                                                             u32 *pair_vals, u64 cap, unsigned long long *cursor,
                                                             int tbits, int abits, u32 split) {
    __shared__ u64 lk[JN_CAP_KEYS];
    __shared__ u32 wlist[JN_THREADS / 64][JK_WLIST]; // per-wave list of the round's pairs (query | index posting << 13)
    __shared__ u32 scan_smem[JN_THREADS / 64 + 1];
    __shared__ unsigned long long base_s;
    const u32 tid = threadIdx.x;
    // `split` workgroups share a bucket: each stages the bucket's keys and joins its slice of the bucket's query postings.  A
    // match-dense search (all-vs-all over a two-letter alphabet: 4 matches per query posting) is the emission of its records,
    // a chain of memory latencies per lane: more, shorter workgroups per bucket hide it (the host picks `split` from the
    // matches of the context's previous search).
    const u32 bucket = split > 1 ? blockIdx.x / split : blockIdx.x, part = split > 1 ? blockIdx.x % split : 0u;
    u64 qs = q_lo[bucket], qe = q_hi[bucket]; // dense postings: q_hi = q_lo + 1 (a directory)
    const u64 ts = dir_t[bucket], te = dir_t[bucket + 1];
    if (qs == qe || ts == te) return;
    if (split > 1) {
        const u64 per = (qe - qs + split - 1) / split;
        const u64 a = qs + per * part;
        qe = a + per < qe ? a + per : qe;
        qs = a;
        if (qs >= qe) return;
    }
    for (u64 c0 = ts; c0 < te; c0 += JN_CAP_KEYS) {
        const u32 n = (u32)((te - c0) < JN_CAP_KEYS ? (te - c0) : JN_CAP_KEYS);
        // JK_FILLU loads of a thread are in flight before their LDS stores (a plain loop waits out one memory latency
        // per 512 keys; all JN_CAP / JN_THREADS at once costs the registers of a third workgroup per CU)
        for (u32 i0 = 0; i0 < n; i0 += JN_THREADS * JK_FILLU) {
            u64 kk[JK_FILLU];
#pragma unroll
            for (int j = 0; j < JK_FILLU; j++) {
                const u32 i = i0 + (u32)j * JN_THREADS + tid;
                kk[j] = i < n ? ikeys[c0 + i] : 0;
            }
#pragma unroll
            for (int j = 0; j < JK_FILLU; j++) {
                const u32 i = i0 + (u32)j * JN_THREADS + tid;
                if (i < n) lk[i] = kk[j];
            }
        }
        __syncthreads();
        for (u64 q0 = qs; q0 < qe; q0 += (u64)JN_THREADS * JN_E) {
            u64 h[JN_E];
            u32 info[JN_E]; // position | run length << 16
#pragma unroll
            for (int e = 0; e < JN_E; e++) {
                const u64 i = q0 + (u64)e * JN_THREADS + tid;
                h[e] = i < qe ? qkeys[i] : 0;
            }
            u32 mine = 0, hit = 0; // hit: bit e set = query e of this thread matched
#pragma unroll
            for (int e = 0; e < JN_E; e++) {
                const u64 i = q0 + (u64)e * JN_THREADS + tid;
                info[e] = 0;
                // (uniform: a bucket with few query postings — small batches, query shards of a strong-scaling run — fills
                // only the first slots of the round, and an empty slot would cost the same 13 LDS probes as a full one)
                if (q0 + (u64)e * JN_THREADS >= qe) continue;
                u32 lo = jn_lower_bound_keys(lk, n, h[e]);
                u32 c = 0;
                if (i < qe)
                    while (lo + c < n && lk[lo + c] == h[e]) c++;
                info[e] = lo | (c << 16);
                mine += c;
                hit |= c ? (1u << e) : 0u;
            }
            u32 total;
            const u32 off = ks_block_excl_scan(mine, scan_smem, &total);
            if (total) { // uniform
                if (tid == 0) base_s = atomicAdd(cursor, (unsigned long long)total); // (64 sharded cursors measured no faster)
                __syncthreads();
                // Matches are sparse (a few per cent of the queries) and sit anywhere among a thread's JN_E queries: walking
                // e = 0..JN_E-1 would run JN_E rounds of memory operations with a handful of lanes active in each.  Instead
                // every thread walks ITS OWN matches (lowest e first: slot order unchanged), and — when the wave's matches
                // fit its LDS list — only to LIST them: one 4-byte entry (query, index posting) per pair at the pair's
                // position inside the wave's slice, so that afterwards lane k emits pair k: full lanes, contiguous stores.
                const u32 lane = tid & 63u, wave = tid >> 6;
                const u32 wbase = __shfl(off, 0, 64);                           // first pair of this wave inside the round
                const u32 wtotal = __shfl(off + mine, 63, 64) - wbase;          // pairs of this wave
                if (wtotal <= JK_WLIST) { // uniform per wave
                    u32 p = off - wbase;
                    while (hit) {
                        const int e = __builtin_ctz(hit);
                        hit &= hit - 1;
                        u32 inf = info[0];
#pragma unroll
                        for (int k = 1; k < JN_E; k++) inf = e == k ? info[k] : inf; // (register array: select, no indexing)
                        const u32 c = inf >> 16, qi = (u32)e * JN_THREADS + tid, pos = inf & 0xffffu;
                        for (u32 j = 0; j < c; j++) wlist[wave][p++] = qi | ((pos + j) << 13);
                    }
                    __builtin_amdgcn_wave_barrier(); // same wave: LDS operations execute in order
                    for (u32 k = lane; k < wtotal; k += 64) {
                        const u32 en = wlist[wave][k];
                        const u64 slot = base_s + wbase + k;
                        if (slot < cap) {
                            const u32 q = qids[q0 + (en & 0x1fffu)];
                            const u64 jp = c0 + (en >> 13);
                            const u64 ids = ((u64)q << tbits) | itids[jp]; // ids packed tight: fewer sort passes
                            if (pair_vals) { pair_keys[slot] = ids; pair_vals[slot] = iabunds[jp]; }
                            else pair_keys[slot] = (ids << abits) | iabunds[jp]; // one 8-byte record per match
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                } else { // a hash shared by many targets: emit straight from the registers
                    u64 slot = base_s + off;
                    while (hit) {
                        const int e = __builtin_ctz(hit);
                        hit &= hit - 1;
                        u32 inf = info[0];
#pragma unroll
                        for (int k = 1; k < JN_E; k++) inf = e == k ? info[k] : inf;
                        const u32 c = inf >> 16;
                        const u32 q = qids[q0 + (u64)e * JN_THREADS + tid];
                        const u64 j0 = c0 + (inf & 0xffffu);
                        // (four postings of the run at a time: their id / abundance loads are in flight together — an
                        // all-vs-all emits ~20 records per lane and round, and one memory latency per record was the kernel)
                        for (u32 j = 0; j < c; j += 4) {
                            u32 tt[4], aa[4];
#pragma unroll
                            for (u32 u = 0; u < 4; u++) {
                                const bool on = j + u < c && slot + u < cap;
                                tt[u] = on ? itids[j0 + j + u] : 0u;
                                aa[u] = on ? iabunds[j0 + j + u] : 0u;
                            }
#pragma unroll
                            for (u32 u = 0; u < 4; u++)
                                if (j + u < c && slot + u < cap) {
                                    const u64 ids = ((u64)q << tbits) | tt[u];
                                    if (pair_vals) { pair_keys[slot + u] = ids; pair_vals[slot + u] = aa[u]; }
                                    else pair_keys[slot + u] = (ids << abits) | aa[u];
                                }
                            slot += (c - j) < 4u ? (c - j) : 4u;
                        }
                    }
                }
                __syncthreads(); // base_s is rewritten next round
            }
        }
        __syncthreads();
    }
}

// ---- the join of big indexes: 32-bit fingerprints streamed, candidates confirmed on 16-byte postings
// Fingerprint of a key inside its join bucket (first key `base`): monotone in the key, 32 bits.  A bucket spans at most
// 2^(64 - pbits) hash values, so with shift = 32 - pbits the clamp never bites for scaled = 1; it keeps the function total
// (and monotone) whatever the prefix rounding does for other `scaled`.  Equal fingerprints are CANDIDATES: the join confirms
// them on the full key.
KS_DEV u32 jn_fingerprint(u64 key, u64 base, int shift) {
    const u64 f = (key - base) >> shift;
    return f > 0xffffffffULL ? 0xffffffffu : (u32)f;
}

__global__ __launch_bounds__(256) void k_index_finish(const u64 *keys, const u32 *tids, const u32 *abunds, const u64 *dir, u64 n, int pbits,
                                                      u32 pfxK, int fp_shift, u32 *fp, ks_post *post) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 key = keys[i];
    const u32 b = pbits ? ks_join_prefix(key, pfxK) : 0u;
    const u64 base = keys[dir[b]]; // (the same two cached words for the ~4k consecutive postings of a bucket)
    fp[i] = jn_fingerprint(key, base, fp_shift);
    ks_post p;
    p.key = key; p.tid = tids[i]; p.abund = abunds[i];
    post[i] = p;
}

#ifndef JN_DIR
#define JN_DIR 4096
#endif
#ifndef JN_PROBES
#define JN_PROBES 4 // entries of a directory slot a query looks at without a loop (more in the slot: the searched path)
#endif
#define JN_SEGS 64       // most segments (cursors) of the pair list
#define JN_CUR_STRIDE 64 // u64 words between two cursors
// Diagnostic build only (-DJN_STAMP): per-phase shader-clock shares of k_join_buckets, summed over workgroups by thread 0
// (tools/microbench.py --variants JN_STAMP).  Never compiled into the shipped library.
#ifdef JN_STAMP
#define JN_STAMP_SLOTS 4096
__device__ unsigned long long jn_stamp_acc[JN_STAMP_SLOTS][16];
#define JN_STAMP_AT(i) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); \
    atomicAdd(&jn_stamp_acc[blockIdx.x % JN_STAMP_SLOTS][i], t_ - jn_t_prev); jn_t_prev = clock64(); } } while (0)
extern "C" void ks_debug_read_join_stamps(unsigned long long *out, int reset) {
    static unsigned long long host[JN_STAMP_SLOTS][16];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(jn_stamp_acc), sizeof host);
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int s = 0; s < JN_STAMP_SLOTS; s++) out[i] += host[s][i]; }
    if (reset) { memset(host, 0, sizeof host); (void)hipMemcpyToSymbol(HIP_SYMBOL(jn_stamp_acc), host, sizeof host); }
}
#else
#define JN_STAMP_AT(i) do { } while (0)
#endif
// directory slot of a fingerprint: floor(f * JN_DIR / (largest staged fingerprint + 1)), capped; monotone in f
KS_DEV u32 jn_slot_mul(u32 largest) {
    const u64 m = ((u64)JN_DIR << 32) / ((u64)largest + 1ULL);
    return m > 0xffffffffULL ? 0xffffffffu : (u32)m;
}
KS_DEV u32 jn_slot(u32 f, u32 mul) {
    const u32 j = __umulhi(f, mul);
    return j < (u32)JN_DIR - 1u ? j : (u32)JN_DIR - 1u;
}

// per join bucket: what the join kernel needs besides the directory words
__global__ __launch_bounds__(256) void k_index_bmeta(const u64 *keys, const u64 *dir, u32 n_buckets, int fp_shift, ks_bmeta *bmeta) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const u64 ts = dir[b], te = dir[b + 1];
    ks_bmeta m;
    m.base = 0; m.slot_mul = 0; m.pad = 0;
    if (ts < te) {
        m.base = keys[ts];
        m.slot_mul = jn_slot_mul(jn_fingerprint(keys[te - 1], m.base, fp_shift));
    }
    bmeta[b] = m;
}

// number of staged fingerprints < f.  Branchless halving on the ACTUAL bucket size: the trip count ceil(log2 n) is uniform
// across the workgroup, and the probe positions are multiples of n/2, n/4, ... rather than of powers of two —
// power-of-two probe strides put every lane of a wave on ONE LDS bank (measured: 86 % of the LDS cycles of this
// kernel were bank-conflict cycles with the 4096/2048/... ladder).
KS_DEV u32 jn_lower_bound_lds(const u32 *lk, u32 n, u32 f) {
    u32 base = 0, len = n;
    while (len > 1) { // uniform: n is the same for every lane
        const u32 half = len >> 1;
        base = (lk[base + half - 1] < f) ? base + half : base;
        len -= half;
    }
    return base + (lk[base] < f ? 1u : 0u);
}

// A candidate run [p0, p0 + c) of equal fingerprints, confirmed on the full keys (sorted): the postings whose key IS h.
// Almost always c == 1 and the key matches; distinct keys under one fingerprint (2^-32-ish per probe) take the searches.
KS_DEV void jn_confirm_slow(const ks_post *post, u32 p0, u32 c, u64 h, u32 *first, u32 *count) {
    u32 lo = 0, hi = c; // first key >= h
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (post[p0 + m].key < h) lo = m + 1; else hi = m; }
    u32 lo2 = lo, hi2 = c; // first key > h
    while (lo2 < hi2) { const u32 m = (lo2 + hi2) >> 1; if (post[p0 + m].key <= h) lo2 = m + 1; else hi2 = m; }
    *first = lo; *count = lo2 - lo;
}

// Query postings as the join reads them.  12-byte form (F10 = 0): (hash u64, sequence id u32).  10-byte form (F10 = 1, see
// ks_sketches::part_s): in join bucket b the hash bits [s, s + 8), s = 32 + fp_shift, are b's low byte, so the key column
// carries the low 8 bits of the sequence id there and the value column is 16 bits wide.  The field sits right above the 32 bits
// the fingerprint is made of, and every index posting of the bucket holds b's byte there: fingerprints are taken from the raw
// word and a candidate is confirmed on the other 56 bits — the 10-byte form costs no instruction per posting, one AND per
// candidate and one extra key read per match.
// 9-byte form (F10 = 2; behind the bucket scatter of a join on 16 prefix bits): in join bucket b ALL 16 prefix bits [48, 64) are
// b, so the scatter moves the second byte of the sequence id into the high digit's field too and the value column shrinks to
// 8 bits: 9 instead of 10 bytes out of the scatter and into the join (0.59 GB less per 1M x 1M step).
template <int F10> struct jn_qfmt {
    u64 keep, fill;
    u32 s;
    KS_DEV jn_qfmt(int fp_shift, u32 bucket)
        : keep(F10 == 2 ? ~(0xffffULL << 48) : (F10 ? ~(0xffULL << (32 + fp_shift)) : ~0ULL)),
          fill(F10 == 2 ? (u64)(bucket & 0xffffu) << 48 : (F10 ? (u64)(bucket & 0xffu) << (32 + fp_shift) : 0ULL)), s(32u + (u32)fp_shift) {}
    KS_DEV u64 hash(u64 raw) const { return F10 ? ((raw & keep) | fill) : raw; }
    // fingerprint of a raw key word inside the bucket whose first key is `base`
    KS_DEV u32 fp(u64 raw, u64 base, int shift) const { return F10 ? (u32)((raw - base) >> shift) : jn_fingerprint(raw, base, shift); }
    // can the raw word match anything in the bucket?  (10-byte form: a word below the base only yields a false candidate)
    KS_DEV bool in_range(u64 raw, u64 base) const { return F10 ? true : raw >= base; }
    KS_DEV bool same(u64 post_key, u64 raw) const { return F10 ? (((post_key ^ raw) & keep) == 0) : post_key == raw; }
    KS_DEV u32 qid(const u64 *qk, const u32 *qv, u32 i) const {
        if (F10 == 2) return ((u32)(qk[i] >> 48) & 0xffffu) | ((u32)((const u8 *)qv)[i] << 16);
        return F10 ? (((u32)(qk[i] >> s) & 0xffu) | ((u32)((const u16 *)qv)[i] << 8)) : qv[i];
    }
    // the value column from posting q0 on (its entries are 4, 2 or 1 bytes wide)
    KS_DEV const u32 *vals_at(const u32 *qv, u64 q0) const {
        return F10 == 2 ? (const u32 *)((const u8 *)qv + q0) : (F10 ? (const u32 *)((const u16 *)qv + q0) : qv + q0);
    }
};

// cursor[0] = matches appended so far (keeps counting past `cap` so the host can size a retry).
// Per round of JN_THREADS*JN_E query postings: every thread searches the staged fingerprints for its queries (LDS only);
// the candidates of a wave — a few per cent of its queries, anywhere among a thread's JN_E slots — are LISTED in LDS
// (query slot, index posting, full hash), so that lane k confirms and emits candidate k: one 16-byte posting fetch and one
// qid fetch per candidate in full lanes, the confirmed ones ranked by a wave scan (segmented pair list: one reservation per
// wave on the segment's cursor) or a block scan (one cursor: one reservation per workgroup round), contiguous stores.
template <int F10>
__global__ __launch_bounds__(JN_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_join_buckets(
    const u64 *qkeys, const u32 *qids, const u32 *ifp, const ks_post *ipost, const ks_bmeta *bmeta, const u64 *q_lo, const u64 *q_hi,
    const u64 *dir_t, u64 *pair_keys, u32 *pair_vals, u64 cap, unsigned long long *cursors, u32 seg_mask, int tbits, int abits,
    int fp_shift) {
    __shared__ u32 lk[JN_CAP + JN_PROBES]; // (+ slack: the probes of a slot read JN_PROBES entries whatever it holds)
    __shared__ unsigned short ldir[JN_DIR + 2]; // ldir[j] = staged fingerprints whose slot (jn_slot) is < j
    __shared__ u32 wlist[JN_THREADS / 64][JN_WLIST]; // per-wave candidate list: query slot | index posting << 13
    __shared__ u64 wlist_h[JN_THREADS / 64][JN_WLIST]; // ... and the query's hash
    __shared__ u32 scan_smem[JN_THREADS / 64 + 1];
    __shared__ unsigned long long base_s;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef JN_STAMP
    unsigned long long jn_t_prev = clock64();
#endif
    // (Measured, round 4: XCD-contiguous bucket numbers — ks_xcd_block, so that neighbouring buckets' 48 bytes of bucket words come
    // through one L2 — change nothing: 1.38 - 1.40 ms either way.)
    const u32 bkt = blockIdx.x;
    const u64 qs = q_lo[bkt], qe = q_hi[bkt]; // dense postings: q_hi = q_lo + 1 (a directory)
    const u64 ts = dir_t[bkt], te = dir_t[bkt + 1];
    const ks_bmeta bm = bmeta[bkt]; // (with the directory words: one memory latency)
    if (qs == qe || ts == te) return;
    const jn_qfmt<F10> QF(fp_shift, bkt);
    // The pair list is cut into seg_mask + 1 segments of `cap` records, each with its own cursor (JN_CUR_STRIDE words apart:
    // different memory channels): atomics on ONE address are served one at a time, ~12 ns each on this chip — 65,536 buckets
    // on one cursor are 0.8 ms of queueing whatever else the kernel does.  A bucket appends to segment (bucket mod segments).
    const u32 seg = bkt & seg_mask;
    unsigned long long *cursor = cursors + (size_t)seg * JN_CUR_STRIDE;
    pair_keys += (u64)seg * cap;
    if (pair_vals) pair_vals += (u64)seg * cap;
    const u64 kbase = bm.base; // the bucket's first key: the fingerprints count from it
    const u32 dirM = bm.slot_mul;
    // The first round's query postings are requested before the fingerprints are staged: the two latencies overlap.
    u64 h[JN_E];
    {
        const u32 nq0 = (u32)((qe - qs) < (u64)JN_THREADS * JN_E ? (qe - qs) : (u64)JN_THREADS * JN_E);
#pragma unroll
        for (int e = 0; e < JN_E; e++) {
            const u32 i = (u32)e * JN_THREADS + tid;
            h[e] = i < nq0 ? qkeys[qs + i] : 0; // (raw words: jn_qfmt)
        }
    }
    for (u64 c0 = ts; c0 < te; c0 += JN_CAP) {
        const u32 n = (u32)((te - c0) < JN_CAP ? (te - c0) : JN_CAP);
        const u32 *fpc = ifp + c0;       // uniform bases + 32-bit lane offsets: scalar base, one VGPR per address
        const ks_post *postc = ipost + c0;
        JN_STAMP_AT(0); // directory words + first key
        // Staging, and in the same pass a DIRECTORY over the staged fingerprints: they are uniform inside the bucket, so slot
        // j = floor(f * JN_DIR / (the bucket's largest + 1)) holds n / JN_DIR of them on average, and a query's search is two
        // directory reads and a search of its slot's few entries instead of log2(n) dependent LDS probes.  ldir[j] = staged
        // fingerprints whose slot is < j: entry j is written by the first fingerprint whose slot is >= j — a thread sees that
        // from its fingerprint and the one before it (the lane below; a load for lane 0).  JN_FILLU of each are in flight before the
        // LDS stores.
        {
            const u32 s_first = jn_slot(fpc[0], dirM), s_last = jn_slot(fpc[n - 1], dirM); // (uniform; slot 0 unless a later chunk)
            for (u32 j = tid; j <= s_first; j += JN_THREADS) ldir[j] = 0;
            for (u32 j = s_last + 1u + tid; j <= (u32)JN_DIR; j += JN_THREADS) ldir[j] = (unsigned short)n;
        }
        for (u32 i0 = 0; i0 < n; i0 += JN_THREADS * JN_FILLU) {
            u32 kk[JN_FILLU];
#pragma unroll
            for (int j = 0; j < JN_FILLU; j++) {
                const u32 i = i0 + (u32)j * JN_THREADS + tid;
                kk[j] = i < n ? fpc[i] : 0;
            }
            // The fingerprint before mine: consecutive threads hold consecutive fingerprints, so it comes from the lane below (round 3
            // loaded it a second time, 9 more global loads per thread: join 1.42 -> 1.40 ms); lane 0's sits in the wave before: lane j
            // of the wave loads it for slot j — ONE register for all slots instead of one per slot, which is what lets JN_FILLU
            // slots be in flight at once within the register budget (a bucket's ~4.5k fingerprints in one round trip, not two).
            u32 ppx = 0;
            {
                const u32 iw = i0 + lane * JN_THREADS + (tid & ~63u); // lane j: where this wave's slot j starts
                if (lane < (u32)JN_FILLU && iw < n && iw) ppx = fpc[iw - 1];
            }
#pragma unroll
            for (int j = 0; j < JN_FILLU; j++) {
                const u32 i = i0 + (u32)j * JN_THREADS + tid;
                const u32 below = ks_lane_below(kk[j]);
                const u32 first = (u32)__builtin_amdgcn_readlane((int)ppx, j);
                if (i < n) {
                    lk[i] = kk[j];
                    if (i) {
                        const u32 sa = jn_slot(kk[j], dirM);
                        for (u32 sl = jn_slot(lane ? below : first, dirM) + 1u; sl <= sa; sl++) ldir[sl] = (unsigned short)i;
                    }
                }
            }
        }
        __syncthreads();
        JN_STAMP_AT(1); // fingerprints staged, directory built
        for (u64 q0 = qs; q0 < qe; q0 += (u64)JN_THREADS * JN_E) {
            u32 info[JN_E]; // position | run length << 16
            const u64 *qkr = qkeys + q0;
            const u32 *qir = QF.vals_at(qids, q0); // (narrow value columns: jn_qfmt)
            const u32 nq = (u32)((qe - q0) < (u64)JN_THREADS * JN_E ? (qe - q0) : (u64)JN_THREADS * JN_E); // query postings of this round
            if (q0 != qs || c0 != ts) { // (uniform; the first round's are on their way since the top)
#pragma unroll
                for (int e = 0; e < JN_E; e++) {
                    const u32 i = (u32)e * JN_THREADS + tid;
                    h[e] = i < nq ? qkr[i] : 0;
                }
            }
#ifdef JN_STAMP
            if (tid == 0 && h[0] == 0x123456789abcdefULL) jn_t_prev++; // (waits for the loads)
#endif
            JN_STAMP_AT(3); // query postings loaded
            u32 cand = 0, mine = 0; // cand: bit e set = query e of this thread has candidates; mine: how many postings
#pragma unroll
            for (int e = 0; e < JN_E; e++) {
                const u32 i = (u32)e * JN_THREADS + tid;
                info[e] = 0;
                // (uniform: a bucket with few query postings — small batches, query shards of a strong-scaling run — fills
                // only the first slots of the round, and an empty slot would cost the same directory reads and probes as a full one)
                if ((u32)e * JN_THREADS >= nq) continue;
                const u32 f = QF.fp(h[e], kbase, fp_shift);
                const u32 sl = jn_slot(f, dirM);
                const u32 d0 = ldir[sl], d1 = ldir[sl + 1];
                // The slot holds n / JN_DIR fingerprints on average — one, give or take: its first JN_PROBES entries are read at once
                // (no loop, no wait between dependent LDS reads: the per-entry loop this replaces ran for the fullest slot among
                // the wave's 64 queries and waited for LDS in every turn) and compared under the slot's count; equal fingerprints
                // are neighbours, so the matches form one run.  A slot with more entries (a crowded slot, a hash many targets share)
                // is walked or searched.
                u32 lo = d0, c = 0;
                if (i < nq && QF.in_range(h[e], kbase)) { // (a hash below the bucket's first key matches nothing)
                    const u32 cnt = d1 - d0;
                    u32 pv[JN_PROBES];
#pragma unroll
                    for (int j = 0; j < JN_PROBES; j++) pv[j] = lk[d0 + j];
                    u32 first = JN_PROBES;
#pragma unroll
                    for (int j = JN_PROBES - 1; j >= 0; j--) {
                        const bool mt = (u32)j < cnt && pv[j] == f;
                        c += mt ? 1u : 0u;
                        first = mt ? (u32)j : first;
                    }
                    lo = d0 + (c ? first : 0u);
                    if (cnt > (u32)JN_PROBES) { // rare: start over on the whole slot
                        lo = d0; c = 0;
                        if (cnt <= 8u) {
                            for (u32 r = d0; r < d1; r++)
                                if (lk[r] == f) { lo = c ? lo : r; c++; }
                        } else {
                            lo = d0 + jn_lower_bound_lds(lk + d0, cnt, f);
                            while (lo + c < d1 && lk[lo + c] == f) c++;
                        }
                    }
                }
                info[e] = lo | (c << 16);
                mine += c;
                cand |= c ? (1u << e) : 0u;
            }
            JN_STAMP_AT(4); // searched
            const u32 wincl = ks_wave_incl_scan(mine);
            const u32 wtotal = (u32)__builtin_amdgcn_readlane((int)wincl, 63); // candidates of this wave (a scalar: v_readlane, no LDS round trip)
            u32 conf = 0;                             // confirmed pairs this lane will write
            u64 rk[JN_WLIST / 64];
            u32 rv[JN_WLIST / 64], okm = 0; // okm: bit it set = slot it of this lane holds a confirmed pair
            if (wtotal <= JN_WLIST) { // uniform per wave
                u32 p = wincl - mine;
                while (cand) {
                    const int e = __builtin_ctz(cand);
                    cand &= cand - 1;
                    u32 inf = info[0];
                    u64 he = h[0];
#pragma unroll
                    for (int k = 1; k < JN_E; k++) { inf = e == k ? info[k] : inf; he = e == k ? h[k] : he; } // (register arrays: select, no indexing)
                    const u32 c = inf >> 16, qi = (u32)e * JN_THREADS + tid, pos = inf & 0xffffu;
                    for (u32 j = 0; j < c; j++, p++) {
                        wlist[wave][p] = qi | ((pos + j) << 13); wlist_h[wave][p] = he;
                    }
                }
                __builtin_amdgcn_wave_barrier(); // same wave: LDS operations execute in order
                // Everything a candidate needs is REQUESTED before anything is compared: the index posting as one 16-byte load, the
                // query's id (its key word and its value byte), for both rounds of the list.  (Written as "load the posting, compare,
                // then take tid / abundance / id" the compiler loaded the posting's key alone, waited, and fetched the rest inside the
                // branch: two dependent memory latencies per round — and nearly every candidate IS a match.)
                uint4 praw[JN_WLIST / 64];
                u32 qcand[JN_WLIST / 64];
                u64 hcand[JN_WLIST / 64];
#pragma unroll
                for (int it = 0; it < JN_WLIST / 64; it++) {
                    const u32 k = (u32)it * 64u + lane;
                    const u32 en = k < wtotal ? wlist[wave][k] : 0u; // (a lane without a candidate reads posting 0 and query 0 of the bucket)
                    praw[it] = *(const uint4 *)&postc[en >> 13];
                    qcand[it] = QF.qid(qkr, qir, en & 0x1fffu);
                    hcand[it] = wlist_h[wave][k < wtotal ? k : 0u];
                }
#pragma unroll
                for (int it = 0; it < JN_WLIST / 64; it++) {
                    const u32 k = (u32)it * 64u + lane;
                    const u64 pkey = ((u64)praw[it].y << 32) | praw[it].x;
                    const u32 ptid = praw[it].z, pab = praw[it].w;
                    const bool ok = k < wtotal && QF.same(pkey, hcand[it]); // confirmed
                    const u64 ids = ((u64)qcand[it] << tbits) | ptid; // ids packed tight: fewer sort passes
                    rk[it] = ok ? (pair_vals ? ids : ((ids << abits) | pab)) : 0ULL; // packed: one 8-byte record per match
                    rv[it] = ok ? pab : 0u;
                    okm |= ok ? (1u << it) : 0u;
                    conf += ok ? 1u : 0u;
                }
                __builtin_amdgcn_wave_barrier();
            } else { // a hash shared by many targets: confirm whole runs, emit straight from the registers
#pragma unroll
                for (int e = 0; e < JN_E; e++) {
                    if (!((cand >> e) & 1u)) continue;
                    u32 lo = info[e] & 0xffffu, c = info[e] >> 16;
                    const u64 hh = QF.hash(h[e]);
                    if (postc[lo].key != hh || postc[lo + c - 1].key != hh) {
                        u32 first, cnt;
                        jn_confirm_slow(postc, lo, c, hh, &first, &cnt);
                        lo += first; c = cnt;
                    }
                    info[e] = lo | (c << 16);
                    conf += c;
                    if (!c) cand &= ~(1u << e);
                }
            }
#ifdef JN_STAMP
            if (tid == 0 && rk[0] == 0x123456789abcdefULL) jn_t_prev++;
#endif
            JN_STAMP_AT(5); // candidates listed, fetched, confirmed
            // Reserving the round's slice of the pair list.  Segmented list (many buckets): every wave reserves for itself — no
            // workgroup barrier between the search and the stores, the waves of a bucket run free until the next chunk is
            // staged (the atomics spread over the segment cursors).  One cursor (few buckets): one atomic per workgroup round.
            u32 total;
            u64 slot;
            if (seg_mask) { // uniform
                const u32 cincl = ks_wave_incl_scan(conf);
                total = (u32)__builtin_amdgcn_readlane((int)cincl, 63);
                JN_STAMP_AT(6); // wave scan
                unsigned long long wb = 0;
                if (total) { // uniform per wave
                    if (lane == 0) wb = atomicAdd(cursor, (unsigned long long)total);
                    wb = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(wb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)wb);
                }
                slot = wb + (cincl - conf);
            } else {
                const u32 off = ks_block_excl_scan(conf, scan_smem, &total);
                JN_STAMP_AT(6); // block scan
                if (total) { // uniform
                    if (tid == 0) base_s = atomicAdd(cursor, (unsigned long long)total);
                    __syncthreads();
                }
                slot = base_s + off;
            }
            JN_STAMP_AT(7); // slice reserved
            if (total) {
                if (wtotal <= JN_WLIST) {
#pragma unroll
                    for (int it = 0; it < JN_WLIST / 64; it++) {
                        if ((okm >> it) & 1u) {
                            if (slot < cap) {
                                pair_keys[slot] = rk[it];
                                if (pair_vals) pair_vals[slot] = rv[it];
                            }
                            slot++;
                        }
                    }
                } else {
                    while (cand) {
                        const int e = __builtin_ctz(cand);
                        cand &= cand - 1;
                        u32 inf = info[0];
#pragma unroll
                        for (int k = 1; k < JN_E; k++) inf = e == k ? info[k] : inf;
                        const u32 c = inf >> 16;
                        const u32 q = QF.qid(qkr, qir, (u32)e * JN_THREADS + tid);
                        const u32 j0 = inf & 0xffffu;
                        for (u32 j = 0; j < c; j++, slot++) {
                            if (slot < cap) {
                                const ks_post pt = postc[j0 + j];
                                const u64 ids = ((u64)q << tbits) | pt.tid;
                                if (pair_vals) { pair_keys[slot] = ids; pair_vals[slot] = pt.abund; }
                                else pair_keys[slot] = (ids << abits) | pt.abund;
                            }
                        }
                    }
                }
                if (!seg_mask) __syncthreads(); // base_s is rewritten next round
                JN_STAMP_AT(8); // pairs written
            }
        }
        __syncthreads();
    }
}

// ---- the join of big indexes when the buckets hold FEW query postings (a query shard of a strong-scaling run, a small
// batch against a big index): the roles swap.  The bucket's queries — a few hundred — are put in fingerprint-slot order in a
// 16-KB LDS table with a directory over it; the index fingerprints are not staged at all: every thread takes its share of
// them straight into registers and probes the table.  Same candidates, same confirmation on the 16-byte postings, same
// per-wave reservation as k_join_buckets — but 256 threads and a third of the LDS per bucket, so 8 workgroups per CU instead
// of 3 work on the chain of memory latencies a sparse bucket is.
#define JS_THREADS 256
#define JS_QCAP 1024  // query postings per table build (more: the bucket is joined in slices)
#define JS_QE (JS_QCAP / JS_THREADS)
#ifndef JS_IPT
#define JS_IPT 18     // index fingerprints per thread and pass (4,608: a bucket of ~4.5k in one pass)
#endif
#ifndef JS_G
#define JS_G 6        // ... of which this many are in flight per group (with the group's slots looked up together: 4 -> 0.512 ms,
#endif                //     5 -> 0.495, 6 -> 0.471, 7 -> 0.473 for the 125k-query shard; 8 spills: 0.56)
#define JS_DIR 1024   // directory slots over the table
#define JS_WLIST 128  // candidates a wave lists per pass
KS_DEV u32 js_slot(u32 f, u32 mul) { // slot of a fingerprint: the bucket's JN_DIR slot mapping, coarsened
    const u32 j = __umulhi(f, mul) / (u32)(JN_DIR / JS_DIR);
    return j < (u32)JS_DIR - 1u ? j : (u32)JS_DIR - 1u;
}
static_assert(JN_DIR % JS_DIR == 0 && ((JN_DIR / JS_DIR) & (JN_DIR / JS_DIR - 1)) == 0, "js_slot coarsens the JN_DIR slot by a power of two");
template <int F10>
__global__ __launch_bounds__(JS_THREADS) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_join_sparse(
    const u64 *qkeys, const u32 *qids, const u32 *ifp, const ks_post *ipost, const ks_bmeta *bmeta, const u64 *q_lo, const u64 *q_hi,
    const u64 *dir_t, u64 *pair_keys, u32 *pair_vals, u64 cap, unsigned long long *cursors, u32 seg_mask, int tbits, int abits,
    int fp_shift) {
    __shared__ u64 qh[JS_QCAP];                       // query hashes, in slot order
    __shared__ u32 qf[JS_QCAP + 2];                   // their fingerprints (+ 2: the probes read two entries of a slot whatever it holds)
    __shared__ unsigned short qi[JS_QCAP];            // their place in the bucket's posting list
    __shared__ u32 qdir[JS_DIR + 1];                  // counts, then first table entry of every slot
    __shared__ u32 wlist[JS_THREADS / 64][JS_WLIST];  // per-wave candidate list: index posting (this pass) << 10 | table entry
    __shared__ u32 wcount[JS_THREADS / 64];
    __shared__ u32 scan_smem[JS_THREADS / 64 + 1];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 bkt = blockIdx.x;
    const u64 qs = q_lo[bkt], qe = q_hi[bkt];
    const u64 ts = dir_t[bkt], te = dir_t[bkt + 1];
    const ks_bmeta bm = bmeta[bkt];
    if (qs == qe || ts == te) return;
    const jn_qfmt<F10> QF(fp_shift, bkt);
    const u32 seg = bkt & seg_mask;
    unsigned long long *cursor = cursors + (size_t)seg * JN_CUR_STRIDE;
    pair_keys += (u64)seg * cap;
    if (pair_vals) pair_vals += (u64)seg * cap;
    const u64 kbase = bm.base;
    const u32 dirM = bm.slot_mul;
    for (u64 q0 = qs; q0 < qe; q0 += JS_QCAP) { // slices of the bucket's queries (one, normally)
        const u32 nq = (u32)((qe - q0) < (u64)JS_QCAP ? (qe - q0) : (u64)JS_QCAP);
        const u64 *qkr = qkeys + q0;
        const u32 *qir = QF.vals_at(qids, q0); // (narrow value columns: jn_qfmt)
        // ---- the table: counting sort of the slice by fingerprint slot
        u64 h[JS_QE];
#pragma unroll
        for (int e = 0; e < JS_QE; e++) {
            const u32 i = (u32)e * JS_THREADS + tid;
            h[e] = i < nq ? QF.hash(qkr[i]) : 0;
        }
        for (u32 i = tid; i <= (u32)JS_DIR; i += JS_THREADS) qdir[i] = 0;
        __syncthreads();
        u32 code[JS_QE]; // slot << 16 | arrival inside the slot; 0xffffffff = no query / below the bucket's first key
#pragma unroll
        for (int e = 0; e < JS_QE; e++) {
            const u32 i = (u32)e * JS_THREADS + tid;
            code[e] = 0xffffffffu;
            if (i < nq && h[e] >= kbase) {
                const u32 sl = js_slot(jn_fingerprint(h[e], kbase, fp_shift), dirM);
                code[e] = (sl << 16) | atomicAdd(&qdir[sl], 1u);
            }
        }
        __syncthreads();
        {
            u32 c[JS_DIR / JS_THREADS], sum = 0;
#pragma unroll
            for (int j = 0; j < JS_DIR / JS_THREADS; j++) { c[j] = qdir[tid * (JS_DIR / JS_THREADS) + j]; sum += c[j]; }
            u32 total;
            u32 ex = ks_block_excl_scan(sum, scan_smem, &total);
#pragma unroll
            for (int j = 0; j < JS_DIR / JS_THREADS; j++) { qdir[tid * (JS_DIR / JS_THREADS) + j] = ex; ex += c[j]; }
            if (tid == JS_THREADS - 1) qdir[JS_DIR] = total;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < JS_QE; e++)
            if (code[e] != 0xffffffffu) {
                const u32 pos = qdir[code[e] >> 16] + (code[e] & 0xffffu);
                qh[pos] = h[e];
                qf[pos] = jn_fingerprint(h[e], kbase, fp_shift);
                qi[pos] = (unsigned short)((u32)e * JS_THREADS + tid);
            }
        __syncthreads();
        // ---- the index fingerprints of the bucket stream past the table, JS_THREADS * JS_IPT per pass
        for (u64 c0 = ts; c0 < te; c0 += (u64)JS_THREADS * JS_IPT) {
            const u32 n = (u32)((te - c0) < (u64)JS_THREADS * JS_IPT ? (te - c0) : (u64)JS_THREADS * JS_IPT);
            const u32 *fpc = ifp + c0;
            const ks_post *postc = ipost + c0;
            if (lane == 0) wcount[wave] = 0;
            __builtin_amdgcn_wave_barrier();
            // groups of JS_G fingerprints per thread, the next group's loads in flight while this one probes the table
            u32 fa[JS_G], fb[JS_G];
#pragma unroll
            for (int j = 0; j < JS_G; j++) {
                const u32 i = (u32)j * JS_THREADS + tid;
                fa[j] = i < n ? fpc[i] : 0;
            }
#pragma unroll 1
            for (u32 g = 0; g < (u32)(JS_IPT / JS_G); g++) {
                const u32 i0 = g * (u32)(JS_G * JS_THREADS);
                if (i0 >= n) break; // (uniform)
#pragma unroll
                for (int j = 0; j < JS_G; j++) {
                    const u32 i = i0 + (u32)(JS_G + j) * JS_THREADS + tid;
                    fb[j] = (g + 1 < (u32)(JS_IPT / JS_G) && i < n) ? fpc[i] : 0;
                }
                // The group's JS_G slots are looked up together: their directory words (2 x JS_G LDS reads in flight), then the first
                // two table entries of every slot (a slot holds 0.55 queries on average), then the compares — the per-fingerprint
                // form (directory words, wait, entry, wait, entry ...) was two to three dependent LDS latencies per fingerprint,
                // twenty fingerprints per thread and pass in a row.
                u32 d0[JS_G], d1[JS_G], e0[JS_G], e1[JS_G];
#pragma unroll
                for (int j = 0; j < JS_G; j++) {
                    const u32 sl = js_slot(fa[j], dirM); // (a lane behind n holds fingerprint 0: slot 0, dropped below)
                    d0[j] = qdir[sl]; d1[j] = qdir[sl + 1];
                }
#pragma unroll
                for (int j = 0; j < JS_G; j++) { e0[j] = qf[d0[j]]; e1[j] = qf[d0[j] + 1u]; }
#pragma unroll
                for (int j = 0; j < JS_G; j++) {
                    const u32 i = i0 + (u32)j * JS_THREADS + tid;
                    const u32 cnt = i < n ? d1[j] - d0[j] : 0u;
                    if (cnt >= 1u && e0[j] == fa[j]) {
                        const u32 p = atomicAdd(&wcount[wave], 1u);
                        if (p < (u32)JS_WLIST) wlist[wave][p] = (i << 10) | d0[j];
                    }
                    if (cnt >= 2u && e1[j] == fa[j]) {
                        const u32 p = atomicAdd(&wcount[wave], 1u);
                        if (p < (u32)JS_WLIST) wlist[wave][p] = (i << 10) | (d0[j] + 1u);
                    }
                    for (u32 r = d0[j] + 2u; r < d0[j] + cnt; r++) // (rare: a crowded slot)
                        if (qf[r] == fa[j]) {
                            const u32 p = atomicAdd(&wcount[wave], 1u);
                            if (p < (u32)JS_WLIST) wlist[wave][p] = (i << 10) | r;
                        }
                }
#pragma unroll
                for (int j = 0; j < JS_G; j++) fa[j] = fb[j];
            }
            __builtin_amdgcn_wave_barrier();
            const u32 wtotal = wcount[wave]; // candidates of this wave (uniform)
            if (wtotal == 0) continue;
            if (wtotal <= (u32)JS_WLIST) {
                // lane k confirms and emits candidate k
                u64 rk[JS_WLIST / 64];
                u32 rv[JS_WLIST / 64], okm = 0, conf = 0;
                // (posting — one 16-byte load — and query id requested for both rounds before anything is compared: see k_join_buckets)
                uint4 praw[JS_WLIST / 64];
                u32 qcand[JS_WLIST / 64], rcand[JS_WLIST / 64];
#pragma unroll
                for (int it = 0; it < JS_WLIST / 64; it++) {
                    const u32 k = (u32)it * 64u + lane;
                    const u32 en = k < wtotal ? wlist[wave][k] : 0u; // (a lane without a candidate reads posting 0 and table entry 0)
                    rcand[it] = en & 1023u;
                    praw[it] = *(const uint4 *)&postc[en >> 10];
                    qcand[it] = QF.qid(qkr, qir, qi[rcand[it]]);
                }
#pragma unroll
                for (int it = 0; it < JS_WLIST / 64; it++) {
                    const u32 k = (u32)it * 64u + lane;
                    const u64 pkey = ((u64)praw[it].y << 32) | praw[it].x;
                    const u32 ptid = praw[it].z, pab = praw[it].w;
                    const bool ok = k < wtotal && pkey == qh[rcand[it]];
                    const u64 ids = ((u64)qcand[it] << tbits) | ptid;
                    rk[it] = ok ? (pair_vals ? ids : ((ids << abits) | pab)) : 0ULL;
                    rv[it] = ok ? pab : 0u;
                    okm |= ok ? (1u << it) : 0u;
                    conf += ok ? 1u : 0u;
                }
                const u32 cincl = ks_wave_incl_scan(conf);
                const u32 total = (u32)__builtin_amdgcn_readlane((int)cincl, 63);
                if (total) { // uniform per wave
                    unsigned long long wb = 0;
                    if (lane == 0) wb = atomicAdd(cursor, (unsigned long long)total);
                    wb = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(wb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)wb);
                    u64 slot = wb + (cincl - conf);
#pragma unroll
                    for (int it = 0; it < JS_WLIST / 64; it++)
                        if ((okm >> it) & 1u) {
                            if (slot < cap) {
                                pair_keys[slot] = rk[it];
                                if (pair_vals) pair_vals[slot] = rv[it];
                            }
                            slot++;
                        }
                }
            } else {
                // more candidates than the list holds (a hash shared by many queries and targets): every lane confirms and
                // emits its own, one reservation per pair — slow and exact
#pragma unroll 1
                for (u32 i = tid; i < n; i += JS_THREADS) {
                    const u32 fj = fpc[i];
                    const u32 sl = js_slot(fj, dirM);
                    for (u32 r = qdir[sl]; r < qdir[sl + 1]; r++)
                        if (qf[r] == fj) {
                            const ks_post pt = postc[i];
                            if (pt.key == qh[r]) {
                                const u64 slot = atomicAdd(cursor, 1ULL);
                                if (slot < cap) {
                                    const u64 ids = ((u64)QF.qid(qkr, qir, qi[r]) << tbits) | pt.tid;
                                    if (pair_vals) { pair_keys[slot] = ids; pair_vals[slot] = pt.abund; }
                                    else pair_keys[slot] = (ids << abits) | pt.abund;
                                }
                            }
                        }
                }
            }
        }
        __syncthreads(); // the table is rebuilt for the next slice
    }
}

// the segments of the pair list (seg_cap records apart, prefix[s+1] - prefix[s] of them filled) -> one dense list
struct jn_seg_table { u64 prefix[JN_SEGS + 1]; };
__global__ __launch_bounds__(256) void k_pairs_compact(const u64 *src_k, const u32 *src_v, u64 seg_cap, jn_seg_table tab, u64 *dst_k, u32 *dst_v) {
    const u32 s = blockIdx.y;
    const u64 n = tab.prefix[s + 1] - tab.prefix[s], i0 = (u64)blockIdx.x * 2048;
    const u64 *sk = src_k + (u64)s * seg_cap;
    u64 *dk = dst_k + tab.prefix[s];
    u64 kk[8]; // (requested together, then stored)
    u32 vv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u64 i = i0 + (u64)j * 256 + threadIdx.x;
        kk[j] = i < n ? sk[i] : 0ULL;
        vv[j] = (src_v && i < n) ? src_v[(u64)s * seg_cap + i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u64 i = i0 + (u64)j * 256 + threadIdx.x;
        if (i < n) {
            dk[i] = kk[j];
            if (src_v) dst_v[tab.prefix[s] + i] = vv[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pair reduce: sorted (qid<<32|tid, abund) -> COO rows
// ---------------------------------------------------------------------------------------------
// `abits` low bits of a key are payload (the packed abundance, 0 when the values travel separately): rows are runs of
// equal keys >> abits
__global__ __launch_bounds__(256) void k_pair_heads(const u64 *keys, u64 n, u32 *heads, int abits) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    heads[i] = (i == 0 || (keys[i] >> abits) != (keys[i - 1] >> abits)) ? 1u : 0u;
}

// hidx = exclusive scan of heads; row r starts where hidx steps from r to r+1
__global__ __launch_bounds__(256) void k_pair_rows(const u64 *keys, const u32 *hidx, u64 n, u32 n_rows, u64 *row_start, int abits) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool head = (i == 0) || (keys[i] >> abits) != (keys[i - 1] >> abits);
    if (head) row_start[hidx[i]] = i;
    if (i == 0) row_start[n_rows] = n;
}

// One pass over the sorted match list, coalesced: element i belongs to row hidx[i] (+1 if it is not a head, -1 based);
// a wave sums abundance and count per row with a segmented shuffle scan and the last lane of each row segment adds the
// partial to the row (rows span waves, so the adds are atomic: ~2 per wave).  Heads write the ids.
// Rows beyond `rows_cap` are dropped (the row arrays are sized from the previous search's row count; the host repeats
// this launch with exact arrays when the true count — known only after the scan — is larger).
__global__ __launch_bounds__(256) void k_pair_reduce(const u64 *keys, const u32 *vals, const u32 *hidx, u64 n, u32 *qid,
                                                     u32 *tid, u32 *isect, unsigned long long *nw, int tbits, int abits, u32 rows_cap) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    const bool live = i < n;
    u64 k = 0, w = 0;
    u32 row = 0xffffffffu, c = 0;
    bool head = false;
    if (live) {
        k = keys[i];
        head = i == 0 || (keys[i - 1] >> abits) != (k >> abits);
        row = hidx[i] - (head ? 0u : 1u);
        w = vals ? (u64)vals[i] : (k & ((1ULL << abits) - 1ULL));
        c = 1;
        if (head && row < rows_cap) {
            const u64 ids = k >> abits;
            qid[row] = (u32)(ids >> tbits);
            tid[row] = (u32)(ids & ((1ULL << tbits) - 1ULL));
        }
    }
    // inclusive segmented scan (segments = equal row, rows ascend)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 ow = __shfl_up(w, d, 64);
        const u32 oc = __shfl_up(c, d, 64), orow = __shfl_up(row, d, 64);
        if (lane >= (u32)d && orow == row) { w += ow; c += oc; }
    }
    const u32 nrow = __shfl_down(row, 1, 64);
    // (plain stores for rows that sit wholly inside a wave were measured slower than these fire-and-forget adds)
    if (live && (lane == 63 || nrow != row) && row < rows_cap) {
        atomicAdd(&isect[row], c);
        atomicAdd(&nw[row], (unsigned long long)w);
    }
}

// ---------------------------------------------------------------------------------------------
// Sorted match list -> COO rows in ONE pass (packed records): heads, their prefix over the whole list and the per-row
// sums used to be three passes (k_pair_heads, a device-wide scan, k_pair_reduce) — 0.23 ms of the 1M x 1M step and a quarter
// of the all-vs-all one, where two records in three open a row.  Here a tile of PF_TILE records keeps its keys in
// registers: sweep 1 marks the heads (bit per round) and counts them per (round, wave); wave 0 turns the 32 counts into
// offsets and chains the tile's total through a decoupled look-back (ticket-ordered tiles, 8-byte {flag, value} words);
// sweep 2 gives every record its row and sums count / abundance per row with the segmented shuffle scan of k_pair_reduce
// (rows span waves and tiles, so the partial sums are added atomically: ~2 atomics per wave and round).
// ---------------------------------------------------------------------------------------------
#define PF_THREADS 256
#define PF_IPT 8
#define PF_TILE (PF_THREADS * PF_IPT)
#define PF_WAVES (PF_THREADS / 64)
#define PF_FLAG_AGG (1ULL << 62)
#define PF_FLAG_PRE (2ULL << 62)
#define PF_VAL_MASK ((1ULL << 62) - 1)

__global__ __launch_bounds__(PF_THREADS) void k_pair_rows_fused(const u64 *keys, u64 n, u32 *qid, u32 *tid, u32 *isect, unsigned long long *nw,
                                                                int tbits, int abits, u32 rows_cap, unsigned long long *status,
                                                                u32 *ticket /* [0] tile ids, [1] a look-back gave up */, u32 *n_rows_out,
                                                                int use_ticket) {
    __shared__ u32 tile_s;
    __shared__ u32 wcount[PF_IPT][PF_WAVES]; // heads per (round, wave), then exclusive offsets inside the tile
    __shared__ unsigned long long base_s;
    const u32 tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6;
    // Tile ids in dispatch order (blockIdx.x): workgroups start in that order on this hardware, so a tile's predecessors are
    // running or done when it looks back.  That is not a documented guarantee: the look-back spins for a bounded time, and a
    // launch in which one gave up is repeated with ids from an atomic ticket (order of arrival; ~12 ns per ticket on one
    // address — 10^4 tiles are 0.1 ms of queueing, most of this kernel's former run time).
    if (use_ticket) { // uniform
        if (tid_ == 0) tile_s = atomicAdd(&ticket[0], 1u);
        __syncthreads();
    }
    const u32 tile = use_ticket ? tile_s : blockIdx.x;
    const u64 b0 = (u64)tile * PF_TILE;
    u64 key[PF_IPT];
    u32 headbits = 0;
    // sweep 1: record i of round r is b0 + r * PF_THREADS + tid (coalesced); a head opens a row
#pragma unroll
    for (int r = 0; r < PF_IPT; r++) {
        const u64 i = b0 + (u64)r * PF_THREADS + tid_;
        key[r] = i < n ? keys[i] : 0;
    }
    // The previous record: the lane below's (two DPP moves, no LDS crossbar), except for lane 0, whose predecessor sits in the
    // previous wave / round / tile: lane r of the wave loads it for round r — one register pair and ONE memory latency for all
    // rounds (a load inside the loop was waited for in every round: eight dependent latencies per tile).
    u64 pp = 0;
    {
        const u64 iw = b0 + (u64)lane * PF_THREADS + (tid_ & ~63u); // lane r: where this wave's round r starts
        if (lane < PF_IPT && iw > 0 && iw < n) pp = keys[iw - 1];
    }
#pragma unroll
    for (int r = 0; r < PF_IPT; r++) {
        const u64 i = b0 + (u64)r * PF_THREADS + tid_;
        const u64 below = ((u64)ks_lane_below((u32)(key[r] >> 32)) << 32) | ks_lane_below((u32)key[r]);
        const u64 first = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(pp >> 32), r) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)pp, r);
        const u64 prev = lane ? below : first;
        const bool head = i < n && (i == 0 || (prev >> abits) != (key[r] >> abits));
        headbits |= head ? (1u << r) : 0u;
        const u64 m = __ballot(head);
        if (lane == 0) wcount[r][wave] = (u32)__popcll(m);
    }
    __syncthreads();
    // wave 0: exclusive offsets of the 32 (round, wave) groups in record order, tile total, look-back
    if (wave == 0) {
        const u32 c = lane < PF_IPT * PF_WAVES ? wcount[lane / PF_WAVES][lane % PF_WAVES] : 0u;
        const u32 incl = ks_wave_incl_scan(c);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane < PF_IPT * PF_WAVES) wcount[lane / PF_WAVES][lane % PF_WAVES] = incl - c;
        if (lane == 0)
            __hip_atomic_store(&status[tile], (tile == 0 ? PF_FLAG_PRE : PF_FLAG_AGG) | (u64)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64 excl = 0;
        if (tile > 0) {
            i64 idx = (i64)tile - 1;
            bool done = false;
            u32 polls = 0;
            const long long t0 = wall_clock64();
            while (!done) {
                const i64 mine = idx - (i64)lane;
                u64 v = PF_FLAG_PRE; // before tile 0: inclusive prefix 0
                if (mine >= 0) {
                    v = __hip_atomic_load(&status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while ((v >> 62) == 0 && !ks_spin_expired(t0, polls)) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(&status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if ((v >> 62) == 0) { atomicOr(&ticket[1], 1u); v = PF_FLAG_PRE; } // gave up: the host reports it
                const u64 is_pre = __ballot((v >> 62) == 2);
                const u32 first = is_pre ? (u32)__ffsll((long long)is_pre) - 1u : 64u;
                u64 contrib = lane <= first ? (v & PF_VAL_MASK) : 0;
                contrib = ks_wave_sum64(contrib);
                excl += contrib;
                if (is_pre) done = true; else idx -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&status[tile], PF_FLAG_PRE | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            base_s = excl;
            if (b0 + PF_TILE >= n) *n_rows_out = (u32)(excl + total); // the last tile knows the row count
        }
    }
    __syncthreads();
    const u64 base = base_s;
    // sweep 2: rows and per-row sums
#pragma unroll
    for (int r = 0; r < PF_IPT; r++) {
        const u64 i = b0 + (u64)r * PF_THREADS + tid_;
        const bool live = i < n;
        const bool head = (headbits >> r) & 1u;
        const u64 m = __ballot(head);
        // heads at or before this record, over the whole list, minus one = its row
        const u64 row64 = base + wcount[r][wave] + ks_lane_lt_count(m) + (head ? 1u : 0u) - 1u;
        const u32 row = live ? (u32)row64 : 0xffffffffu;
        u64 w = live ? (key[r] & ((1ULL << abits) - 1ULL)) : 0;
        u32 c = live ? 1u : 0u;
        if (head && row < rows_cap) {
            const u64 ids = key[r] >> abits;
            qid[row] = (u32)(ids >> tbits);
            tid[row] = (u32)(ids & ((1ULL << tbits) - 1ULL));
        }
        // Inclusive segmented scan (segments = equal row; rows ascend, so a lane whose source has my row continues my segment) with DPP
        // moves only: four row_shr steps inside the rows of 16, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 — the
        // __shfl_up form was 25 ds_bpermute per round, 213 per thread: the LDS crossbar was this kernel (19 M wave-level permutes
        // for the 49 M records of a 200k all-vs-all).  A lane without a source reads `rowx`, which is not its row: it adds nothing.
        {
            const u32 rowx = row ^ 1u;
            u32 wl = (u32)w, wh = (u32)(w >> 32);
#define PF_SEG_STEP(CTRL, RMASK) do { \
                const u32 orow = (u32)__builtin_amdgcn_update_dpp((int)rowx, (int)row, CTRL, RMASK, 0xf, false); \
                const u32 oc = (u32)__builtin_amdgcn_update_dpp(0, (int)c, CTRL, RMASK, 0xf, false); \
                const u32 ol = (u32)__builtin_amdgcn_update_dpp(0, (int)wl, CTRL, RMASK, 0xf, false); \
                const u32 oh = (u32)__builtin_amdgcn_update_dpp(0, (int)wh, CTRL, RMASK, 0xf, false); \
                const bool same = orow == row; \
                const u64 nw_ = (((u64)wh << 32) | wl) + (same ? (((u64)oh << 32) | ol) : 0ULL); \
                c += same ? oc : 0u; wl = (u32)nw_; wh = (u32)(nw_ >> 32); } while (0)
            PF_SEG_STEP(0x111, 0xf); // row_shr:1
            PF_SEG_STEP(0x112, 0xf); // row_shr:2
            PF_SEG_STEP(0x114, 0xf); // row_shr:4
            PF_SEG_STEP(0x118, 0xf); // row_shr:8
            PF_SEG_STEP(0x142, 0xa); // row_bcast:15 -> rows 1 and 3
            PF_SEG_STEP(0x143, 0xc); // row_bcast:31 -> rows 2 and 3
#undef PF_SEG_STEP
            w = ((u64)wh << 32) | wl;
        }
        const u32 nrow = (u32)__builtin_amdgcn_update_dpp((int)(row ^ 1u), (int)row, 0x130, 0xf, 0xf, false); // wave_shl:1 (lane 63: not its row)
        // (Measured dead end: STORING the rows whose records all sit inside one wave — most rows of an all-vs-all search are one or
        // two records — instead of adding them: 0.45 -> 0.83 ms at 200k x 200k hp.  Scattered 4- / 8-byte stores of a wave cost
        // more than the same no-return atomics, which the L2 merges line by line.  Handing the head lanes' ids to lanes 0 .. heads-1
        // through LDS so that qid[] / tid[] leave coalesced: no change, 0.453 ms either way.)
        if (live && (lane == 63 || nrow != row) && row < rows_cap) {
            atomicAdd(&isect[row], c);
            atomicAdd(&nw[row], (unsigned long long)w);
        }
    }
}

// One search with the whole query batch in one match list.  *split_pairs != 0 on return (with KS_OK and *out == NULL) means
// the list would hold that many records — more than one list can (2^32) — and nothing was produced: the caller splits.
// q->pending (ks_sketch_search_device): the sketch launches are queued and nobody has waited for them; the first wait of
// the search stands in, and q's counts are upper bounds until then.  *sketch_redo != 0 on return (KS_OK, *out == NULL): the
// sketch has to be repeated the plain way (ks_sketch_finish_pending) and nothing was produced.
static int search_core(ks_ctx *ctx, const ks_index *ix, const ks_sketches *q, ks_hits **out, u64 *split_pairs, int *sketch_redo = nullptr) {
    *split_pairs = 0;
    *out = nullptr;
    if (sketch_redo) *sketch_redo = 0;
    if (!ix || !q || !out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    if (ix->params.ksize != q->params.ksize || ix->params.scaled != q->params.scaled ||
        ix->params.moltype != q->params.moltype || ix->params.seed != q->params.seed)
        return ks_fail(ctx, KS_ERR_INVALID_ARG, "query sketches and index were built with different parameters");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    ks_hits *H = new ks_hits();
    memset(H, 0, sizeof *H);
    H->ctx = ctx;
    u64 n_q = q->n_hashes;
    const u64 n_t = ix->n_postings;
    u64 *qk0 = nullptr, *qk1 = nullptr, *pk0 = nullptr, *pk1 = nullptr, *row_start = nullptr, *dir_q = nullptr;
    u32 *qv0 = nullptr, *qv1 = nullptr, *pv0 = nullptr, *pv1 = nullptr, *heads = nullptr, *d_nrows = nullptr;
    unsigned long long *cursor = nullptr, *pf_status = nullptr;
    u32 *pf_ticket = nullptr;
    int st = KS_OK;
    bool split = false;
    // the exact counts of a pending sketch, behind a wait; a sketch that has to be repeated ends the search (split: nothing produced)
#define SE_FINISH_PENDING() do { if (q->pending) { int redo_ = 0; st = ks_sketch_finish_pending(const_cast<ks_sketches *>(q), &redo_); \
        if (st != KS_OK) goto done; n_q = q->n_hashes; if (redo_) { if (sketch_redo) *sketch_redo = redo_; split = true; goto done; } } } while (0)
#define SE_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
#define SE_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
    if (q->pending && (n_t == 0 || !(q->part_keys && q->part_pbits == ix->pbits && ix->pbits > 0))) {
        // (no postings for this index: the partition starts from the CSR and needs the exact counts)
        { const ks_fetch_seg f = ks_sketch_pending_seg(q); SE_CHECK(ks_stream_wait_fetch(ctx, &f, 1)); }
        SE_FINISH_PENDING();
    }
    if (n_q == 0 || n_t == 0) {
        SE_CHECK(ks_alloc(ctx, &H->d_qid, 1)); SE_CHECK(ks_alloc(ctx, &H->d_tid, 1));
        SE_CHECK(ks_alloc(ctx, &H->d_isect, 1)); SE_CHECK(ks_alloc(ctx, &H->d_nw, 1));
        *out = H;
        return KS_OK;
    }
    {
        // ---- query postings grouped on the top pbits hash bits (the join needs locality, not order)
        const int pbits = ix->pbits;
        const u32 n_buckets = 1u << pbits;
        const int tbits = bits_for(ix->n_targets); // pair key = qid << tbits | tid
        // a match is one 8-byte record (qid, tid, target abundance) whenever the three fit 64 bits — always, short of
        // ~10^5 x 10^5 proteins with 2^30-fold repeats — so the match sort moves keys only; else ids and abundance travel apart
        const int qbits = bits_for(q->n_seqs);
        int abits = bits_for_value(ix->max_abund);
        const bool packed = tbits + qbits + abits <= 64 && !ks_dbg(ctx, KS_DBG_UNPACKED_PAIRS);
        if (!packed) abits = 0;
        const u32 pfxK = ks_join_prefix_mul(pbits, ks_max_hash(ix->params.scaled));
        SE_CHECK(ks_alloc(ctx, &dir_q, (size_t)2 * n_buckets + 2));
        const u64 *dir_t = ix->d_dir; // built with the index
        // cursor block: segment s of the pair list counts at word s * JN_CUR_STRIDE; word 1 = "a query bucket overflowed"
        // (k_bucket_scatter)
        // (+ the join buckets' fill counts right behind it: one allocation, one memset)
        SE_CHECK(ks_alloc(ctx, (u64 **)&cursor, (size_t)JN_SEGS * JN_CUR_STRIDE + (n_buckets + 1) / 2));
        u32 *const bcur = (u32 *)(cursor + (size_t)JN_SEGS * JN_CUR_STRIDE);
        const bool pre_any = q->part_keys && q->part_pbits == pbits && q->part_K == pfxK && pbits > 0;
        // (10-byte postings are read by the bucket scatter and the fingerprint joins only: the dense fall-back starts from the
        // CSR, and so does a search against an index they were not made for)
        const bool f10_fits = ix->fp_layout && pbits > 8 && q->part_s == 32u + (u32)ix->fp_shift;
        bool pre = pre_any && (q->part_s == 0 || f10_fits);
        const bool f10 = pre && q->part_s != 0;
        if (q->pending && !pre) {
            const ks_fetch_seg f = ks_sketch_pending_seg(q);
            SE_CHECK(ks_stream_wait_fetch(ctx, &f, 1));
            SE_FINISH_PENDING();
        }
        u64 cap = n_q < (1u << 20) ? (1u << 20) : n_q;
        if (ctx->pair_cap_hint > cap) cap = ctx->pair_cap_hint; // a workload that matched heavily last time will again
        u64 n_pairs = 0, seg_cap = 0, seg_count[JN_SEGS];
        u32 n_segs = 1;
        // way 0: histogram-free bucket scatter of the sketch kernel's regions (may overflow on skewed hashes);
        // way 1: the dense, always-correct partition
        for (int way = (pre && pbits > 8) ? 0 : 1; way < 2; way++) {
            if (way == 1 && f10) pre = false;
            const u32 q_s = (way == 0 && f10) ? q->part_s : 0u;
            // behind the bucket scatter of a 16-bit join prefix both prefix bytes are implied by the bucket: 9-byte postings
            // (KS_DEBUG_POSTINGS10 keeps the 10-byte form there)
            const int q_fmt = q_s ? ((pbits == 16 && q_s == 48u && !ks_dbg(ctx, KS_DBG_POSTINGS10)) ? 2 : 1) : 0;
            u64 *qk = nullptr;
            u32 *qv = nullptr;
            const u64 *q_lo = nullptr, *q_hi = nullptr;
            SE_HIP(hipMemsetAsync(cursor, 0, ((size_t)JN_SEGS * JN_CUR_STRIDE + (way == 0 ? (n_buckets + 1) / 2 : 0)) * sizeof(u64), ctx->stream));
            if (way == 0) {
                const u64 per = n_q / n_buckets;
                const u32 bcap = (u32)(per + per / 8 + 512);
                SE_CHECK(ks_alloc(ctx, &qk0, (size_t)n_buckets * bcap));
                SE_CHECK(ks_alloc(ctx, &qv0, (size_t)n_buckets * bcap));
                ks_rs_segments seg{q->part_len, q->part_cap, q->part_regions << q->part_sub_shift, q->part_sub_shift};
                SE_CHECK(ks_bucket_scatter_u32(ctx, q->part_keys, q->part_vals, &seg, 8, pfxK, qk0, qv0, bcur, bcap, cursor, n_buckets >> 8,
                                               q_fmt)); // (10-byte postings stay 10 bytes, or lose one more: the join decodes them)
                H->bucket_posting_bytes = q_fmt == 2 ? 9 : (q_fmt == 1 ? 10 : 12);
                ks_timer_begin(ctx, "bucket_dir");
                hipLaunchKernelGGL(k_region_dir, dim3((n_buckets + 255) / 256), dim3(256), 0, ctx->stream, (const u32 *)bcur, (u64)bcap,
                                   n_buckets, dir_q, dir_q + n_buckets, n_buckets >> 8);
                ks_timer_end(ctx);
                qk = qk0; qv = qv0; q_lo = dir_q; q_hi = dir_q + n_buckets;
            } else if (pre && pbits <= 8) {
                // the sketch kernel already wrote one region per bucket
                qk = q->part_keys; qv = q->part_vals;
                ks_timer_begin(ctx, "bucket_dir");
                hipLaunchKernelGGL(k_region_dir, dim3((n_buckets + 255) / 256), dim3(256), 0, ctx->stream, (const u32 *)q->part_len,
                                   q->part_cap, n_buckets, dir_q, dir_q + n_buckets, 0u);
                ks_timer_end(ctx);
                q_lo = dir_q; q_hi = dir_q + n_buckets;
            } else {
                if (!pre) SE_CHECK(ks_sketches_make_dense(ctx, const_cast<ks_sketches *>(q))); // (this partition starts from the CSR)
                SE_CHECK(ks_alloc(ctx, &qk0, (size_t)n_q)); SE_CHECK(ks_alloc(ctx, &qk1, (size_t)n_q));
                SE_CHECK(ks_alloc(ctx, &qv0, (size_t)n_q)); SE_CHECK(ks_alloc(ctx, &qv1, (size_t)n_q));
                if (pre) {
                    // the sketch kernel did the low digit; segmented (histogram + scan) passes on the high bits finish
                    int shifts[2], nsh = 0;
                    for (int sh = 8; sh < pbits; sh += 8) shifts[nsh++] = sh; // bits [8, pbits) of the join prefix
                    ks_rs_segments seg{q->part_len, q->part_cap, q->part_regions << q->part_sub_shift, q->part_sub_shift};
                    SE_CHECK(ks_radix_sort_u32(ctx, KS_SORT_QPART, q->part_keys, q->part_vals, qk1, qv1, qk0, qv0, n_q, shifts, nsh,
                                               &qk, &qv, &seg, pfxK));
                } else {
                    ks_timer_begin(ctx, "fill_query_vals");
                    hipLaunchKernelGGL(k_fill_query_vals, dim3((q->n_seqs + 3) / 4), dim3(256), 0, ctx->stream, (const u64 *)q->d_offsets, q->n_seqs, qv0);
                    ks_timer_end(ctx);
                    SE_HIP(hipGetLastError());
                    // hashes are read straight from the query sketches on the first pass (no staging copy);
                    // qv0 holds the input qids, so the first pass lands in (qk1, qv1)
                    int shifts[3], ns = 0;
                    for (int sh = 0; sh < pbits; sh += 8) shifts[ns++] = sh; // digits of the join prefix, low first
                    SE_CHECK(ks_radix_sort_u32(ctx, KS_SORT_QPART, q->d_hashes, qv0, qk1, qv1, qk0, qv0, n_q, shifts, ns, &qk, &qv,
                                               nullptr, pfxK));
                }
                ks_timer_begin(ctx, "bucket_dir");
                hipLaunchKernelGGL(k_bucket_dir, dim3((n_buckets + 256) / 256), dim3(256), 0, ctx->stream, (const u64 *)qk, n_q, pbits, pfxK, dir_q);
                ks_timer_end(ctx);
                q_lo = dir_q; q_hi = dir_q + 1;
            }
            SE_HIP(hipGetLastError());

            // join, with a retry if the match list outgrows its first guess.  Many (bucket, round) reservations: the list is cut
            // into segments with a cursor each (see the kernel) and made dense by one copy afterwards; few: one cursor, no copy.
            bool overflowed = false;
            {
                const u64 per_bucket = n_q / n_buckets, round = (u64)JN_THREADS * JN_E;
                const u64 reservations = (u64)n_buckets * ((per_bucket + round - 1) / round ? (per_bucket + round - 1) / round : 1);
                n_segs = (ix->fp_layout && reservations >= 8192 && n_buckets >= JN_SEGS && !ks_dbg(ctx, KS_DBG_ONE_CURSOR)) ? JN_SEGS : 1;
                if (ix->fp_layout && ks_dbg(ctx, KS_DBG_JOIN_SEGS) && n_buckets >= JN_SEGS) n_segs = JN_SEGS; // (tests: small inputs through the segmented path)
            }
            seg_cap = n_segs == 1 ? cap : cap / n_segs + cap / n_segs / 8 + 4096;
            if (ks_dbg(ctx, KS_DBG_JOIN_SEG_CAP)) seg_cap = strtoull(ks_dbg(ctx, KS_DBG_JOIN_SEG_CAP), nullptr, 10); // (tests: the retry)
            for (int attempt = 0; attempt < 2; attempt++) {
                SE_CHECK(ks_alloc(ctx, &pk0, (size_t)(seg_cap * n_segs)));
                if (!packed) SE_CHECK(ks_alloc(ctx, &pv0, (size_t)(seg_cap * n_segs)));
                if (attempt > 0) // (attempt 0: cleared with the flag word above; the flag word survives)
                    SE_HIP(hipMemset2DAsync(cursor, (size_t)JN_CUR_STRIDE * sizeof(u64), 0, sizeof(u64), JN_SEGS, ctx->stream));
                ks_timer_begin(ctx, "join_buckets");
                // (few query postings per bucket: the table kernel; KS_DEBUG_JOIN_SPARSE = 0 / 1 forces the choice in the tests)
                const bool sparse = ix->fp_layout && (ks_dbg(ctx, KS_DBG_JOIN_SPARSE) ? atoi(ks_dbg(ctx, KS_DBG_JOIN_SPARSE)) != 0
                                                                                     : n_q / n_buckets <= (u64)JS_QCAP * 3 / 4);
                if (sparse)
                    hipLaunchKernelGGL(q_fmt == 2 ? k_join_sparse<2> : (q_fmt ? k_join_sparse<1> : k_join_sparse<0>), dim3(n_buckets), dim3(JS_THREADS), 0, ctx->stream, (const u64 *)qk,
                                       (const u32 *)qv, (const u32 *)ix->d_fp, (const ks_post *)ix->d_post, (const ks_bmeta *)ix->d_bmeta, q_lo,
                                       q_hi, dir_t, pk0, pv0, seg_cap, cursor, n_segs - 1, tbits, abits, ix->fp_shift);
                else if (ix->fp_layout)
                    hipLaunchKernelGGL(q_fmt == 2 ? k_join_buckets<2> : (q_fmt ? k_join_buckets<1> : k_join_buckets<0>), dim3(n_buckets), dim3(JN_THREADS), 0, ctx->stream, (const u64 *)qk,
                                       (const u32 *)qv, (const u32 *)ix->d_fp, (const ks_post *)ix->d_post, (const ks_bmeta *)ix->d_bmeta, q_lo,
                                       q_hi, dir_t, pk0, pv0, seg_cap, cursor, n_segs - 1, tbits, abits, ix->fp_shift);
                else {
                    // workgroups per bucket: the matches per query posting of this context's previous search, when the buckets
                    // hold enough query postings to share (KS_DEBUG_JOIN_SPLIT forces it)
                    u32 split = 1;
                    if (ctx->pair_cap_hint > n_q && n_q / n_buckets >= 1024) { // (n_q may still be the pending sketch's upper bound)
                        const u64 dens2 = 2 * ctx->pair_cap_hint / n_q;
                        split = dens2 >= 6 ? 4u : (dens2 >= 3 ? 2u : 1u); // 200k x 200k hp: 0.73 / 0.59 / 0.53 / 0.59 ms with 1 / 2 / 4 / 8
                    }
                    if (const char *f = ks_dbg(ctx, KS_DBG_JOIN_SPLIT)) { const int v = atoi(f); if (v >= 1 && v <= 16) split = (u32)v; }
                    hipLaunchKernelGGL(k_join_buckets_keys, dim3(n_buckets * split), dim3(JN_THREADS), 0, ctx->stream, (const u64 *)qk,
                                       (const u32 *)qv, (const u64 *)ix->d_keys, (const u32 *)ix->d_tids, (const u32 *)ix->d_abunds,
                                       q_lo, q_hi, dir_t, pk0, pv0, seg_cap, cursor, tbits, abits, split);
                }
                ks_timer_end(ctx);
                SE_HIP(hipGetLastError());
                // the segment counts (+ the flag word beside the first): one copy, strided when the list is segmented
                // (the kernel that stamps the host's flag writes them to the pinned block itself — with the control block of a
                // pending sketch — instead of a copy dispatch per block in front of it)
                {
                    ks_fetch_seg f[2];
                    f[0] = ks_fetch_seg{cursor, (u32 *)ctx->h_pin, n_segs == 1 ? 1u : (u32)JN_SEGS, 4u, (u32)JN_CUR_STRIDE * 2u};
                    int nf = 1;
                    if (q->pending) f[nf++] = ks_sketch_pending_seg(q);
                    SE_CHECK(ks_stream_wait_fetch(ctx, f, nf));
                }
                SE_FINISH_PENDING();
                n_pairs = 0;
                u64 seg_max = 0;
                for (u32 s_ = 0; s_ < n_segs; s_++) {
                    seg_count[s_] = ctx->h_pin[2 * s_];
                    n_pairs += seg_count[s_];
                    if (seg_count[s_] > seg_max) seg_max = seg_count[s_];
                }
                overflowed = way == 0 && ctx->h_pin[1] != 0;
                if (!overflowed && n_pairs >= KS_PAIR_LIMIT) { // saturated alphabets: the caller searches the queries in slices
                    *split_pairs = n_pairs;
                    split = true;
                    goto done;
                }
                if (overflowed || seg_max <= seg_cap) break;
                if (attempt == 1) {
                    st = ks_fail(ctx, KS_ERR_CAPACITY, "search produced %llu matched posting pairs (segment cap %llu)",
                                 (unsigned long long)n_pairs, (unsigned long long)seg_cap);
                    goto done;
                }
                ks_pool_free(ctx, pk0); ks_pool_free(ctx, pv0); pk0 = nullptr; pv0 = nullptr;
                seg_cap = seg_max; // (same postings, same buckets, same segments: the repeat fits exactly)
                ctx->join_retries++;
            }
            ks_pool_free(ctx, qk0); ks_pool_free(ctx, qk1); ks_pool_free(ctx, qv0); ks_pool_free(ctx, qv1);
            qk0 = qk1 = nullptr; qv0 = qv1 = nullptr;
            if (!overflowed) {
                H->partition_path = way == 0 ? 1 : (pre ? (pbits <= 8 ? 0 : 2) : 3);
                break;
            }
            // a bucket overflowed (skewed hashes): drop the partial result and partition the dense way
            ks_pool_free(ctx, pk0); ks_pool_free(ctx, pv0); pk0 = nullptr; pv0 = nullptr;
        }
        H->n_pair_instances = n_pairs;
        {
            const u64 want = n_pairs + n_pairs / 8;
            ctx->pair_cap_hint = want > ctx->pair_cap_hint / 2 ? want : ctx->pair_cap_hint / 2; // follows growth at once, decays slowly
        }
        if (n_pairs == 0) {
            SE_CHECK(ks_alloc(ctx, &H->d_qid, 1)); SE_CHECK(ks_alloc(ctx, &H->d_tid, 1));
            SE_CHECK(ks_alloc(ctx, &H->d_isect, 1)); SE_CHECK(ks_alloc(ctx, &H->d_nw, 1));
            goto done;
        }
        // sort matches by (qid, tid) on the live id bits only
        SE_CHECK(ks_alloc(ctx, &pk1, (size_t)n_pairs));
        if (!packed) SE_CHECK(ks_alloc(ctx, &pv1, (size_t)n_pairs));
        // Packed records go into the match sort as they lie: its first partition level reads the segments in place.  Only a list
        // that sort declines (short lists, narrow keys: the LSD passes) or unpacked records are made dense by a copy first.
        int msd = 0;
        if (n_segs > 1 && packed) {
            static_assert(JN_SEGS <= KS_MSD_MAX_SEGS, "segment table of the match sort");
            ks_msd_segs sg;
            sg.n = n_segs; sg.seg_cap = seg_cap;
            u32 t = 0;
            for (u32 s_ = 0; s_ < n_segs; s_++) {
                sg.tile_start[s_] = t; sg.count[s_] = (u32)seg_count[s_];
                t += (u32)((seg_count[s_] + 8191) / 8192); // (MS_TILE records per level-1 tile, ks_msd.hip)
            }
            sg.tile_start[n_segs] = t;
            for (u32 s_ = n_segs + 1; s_ <= KS_MSD_MAX_SEGS; s_++) sg.tile_start[s_] = t;
            SE_CHECK(ks_sort_pairs_msd(ctx, pk0, pk1, n_pairs, abits, tbits + qbits, &msd, &sg));
        }
        if (n_segs > 1 && !msd) { // the segments -> one dense list (then the roles of the two buffers swap: the segmented one is the scratch)
            jn_seg_table tab;
            u64 acc = 0;
            for (u32 s_ = 0; s_ < n_segs; s_++) { tab.prefix[s_] = acc; acc += seg_count[s_]; }
            tab.prefix[n_segs] = acc;
            u64 seg_max = 0;
            for (u32 s_ = 0; s_ < n_segs; s_++) if (seg_count[s_] > seg_max) seg_max = seg_count[s_];
            ks_timer_begin(ctx, "pairs_compact");
            hipLaunchKernelGGL(k_pairs_compact, dim3((u32)((seg_max + 2047) / 2048), n_segs), dim3(256), 0, ctx->stream, (const u64 *)pk0,
                               (const u32 *)pv0, seg_cap, tab, pk1, pv1);
            ks_timer_end(ctx);
            SE_HIP(hipGetLastError());
            u64 *tk = pk0; pk0 = pk1; pk1 = tk;
            u32 *tv = pv0; pv0 = pv1; pv1 = tv;
        }
        u64 *pk = nullptr;
        u32 *pv = nullptr;
        {
            int shifts[8], ns = 0;
            for (int sh = 0; sh < tbits + qbits; sh += 8) shifts[ns++] = abits + sh;
            // the match list (pk0, pv0) is scratch from here on: ping-pong with (pk1, pv1).  Packed records: three moves
            // (two exact MSD partition levels + in-LDS bucket sort, ks_msd.hip) instead of one per 8 key bits
            if (packed && !msd) SE_CHECK(ks_sort_pairs_msd(ctx, pk0, pk1, n_pairs, abits, tbits + qbits, &msd));
            if (msd) pk = pk0;
            else if (packed) SE_CHECK(ks_radix_sort_keys(ctx, KS_SORT_PAIRS, pk0, pk0, pk1, n_pairs, shifts, ns, &pk));
            else SE_CHECK(ks_radix_sort_u32(ctx, KS_SORT_PAIRS, pk0, pv0, pk0, pv0, pk1, pv1, n_pairs, shifts, ns, &pk, &pv));
        }
        // run-length reduce: one fused pass for packed records (k_pair_rows_fused); heads + scan + reduce when the
        // abundances travel apart (ids + abundance wider than 64 bits)
        const bool fused = packed && !ks_dbg(ctx, KS_DBG_UNFUSED_ROWS);
        const u32 gp = (u32)((n_pairs + 255) / 256);
        const u32 pf_tiles = (u32)((n_pairs + PF_TILE - 1) / PF_TILE);
        u32 *nrows_dev = nullptr;
        if (fused) { // status words, the ticket pair and the row count in one block WITH the two row arrays the pass adds into
                     // (allocated below, per attempt): one memset, one read-back
        } else {
            SE_CHECK(ks_alloc(ctx, &d_nrows, 1));
            nrows_dev = d_nrows;
            SE_CHECK(ks_alloc(ctx, &heads, (size_t)n_pairs));
            ks_timer_begin(ctx, "pair_heads");
            hipLaunchKernelGGL(k_pair_heads, dim3(gp), dim3(256), 0, ctx->stream, (const u64 *)pk, n_pairs, heads, abits);
            ks_timer_end(ctx);
            SE_CHECK(ks_scan_u32_inplace(ctx, heads, n_pairs, d_nrows));
        }
        // The row count is only known on the device here.  Instead of a round trip before the reduce, the row arrays take
        // their size from the previous search of this context (+ 25 %) and the count is read with the final
        // synchronisation; a search that produced more rows than that repeats the (cheap) reduce with exact arrays.
        u64 rows_cap = n_pairs;
        if (ctx->rows_hint && ctx->rows_hint < rows_cap && !ks_dbg(ctx, KS_DBG_NO_ROWS_HINT)) rows_cap = ctx->rows_hint;
        u32 n_rows = 0;
        for (int attempt = 0; attempt < 3; attempt++) { // (repeats: more rows than the guess; a look-back that gave up)
            SE_CHECK(ks_alloc(ctx, &H->d_qid, (size_t)rows_cap)); SE_CHECK(ks_alloc(ctx, &H->d_tid, (size_t)rows_cap));
            {
                // n_weighted (u64) | status words + ticket pair + row count (u64) | intersect (u32): zeroed together
                const size_t st_words = fused ? (size_t)pf_tiles + 2 : 0, is_words = ((size_t)rows_cap + 1) / 2;
                SE_CHECK(ks_alloc(ctx, &H->d_block, (size_t)rows_cap + st_words + is_words));
                H->d_nw = H->d_block;
                H->d_isect = (u32 *)(H->d_block + rows_cap + st_words);
                if (fused) {
                    pf_status = (unsigned long long *)(H->d_block + rows_cap);
                    pf_ticket = (u32 *)(pf_status + pf_tiles);
                    nrows_dev = pf_ticket + 2;
                }
                SE_HIP(hipMemsetAsync(H->d_block, 0, ((size_t)rows_cap + st_words + is_words) * sizeof(u64), ctx->stream));
            }
            if (fused) {
                ks_timer_begin(ctx, "pair_rows");
                hipLaunchKernelGGL(k_pair_rows_fused, dim3(pf_tiles), dim3(PF_THREADS), 0, ctx->stream, (const u64 *)pk, n_pairs, H->d_qid, H->d_tid,
                                   H->d_isect, (unsigned long long *)H->d_nw, tbits, abits, (u32)rows_cap, pf_status, pf_ticket, nrows_dev,
                                   (ctx->rows_use_ticket || ks_dbg(ctx, KS_DBG_ROWS_TICKET)) ? 1 : 0);
                ks_timer_end(ctx);
            } else {
                ks_timer_begin(ctx, "pair_reduce");
                hipLaunchKernelGGL(k_pair_reduce, dim3(gp), dim3(256), 0, ctx->stream, (const u64 *)pk, (const u32 *)pv, (const u32 *)heads,
                                   n_pairs, H->d_qid, H->d_tid, H->d_isect, (unsigned long long *)H->d_nw, tbits, abits, (u32)rows_cap);
                ks_timer_end(ctx);
            }
            SE_HIP(hipGetLastError());
            {
                ks_fetch_seg f[2];
                f[0] = fused ? ks_fetch_words(pf_ticket, ctx->h_pin + 1, 4) // ticket pair + row count
                             : ks_fetch_words(d_nrows, ctx->h_pin + 2, 1);
                const int nf = ks_scan_status_seg(ctx, &f[1]) ? 2 : 1;
                SE_CHECK(ks_stream_wait_fetch(ctx, f, nf));
            }
            SE_CHECK(ks_scan_status_check(ctx));
            bool gave_up = fused && ((u32 *)(ctx->h_pin + 1))[1] != 0;
            if (fused && ks_dbg(ctx, KS_DBG_FORCE_ROWS_TICKET_RETRY) && !ctx->rows_use_ticket) gave_up = true; // (tests)
            if (gave_up) {
                if (ctx->rows_use_ticket || attempt == 2) { st = ks_fail(ctx, KS_ERR_HIP, "search: row look-back gave up waiting for a predecessor tile"); goto done; }
                ctx->rows_use_ticket = true; // dispatch order did not hold here: tickets from now on
                ctx->rows_ticket_fallbacks++;
            } else {
                n_rows = *(u32 *)(ctx->h_pin + 2); // (fused: third word of the block copied to h_pin + 1; else copied there)
                if (n_rows <= rows_cap) break;
                rows_cap = n_rows;
            }
            ks_pool_free(ctx, H->d_qid); ks_pool_free(ctx, H->d_tid); ks_pool_free(ctx, H->d_block);
            H->d_qid = H->d_tid = H->d_isect = nullptr; H->d_nw = nullptr; H->d_block = nullptr;
        }
        if (!H->d_qid) { st = ks_fail(ctx, KS_ERR_HIP, "search: the row pass did not settle"); goto done; }
        H->n_hits = n_rows;
        {
            const u64 want = (u64)n_rows + n_rows / 4 + 4096;
            ctx->rows_hint = want > ctx->rows_hint / 2 ? want : ctx->rows_hint / 2; // follows growth at once, decays slowly
        }
    }
done:
    ks_pool_free(ctx, qk0); ks_pool_free(ctx, qk1); ks_pool_free(ctx, qv0); ks_pool_free(ctx, qv1);
    ks_pool_free(ctx, pk0); ks_pool_free(ctx, pk1); ks_pool_free(ctx, pv0); ks_pool_free(ctx, pv1);
    ks_pool_free(ctx, heads); ks_pool_free(ctx, d_nrows); ks_pool_free(ctx, row_start); ks_pool_free(ctx, cursor);
    ks_pool_free(ctx, dir_q); // (pf_status lies in the hit object's block)
    if (st != KS_OK || split) { (void)hipStreamSynchronize(ctx->stream); ks_hits_free(H); return st; }
    *out = H;
    return KS_OK;
#undef SE_CHECK
#undef SE_HIP
#undef SE_FINISH_PENDING
}

__global__ __launch_bounds__(256) void k_rebase_offsets(const u64 *offs, u64 base, u32 n, u64 *out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = offs[i] - base;
}
__global__ __launch_bounds__(256) void k_add_u32(u32 *a, u64 n, u32 v) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += v;
}

// ks_search: one match list when it fits; otherwise the query sequences are searched in contiguous slices whose lists
// fit (a slice is a view of the batch's CSR: hits of different query ranges are disjoint and stay ordered by qid).
int ks_search_impl(ks_ctx *ctx, const ks_index *ix, const ks_sketches *q, ks_hits **out, int *sketch_redo) {
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    u64 need = 0;
    int st = search_core(ctx, ix, q, out, &need, sketch_redo);
    if (st != KS_OK || need == 0) return st;
    KS_TRY(ks_sketches_make_dense(ctx, const_cast<ks_sketches *>(q))); // (a slice is a view of the batch's plain CSR)
    // ---- slices of roughly equal posting counts, each expected to produce KS_PAIR_LIMIT / 4 records
    std::vector<u64> offs((size_t)q->n_seqs + 1);
    KS_HIP(ctx, hipMemcpyAsync(offs.data(), q->d_offsets, offs.size() * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 n_slices = need / (KS_PAIR_LIMIT / 4) + 1;
    std::vector<ks_hits *> parts;
    std::vector<u32> firsts;
    auto cleanup = [&]() { for (auto *h : parts) ks_hits_free(h); };
    u32 a = 0;
    while (a < q->n_seqs) {
        const u64 want = q->n_hashes / n_slices + 1;
        u32 b = a + 1;
        while (b < q->n_seqs && offs[b + 1] - offs[a] <= want) b++;
        for (;;) { // search sequences [a, b); halve the slice while it still overflows
            ks_sketches V;
            memset(&V, 0, sizeof V);
            V.ctx = ctx; V.params = q->params; V.n_seqs = b - a; V.n_hashes = offs[b] - offs[a]; V.n_windows = q->n_windows;
            V.d_hashes = q->d_hashes + offs[a]; V.d_abunds = q->d_abunds + offs[a];
            st = ks_alloc(ctx, &V.d_offsets, (size_t)V.n_seqs + 1);
            if (st != KS_OK) { cleanup(); return st; }
            hipLaunchKernelGGL(k_rebase_offsets, dim3((V.n_seqs + 256) / 256), dim3(256), 0, ctx->stream, (const u64 *)q->d_offsets + a, offs[a],
                               V.n_seqs + 1, V.d_offsets);
            ks_hits *h = nullptr;
            u64 more = 0;
            st = search_core(ctx, ix, &V, &h, &more);
            (void)hipStreamSynchronize(ctx->stream);
            ks_pool_free(ctx, V.d_offsets);
            if (st != KS_OK) { cleanup(); return st; }
            if (more == 0) { parts.push_back(h); firsts.push_back(a); break; }
            if (b - a == 1) { cleanup(); return ks_fail(ctx, KS_ERR_CAPACITY, "one query sequence matches %llu postings: beyond the match-list limit", (unsigned long long)more); }
            b = a + (b - a) / 2;
            n_slices *= 2;
        }
        a = b;
    }
    // ---- concatenate (query ids back to batch numbering)
    ks_hits *H = new ks_hits();
    memset(H, 0, sizeof *H);
    H->ctx = ctx;
    H->partition_path = 3;
    for (auto *h : parts) { H->n_hits += h->n_hits; H->n_pair_instances += h->n_pair_instances; }
    const size_t tot = H->n_hits ? (size_t)H->n_hits : 1;
    st = ks_alloc(ctx, &H->d_qid, tot);
    if (st == KS_OK) st = ks_alloc(ctx, &H->d_tid, tot);
    if (st == KS_OK) st = ks_alloc(ctx, &H->d_isect, tot);
    if (st == KS_OK) st = ks_alloc(ctx, &H->d_nw, tot);
    u64 at = 0;
    for (size_t i = 0; i < parts.size() && st == KS_OK; i++) {
        const u64 n = parts[i]->n_hits;
        if (n) {
            hipLaunchKernelGGL(k_add_u32, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, parts[i]->d_qid, n, firsts[i]);
            if (hipMemcpyAsync(H->d_qid + at, parts[i]->d_qid, n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(H->d_tid + at, parts[i]->d_tid, n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(H->d_isect + at, parts[i]->d_isect, n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(H->d_nw + at, parts[i]->d_nw, n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                st = ks_fail(ctx, KS_ERR_HIP, "hit concatenation failed");
            at += n;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    cleanup();
    if (st != KS_OK) { ks_hits_free(H); return st; }
    *out = H;
    return KS_OK;
}

// ---------------------------------------------------------------------------------------------
// union of all sketches with summed abundances: the "combined minhash" of ProteomeIndex::store_signatures
// (src/rust/index.rs:800-830, add_many_with_abund under a mutex there; here one sort + run-length reduce)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_union_emit(const u64 *keys, const u32 *vals, const u64 *row_start, u32 n_rows,
                                                    u64 *hashes, u32 *abunds) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    u64 b = row_start[r], e = row_start[r + 1];
    u64 w = 0;
    for (u64 j = b; j < e; j++) w += vals[j];
    hashes[r] = keys[b];
    abunds[r] = w > 0xffffffffULL ? 0xffffffffu : (u32)w;
}

int ks_union_impl(ks_ctx *ctx, const ks_sketches *in, ks_sketches **out) {
    if (!in || !out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_sketches_make_dense(ctx, const_cast<ks_sketches *>(in)));
    ks_sketches *U = new ks_sketches();
    memset(U, 0, sizeof *U);
    U->ctx = ctx; U->params = in->params; U->n_seqs = 1; U->n_windows = in->n_windows;
    const u64 n = in->n_hashes;
    u64 *k0 = nullptr, *k1 = nullptr, *row_start = nullptr;
    u32 *v0 = nullptr, *v1 = nullptr, *heads = nullptr, *d_nrows = nullptr;
    int st = KS_OK;
#define UN_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
#define UN_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
    UN_CHECK(ks_alloc(ctx, &U->d_offsets, 2));
    if (n == 0) {
        UN_HIP(hipMemsetAsync(U->d_offsets, 0, 2 * sizeof(u64), ctx->stream));
        UN_CHECK(ks_alloc(ctx, &U->d_hashes, 1)); UN_CHECK(ks_alloc(ctx, &U->d_abunds, 1));
        UN_HIP(hipStreamSynchronize(ctx->stream));
        goto done;
    }
    {
        UN_CHECK(ks_alloc(ctx, &k0, (size_t)n)); UN_CHECK(ks_alloc(ctx, &k1, (size_t)n));
        UN_CHECK(ks_alloc(ctx, &v0, (size_t)n)); UN_CHECK(ks_alloc(ctx, &v1, (size_t)n));
        const int shifts[8] = {0, 8, 16, 24, 32, 40, 48, 56};
        u64 *ks = nullptr;
        u32 *vs = nullptr;
        UN_CHECK(ks_radix_sort_u32(ctx, KS_SORT_PAIRS, in->d_hashes, in->d_abunds, k0, v0, k1, v1, n, shifts, 8, &ks, &vs));
        UN_CHECK(ks_alloc(ctx, &heads, (size_t)n));
        UN_CHECK(ks_alloc(ctx, &d_nrows, 1));
        const u32 g = (u32)((n + 255) / 256);
        ks_timer_begin(ctx, "pair_heads");
        hipLaunchKernelGGL(k_pair_heads, dim3(g), dim3(256), 0, ctx->stream, (const u64 *)ks, n, heads, 0);
        ks_timer_end(ctx);
        UN_CHECK(ks_scan_u32_inplace(ctx, heads, n, d_nrows));
        UN_HIP(hipMemcpyAsync(ctx->h_pin, d_nrows, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        UN_HIP(hipStreamSynchronize(ctx->stream));
        const u32 n_rows = *(u32 *)ctx->h_pin;
        U->n_hashes = U->n_slots = n_rows;
        UN_CHECK(ks_alloc(ctx, &row_start, (size_t)n_rows + 1));
        UN_CHECK(ks_alloc(ctx, &U->d_hashes, (size_t)n_rows)); UN_CHECK(ks_alloc(ctx, &U->d_abunds, (size_t)n_rows));
        ks_timer_begin(ctx, "pair_rows");
        hipLaunchKernelGGL(k_pair_rows, dim3(g), dim3(256), 0, ctx->stream, (const u64 *)ks, (const u32 *)heads, n, n_rows, row_start, 0);
        ks_timer_end(ctx);
        ks_timer_begin(ctx, "union_emit");
        hipLaunchKernelGGL(k_union_emit, dim3((n_rows + 255) / 256), dim3(256), 0, ctx->stream, (const u64 *)ks, (const u32 *)vs,
                           (const u64 *)row_start, n_rows, U->d_hashes, U->d_abunds);
        ks_timer_end(ctx);
        UN_HIP(hipGetLastError());
        ctx->h_pin[0] = 0; ctx->h_pin[1] = n_rows;
        UN_HIP(hipMemcpyAsync(U->d_offsets, ctx->h_pin, 2 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        UN_CHECK(ks_scan_status_fetch(ctx));
        UN_HIP(hipStreamSynchronize(ctx->stream));
        UN_CHECK(ks_scan_status_check(ctx));
    }
done:
    ks_pool_free(ctx, k0); ks_pool_free(ctx, k1); ks_pool_free(ctx, v0); ks_pool_free(ctx, v1);
    ks_pool_free(ctx, heads); ks_pool_free(ctx, d_nrows); ks_pool_free(ctx, row_start);
    if (st != KS_OK) { (void)hipStreamSynchronize(ctx->stream); ks_sketches_free(U); return st; }
    *out = U;
    return KS_OK;
#undef UN_CHECK
#undef UN_HIP
}
