// ks_api.hip — the extern "C" entry points of include/kmerseek_amd.h that move data across the
// boundary (host buffers <-> HBM) and own the opaque result objects, plus the k-mer position kernel.
#include "ks_device.h"

// ---------------------------------------------------------------------------------------------
// sketches
// ---------------------------------------------------------------------------------------------
extern "C" int ks_sketch_batch_device(ks_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_seq_offsets,
                                      uint32_t n_seqs, uint64_t n_residues, uint32_t max_seq_len,
                                      const ks_params *params, ks_sketches **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!out || (!d_seq_offsets) || (!d_residues && n_residues)) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    return ks_sketch_device_impl(ctx, d_residues, d_seq_offsets, n_seqs, n_residues, max_seq_len, params, 0, 0, 0, out);
    });
}

extern "C" int ks_sketch_queries_device(ks_ctx *ctx, const ks_index *index, const uint8_t *d_residues,
                                        const uint64_t *d_seq_offsets, uint32_t n_seqs, uint64_t n_residues,
                                        uint32_t max_seq_len, ks_sketches **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!index || !out || (!d_seq_offsets) || (!d_residues && n_residues)) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    return ks_sketch_device_impl(ctx, d_residues, d_seq_offsets, n_seqs, n_residues, max_seq_len, &index->params,
                                 index->pbits, (index->fp_layout && index->fp_shift == 32 - index->pbits) ? 1 : 0, 0, out);
    });
}

static int upload_batch(ks_ctx *ctx, const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_seqs,
                        u8 **d_res, u64 **d_offs, u64 *n_res, u32 *max_len) {
    if (!seq_offsets) return ks_fail(ctx, KS_ERR_INVALID_ARG, "seq_offsets is NULL");
    u64 total = seq_offsets[n_seqs];
    u64 mx = 0;
    for (u32 s = 0; s < n_seqs; s++) {
        if (seq_offsets[s + 1] < seq_offsets[s]) return ks_fail(ctx, KS_ERR_INVALID_ARG, "seq_offsets must be ascending (record %u)", s);
        u64 l = seq_offsets[s + 1] - seq_offsets[s];
        mx = l > mx ? l : mx;
    }
    if (seq_offsets[0] != 0) return ks_fail(ctx, KS_ERR_INVALID_ARG, "seq_offsets[0] must be 0");
    if (mx > 0xfffffff0ULL) return ks_fail(ctx, KS_ERR_INVALID_ARG, "sequence longer than 2^32 residues");
    if (total && !residues) return ks_fail(ctx, KS_ERR_INVALID_ARG, "residues is NULL");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_alloc(ctx, d_res, (size_t)total + 16));
    KS_TRY(ks_alloc(ctx, d_offs, (size_t)n_seqs + 1));
    if (total) KS_TRY(ks_copy_h2d(ctx, *d_res, residues, (size_t)total));
    KS_HIP(ctx, hipMemcpyAsync(*d_offs, seq_offsets, ((size_t)n_seqs + 1) * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    *n_res = total;
    *max_len = (u32)mx;
    return KS_OK;
}

extern "C" int ks_sketch_batch(ks_ctx *ctx, const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_seqs,
                               const ks_params *params, ks_sketches **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "out is NULL");
    KS_TRY(ks_check_params(ctx, params));
    u8 *d_res = nullptr;
    u64 *d_offs = nullptr;
    u64 n_res = 0;
    u32 max_len = 0;
    int st = upload_batch(ctx, residues, seq_offsets, n_seqs, &d_res, &d_offs, &n_res, &max_len);
    if (st == KS_OK) st = ks_sketch_device_impl(ctx, d_res, d_offs, n_seqs, n_res, max_len, params, 0, 0, 0, out);
    (void)hipStreamSynchronize(ctx->stream);
    ks_pool_free(ctx, d_res);
    ks_pool_free(ctx, d_offs);
    return st;
    });
}

// ---- slots -> plain CSR --------------------------------------------------------------------------------------------------
// A sketch call leaves every sequence a slot as long as its KEPT hashes (ks_common.h: ks_sketches); where a sequence repeats a
// k-mer its distinct hashes fill only the head of the slot.  One gather (one wave per sequence) closes the gaps: new offsets =
// exclusive scan of the distinct counts.  Only batches with repeats pay for it, and only when something reads the arrays as a
// plain CSR (copies to the host, the device accessors of the ABI, an index build, a search that starts from the CSR, a union).
__global__ __launch_bounds__(256) void k_dense_gather(const u64 *old_offs, const u64 *new_offs, const u64 *oh, const u32 *oa, u32 n_seqs,
                                                      u64 *nh, u32 *na) {
    const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seqs) return;
    const u64 src = old_offs[s], dst = new_offs[s], n = new_offs[s + 1] - dst;
    for (u64 j = threadIdx.x & 63; j < n; j += 64) { nh[dst + j] = oh[src + j]; na[dst + j] = oa[src + j]; }
}

int ks_sketches_make_dense(ks_ctx *ctx, ks_sketches *S) {
    if (!S || !S->gapped) return KS_OK;
    if (S->pending) return ks_fail(ctx, KS_ERR_INVALID_ARG, "sketches with a pending read-back cannot be made dense");
    if (!S->d_counts) return ks_fail(ctx, KS_ERR_HIP, "internal error: gapped sketches without per-sequence counts");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    u64 *no = nullptr, *nh = nullptr;
    u32 *na = nullptr;
    int st = ks_alloc(ctx, &no, (size_t)S->n_seqs + 1);
    if (st == KS_OK) st = ks_alloc(ctx, &nh, (size_t)S->n_hashes);
    if (st == KS_OK) st = ks_alloc(ctx, &na, (size_t)S->n_hashes);
    if (st == KS_OK) st = ks_scan_u32_to_u64(ctx, S->d_counts, no, S->n_seqs);
    if (st == KS_OK && S->n_seqs) {
        ks_timer_begin(ctx, "dense_gather");
        hipLaunchKernelGGL(k_dense_gather, dim3((S->n_seqs + 3) / 4), dim3(256), 0, ctx->stream, (const u64 *)S->d_offsets, (const u64 *)no,
                           (const u64 *)S->d_hashes, (const u32 *)S->d_abunds, S->n_seqs, nh, na);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "dense gather launch failed");
    }
    if (st == KS_OK) st = ks_scan_status_fetch(ctx);
    if (st == KS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "dense gather failed");
    if (st == KS_OK) st = ks_scan_status_check(ctx);
    if (st != KS_OK) { ks_pool_free(ctx, no); ks_pool_free(ctx, nh); ks_pool_free(ctx, na); return st; }
    ks_pool_free(ctx, S->d_offsets); ks_pool_free(ctx, S->d_hashes); ks_pool_free(ctx, S->d_abunds);
    S->d_offsets = no; S->d_hashes = nh; S->d_abunds = na;
    S->n_slots = S->n_hashes;
    S->gapped = false;
    return KS_OK;
}

extern "C" uint32_t ks_sketches_n_seqs(const ks_sketches *s) { return s ? s->n_seqs : 0; }
extern "C" uint64_t ks_sketches_n_hashes(const ks_sketches *s) { return s ? s->n_hashes : 0; }
extern "C" uint64_t ks_sketches_n_windows(const ks_sketches *s) { return s ? s->n_windows : 0; }
extern "C" int ks_sketches_has_postings(const ks_sketches *s) {
    return ks_guard(nullptr, [&]() -> int { return (s && s->part_keys) ? (s->part_s ? 2 : 1) : 0;
    });
}
extern "C" void ks_sketches_params(const ks_sketches *s, ks_params *out) { if (s && out) *out = s->params; }
// (the arrays the ABI shows are a plain CSR: a batch whose slots have gaps is made dense on first sight — the object is
// logically const: same sketches — and NULL comes back if that fails, with the reason in ks_last_error)
static const ks_sketches *dense_view(const ks_sketches *s) {
    if (!s || !s->gapped) return s;
    ks_sketches *m = const_cast<ks_sketches *>(s);
    return ks_guard(m->ctx, [&]() -> int { return ks_sketches_make_dense(m->ctx, m); }) == KS_OK ? s : nullptr;
}
extern "C" const uint64_t *ks_sketches_device_offsets(const ks_sketches *s) { s = dense_view(s); return s ? s->d_offsets : nullptr; }
extern "C" const uint64_t *ks_sketches_device_hashes(const ks_sketches *s) { s = dense_view(s); return s ? s->d_hashes : nullptr; }
extern "C" const uint32_t *ks_sketches_device_abunds(const ks_sketches *s) { s = dense_view(s); return s ? s->d_abunds : nullptr; }

extern "C" int ks_sketches_copy_to_host(ks_ctx *ctx, const ks_sketches *s, uint64_t *offsets, uint64_t *hashes,
                                        uint32_t *abunds) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !s) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_TRY(ks_sketches_make_dense(ctx, const_cast<ks_sketches *>(s)));
    if (offsets) KS_HIP(ctx, hipMemcpyAsync(offsets, s->d_offsets, ((size_t)s->n_seqs + 1) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    if (hashes && s->n_hashes) KS_TRY(ks_copy_d2h(ctx, hashes, s->d_hashes, (size_t)s->n_hashes * sizeof(u64)));
    if (abunds && s->n_hashes) KS_TRY(ks_copy_d2h(ctx, abunds, s->d_abunds, (size_t)s->n_hashes * sizeof(u32)));
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}

extern "C" int ks_sketches_from_host(ks_ctx *ctx, const uint64_t *offsets, const uint64_t *hashes, const uint32_t *abunds,
                                     uint32_t n_seqs, const ks_params *params, ks_sketches **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!offsets || !out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    KS_TRY(ks_check_params(ctx, params));
    if (offsets[0] != 0) return ks_fail(ctx, KS_ERR_INVALID_ARG, "offsets[0] must be 0");
    const u64 n = offsets[n_seqs];
    if (n && (!hashes || !abunds)) return ks_fail(ctx, KS_ERR_INVALID_ARG, "hashes/abunds is NULL");
    // the index / join arithmetic assumes 0 < h <= max_hash(scaled) (ks_join_prefix < 2^pbits, sort prefix < 2^S): a
    // sketch made with a smaller `scaled` than the one passed here would wrap into foreign buckets and lose matches silently
    const u64 max_hash = ks_max_hash(params->scaled);
    for (u32 s = 0; s < n_seqs; s++) {
        if (offsets[s + 1] < offsets[s]) return ks_fail(ctx, KS_ERR_INVALID_ARG, "offsets must be ascending");
        for (u64 j = offsets[s]; j < offsets[s + 1]; j++) {
            if (hashes[j] == 0 || hashes[j] > max_hash)
                return ks_fail(ctx, KS_ERR_INVALID_ARG, "sketch %u holds hash %llu outside (0, max_hash(scaled=%u)]", s,
                               (unsigned long long)hashes[j], params->scaled);
            if (j > offsets[s] && hashes[j] <= hashes[j - 1])
                return ks_fail(ctx, KS_ERR_INVALID_ARG, "sketch %u is not strictly ascending", s);
        }
    }
    KS_HIP(ctx, hipSetDevice(ctx->device));
    ks_sketches *S = new ks_sketches();
    memset(S, 0, sizeof *S);
    S->ctx = ctx; S->params = *params; S->n_seqs = n_seqs; S->n_hashes = S->n_slots = n; S->n_windows = 0;
    int st = ks_alloc(ctx, &S->d_offsets, (size_t)n_seqs + 1);
    if (st == KS_OK) st = ks_alloc(ctx, &S->d_hashes, (size_t)n);
    if (st == KS_OK) st = ks_alloc(ctx, &S->d_abunds, (size_t)n);
    if (st != KS_OK) { ks_sketches_free(S); return st; }
    hipError_t e = hipMemcpyAsync(S->d_offsets, offsets, ((size_t)n_seqs + 1) * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(S->d_hashes, hashes, (size_t)n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(S->d_abunds, abunds, (size_t)n * sizeof(u32), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ks_sketches_free(S); return ks_fail(ctx, KS_ERR_HIP, "upload failed: %s", hipGetErrorString(e)); }
    *out = S;
    return KS_OK;
    });
}

extern "C" int ks_sketches_union(ks_ctx *ctx, const ks_sketches *in, ks_sketches **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    return ks_union_impl(ctx, in, out);
    });
}

extern "C" void ks_sketches_free(ks_sketches *s) {
    if (!s) return;
    ks_pool_free(s->ctx, s->d_offsets);
    ks_pool_free(s->ctx, s->d_hashes);
    ks_pool_free(s->ctx, s->d_abunds);
    ks_pool_free(s->ctx, s->d_counts);
    ks_pool_free(s->ctx, s->part_keys);
    ks_pool_free(s->ctx, s->part_vals);
    if (s->ctl_block) ks_pool_free(s->ctx, s->ctl_block); // (part_len lies inside it)
    else ks_pool_free(s->ctx, s->part_len);
    delete s;
}

// ---------------------------------------------------------------------------------------------
// k-mer positions (ProteomeIndex::process_kmers, src/rust/index.rs:749-786): k_kmerpos_tiles in ks_sketch.hip.
// NOTE: the Rust path hashes the validated sequence as given (no upper-casing inside process_kmers);
// inputs that reach it are already upper-case, so the LUT's case folding is unobservable.
// ---------------------------------------------------------------------------------------------
int ks_kmerpos_device_impl(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, const ks_params *p,
                           ks_kmerpos **out) {
    KS_TRY(ks_check_params(ctx, p));
    if (((uintptr_t)d_res & 15) != 0) return ks_fail(ctx, KS_ERR_INVALID_ARG, "d_residues must be 16-byte aligned");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    ks_kmerpos *K = new ks_kmerpos();
    memset(K, 0, sizeof *K);
    K->ctx = ctx;
    int st = KS_OK;
    // every residue position starts at most one window: n_res bounds the table
    const size_t cap = (n_res == 0 || n_seqs == 0) ? 1 : (size_t)n_res;
    st = ks_alloc(ctx, &K->d_seq, cap);
    if (st == KS_OK) st = ks_alloc(ctx, &K->d_start, cap);
    if (st == KS_OK) st = ks_alloc(ctx, &K->d_hash, cap);
    if (st == KS_OK && n_res != 0 && n_seqs != 0)
        st = ks_kmerpos_tiles_launch(ctx, d_res, d_offs, n_seqs, n_res, p, K->d_seq, K->d_start, K->d_hash, &K->n);
    if (st != KS_OK) { (void)hipStreamSynchronize(ctx->stream); ks_kmerpos_free(K); return st; }
    *out = K;
    return KS_OK;
}

extern "C" int ks_kmer_positions(ks_ctx *ctx, const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_seqs,
                                 const ks_params *params, ks_kmerpos **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "out is NULL");
    KS_TRY(ks_check_params(ctx, params));
    u8 *d_res = nullptr;
    u64 *d_offs = nullptr;
    u64 n_res = 0;
    u32 max_len = 0;
    int st = upload_batch(ctx, residues, seq_offsets, n_seqs, &d_res, &d_offs, &n_res, &max_len);
    if (st == KS_OK) st = ks_kmerpos_device_impl(ctx, d_res, d_offs, n_seqs, n_res, params, out);
    (void)hipStreamSynchronize(ctx->stream);
    ks_pool_free(ctx, d_res);
    ks_pool_free(ctx, d_offs);
    return st;
    });
}

extern "C" int ks_kmer_positions_device(ks_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_seq_offsets, uint32_t n_seqs,
                                        uint64_t n_residues, const ks_params *params, ks_kmerpos **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "out is NULL");
    if (n_seqs && (!d_seq_offsets || (n_residues && !d_residues))) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL device buffer");
    return ks_kmerpos_device_impl(ctx, d_residues, d_seq_offsets, n_seqs, n_residues, params, out);
    });
}

extern "C" uint64_t ks_kmerpos_count(const ks_kmerpos *p) { return p ? p->n : 0; }

extern "C" int ks_kmerpos_copy_to_host(ks_ctx *ctx, const ks_kmerpos *p, uint32_t *seq, uint32_t *start, uint64_t *hash) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !p) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    if (p->n) {
        if (seq) KS_TRY(ks_copy_d2h(ctx, seq, p->d_seq, (size_t)p->n * sizeof(u32)));
        if (start) KS_TRY(ks_copy_d2h(ctx, start, p->d_start, (size_t)p->n * sizeof(u32)));
        if (hash) KS_TRY(ks_copy_d2h(ctx, hash, p->d_hash, (size_t)p->n * sizeof(u64)));
    }
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}

extern "C" void ks_kmerpos_free(ks_kmerpos *p) {
    if (!p) return;
    ks_pool_free(p->ctx, p->d_seq);
    ks_pool_free(p->ctx, p->d_start);
    ks_pool_free(p->ctx, p->d_hash);
    delete p;
}

// ---------------------------------------------------------------------------------------------
// index + search
// ---------------------------------------------------------------------------------------------
extern "C" int ks_index_build(ks_ctx *ctx, const ks_sketches *targets, ks_index **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    return ks_index_build_impl(ctx, targets, out);
    });
}
extern "C" uint32_t ks_index_n_targets(const ks_index *ix) { return ix ? ix->n_targets : 0; }
extern "C" uint64_t ks_index_n_postings(const ks_index *ix) { return ix ? ix->n_postings : 0; }
extern "C" void ks_index_free(ks_index *ix) {
    if (!ix) return;
    ks_pool_free(ix->ctx, ix->d_keys);
    ks_pool_free(ix->ctx, ix->d_tids);
    ks_pool_free(ix->ctx, ix->d_abunds);
    ks_pool_free(ix->ctx, ix->d_fp);
    ks_pool_free(ix->ctx, ix->d_post);
    ks_pool_free(ix->ctx, ix->d_bmeta);
    ks_pool_free(ix->ctx, ix->d_dir);
    delete ix;
}

extern "C" int ks_search(ks_ctx *ctx, const ks_index *index, const ks_sketches *queries, ks_hits **out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    return ks_search_impl(ctx, index, queries, out);
    });
}
// One call for "sketch this query batch and search it": the sketch launches are queued WITHOUT the wait at their end, the
// search's partition and join follow on the same stream with sizes taken from upper bounds, and the search's first wait
// brings the sketch's control block back together with the join's counts — two waits per step instead of three.  A batch whose
// sketch has to be repeated (an economy that did not fit, dropped postings, a look-back that gave up: all rare) is simply
// done again with the two plain calls.
static int sketch_search_impl(ks_ctx *ctx, const ks_index *index, const uint8_t *d_residues, const uint64_t *d_seq_offsets, uint32_t n_seqs,
                              uint64_t n_residues, uint32_t max_seq_len, ks_sketches **sketches_out, ks_hits **hits_out) {
    *hits_out = nullptr;
    if (sketches_out) *sketches_out = nullptr;
    const int fmt10 = (index->fp_layout && index->fp_shift == 32 - index->pbits) ? 1 : 0;
    ks_sketches *S = nullptr;
    ks_hits *H = nullptr;
    int st = ks_sketch_device_impl(ctx, d_residues, d_seq_offsets, n_seqs, n_residues, max_seq_len, &index->params, index->pbits, fmt10,
                                   ks_dbg(ctx, KS_DBG_NO_DEFER) ? 0 : 1, &S);
    if (st != KS_OK) return st;
    const bool deferred = S->pending != 0;
    int redo = 0;
    st = ks_search_impl(ctx, index, S, &H, &redo);
    if (S->pending) { // (the search failed before its first wait)
        const ks_fetch_seg f = ks_sketch_pending_seg(S);
        int r2 = 0;
        int st2 = ks_stream_wait_fetch(ctx, &f, 1);
        if (st2 == KS_OK) st2 = ks_sketch_finish_pending(S, &r2);
        if (st == KS_OK) { st = st2; redo = r2; }
    }
    if (st == KS_OK && redo) { // the plain way: the sketch call repeats what it has to, the search starts from what it gets
        ks_hits_free(H); H = nullptr;
        ks_sketches_free(S); S = nullptr;
        ctx->fused_redos++;
        st = ks_sketch_device_impl(ctx, d_residues, d_seq_offsets, n_seqs, n_residues, max_seq_len, &index->params, index->pbits, fmt10, 0, &S);
        if (st == KS_OK) st = ks_search_impl(ctx, index, S, &H);
    }
    if (st != KS_OK) { ks_hits_free(H); ks_sketches_free(S); return st; }
    if (deferred) ctx->fused_deferred++;
    *hits_out = H;
    if (sketches_out) *sketches_out = S; else ks_sketches_free(S);
    return KS_OK;
}

extern "C" int ks_sketch_search_device(ks_ctx *ctx, const ks_index *index, const uint8_t *d_residues,
                                       const uint64_t *d_seq_offsets, uint32_t n_seqs, uint64_t n_residues,
                                       uint32_t max_seq_len, ks_sketches **sketches_out, ks_hits **hits_out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!index || !hits_out || (!d_seq_offsets) || (!d_residues && n_residues)) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    return sketch_search_impl(ctx, index, d_residues, d_seq_offsets, n_seqs, n_residues, max_seq_len, sketches_out, hits_out);
    });
}

// ... and from host arrays (the batch is uploaded, its longest record is known from the offsets: the read-back is always folded)
extern "C" int ks_sketch_search(ks_ctx *ctx, const ks_index *index, const uint8_t *residues, const uint64_t *seq_offsets,
                                uint32_t n_seqs, ks_sketches **sketches_out, ks_hits **hits_out) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!index || !hits_out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    u8 *d_res = nullptr;
    u64 *d_offs = nullptr;
    u64 n_res = 0;
    u32 max_len = 0;
    int st = upload_batch(ctx, residues, seq_offsets, n_seqs, &d_res, &d_offs, &n_res, &max_len);
    if (st == KS_OK) st = sketch_search_impl(ctx, index, d_res, d_offs, n_seqs, n_res, max_len, sketches_out, hits_out);
    (void)hipStreamSynchronize(ctx->stream);
    ks_pool_free(ctx, d_res);
    ks_pool_free(ctx, d_offs);
    return st;
    });
}
extern "C" uint64_t ks_hits_count(const ks_hits *h) { return h ? h->n_hits : 0; }
extern "C" uint64_t ks_hits_n_pair_instances(const ks_hits *h) { return h ? h->n_pair_instances : 0; }
extern "C" int ks_hits_partition_path(const ks_hits *h) {
    return ks_guard(nullptr, [&]() -> int { return h ? h->partition_path : -1; 
    });
}
extern "C" int ks_hits_bucket_posting_bytes(const ks_hits *h) {
    return ks_guard(nullptr, [&]() -> int { return h ? h->bucket_posting_bytes : -1;
    });
}
extern "C" int ks_hits_copy_to_host(ks_ctx *ctx, const ks_hits *h, uint32_t *qid, uint32_t *tid, uint32_t *intersect,
                                    uint64_t *n_weighted) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !h) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)h->n_hits;
    if (n) {
        if (qid) KS_TRY(ks_copy_d2h(ctx, qid, h->d_qid, n * sizeof(u32)));
        if (tid) KS_TRY(ks_copy_d2h(ctx, tid, h->d_tid, n * sizeof(u32)));
        if (intersect) KS_TRY(ks_copy_d2h(ctx, intersect, h->d_isect, n * sizeof(u32)));
        if (n_weighted) KS_TRY(ks_copy_d2h(ctx, n_weighted, h->d_nw, n * sizeof(u64)));
    }
    KS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KS_OK;
    });
}
extern "C" const uint32_t *ks_hits_device_qid(const ks_hits *h) { return h ? h->d_qid : nullptr; }
extern "C" const uint32_t *ks_hits_device_tid(const ks_hits *h) { return h ? h->d_tid : nullptr; }
extern "C" const uint32_t *ks_hits_device_intersect(const ks_hits *h) { return h ? h->d_isect : nullptr; }
extern "C" const uint64_t *ks_hits_device_n_weighted(const ks_hits *h) { return h ? h->d_nw : nullptr; }

__global__ __launch_bounds__(256) void k_copy_add_u32(const u32 *in, u32 *out, u64 n, u32 add) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}

extern "C" int ks_hits_copy_to_device(ks_ctx *ctx, const ks_hits *h, uint32_t qid_base, uint32_t tid_base, uint32_t *d_qid,
                                      uint32_t *d_tid, uint32_t *d_intersect, uint64_t *d_n_weighted) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !h) return KS_ERR_INVALID_ARG;
    KS_HIP(ctx, hipSetDevice(ctx->device));
    const u64 n = h->n_hits;
    if (n == 0) return KS_OK;
    const u32 g = (u32)((n + 255) / 256);
    if (d_qid) KS_LAUNCH(ctx, "hits_copy", k_copy_add_u32, g, 256, (const u32 *)h->d_qid, d_qid, n, qid_base);
    if (d_tid) KS_LAUNCH(ctx, "hits_copy", k_copy_add_u32, g, 256, (const u32 *)h->d_tid, d_tid, n, tid_base);
    if (d_intersect) KS_HIP(ctx, hipMemcpyAsync(d_intersect, h->d_isect, (size_t)n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
    if (d_n_weighted) KS_HIP(ctx, hipMemcpyAsync(d_n_weighted, h->d_nw, (size_t)n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    return KS_OK;
    });
}

// ---- packed transport records (multi-GPU hit exchange) ---------------------------------------------------------------
// A COO row is 20 bytes (qid u32, tid u32, intersect u32, n_weighted u64); an all-vs-all of 200k proteins gathers 31 M of
// them, and over xGMI the exchange — not the kernels — is the step.  For transport a row travels as ONE 64-bit word
//     qid << (tbits + 2v) | tid << 2v | intersect << v | n_weighted,      v = (64 - qbits - tbits) / 2 value bits,
// ids in global numbering.  A row whose intersect or n_weighted does not fit v bits carries all-ones in both value fields and
// its true values go to a short escape list (row index, intersect, n_weighted), gathered beside the words.
__global__ __launch_bounds__(256) void k_hits_pack64(const u32 *qid, const u32 *tid, const u32 *isect, const u64 *nw, u64 n, u32 qid_base,
                                                     u32 tid_base, int qbits, int tbits, int vbits, u64 *packed, u32 *esc_row, u32 *esc_isect,
                                                     u64 *esc_nw, u32 *n_esc, u32 esc_cap) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 vmax = (1ULL << vbits) - 1ULL;
    u64 a = isect[i], b = nw[i];
    if (a >= vmax || b >= vmax) { // (all-ones itself is the escape marker, so a value equal to it escapes too)
        const u32 e = atomicAdd(n_esc, 1u);
        if (e < esc_cap) { esc_row[e] = (u32)i; esc_isect[e] = (u32)a; esc_nw[e] = b; }
        a = vmax; b = vmax;
    }
    const u64 q = (u64)qid[i] + qid_base, t = (u64)tid[i] + tid_base;
    // an id that does not fit its field (id_counts smaller than the real global range) would spill into its neighbour:
    // the escape count is pushed beyond any capacity instead, so every rank takes the unpacked exchange
    // (tested against qbits itself: 64 - tbits - 2v is one bit wider when 64 - qbits - tbits is odd, and the receiving side
    // masks with qbits)
    if ((q >> qbits) != 0 || (t >> tbits) != 0) atomicOr(n_esc, 0x80000000u);
    packed[i] = (((q << tbits) | t) << (2 * vbits)) | (a << vbits) | b;
}

extern "C" int ks_hits_pack64_to_device(ks_ctx *ctx, const ks_hits *h, uint32_t qid_base, uint32_t tid_base, int qbits, int tbits,
                                        uint64_t *d_packed, uint32_t *d_esc_row, uint32_t *d_esc_intersect, uint64_t *d_esc_n_weighted,
                                        uint32_t *d_n_esc, uint32_t esc_cap) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx || !h) return KS_ERR_INVALID_ARG;
    if (qbits < 1 || tbits < 1 || qbits + tbits > 48) return ks_fail(ctx, KS_ERR_INVALID_ARG, "pack64: %d + %d id bits leave fewer than 8 value bits", qbits, tbits);
    KS_HIP(ctx, hipSetDevice(ctx->device));
    const u64 n = h->n_hits;
    if (!d_n_esc) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    // the escape counter is cleared HERE, on the context's stream: a caller that zeroes its buffer on another stream
    // (torch's) is not ordered before this launch
    KS_HIP(ctx, hipMemsetAsync(d_n_esc, 0, sizeof(u32), ctx->stream));
    if (n == 0) return KS_OK;
    if (n >= 0xffffffffULL) return ks_fail(ctx, KS_ERR_CAPACITY, "pack64: %llu rows (escape rows are 32-bit indices)", (unsigned long long)n);
    if (!d_packed || (esc_cap && (!d_esc_row || !d_esc_intersect || !d_esc_n_weighted))) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    KS_LAUNCH(ctx, "hits_pack", k_hits_pack64, (u32)((n + 255) / 256), 256, (const u32 *)h->d_qid, (const u32 *)h->d_tid, (const u32 *)h->d_isect,
              (const u64 *)h->d_nw, n, qid_base, tid_base, qbits, tbits, (64 - qbits - tbits) / 2, d_packed, d_esc_row, d_esc_intersect, d_esc_n_weighted,
              d_n_esc, esc_cap);
    return KS_OK;
    });
}

// the inverse, for the receiving side of the exchange: words -> columns (escaped rows keep the all-ones markers: the caller
// patches them from the escape lists)
__global__ __launch_bounds__(256) void k_hits_unpack64(const u64 *packed, u64 n, int qbits, int tbits, int vbits, u32 *qid, u32 *tid,
                                                       u32 *isect, u64 *nw) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 w = packed[i], vmax = (1ULL << vbits) - 1ULL;
    qid[i] = (u32)((w >> (tbits + 2 * vbits)) & ((1ULL << qbits) - 1ULL));
    tid[i] = (u32)((w >> (2 * vbits)) & ((1ULL << tbits) - 1ULL));
    isect[i] = (u32)((w >> vbits) & vmax);
    nw[i] = w & vmax;
}

extern "C" int ks_hits_unpack64_device(ks_ctx *ctx, const uint64_t *d_packed, uint64_t n, int qbits, int tbits, uint32_t *d_qid,
                                       uint32_t *d_tid, uint32_t *d_intersect, uint64_t *d_n_weighted) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (qbits < 1 || tbits < 1 || qbits + tbits > 48) return ks_fail(ctx, KS_ERR_INVALID_ARG, "unpack64: %d + %d id bits leave fewer than 8 value bits", qbits, tbits);
    if (n == 0) return KS_OK;
    if (!d_packed || !d_qid || !d_tid || !d_intersect || !d_n_weighted) return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    if (n > 0xffffffffULL * 256ULL) return ks_fail(ctx, KS_ERR_INVALID_ARG, "unpack64: too many rows");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    KS_LAUNCH(ctx, "hits_unpack", k_hits_unpack64, (u32)((n + 255) / 256), 256, (const u64 *)d_packed, n, qbits, tbits, (64 - qbits - tbits) / 2,
              d_qid, d_tid, d_intersect, (u64 *)d_n_weighted);
    return KS_OK;
    });
}

// ---------------------------------------------------------------------------------------------
// Index-sharded exchange, global (qid, tid) order: the gathered list is W rank blocks, each ordered by (qid, tid) with the
// ranks' target ranges ascending — so the merged order is "by qid, then by rank", and a row's place follows from counting,
// not from sorting 10^7 keys: per (query, rank) the block's rows of that query are one run (found by binary search),
// an exclusive scan over (qid-major, rank-minor) run lengths gives every run its start, one pass moves the rows.
// Replaces a stable device sort on qid (torch.sort: ~8 passes over keys + 4 gathers).  Reference semantics: branchwater
// manysearch rows per query (src/python/kmerseek/search.py:125-141).
// ---------------------------------------------------------------------------------------------
struct hm_blocks { u64 base[65]; u32 n; }; // rank r's rows are [base[r], base[r + 1])

// first[r * (nq + 1) + q] = first row of block r whose qid is >= q (local to the block)
__global__ __launch_bounds__(256) void k_hm_starts(const u32 *qid, hm_blocks B, u32 nq, u32 *first) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 per = (u64)nq + 1;
    if (i >= per * B.n) return;
    const u32 r = (u32)(i / per), q = (u32)(i % per);
    u64 lo = B.base[r], hi = B.base[r + 1];
    while (lo < hi) {
        const u64 mid = lo + ((hi - lo) >> 1);
        if (qid[mid] < q) lo = mid + 1; else hi = mid;
    }
    first[i] = (u32)(lo - B.base[r]);
}
// run[q * W + r] = rows of query q in block r
__global__ __launch_bounds__(256) void k_hm_runs(const u32 *first, u32 W, u32 nq, u32 *run) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)nq * W) return;
    const u32 q = (u32)(i / W), r = (u32)(i % W);
    const u64 at = (u64)r * ((u64)nq + 1) + q;
    run[i] = first[at + 1] - first[at];
}
__global__ __launch_bounds__(256) void k_hm_move(const u32 *qid, const u32 *tid, const u32 *isect, const u64 *nw, hm_blocks B, u32 nq,
                                                 const u32 *first, const u32 *start, u32 *o_qid, u32 *o_tid, u32 *o_isect, u64 *o_nw,
                                                 u32 *n_dropped) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.base[B.n]) return;
    u32 r = 0;
    while (r + 1 < B.n && i >= B.base[r + 1]) r++; // (W <= 64 blocks)
    const u32 q = qid[i];
    if (q >= nq) { atomicAdd(n_dropped, 1u); return; } // (an id beyond the declared range: not written out of bounds, but counted — the call fails)
    const u64 pos = (u64)start[(u64)q * B.n + r] + ((i - B.base[r]) - first[(u64)r * ((u64)nq + 1) + q]);
    o_qid[pos] = q; o_tid[pos] = tid[i]; o_isect[pos] = isect[i]; o_nw[pos] = nw[i];
}

extern "C" int ks_hits_merge_by_qid_device(ks_ctx *ctx, const uint32_t *d_qid, const uint32_t *d_tid, const uint32_t *d_intersect,
                                           const uint64_t *d_n_weighted, const uint64_t *block_rows, uint32_t n_blocks, uint32_t n_queries,
                                           uint32_t *d_out_qid, uint32_t *d_out_tid, uint32_t *d_out_intersect, uint64_t *d_out_n_weighted) {
    return ks_guard(ctx, [&]() -> int {
    if (!ctx) return KS_ERR_INVALID_ARG;
    if (!block_rows || n_blocks == 0 || n_blocks > 64) return ks_fail(ctx, KS_ERR_INVALID_ARG, "merge: 1 .. 64 rank blocks");
    hm_blocks B;
    B.n = n_blocks; B.base[0] = 0;
    for (u32 r = 0; r < n_blocks; r++) B.base[r + 1] = B.base[r] + block_rows[r];
    const u64 n = B.base[n_blocks];
    if (n == 0) return KS_OK;
    if (n >= 0xffffffffULL) return ks_fail(ctx, KS_ERR_CAPACITY, "merge: %llu rows", (unsigned long long)n);
    if (!d_qid || !d_tid || !d_intersect || !d_n_weighted || !d_out_qid || !d_out_tid || !d_out_intersect || !d_out_n_weighted)
        return ks_fail(ctx, KS_ERR_INVALID_ARG, "NULL argument");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    u32 *first = nullptr, *run = nullptr;
    const u64 n_first = ((u64)n_queries + 1) * n_blocks, n_run = (u64)n_queries * n_blocks;
    int st = ks_alloc(ctx, &first, (size_t)n_first);
    if (st == KS_OK) st = ks_alloc(ctx, &run, (size_t)n_run + 2); // (+ the scan's total, + the count of rows with an id out of range)
    u32 *const n_dropped = run ? run + n_run + 1 : nullptr;
    if (st == KS_OK && hipMemsetAsync(n_dropped, 0, sizeof(u32), ctx->stream) != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "merge: memset failed");
    if (st == KS_OK) {
        ks_timer_begin(ctx, "hits_merge");
        hipLaunchKernelGGL(k_hm_starts, dim3((u32)((n_first + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_qid, B, n_queries, first);
        if (n_run) hipLaunchKernelGGL(k_hm_runs, dim3((u32)((n_run + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)first, n_blocks, n_queries, run);
        ks_timer_end(ctx);
        if (n_run) st = ks_scan_u32_inplace(ctx, run, n_run, nullptr);
    }
    if (st == KS_OK) {
        ks_timer_begin(ctx, "hits_merge");
        hipLaunchKernelGGL(k_hm_move, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_qid, (const u32 *)d_tid,
                           (const u32 *)d_intersect, (const u64 *)d_n_weighted, B, n_queries, (const u32 *)first, (const u32 *)run, d_out_qid,
                           d_out_tid, d_out_intersect, (u64 *)d_out_n_weighted, n_dropped);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "merge launch failed");
    }
    if (st == KS_OK) { // a row whose query id lies beyond n_queries has no place in the merged order: the outputs would hold a gap
        const ks_fetch_seg f = ks_fetch_words(n_dropped, ctx->h_pin, 1);
        st = ks_stream_wait_fetch(ctx, &f, 1);
        if (st == KS_OK && *(const u32 *)ctx->h_pin != 0)
            st = ks_fail(ctx, KS_ERR_INVALID_ARG, "merge: %u rows carry a query id >= n_queries = %u", *(const u32 *)ctx->h_pin, n_queries);
    }
    // (the scratch blocks go back to the pool in stream order: the next allocation on this context's stream comes behind the kernels)
    ks_pool_free(ctx, first); ks_pool_free(ctx, run);
    return st;
    });
}

extern "C" void ks_hits_free(ks_hits *h) {
    if (!h) return;
    ks_pool_free(h->ctx, h->d_qid);
    ks_pool_free(h->ctx, h->d_tid);
    if (h->d_block) ks_pool_free(h->ctx, h->d_block); // (d_isect and d_nw lie inside it)
    else { ks_pool_free(h->ctx, h->d_isect); ks_pool_free(h->ctx, h->d_nw); }
    delete h;
}
