/*
 * kmerseek_amd.h — C ABI of the MI355X (gfx950) protein k-mer sketch-and-search engine.
 *
 * This is the drop-in boundary for the one hot path of seanome/kmerseek:
 *   sketch : sliding-window k-mer -> reduced-alphabet re-encode -> MurmurHash3_x64_128.h1 (seed 42)
 *            -> FracMinHash keep-below-threshold -> per-sequence sorted unique hashes + abundances
 *   search : sorted-hash set intersection of every query sketch against an index of target sketches
 *
 * The reference has no FFI seam of its own for this path (sourmash / branchwater are linked-in
 * Rust crates).  Each entry point below names the reference interface it replaces; a Rust
 * `extern "C"` block or a Python ctypes stub binds them 1:1.
 *
 * Conventions
 *   - every function returns a ks_status (0 = ok) unless documented otherwise; nothing throws
 *     or aborts across this boundary — every entry point that can allocate runs its body inside an
 *     exception guard (std::bad_alloc -> KS_ERR_OOM, anything else -> KS_ERR_HIP "internal error");
 *     ks_last_error(ctx) gives the message of the last failure.  Reference convention: every failure
 *     is a value (src/rust/errors.rs:8-24).
 *   - a ks_ctx owns one HIP device + one stream + a grow-only device workspace.  It is NOT
 *     thread-safe: use one context per host thread (the reference's `&self` + rayon fan-out,
 *     src/rust/index.rs:990-1005, becomes one batched call).
 *   - inputs are caller-owned and only read during the call; outputs are library-owned opaque
 *     objects, released with the matching *_free.  All calls are synchronous on return unless
 *     the name ends in _async.
 *   - there is no CPU fallback: without a HIP device ks_ctx_create fails with KS_ERR_NO_DEVICE.
 */
#ifndef KMERSEEK_AMD_H
#define KMERSEEK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KS_ABI_VERSION 1

typedef enum ks_status {
    KS_OK = 0,
    KS_ERR_INVALID_MOLTYPE = 1, /* IndexError::InvalidMoltype, src/rust/errors.rs:8-9; text of encoding.rs:22-25 */
    KS_ERR_INVALID_KSIZE = 2,   /* IndexError::InvalidKsize, src/rust/errors.rs:20-21 */
    KS_ERR_INVALID_RESIDUE = 3, /* IndexError::InvalidAminoAcid(char, pos), src/rust/errors.rs:14-15 */
    KS_ERR_INVALID_ARG = 4,
    KS_ERR_OOM = 5,
    KS_ERR_HIP = 6,
    KS_ERR_NO_DEVICE = 7,
    KS_ERR_CAPACITY = 8,        /* a device-side list (hit pairs) outgrew its hard cap */
    KS_ERR_INVALID_SCALED = 9
} ks_status;

/* get_hash_function_from_moltype / get_encoding_fn_from_moltype, src/rust/encoding.rs:17-53;
 * PyProteinEncoding Raw/Dayhoff/HP, src/rust/lib.rs:29-65. */
typedef enum ks_moltype { KS_PROTEIN = 0, KS_DAYHOFF = 1, KS_HP = 2 } ks_moltype;

#define KS_SEED_DEFAULT 42ull /* pub const SEED, src/rust/signature.rs:12 */
#define KS_MAX_KSIZE 128u

/* KmerMinHash::new(scaled, 3*ksize, hashfn, seed, track_abundance=true, num=0),
 * src/rust/signature.rs:120-131.  ksize is the PROTEIN k (the sourmash ksize is 3*ksize). */
typedef struct ks_params {
    uint32_t ksize;   /* 1..KS_MAX_KSIZE */
    uint32_t scaled;  /* >= 1; max_hash as ks_max_hash() */
    uint32_t moltype; /* ks_moltype */
    uint32_t flags;   /* reserved, 0 */
    uint64_t seed;    /* KS_SEED_DEFAULT */
} ks_params;

typedef struct ks_ctx ks_ctx;
typedef struct ks_sketches ks_sketches; /* device-resident CSR: offsets[n+1], hashes u64, abund u32 */
typedef struct ks_index ks_index;       /* device-resident inverted index: postings sorted by hash */
typedef struct ks_hits ks_hits;         /* device-resident COO (qid, tid, intersect, n_weighted) */
typedef struct ks_kmerpos ks_kmerpos;   /* device-resident (seq, start, hash) triples */

/* ---- library / context ------------------------------------------------------------------ */

uint32_t ks_abi_version(void);
const char *ks_status_string(int status);

/* "protein" | "raw" | "hp" | "dayhoff" -> ks_moltype (src/rust/encoding.rs:17-27).
 * Returns KS_ERR_INVALID_MOLTYPE for anything else. */
int ks_moltype_from_string(const char *name, uint32_t *moltype_out);

/* sourmash max_hash_for_scaled as used by KmerMinHash::new (src/rust/signature.rs:124-131). */
uint64_t ks_max_hash(uint32_t scaled);

/* hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the context create its own non-blocking stream.  The device's default (null) stream cannot be named by NULL here:
 * pass hipStreamLegacy ((hipStream_t)1) for it — kmerseek_amd/engine.py does that for a caller that hands over the 0
 * torch reports for its default stream. */
int ks_ctx_create(int device, void *hip_stream, ks_ctx **out);
void ks_ctx_destroy(ks_ctx *ctx);
const char *ks_last_error(const ks_ctx *ctx);
void *ks_ctx_stream(const ks_ctx *ctx); /* the hipStream_t all kernels are launched on */
int ks_ctx_synchronize(ks_ctx *ctx);
/* device workspace pool: blocks held, bytes held / in use, hipMalloc calls so far (0 new ones in steady state) */
int ks_ctx_pool_stats(const ks_ctx *ctx, uint64_t *n_blocks, uint64_t *bytes_held, uint64_t *bytes_in_use,
                      uint64_t *n_mallocs);

/* How often this context had to repeat a sketch batch (results are identical either way; a repeat only costs time):
 * out[0] = launches repeated because a tile look-back gave up waiting (dispatch order was not blockIdx order; the context
 *          then takes tile ids from an atomic ticket for good: out[1] = 1),
 * out[2] = batches repeated with plain tiles because a compacting tile (scaled > 1) kept more hashes than its LDS lists take,
 * out[3] = batches repeated with window-count sized outputs because they kept more hashes than the expected 1/scaled. */
int ks_ctx_sketch_stats(const ks_ctx *ctx, uint64_t out[4]);
/* The same for ks_search: out[0] = searches that ran their join twice because the match list outgrew its first guess
 * (the list is sized from the previous search of the context), out[1] = searches whose row pass was repeated with
 * ticket-ordered tiles because a look-back gave up (the context then uses tickets for good). */
int ks_ctx_search_stats(const ks_ctx *ctx, uint64_t out[2]);
/* ks_sketch_search_device on this context: out[0] = calls whose sketch read-back was folded into the search's first wait,
 * out[1] = calls that had to be repeated with the two plain calls (skewed hashes, an economy that did not fit). */
int ks_ctx_fused_stats(const ks_ctx *ctx, uint64_t out[2]);
/* Diagnostics: the KS_DEBUG_* environment variables (they force the rarely taken paths in the tests; results never
 * depend on them) are read once, when the context is created — never on the per-call path.  This reads them again. */
int ks_ctx_reload_debug_env(ks_ctx *ctx);
/* Self-test of the exception guard, callable without a device: a body that throws `what` ("bad_alloc", "runtime", "other";
 * anything else throws nothing) runs behind the guard; returns the status that came out of it. */
int ks_debug_guard_selftest(const char *what);

/* Plain device buffers for callers that have no HIP binding of their own (the *_device entry points take raw
 * device pointers): 256-byte aligned allocations on ctx's device, stream-ordered copies that return when done. */
int ks_dev_malloc(ks_ctx *ctx, uint64_t bytes, void **out);
int ks_dev_free(ks_ctx *ctx, void *ptr);
int ks_dev_upload(ks_ctx *ctx, void *dst_device, const void *src_host, uint64_t bytes);
int ks_dev_download(ks_ctx *ctx, void *dst_host, const void *src_device, uint64_t bytes);

/* Pinned (page-locked) host memory.  Host buffers handed to this library are copied in one DMA at link rate when they are
 * pinned (ks_host_alloc, hipHostMalloc, hipHostRegister); pageable ones go through the context's double-buffered pinned
 * staging with a few host copy threads.  Callers that want sketches / k-mer tables back should receive them in pinned arrays. */
int ks_host_alloc(ks_ctx *ctx, uint64_t bytes, void **out);
int ks_host_free(ks_ctx *ctx, void *ptr);

/* ---- host-side pre-step ------------------------------------------------------------------ */

typedef struct ks_residue_error {
    uint32_t seq_index; /* index of the offending record in the batch */
    uint32_t position;  /* 1-based, in the OUTPUT so far (src/rust/aminoacid.rs:86) */
    uint8_t residue;
} ks_residue_error;

/* AminoAcidAmbiguity::validate_and_resolve (src/rust/aminoacid.rs:74-105) on one record:
 * accepts the 20 standard residues + X U O * + B Z J; truncates after the first '*' (inclusive);
 * B->D|N, Z->E|Q, J->I|L chosen by a SplitMix64 stream seeded with rng_seed (the reference draws
 * from rand::rng(), aminoacid.rs:48 — non-reproducible by construction).
 * upper != 0 first upper-cases ASCII (the FASTA path, src/rust/index.rs:1000).
 * out must hold len bytes.  Returns KS_OK or KS_ERR_INVALID_RESIDUE (err filled, seq_index = 0). */
int ks_validate_and_resolve(const uint8_t *seq, uint64_t len, int upper, uint64_t rng_seed,
                            uint8_t *out, uint64_t *out_len, ks_residue_error *err);

/* ---- sketch ------------------------------------------------------------------------------- */

/* Replaces the rayon loop of process_batch_parallel (src/rust/index.rs:984-1016) around
 * ProteinSignature::add_protein -> sourmash KmerMinHash::add_protein (src/rust/signature.rs:273-282),
 * and branchwater do_manysketch(singleton=True) (src/python/kmerseek/sketch.py:33-39).
 *
 * residues   : n_seqs sequences concatenated, 1 byte per residue (ASCII), HOST memory
 * seq_offsets: n_seqs+1 ascending byte offsets into residues, HOST memory
 * Residues are hashed as given after ASCII upper-casing (what sourmash does internally); run
 * ks_validate_and_resolve first for the Rust-path semantics.
 * Result: per sequence, ascending unique hashes h with 0 < h <= max_hash and abund = number of
 * windows of that sequence hashing to h. */
int ks_sketch_batch(ks_ctx *ctx, const uint8_t *residues, const uint64_t *seq_offsets,
                    uint32_t n_seqs, const ks_params *params, ks_sketches **out);

/* Same, with residues / seq_offsets already resident in device memory (HBM) on ctx's device.
 * n_residues = seq_offsets[n_seqs].  max_seq_len: an upper bound on the longest sequence of the batch, or 0.
 * With a bound the launches are planned on the host and the call synchronises once, at its end; with 0 the
 * library measures the batch first (one more device->host round trip, and a tile stride fitted to the batch).
 * A bound smaller than the longest sequence is detected and fails with KS_ERR_INVALID_ARG (no output). */
int ks_sketch_batch_device(ks_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_seq_offsets,
                           uint32_t n_seqs, uint64_t n_residues, uint32_t max_seq_len,
                           const ks_params *params, ks_sketches **out);

/* Sketch a QUERY batch that is about to be searched against `index` (sketch parameters are the index's).  Same result
 * as ks_sketch_batch_device; in addition the sketch kernel writes the batch's postings already partitioned on the hash
 * bits the join against this index consumes, so the following ks_search skips its first partition pass (the sketch
 * kernel is ALU-bound: the extra writes ride on idle memory pipes).  If the hashes are too skewed for the fixed-size
 * regions the postings are dropped and ks_search partitions as usual — results are identical either way. */
int ks_sketch_queries_device(ks_ctx *ctx, const ks_index *index, const uint8_t *d_residues,
                             const uint64_t *d_seq_offsets, uint32_t n_seqs, uint64_t n_residues,
                             uint32_t max_seq_len, ks_sketches **out);

/* ks_sketch_queries_device + ks_search in ONE call, for callers that sketch a batch only to search it — what the
 * reference does per query file (src/python/kmerseek/search.py:125-141 after sketch.py:28-40; batches of 1000 records in
 * src/rust/main.rs:130).  Same sketches, same hits as the two calls; the host waits for the device twice instead of three
 * times (the wait at the end of the sketch is folded into the search's first one: ~25 us per call, which is what a small
 * batch or a 1/8 query shard notices).  *sketches_out may be NULL: the sketches are then freed before returning. */
int ks_sketch_search_device(ks_ctx *ctx, const ks_index *index, const uint8_t *d_residues,
                            const uint64_t *d_seq_offsets, uint32_t n_seqs, uint64_t n_residues,
                            uint32_t max_seq_len, ks_sketches **sketches_out, ks_hits **hits_out);
/* The same from host arrays (the layout ks_sketch_batch takes): upload, sketch, search. */
int ks_sketch_search(ks_ctx *ctx, const ks_index *index, const uint8_t *residues, const uint64_t *seq_offsets,
                     uint32_t n_seqs, ks_sketches **sketches_out, ks_hits **hits_out);

uint32_t ks_sketches_n_seqs(const ks_sketches *s);
uint64_t ks_sketches_n_hashes(const ks_sketches *s);
uint64_t ks_sketches_n_windows(const ks_sketches *s); /* k-mer windows hashed to build it */
/* != 0 if ks_sketch_queries_device's partitioned postings are attached: 1 = 12-byte postings (hash, sequence id),
 * 2 = 10-byte postings (the fingerprint join of big indexes at scaled = 1: 8 hash bits are implied by the region) */
int ks_sketches_has_postings(const ks_sketches *s);
void ks_sketches_params(const ks_sketches *s, ks_params *out);
/* device pointers (valid until ks_sketches_free): offsets u64[n+1], hashes u64[], abund u32[] — the plain CSR.  (Inside the
 * library a fresh batch is a CSR of slots: a sequence that repeats a k-mer leaves a gap behind its distinct hashes.  The first
 * of these calls — like ks_sketches_copy_to_host, ks_index_build, ks_sketches_union — closes the gaps of such a batch with one
 * gather on the context's stream and waits for it; a batch without repeats is a plain CSR as it stands.  NULL with the reason
 * in ks_last_error if that pass fails.) */
const uint64_t *ks_sketches_device_offsets(const ks_sketches *s);
const uint64_t *ks_sketches_device_hashes(const ks_sketches *s);
const uint32_t *ks_sketches_device_abunds(const ks_sketches *s);
/* any of the three host pointers may be NULL */
int ks_sketches_copy_to_host(ks_ctx *ctx, const ks_sketches *s, uint64_t *offsets,
                             uint64_t *hashes, uint32_t *abunds);
/* upload an existing CSR (e.g. sketches loaded from a .sig.zip) so it can be indexed / searched */
int ks_sketches_from_host(ks_ctx *ctx, const uint64_t *offsets, const uint64_t *hashes,
                          const uint32_t *abunds, uint32_t n_seqs, const ks_params *params,
                          ks_sketches **out);
/* Union of all sequences' sketches with abundances summed per hash, returned as ONE sequence: the
 * "combined minhash" that ProteomeIndex::store_signatures maintains (src/rust/index.rs:800-830,
 * add_many_with_abund under a mutex there).  Abundance sums saturate at 2^32-1. */
int ks_sketches_union(ks_ctx *ctx, const ks_sketches *in, ks_sketches **out);
void ks_sketches_free(ks_sketches *s);

/* ---- k-mer positions ----------------------------------------------------------------------- */

/* Replaces ProteomeIndex::process_kmers (src/rust/index.rs:749-786): for every window whose hash
 * is in the sequence's sketch emit (seq, start, hash), ordered by (seq, start).  With the
 * FracMinHash rule "in the sketch" == "0 < h <= max_hash", so no membership scan is needed.
 * The host groups triples into KmerInfo{encoded_kmer, original_kmer -> [positions]} (src/rust/kmer.rs:6-12). */
int ks_kmer_positions(ks_ctx *ctx, const uint8_t *residues, const uint64_t *seq_offsets,
                      uint32_t n_seqs, const ks_params *params, ks_kmerpos **out);
/* Same, with the batch already resident in device memory (d_residues 16-byte aligned, n_residues = seq_offsets[n_seqs]). */
int ks_kmer_positions_device(ks_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_seq_offsets, uint32_t n_seqs,
                             uint64_t n_residues, const ks_params *params, ks_kmerpos **out);
uint64_t ks_kmerpos_count(const ks_kmerpos *p);
int ks_kmerpos_copy_to_host(ks_ctx *ctx, const ks_kmerpos *p, uint32_t *seq, uint32_t *start,
                            uint64_t *hash);
void ks_kmerpos_free(ks_kmerpos *p);

/* ---- index + search ------------------------------------------------------------------------ */

/* Builds the inverted index over target sketches: postings (hash, tid, abund) sorted by hash.
 * Stands where `kmerseek index` / do_index builds its search structure (src/python/kmerseek/index.py:55-74);
 * the targets' per-sequence sizes and abundance totals are kept for the ratio columns. */
int ks_index_build(ks_ctx *ctx, const ks_sketches *targets, ks_index **out);
uint32_t ks_index_n_targets(const ks_index *ix);
uint64_t ks_index_n_postings(const ks_index *ix);
void ks_index_free(ks_index *ix);

/* Replaces branchwater do_manysearch(threshold=0, output_all=False) (src/python/kmerseek/search.py:125-141):
 * for every (query, target) pair with at least one shared hash emits
 *   qid, tid, intersect = |mins_q ∩ mins_t|, n_weighted = Σ target abundance over the shared hashes,
 * sorted by (qid, tid).  Pair results are identical to a pairwise sorted merge of the sketches. */
int ks_search(ks_ctx *ctx, const ks_index *index, const ks_sketches *queries, ks_hits **out);
uint64_t ks_hits_count(const ks_hits *h);
uint64_t ks_hits_n_pair_instances(const ks_hits *h); /* Σ_h q(h)·t(h): matched (posting, posting) pairs */
/* how the query postings were grouped for the join (results are identical on every path):
 * 0 the sketch kernel's regions were the buckets; 1 regions + one histogram-free bucket scatter;
 * 2 regions + one dense radix pass (a bucket overflowed); 3 dense radix partition from the CSR (no postings attached) */
int ks_hits_partition_path(const ks_hits *h);
/* Bytes per query posting inside the join buckets of the search that produced h: 12 (hash u64 + sequence id u32), 10 (the
 * sketch kernel's 10-byte form kept through the bucket scatter), 9 (join prefixes of 16 bits: the bucket implies two hash bytes
 * and carries two bytes of the sequence id there), 0 (the search ran without the histogram-free bucket scatter).  Diagnostic. */
int ks_hits_bucket_posting_bytes(const ks_hits *h);
int ks_hits_copy_to_host(ks_ctx *ctx, const ks_hits *h, uint32_t *qid, uint32_t *tid,
                         uint32_t *intersect, uint64_t *n_weighted);
/* Device-resident COO columns (valid until ks_hits_free; ks_hits_count entries each) — what a multi-GPU caller hands to
 * RCCL without a host round trip (hit lists of shards are disjoint and only concatenated). */
const uint32_t *ks_hits_device_qid(const ks_hits *h);
const uint32_t *ks_hits_device_tid(const ks_hits *h);
const uint32_t *ks_hits_device_intersect(const ks_hits *h);
const uint64_t *ks_hits_device_n_weighted(const ks_hits *h);
/* Device-to-device copy of the columns into caller-owned device buffers (e.g. the send block of an all-gather), with
 * qid_base / tid_base added to the ids (a shard's local numbering -> global).  Asynchronous on ctx's stream; any
 * destination may be NULL. */
int ks_hits_copy_to_device(ks_ctx *ctx, const ks_hits *h, uint32_t qid_base, uint32_t tid_base, uint32_t *d_qid,
                           uint32_t *d_tid, uint32_t *d_intersect, uint64_t *d_n_weighted);
/* Transport form of the rows for a multi-GPU exchange: one 64-bit word per row,
 *     qid << (tbits + 2v) | tid << 2v | intersect << v | n_weighted,   v = (64 - qbits - tbits) / 2,
 * ids in global numbering (qid_base / tid_base added; qbits / tbits = bits of the GLOBAL id ranges, qbits + tbits <= 48).
 * A row whose intersect or n_weighted needs more than v bits carries all-ones in both value fields and is appended to the
 * escape list (row index, intersect, n_weighted; *d_n_esc counts them, also past esc_cap — the caller sizes a repeat).
 * 8 bytes per row instead of 20 over xGMI.  Asynchronous on ctx's stream; *d_n_esc must be zero on entry. */
int ks_hits_pack64_to_device(ks_ctx *ctx, const ks_hits *h, uint32_t qid_base, uint32_t tid_base, int qbits, int tbits,
                             uint64_t *d_packed, uint32_t *d_esc_row, uint32_t *d_esc_intersect, uint64_t *d_esc_n_weighted,
                             uint32_t *d_n_esc, uint32_t esc_cap);
/* The receiving side: n transport words (device) -> the four columns (device, caller-owned).  Escaped rows come out with
 * all-ones in both value fields; the caller patches them from the gathered escape lists.  Asynchronous on ctx's stream. */
int ks_hits_unpack64_device(ks_ctx *ctx, const uint64_t *d_packed, uint64_t n, int qbits, int tbits, uint32_t *d_qid,
                            uint32_t *d_tid, uint32_t *d_intersect, uint64_t *d_n_weighted);
/* Index-sharded exchange, global (qid, tid) order: the gathered rows are n_blocks rank blocks (block r = block_rows[r] rows,
 * host array), each ordered by (qid, tid), the ranks' target ranges ascending.  One counting merge (per (query, rank) run
 * lengths by binary search -> exclusive scan -> one move) writes them ordered by (qid, tid) into the caller-owned output
 * columns (device, sum(block_rows) entries; must not alias the inputs).  qid values must be < n_queries: a row that is not
 * has no place in the merged order, and the call fails with KS_ERR_INVALID_ARG (outputs then incomplete) instead of leaving a
 * gap.  Returns after the move has run (one wait on ctx's stream: the count of such rows comes back with it).
 * (Rows per query as branchwater manysearch lists them: src/python/kmerseek/search.py:125-141.) */
int ks_hits_merge_by_qid_device(ks_ctx *ctx, const uint32_t *d_qid, const uint32_t *d_tid, const uint32_t *d_intersect,
                                const uint64_t *d_n_weighted, const uint64_t *block_rows, uint32_t n_blocks, uint32_t n_queries,
                                uint32_t *d_out_qid, uint32_t *d_out_tid, uint32_t *d_out_intersect, uint64_t *d_out_n_weighted);
void ks_hits_free(ks_hits *h);

/* ---- measurement --------------------------------------------------------------------------- */

/* Per-kernel HIP-event timing on ctx's stream.  enable: 0 off; 1 events bracket every launch (~20 us of idle queue
 * around each: for an untimed diagnostic pass); 2 only the sketch tile kernel — the one a step's roofline is quoted
 * for — so that a timed region pays for one bracket per batch. */
typedef struct ks_kernel_time {
    char name[48];
    uint64_t launches;
    double total_ms;
} ks_kernel_time;
int ks_timing_enable(ks_ctx *ctx, int enable);
int ks_timing_reset(ks_ctx *ctx);
/* resolves pending events (synchronizes the stream) and copies up to cap rows; *n = rows available */
int ks_timing_get(ks_ctx *ctx, ks_kernel_time *rows, uint32_t cap, uint32_t *n);

/* The two ceilings SURVEY.md section 8(d) asks bench.py to print beside the HBM roofline, measured on ctx's device:
 *   gmul_per_s    - 64-bit integer multiplies per second / 1e9 (MurmurHash3 needs 8 per window for k <= 16),
 *   copy_gb_per_s - device-to-device hipMemcpy rate, bytes read + bytes written per second / 1e9,
 *   nominal_gb_per_s - memoryClockRate x memoryBusWidth of the device properties (DDR: x2) / 1e9. */
int ks_bench_device_rates(ks_ctx *ctx, double *gmul_per_s, double *copy_gb_per_s, double *nominal_gb_per_s);
/* random 8-byte gathers per second from a 134 MB table ([0]) and from a 2 MB table ([1]): the ceiling of a table-lookup hash
 * for the two-letter hp alphabet (see DESIGN.md for why the multiplies win) */
int ks_bench_gather_rates(ks_ctx *ctx, double gathers_per_s[2]);

#ifdef __cplusplus
}
#endif
#endif /* KMERSEEK_AMD_H */
