// ks_sketch.hip — FracMinHash sketching of a batch of proteins on gfx950.
//
// Replaces, for a whole batch in one pass, the reference's per-protein
//   ProteinSignature::add_protein -> sourmash KmerMinHash::add_protein   (src/rust/signature.rs:273-282)
// under the rayon loop of process_batch_parallel                          (src/rust/index.rs:984-1016)
// and branchwater manysketch(singleton)                                   (src/python/kmerseek/sketch.py:33-39).
//
// Layout / algorithm (integer, HBM-bound by design; no MFMA):
//   * residues: 1 byte each, all sequences concatenated; offsets u64[n+1].
//   * The batch is cut into tiles by residue range: tile t owns the sequences whose START offset lies
//     in [t*R, (t+1)*R).  One 512-thread workgroup stages the tile's residues once into LDS with
//     coalesced 16-B loads, re-encoding through a 256-B LUT (upper-case + protein/dayhoff/hp) on the way.
//   * Every thread hashes 8 consecutive windows from an LDS sliding window (aligned ds_read_b64 +
//     constant funnel shifts rebuild the byte-exact murmur input words), keeps 0 < h <= max_hash.
//   * Per-sequence "sort + unique + count" without a comparison sort: murmur output is uniform, so a
//     kept hash goes to bucket  seq_start + floor(h / max_hash * n_windows)  (monotone in h; ~1 element per
//     bucket).  LDS atomics count buckets and hand out arrival slots, one block scan turns counts into
//     starts, the hashes are scattered into bucket order; the buckets that hold more than one hash are LISTED by the
//     thread that scans them and put in order in place (a pair: one compare; a few: insertion sort; many — repeats of
//     a low-complexity sequence —: ranked by the whole workgroup).  Equal hashes of a sequence share a bucket, so the
//     repeats are counted on the way: the tile's distinct count is known before anything is read back.
//   * Tiles with repeats (or with postings to emit) read the sorted run back position-major: representative flags,
//     abundances and distinct ranks from bitmaps; the others write their sorted run as it stands.  A decoupled
//     look-back over the tiles' distinct counts gives the tile its place in the final CSR: one coalesced stream out.
//   * A sequence that starts inside a tile's residue range but does not END inside the tile's LDS window (at most
//     one per tile, the last) is deferred: if it fits a tile on its own ("medium") it gets one in a second launch of
//     the same kernel; longer ones take the same algorithm with its arrays in a global scratch slab (k_sketch_long),
//     one workgroup per sequence.  (Packed tiles — the plain variant — hold whole sequences: only sequences longer than a
//     tile go to the slab path.)
#include "ks_device.h"

#ifndef SK_THREADS
#define SK_THREADS 512
#endif
#define SK_E 8
#define SK_TILE (SK_THREADS * SK_E) // 4096 LDS positions
#define SK_MED_MAX (SK_TILE - 16)   // longest sequence that still fits one tile on its own ("medium")
// Tile t takes the sequences that START in residues [t * R, (t+1) * R) and stages SK_TILE bytes from t * R: a larger
// stride R fills more of the tile's positions with windows but defers more sequences (one that starts at local position
// p with length L is deferred when p + L > SK_MED_MAX, i.e. with probability ~ E[max(0, L - (SK_MED_MAX - R))] / R).
// R is chosen per batch from these candidates (all multiples of 16) by a measured cost model: a shared tile avoided
// saves ~26 ns, a deferred sequence costs ~45 ns (MI355X, 1M-protein batches).
#define SK_NR 8
static const u32 sk_r_cand_host[SK_NR] = {2544, 2800, 3056, 3312, 3568, 3824, 3952, 4016};
#define SK_PAD 160                  // >= KS_MAX_KSIZE + 24: slack behind the last residue for word reads
#ifndef SK_MINW
#define SK_MINW 6                   // waves per SIMD to compile for: 3 workgroups of 8 waves per CU
#endif
#define SK_NFLAG (SK_TILE / 32)
#ifndef SK_LB_WAVES
#define SK_LB_WAVES 1 // waves of a workgroup that look back (64 predecessors each)
#endif
#define SK_C3MAX 32u                // a bucket of more hashes than this is put in order by the whole workgroup
#define SK_QB_CAP 128u              // >= SK_TILE / (SK_C3MAX + 1): list of those buckets (16-bit entries)

// Diagnostic build only (-DSK_STAMP): per-phase shader-clock shares of k_sketch_tiles, summed over
// workgroups by lane 0.  Never compiled into the shipped library; the numbers are shares, not times.
#ifdef SK_STAMP
#define SK_STAMP_SLOTS 4096
__device__ unsigned long long sk_stamp_acc[SK_STAMP_SLOTS][16];
#define SK_STAMP_AT(i) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); \
    atomicAdd(&sk_stamp_acc[blockIdx.x % SK_STAMP_SLOTS][i], t_ - sk_t_prev); sk_t_prev = clock64(); } } while (0)
extern "C" void ks_debug_read_stamps(unsigned long long *out, int reset) {
    static unsigned long long host[SK_STAMP_SLOTS][16];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(sk_stamp_acc), sizeof host);
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int s = 0; s < SK_STAMP_SLOTS; s++) out[i] += host[s][i]; }
    if (reset) { memset(host, 0, sizeof host); (void)hipMemcpyToSymbol(HIP_SYMBOL(sk_stamp_acc), host, sizeof host); }
}
#elif defined(SK_STOP_AFTER)
// Diagnostic build only (-DSK_STOP_AFTER=n): every workgroup returns after phase n, so a counter pass over variants
// n = 0..8 gives the cumulative instruction mix by phase (tools/phase_counters.py).  The output is garbage.
#define SK_STAMP_AT(i) do { if ((i) == SK_STOP_AFTER) return; } while (0)
#else
#define SK_STAMP_AT(i) do { } while (0)
#endif

struct sk_args {
    const u8 *res;
    const u64 *offs;
    u32 n_seqs;
    u64 n_res;
    u32 k;
    u64 seed;
    u64 max_hash;
    u32 sfix;     // floor(2^48 / ((max_hash >> 32) + 1)): bucket multiplier = (n_windows * sfix) >> 16
    const u8 *lut; // 256-byte encode table for this moltype
    u32 upper_only; // the table only upper-cases (moltype protein): applied arithmetically
    u32 R;         // tile stride in residues (see sk_r_cand); 0 = packed tiles (tile_g0 gives each tile's first residue)
    const u64 *tile_g0; // packed tiles: 16-byte aligned residue offset the tile's LDS window starts at
    u32 span;      // residues a shared tile covers from tile * R: SK_TILE, or more for the compacting variant (scaled > 1)
    u32 c_div, c_rcp; // compacting variant: bucket space is positions / c_div (c_rcp = ceil(2^32 / c_div))
    u64 out_cap;   // capacity of out_hash / out_abund (MODE 0): writes beyond it are dropped and the host repeats larger
    u32 use_ticket; // tile ids from the atomic ticket (1) or from blockIdx.x (0)
    u32 debug_qcap;      // diagnostics (KS_DEBUG_QCAP): capacity of the bucket lists of phase 3 (0 = what fits)
    u32 debug_skip_tile; // diagnostics (KS_DEBUG_LOOKBACK_SKIP): this tile never publishes — its successors' spins really expire
    u32 le_cap;    // a sequence whose LOCAL end lies beyond this is not this launch's business
    u32 max_len_tile; // ... nor is one longer than this (packed tiles: PK_MAX_LEN, so that "long" means the same everywhere)
    const u32 *seq_list; // MODE 0: tile_first[n_tiles + 1] (tile -> first sequence); MODE 1: the medium sequences
    const u32 *n_list;   // MODE 1: device-resident length of seq_list
    u32 n_list_cap;      // ... and the allocated length (the smaller one counts)
    // MODE 0 writes the final CSR directly: hashes / abunds at csr positions, csr[s] per sequence
    u64 *out_hash;  // MODE 0: final hashes [n_windows]; MODE 1: lg_hash [n_res] (run of sequence s starts at offs[s])
    u32 *out_abund;
    u64 *csr;       // [n_seqs + 1] final CSR offsets (MODE 0)
    u64 *total_out; // MODE 0: the batch's kept-hash total once more, next to the other words the host reads back
    u32 *counts;    // [n_seqs] DISTINCT hashes of every sequence (ks_sketches::d_counts): written by whoever sketches the sequence
    u32 *kept;      // [n_seqs] kept hashes (repeats included) of medium / long sequences (written by MODE 1 / k_sketch_long, read by
                    // MODE 0: a deferred sequence's slot in the CSR is as long as its kept count)
    u64 *drops_out; // kept hashes that repeat an earlier one of their sequence, summed over the batch (slots the CSR leaves empty)
    // decoupled look-back across tiles (MODE 0)
    unsigned long long *tile_status; // [n_tiles] (flag << 62) | value; flag 1 = tile aggregate, 2 = inclusive prefix
    u32 *ticket;    // [0] dynamic tile id, [1] status bits: 1 = a bounded spin expired, 2 = postings not emitted for some tile,
                    //     4 = a compacting tile kept more hashes (or holds more sequences) than its LDS lists take
    u32 n_tiles;
    const u32 *n_tiles_dev; // MODE 0, optional: the tiles there really are (a launch sized by an upper bound: the rest return)
    // optional: postings (hash, sequence) partitioned on the low 8 bits of the join prefix into <= 256 fixed-capacity
    // regions, written while the vector ALU is the bottleneck — the query side's first partition pass of ks_search
    u64 *part_keys;   // [256 * part_cap] or NULL
    u32 *part_vals;
    u32 *part_cursor; // [256] records placed per region so far
    u64 part_cap;
    u32 part_K, part_mask; // region = ks_join_prefix(h, part_K) & part_mask
    u32 part_kshift;       // != 0: part_K = 2^(32 - part_kshift), the prefix is a shift (sk_digit)
    u32 part_s;            // != 0: 10-byte postings (ks_sketches::part_s): sequence id bits 0..7 ride in hash bits [part_s, part_s + 8)
    u32 part_sub_shift;    // sub-regions per region = 1 << shift; a workgroup writes sub-region blockIdx.x & (that - 1)
};

#define SK_FLAG_AGG (1ULL << 62)
#define SK_FLAG_PRE (2ULL << 62)
#define SK_VAL_MASK ((1ULL << 62) - 1)

// bucket multiplier: bucket = umulhi(h >> 32, mul) < n_windows for every kept h (h <= max_hash)
KS_DEV u32 sk_bucket_mul(u32 nw, u32 sfix) {
    u64 m = ((u64)nw * sfix) >> 16;
    return m > 0xffffffffULL ? 0xffffffffu : (u32)m;
}

KS_DEV u32 sk_lower_bound(const u64 *a, u32 lo, u32 hi, u64 x) { // first i in [lo,hi) with a[i] >= x
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// hash of the window that starts at LDS byte `pos8 + I` where pos8 is 8-byte aligned.  KC != 0: k is the compile-time
// constant KC (the launches of the common k-mer sizes): the block loop, the tail branches and the byte masks fold away and a
// tail of <= 4 bytes multiplies as a 32-bit value — ~56 instead of ~90 vector instructions per window at k = 10, where
// the hash phase is what the vector ALU is busy with (profiles/).
template <int I, int KC = 0>
KS_DEV u64 sk_hash_window(const u64 *w /* LDS words starting at pos8 */, u32 k_rt, u64 seed) {
    const u32 k = KC ? (u32)KC : k_rt;
    ks_murmur m;
    m.init(seed);
    const u32 nb = k >> 4, t = k & 15;
    u32 j = 0;
    for (u32 b = 0; b < nb; b++, j += 2) {
        u64 a0 = w[j], a1 = w[j + 1], a2 = w[j + 2];
        m.block(ks_funnel<I>(a0, a1), ks_funnel<I>(a1, a2));
    }
    if (t) {
        u64 a0 = w[j], a1 = w[j + 1], a2 = w[j + 2];
        u64 k1 = ks_funnel<I>(a0, a1), k2 = ks_funnel<I>(a1, a2);
        if (t > 8) k2 &= ks_mask_bytes(t - 8); else { k1 &= ks_mask_bytes(t); k2 = 0; }
        m.tail(k1, k2, t);
    }
    return m.finish((u64)k);
}

// does the sequence at residue offset `start` with `len` residues end outside its shared tile's LDS window?
KS_DEV bool sk_deferred(u64 start, u64 len, u32 R, u32 span) { return start % R + len > span - 16; }
// floor(x / d) for x < 2^32 / d with rcp = ceil(2^32 / d) (d = 1: rcp does not fit, handled apart)
KS_DEV u32 sk_div(u32 x, u32 d, u32 rcp) { return d == 1 ? x : __umulhi(x, rcp); }

// Block-wide exclusive scan of two values at once (SK_THREADS threads; `smem` holds 2 * (SK_THREADS / 64 + 1) words).
// The wave totals are combined without a second barrier and without every thread adding them up itself (round 3: 8 x 2 LDS
// reads + a compare / select / add each per thread — a quarter of phase 3's vector instructions): lane j < 8 of every wave
// reads total j, three DPP steps scan the eight, and the wave's own base / the tile total are READ from lanes wave - 1 / 7
// (the wave number is uniform: v_readlane).
KS_DEV u32 sk_wave_totals_scan(const u32 *smem, u32 lane, u32 wave_u /* uniform */, u32 *total) {
    constexpr u32 NW = SK_THREADS / 64;
    static_assert(NW <= 16, "one DPP row");
    u32 v = lane < NW ? smem[lane] : 0u;
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
    if (NW > 8) v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8
    *total = (u32)__builtin_amdgcn_readlane((int)v, NW - 1);
    return wave_u ? (u32)__builtin_amdgcn_readlane((int)v, (int)wave_u - 1) : 0u;
}
KS_DEV u32 sk_block_excl_scan2(u32 a, u32 b, u32 *smem, u32 *total_a, u32 *excl_b) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr u32 NW = SK_THREADS / 64;
    const u32 ia = ks_wave_incl_scan(a), ib = ks_wave_incl_scan(b);
    if (lane == 63) { smem[wave] = ia; smem[NW + 1 + wave] = ib; }
    __syncthreads();
    const u32 wave_u = (u32)__builtin_amdgcn_readfirstlane((int)wave);
    u32 tot_b;
    const u32 base_a = sk_wave_totals_scan(smem, lane, wave_u, total_a);
    const u32 base_b = sk_wave_totals_scan(smem + NW + 1, lane, wave_u, &tot_b);
    *excl_b = base_b + ib - b;
    // tile totals of b for whoever reads them after the caller's next barrier.  (smem[NW] is read by nobody in this scan, and the
    // caller's previous use of it lies behind the barrier above.)
    if (threadIdx.x == 0) smem[NW] = tot_b;
    return base_a + ia - a;
}

// ... and of one value, with ONE barrier (the caller's next barrier must come before scan_smem is written again)
KS_DEV u32 sk_block_excl_scan1(u32 a, u32 *smem) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 ia = ks_wave_incl_scan(a);
    if (lane == 63) smem[wave] = ia;
    __syncthreads();
    u32 total;
    const u32 base = sk_wave_totals_scan(smem, lane, (u32)__builtin_amdgcn_readfirstlane((int)wave), &total);
    return base + ia - a;
}

// partition digit of a kept hash: the join prefix (ks_join_prefix) masked; at scaled = 1 the prefix multiplier is a power of
// two and the quarter-rate multiply is a shift (kshift = 32 - log2 K; 0 = multiply)
KS_DEV u32 sk_digit(u64 h, u32 K, u32 kshift, u32 mask) {
    const u32 hi = (u32)(h >> 32);
    return (kshift ? (hi >> kshift) : __umulhi(hi, K)) & mask;
}

#define SK_CTL_WORDS 32 // control block of a sketch call (see sketch_attempt)
#define SK_SEQ_CAP 254 // sequence boundaries of a tile staged in LDS (tiles with more fall back to global reads)
// per-element code: sequence (relative to the tile's first, 8 bits) | bucket (12 bits) | arrival slot (12 bits)
#define SK_BO_B(x) (((x) >> 12) & 0xfffu)
#define SK_BO_O(x) ((x) & 0xfffu)
#define SK_BO_S(x) ((x) >> 24)

// Sequence boundaries of the tile in LOCAL coordinates (byte position relative to g0, clamped to 2^31-1),
// served from LDS when the tile has <= SK_SEQ_CAP sequences, else straight from the offsets array.
struct sk_bounds {
    const u32 *loff; // LDS copy, entry i = local offset of sequence s_first + i (ns + 1 entries)
    const u64 *goff; // global offsets
    u64 g0;
    u32 s_first;
    bool in_lds;
    KS_DEV u32 at(u32 s) const {
        if (in_lds) return loff[s - s_first];
        u64 v = goff[s] - g0;
        return v > 0x7fffffffULL ? 0x7fffffffu : (u32)v;
    }
};

struct sk_seq { // per-thread view of the sequence its current window belongs to
    u32 s;      // sequence id
    u32 ls, le; // local [start, end) in tile coordinates (le clamped)
    u32 nw;     // windows
    u32 mul;    // bucket multiplier
    bool ok;    // short enough for this launch and has windows
};

KS_DEV void sk_load_seq(sk_seq &q, const sk_args &A, const sk_bounds &B, u32 s_end) {
    if (q.s >= s_end) { q.ok = false; q.ls = 0xffffffffu; q.le = 0xffffffffu; q.nw = 0; q.mul = 0; return; }
    q.ls = B.at(q.s);
    q.le = B.at(q.s + 1);
    const u32 len = q.le - q.ls; // a clamped end only makes a too-long sequence look (still) too long
    q.nw = (len >= A.k && q.le <= A.le_cap && len <= A.max_len_tile) ? (len - A.k + 1) : 0;
    q.mul = sk_bucket_mul(q.nw, A.sfix);
    q.ok = q.nw > 0;
}

// bookkeeping for the window at local position p: which sequence, is it a real window, bucket + arrival slot
KS_DEV u32 sk_place_window(const sk_args &A, u32 p, u64 h, sk_seq &q, const sk_bounds &B, u32 s_end, u32 *cnt) {
    while (q.s < s_end && p >= q.le) { q.s++; sk_load_seq(q, A, B, s_end); }
    const bool keep = q.ok && p >= q.ls && p + A.k <= q.le && h != 0 && h <= A.max_hash;
    u32 bo = 0xffffffffu;
    if (keep) {
        const u32 b = q.ls + __umulhi((u32)(h >> 32), q.mul);
        const u32 sh = (b & 1u) * 16u;
        const u32 o = (atomicAdd(&cnt[b >> 1], 1u << sh) >> sh) & 0xffffu; // two 16-bit counters per word
        const u32 srel = q.s - B.s_first;
        bo = ((srel < 254u ? srel : 254u) << 24) | (b << 12) | o; // b, o < 4096
    }
    return bo;
}

// k >= SK_E ("one run"): a sequence's last k - 1 positions start no window, so the SK_E consecutive positions of a thread hold
// windows of at most ONE sequence — the first whose last window lies at or beyond the thread's first position (found once per
// thread: the binary search of phase 2 with key q0 + k - 1) — and a position p starts a real window of it iff (p - ls) < nw,
// unsigned.  No per-window walk over the boundaries, no per-window sequence code (round 3: a compare + branch + three range
// tests per window, 25 vector + 14 scalar instructions of bookkeeping per window against 66 of hashing;
// profiles/r03_sq_counters.md).  The placement itself is written out in phase 2 (all eight atomics in flight together).

// ---------------------------------------------------------------------------------------------
// Packed tiles (plain variant, scaled = 1): instead of cutting the batch at fixed residue strides — where a stride of 3312
// leaves 19 % of a tile's 4096 positions empty so that few sequences straddle a tile's end — tiles are packed greedily
// with WHOLE sequences: tile = the longest run of consecutive sequences whose residues fit the LDS window from the
// (16-byte aligned) start of the first.  ~95 % of the positions carry windows, ~14 % fewer tiles, and no sequence is
// ever deferred for its position: only those longer than PK_MAX_LEN go to the global-slab path (as the deferred tail of
// the tile they follow).  The greedy cut is sequential, so it runs per chunk of PK_CHUNK sequences (a tile never spans
// chunks: one short tile per 1024 sequences): one wave per chunk walks its offsets in LDS with 64-ary searches;
// k_pack_fill then lays the chunks' tiles out densely.
// ---------------------------------------------------------------------------------------------
#define PK_CHUNK 1024
#define PK_MAX_LEN (SK_MED_MAX - 16) // fits the window wherever its first residue falls inside the aligned 16 bytes

__global__ __launch_bounds__(64) void k_pack_walk(const u64 *offs, u32 n_seqs, u32 chunk /* <= PK_CHUNK sequences per chunk */,
                                                  u32 *chunk_tiles /* [n_seqs]: tile starts of chunk c from c * chunk */, u32 *chunk_cnt) {
    __shared__ u64 lo_s[PK_CHUNK + 1];
    const u32 lane = threadIdx.x;
    const u32 c0 = blockIdx.x * chunk;
    const u32 n = (n_seqs - c0) < chunk ? (n_seqs - c0) : chunk; // sequences of this chunk
    for (u32 i = lane; i <= n; i += 64) lo_s[i] = offs[c0 + i];
    __builtin_amdgcn_wave_barrier();
    u32 s = 0, nt = 0;
    while (s < n) { // (uniform: every lane walks the same chain)
        const u64 limit = (lo_s[s] & ~15ULL) + SK_MED_MAX; // a sequence fits if it ENDS at or before this residue offset
        // largest e in [s, n] with lo_s[e] <= limit (lo_s[s] always is): 64-ary search
        u32 lo = s, hi = n + 1;
        while (hi - lo > 1) {
            const u32 step = (hi - lo + 63) / 64;
            const u32 idx = lo + (lane + 1) * step;
            const bool ok = idx < hi && lo_s[idx] <= limit;
            const u64 m = __ballot(ok);
            const u32 good = m == ~0ULL ? 64u : (u32)__builtin_ctzll(~m); // leading run of lanes that still fit
            const u32 nlo = lo + good * step;
            const u32 nhi = lo + (good + 1) * step;
            lo = nlo < hi ? nlo : hi - 1;
            hi = nhi < hi ? nhi : hi;
        }
        u32 e = lo; // sequences [s, e) fit
        if (e == s) e = s + 1;                           // a long sequence on its own (it is the tile's deferred tail)
        else if (e < n && lo_s[e + 1] - lo_s[e] > PK_MAX_LEN) e++; // the long sequence that follows rides along as the tail
        if (lane == 0) chunk_tiles[c0 + nt] = c0 + s;
        nt++;
        s = e;
    }
    if (lane == 0) chunk_cnt[blockIdx.x] = nt;
}

// dense plan: tile_first[t] / tile_g0[t] for the tiles of all chunks in order; tile_first[n_tiles] = n_seqs; *n_tiles_out
__global__ __launch_bounds__(256) void k_pack_fill(const u64 *offs, u32 n_seqs, u32 chunk, const u32 *chunk_tiles, const u32 *chunk_cnt, u32 n_chunks,
                                                   u32 *tile_first, u64 *tile_g0, u32 *n_tiles_out) {
    __shared__ u32 scan_smem[256 / 64 + 1];
    u32 part = 0;
    for (u32 c = threadIdx.x; c < blockIdx.x; c += 256) part += chunk_cnt[c];
    u32 total;
    (void)ks_block_excl_scan(part, scan_smem, &total); // total = tiles of the chunks before this one
    const u32 base = total, mine = chunk_cnt[blockIdx.x];
    for (u32 i = threadIdx.x; i < mine; i += 256) {
        const u32 sf = chunk_tiles[blockIdx.x * chunk + i];
        tile_first[base + i] = sf;
        tile_g0[base + i] = offs[sf] & ~15ULL;
    }
    if (blockIdx.x == n_chunks - 1 && threadIdx.x == 0) {
        tile_first[base + mine] = n_seqs;
        *n_tiles_out = base + mine;
    }
}

// tile_first[t] = first sequence whose start offset is >= t * R (n_tiles + 1 entries): one parallel
// binary search per tile here instead of a serial, latency-bound one at the head of every workgroup
__global__ __launch_bounds__(256) void k_tile_plan(const u64 *offs, u32 n_seqs, u32 n_tiles, u32 R, u32 *tile_first) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    tile_first[t] = t == n_tiles ? n_seqs : sk_lower_bound(offs, 0, n_seqs, (u64)t * R);
}

// Compacting variant, windows [p0 + H, p0 + H + NW) of a thread: hash, keep what passes the threshold, append to the LDS list
// (hash -> hlist, sequence relative to the tile's first -> slist): wave scan of the per-lane keep counts, one LDS atomic
// per wave.  Entries beyond SK_TILE are dropped (the cursor keeps counting: the tile then reports the overflow).
template <int H, int NW, int KC>
KS_DEV void sk_cmp_half(const sk_args &A, const sk_bounds &B, sk_seq &q, const u64 *wl, u32 p0, bool active, u32 s_first, u32 s_end,
                        u32 lane, u64 *hlist, u8 *slist, u32 *cursor) {
    static_assert(NW == 4 || NW == 8, "a half or the whole of a thread's windows");
    u64 h[NW];
    u32 keep = 0, sr[2] = {0, 0}; // keep mask of the NW windows, their sequences (one byte each)
    if (active) {
        h[0] = sk_hash_window<H + 0, KC>(wl, A.k, A.seed);
        h[1] = sk_hash_window<H + 1, KC>(wl, A.k, A.seed);
        h[2] = sk_hash_window<H + 2, KC>(wl, A.k, A.seed);
        h[3] = sk_hash_window<H + 3, KC>(wl, A.k, A.seed);
        if constexpr (NW == 8) {
            h[4] = sk_hash_window<H + 4, KC>(wl, A.k, A.seed);
            h[5] = sk_hash_window<H + 5, KC>(wl, A.k, A.seed);
            h[6] = sk_hash_window<H + 6, KC>(wl, A.k, A.seed);
            h[7] = sk_hash_window<H + 7, KC>(wl, A.k, A.seed);
        }
        if ((KC ? (u32)KC : A.k) >= SK_E) { // (uniform) one sequence per thread (see sk_place_window)
            const u32 srel = q.s - s_first;
#pragma unroll
            for (int i = 0; i < NW; i++)
                if ((p0 + H + i - q.ls) < q.nw && h[i] != 0 && h[i] <= A.max_hash) keep |= 1u << i;
            sr[0] = sr[1] = srel * 0x01010101u;
        } else {
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const u32 p = p0 + H + i;
                while (q.s < s_end && p >= q.le) { q.s++; sk_load_seq(q, A, B, s_end); }
                if (q.ok && p >= q.ls && p + A.k <= q.le && h[i] != 0 && h[i] <= A.max_hash) {
                    keep |= 1u << i;
                    sr[i >> 2] |= (q.s - s_first) << (8 * (i & 3));
                }
            }
        }
    }
    const u32 nk = (u32)__popc(keep);
    const u32 incl = ks_wave_incl_scan(nk);
    // (the wave's total and its slice of the list travel through scalar registers — v_readlane / v_readfirstlane —, not through
    // two ds_bpermute round trips per half and sub-tile)
    const u32 wtot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    u32 wbase = 0;
    if (wtot) { // (uniform)
        if (lane == 0) wbase = atomicAdd(cursor, wtot);
        wbase = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
    }
    u32 pos = wbase + incl - nk;
#pragma unroll
    for (int i = 0; i < NW; i++)
        if (keep & (1u << i)) {
            if (pos < SK_TILE) {
                hlist[pos] = h[i];
                slist[pos] = (u8)(sr[i >> 2] >> (8 * (i & 3)));
            }
            pos++;
        }
}

// MODE 0: shared tiles cut by residue range; MODE 1: one listed medium sequence per workgroup.
// CMP 1 (MODE 0 only, scaled > 1): the compacting variant.  FracMinHash drops (scaled - 1) / scaled of the windows, so a
// tile of SK_TILE positions would run its sort / unique phases — and its prologue, look-back and ~10 barriers — for a
// fifth of the output.  Instead the tile spans A.span ~ scaled * 3800 residues: it is hashed one SK_TILE-position sub-tile
// at a time (residue buffer re-staged, next sub-tile's 16 bytes per lane prefetched while the current one is hashed),
// the windows that pass the threshold are appended to an LDS list (wave scan of the per-lane keep counts + one LDS
// atomic per wave), and the sort / unique phases run ONCE over the compacted list: the same number of kept hashes per
// tile as at scaled = 1.  Bucket space shrinks with it: sequence s gets buckets [ls / c + srel, ... + ceil(nw / c)).
template <int MODE, int CMP, int KC>
KS_DEV void sk_tile_body(const sk_args &A, const u32 tile_in) {
    static_assert(!(CMP && MODE), "the compacting variant is for shared tiles");
    __shared__ __attribute__((aligned(16))) u64 res_w[(SK_TILE + SK_PAD) / 8];
    __shared__ __attribute__((aligned(16))) u32 cnt[SK_TILE / 2 + 4]; // bucket counts, then starts: 16 bits each
    __shared__ __attribute__((aligned(8))) u16 dseq[SK_SEQ_CAP + 2];                               // sorted position (kept rank) at each sequence start
    __shared__ u16 dsd[SK_SEQ_CAP + 2];                                                            // tiles with repeats: distinct rank there
    __shared__ __attribute__((aligned(16))) u64 tmp[SK_TILE];
    __shared__ u32 flagbits[SK_NFLAG];
    __shared__ u32 flagpre[SK_NFLAG + 1];
    __shared__ u32 scan_smem[2 * (SK_THREADS / 64 + 1)];
    __shared__ u32 loff[SK_SEQ_CAP + 2];
    __shared__ u8 lut_s[256];
#ifdef SK_LDS_PAD // diagnostic builds only: fewer workgroups per CU
    __shared__ u32 lds_pad[SK_LDS_PAD / 4];
    if (threadIdx.x == 0 && A.n_res == 0xfffffffffffffffULL) lds_pad[A.k] = 1;
#endif
    __shared__ u32 bins[256 + 1];  // postings: count per digit, then the digit's start inside the tile (+ a word that takes the adds of 0)
    __shared__ unsigned long long gaddr[256]; // postings: slot of the tile's first element of each digit, minus the digit's start inside the tile

    const u32 tid = threadIdx.x;
    u8 *res_b = (u8 *)res_w;
#ifdef SK_STAMP
    unsigned long long sk_t_prev = clock64();
#endif

    // ---- phase 0: tile -> sequence range (planned ahead), zero LDS state, stage LUT + sequence boundaries
    __shared__ u32 tile_s;
    __shared__ u32 ext_n;
    __shared__ u32 ndup_s; // repeats counted in phase 5
    __shared__ u32 ext_seq[4], ext_cnt[4], ext_d[4];
    __shared__ unsigned long long lb_sum[SK_THREADS / 64]; // look-back: per wave, aggregates up to its first inclusive prefix
    __shared__ u32 lb_pre[SK_THREADS / 64];                // ... and whether it saw one
    u32 tile = tile_in;
    constexpr u32 NCH = (SK_TILE + SK_PAD) / 16; // 16-byte chunks of a staged tile
    static_assert(NCH <= SK_THREADS, "one staging chunk per thread");
    // A workgroup's start is a chain of dependent memory latencies (ticket -> tile plan -> offsets); everything that
    // does not depend on the previous link is issued beside it: LDS zeroing and the LUT ride on the ticket atomic, and
    // the tile's residues (MODE 0: a fixed window of SK_TILE bytes from tile * R) are requested with the plan entry.
    u32 ticket_v = 0;
    // Tile id: workgroups are dispatched in blockIdx order, so blockIdx.x already is an id under which every predecessor
    // a look-back waits for is running or done — and it costs no memory round trip (3.54 -> 3.29 ms).  That order is
    // an observed property of the dispatcher, not a documented one: the look-back's bounded spin turns a violation
    // into a flag, and the host then repeats the launch with ids drawn from an atomic ticket (use_ticket), which
    // guarantees the order by construction.
    if (MODE == 0 && tid == 0) ticket_v = A.use_ticket ? atomicAdd(&A.ticket[0], 1u) : tile_in;
    // The encode table's byte is REQUESTED here and stored behind the plan loads below (stored at once, the workgroup sat out one
    // memory latency before it asked for anything else); moltype protein does not use the table at all.
    u32 lut_v = 0;
    if (!A.upper_only && tid < 256) lut_v = A.lut[tid];
    if (tid < 256) bins[tid] = 0;
    if (tid == 256) bins[256] = 0;
    static_assert(SK_TILE / 2 == SK_THREADS * 4, "one 16-byte store per thread zeroes the bucket counters");
    *(uint4 *)&cnt[tid * 4u] = make_uint4(0, 0, 0, 0);
    if (tid < 4) cnt[SK_TILE / 2 + tid] = 0;
    if (tid < SK_NFLAG) flagbits[tid] = 0;
    __shared__ u32 n_list_s; // CMP: kept windows appended to the list so far
    if (tid == 0) { ext_n = 0; n_list_s = 0; ndup_s = 0; }
    if (MODE == 0 && A.use_ticket) { // (uniform)
        // tiles are handed out in ticket order, so every predecessor a look-back waits for is already running
        if (tid == 0) tile_s = ticket_v;
        __syncthreads();
        tile = tile_s;
    }
    auto load_chunk = [&](u64 g) -> uint4 { // 16 residue bytes at g (16-byte aligned), zero-filled behind the batch
#ifdef SK_NT_RES
        if (g + 16 <= A.n_res) { const u32 *p4 = (const u32 *)(A.res + g); return make_uint4(__builtin_nontemporal_load(p4), __builtin_nontemporal_load(p4 + 1), __builtin_nontemporal_load(p4 + 2), __builtin_nontemporal_load(p4 + 3)); }
#else
        if (g + 16 <= A.n_res) return *(const uint4 *)(A.res + g);
#endif
        u32 t[4] = {0, 0, 0, 0};
        for (u32 b = 0; b < 16 && g + b < A.n_res; b++) t[b >> 2] |= (u32)A.res[g + b] << (8 * (b & 3));
        return make_uint4(t[0], t[1], t[2], t[3]);
    };
    auto stage_chunk = [&](uint4 v) { // through the encode LUT into the tile's residue buffer, 16 B per lane
        const u32 in[4] = {v.x, v.y, v.z, v.w};
        u32 o[4];
        if (A.upper_only) { // (uniform) moltype protein: the table only upper-cases — four bytes at a time, no table, no barrier for it
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const u32 y = in[d] & 0x7f7f7f7fu; // (no carry between bytes: 0x7f + 0x1f < 0x100)
                const u32 lower = (y + 0x1f1f1f1fu) & ~(y + 0x05050505u) & ~in[d] & 0x80808080u; // bytes in 'a' .. 'z'
                o[d] = in[d] ^ (lower >> 2);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 4; d++)
                o[d] = (u32)lut_s[in[d] & 255u] | ((u32)lut_s[(in[d] >> 8) & 255u] << 8) |
                       ((u32)lut_s[(in[d] >> 16) & 255u] << 16) | ((u32)lut_s[in[d] >> 24] << 24);
        }
        *(uint4 *)(res_b + (size_t)tid * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    uint4 rv = make_uint4(0, 0, 0, 0);
    const bool packed = MODE == 0 && !CMP && A.R == 0; // (uniform) packed tiles: the window starts where the plan says
    // The plan entries of the tile — where its residues start, its first and last sequence — are three independent loads: all
    // three are requested before the first is used (in source order "g0, residues(g0), sequences" the residues' request waited
    // for g0 and the sequence range was only asked for behind it: four dependent memory latencies from launch to the first
    // residue in LDS, now two: {table byte, g0, sequence range} -> {residues, sequence boundaries}).
    tile = (u32)__builtin_amdgcn_readfirstlane((int)tile); // (uniform: scalar loads)
    u64 g0p = 0;
    if (packed) g0p = A.tile_g0[tile];
    u32 s_first, s_end;
    if (MODE == 1) { s_first = A.seq_list[tile]; s_end = s_first + 1; }
    else { s_first = A.seq_list[tile]; s_end = A.seq_list[tile + 1]; }
    if (!A.upper_only && tid < 256) lut_s[tid] = (u8)lut_v;
    if (MODE == 0 && tid < NCH) rv = load_chunk((packed ? g0p : (u64)tile * A.R) + (u64)tid * 16); // R is a multiple of 16
    if (s_first >= s_end) {
        if (MODE == 0 && tid == 0)
            __hip_atomic_store(&A.tile_status[tile], SK_FLAG_AGG | 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // local coordinates: MODE 0 counts from the tile's first byte, MODE 1 from the (aligned) start of its sequence
    const u64 g0 = MODE == 0 ? (packed ? g0p : (u64)tile * A.R) : (A.offs[s_first] & ~15ULL); // A.res is 16-byte aligned (checked on the host)
    const u32 ns = s_end - s_first;
    sk_bounds B;
    B.loff = loff; B.goff = A.offs; B.g0 = g0; B.s_first = s_first; B.in_lds = ns <= SK_SEQ_CAP;
    if (B.in_lds)
        for (u32 i = tid; i <= ns; i += SK_THREADS) {
            u64 v = A.offs[s_first + i] - g0;
            loff[i] = v > 0x7fffffffULL ? 0x7fffffffu : (u32)v;
        }

    SK_STAMP_AT(0);
    // ---- phase 1: residues -> LDS through the encode LUT, 16 B per lane
    if (MODE == 1) {
        u64 span_end = A.offs[s_end];
        if (span_end > g0 + SK_TILE) span_end = g0 + SK_TILE;
        if (tid < NCH) {
            const u64 g = g0 + (u64)tid * 16;
            if (g < span_end) rv = load_chunk(g);
        }
        __syncthreads(); // lut_s
    }
    if (MODE == 0 && !A.upper_only) __syncthreads(); // the LUT (the tile's global loads are in flight; the zeroed LDS state is first touched behind the next barrier)
    if (tid < NCH) stage_chunk(rv);
    __syncthreads();

    SK_STAMP_AT(1);
    // ---- phase 2: hash 8 consecutive windows per thread, bucket + arrival slot via LDS atomics
    const u32 q0 = tid * SK_E;
    sk_seq q;
    const u32 kk = KC ? (u32)KC : A.k;
    const bool onerun = kk >= SK_E; // (uniform; compile-time for the folded k-mer sizes) see the note behind sk_place_window
    {
        // first sequence of the tile whose end lies beyond q0 — whose LAST WINDOW lies at or beyond q0 when k >= SK_E
        const u32 key = onerun ? q0 + kk - 1u : q0;
        u32 lo = s_first, hi = s_end;
        while (lo < hi) {
            u32 mid = lo + ((hi - lo) >> 1);
            if (B.at(mid + 1) > key) hi = mid; else lo = mid + 1;
        }
        q.s = lo;
        sk_load_seq(q, A, B, s_end);
    }
    const u64 *wl = res_w + tid; // word at byte q0
    u64 h[SK_E];
    u32 bo[SK_E];
    // (uniform) postings are wanted and this variant ranks them while it hashes (the compacting variant keeps the later pass)
#ifdef SK_NO_EARLY // (diagnostic builds only)
    const bool early = false;
#else
    const bool early = !CMP && A.part_keys != nullptr && B.in_lds;
#endif
    u32 rkp[SK_E / 2] = {0, 0, 0, 0}; // ranks inside (tile, digit) of my windows, 16 bits each
    u32 kept_mask = 0, srel_one = 0;  // one-run threads (k >= SK_E): which of my windows were kept; my sequence (relative, capped)
    if (CMP) {
        // ---- phase 2, compacting variant: hash sub-tile by sub-tile, keep the windows under the threshold in an LDS list
        // (hash in tmp, its sequence in the counter words, which are not counting yet), then bucket the compacted list
        u8 *slist = (u8 *)cnt;
        const u32 lane = tid & 63u;
        u32 end_l = B.at(s_end);
        if (end_l > A.span) end_l = A.span;
        const u32 n_sub = B.in_lds ? (end_l + SK_TILE - 1) / SK_TILE : 0; // (more sequences than the LDS table holds: flagged below)
        for (u32 sub = 0; sub < n_sub; sub++) {
            if (sub > 0) {
                if (tid < NCH) stage_chunk(rv);
                __syncthreads();
            }
            SK_STAMP_AT(9); // (stamps 9 .. 11: the compacting loop's stage / hash + append / closing barrier)
            if (sub + 1 < n_sub && tid < NCH) // next sub-tile's residues travel while this one is hashed
                rv = load_chunk((u64)tile * A.R + (u64)(sub + 1) * SK_TILE + (u64)tid * 16);
            const u32 p0 = sub * SK_TILE + q0;
            const bool active = p0 < end_l;
            if (active) { // the sequence that holds (or follows) position p0 — whose last window lies at or beyond it when k >= SK_E
                const u32 key = onerun ? p0 + kk - 1u : p0;
                u32 lo = s_first, hi = s_end;
                while (lo < hi) {
                    u32 mid = lo + ((hi - lo) >> 1);
                    if (B.at(mid + 1) > key) hi = mid; else lo = mid + 1;
                }
                q.s = lo;
                sk_load_seq(q, A, B, s_end);
            }
            // all eight windows behind one scan and one cursor atomic where the eight live hashes fit the register budget next to
            // the prefetched residues (k = 16: 75 registers, configs[2] launch 0.147 -> 0.143 ms); two halves of four otherwise
            // (k = 24 spills 56 bytes with eight and is 4 % slower)
            if constexpr (KC == 16) {
                sk_cmp_half<0, 8, KC>(A, B, q, wl, p0, active, s_first, s_end, lane, tmp, slist, &n_list_s);
            } else {
                sk_cmp_half<0, 4, KC>(A, B, q, wl, p0, active, s_first, s_end, lane, tmp, slist, &n_list_s);
                sk_cmp_half<4, 4, KC>(A, B, q, wl, p0, active, s_first, s_end, lane, tmp, slist, &n_list_s);
            }
            SK_STAMP_AT(10);
            __syncthreads(); // the sub-tile is hashed (its residues may be overwritten) and its appends are visible
            SK_STAMP_AT(11);
        }
        const u32 n_list = n_list_s;
        if (!B.in_lds || n_list > SK_TILE) {
            // more kept hashes (repeats / skewed input) or more sequences than the LDS lists take: nothing from this launch is
            // used — the host repeats the batch with the plain variant — but the look-back chain must not stall
            if (tid == 0) {
                atomicOr(&A.ticket[1], 4u);
                __hip_atomic_store(&A.tile_status[tile], (tile == 0 ? SK_FLAG_PRE : SK_FLAG_AGG) | 0ULL, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // my 8 entries of the list (strided over the threads: conflict-free LDS reads, and every lane gets its share)
        u32 srl[SK_E];
#pragma unroll
        for (int i = 0; i < SK_E; i++) {
            const u32 idx = (u32)i * SK_THREADS + tid;
            const bool live = idx < n_list;
            h[i] = live ? tmp[idx] : 0;
            srl[i] = live ? (u32)slist[idx] : 0xffffffffu;
        }
        __syncthreads(); // list read: tmp is free for the bucket order, cnt for counting
        for (u32 i = tid; i < SK_TILE / 2 + 4; i += SK_THREADS) cnt[i] = 0;
        // per-sequence {first bucket, bucket multiplier} in the (dead) residue buffer: sequence i owns buckets
        // [ls / c + i, ls / c + i + ceil(nw / c)) — disjoint and ascending, < SK_TILE since span <= c * 3840
        uint2 *stab = (uint2 *)res_w;
        for (u32 i = tid; i < ns; i += SK_THREADS) {
            const u32 ls = loff[i], le = loff[i + 1], len = le - ls;
            const u32 nw = (len >= A.k && le <= A.le_cap) ? len - A.k + 1 : 0;
            const u32 nb = sk_div(nw + A.c_div - 1, A.c_div, A.c_rcp);
            stab[i] = make_uint2(sk_div(ls, A.c_div, A.c_rcp) + i, sk_bucket_mul(nb, A.sfix));
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SK_E; i++) {
            bo[i] = 0xffffffffu;
            if (srl[i] != 0xffffffffu) {
                const uint2 t = stab[srl[i]];
                const u32 b = t.x + __umulhi((u32)(h[i] >> 32), t.y);
                const u32 sh = (b & 1u) * 16u;
                const u32 o = (atomicAdd(&cnt[b >> 1], 1u << sh) >> sh) & 0xffffu;
                bo[i] = (srl[i] << 24) | (b << 12) | o;
            }
        }
    } else
    // a tile's sequences end, on average, two thirds of the way through its SK_TILE positions: the threads behind
    // the last residue (whole waves, mostly) have nothing to hash
    if (q0 < B.at(s_end)) {
        h[0] = sk_hash_window<0, KC>(wl, A.k, A.seed);
        h[1] = sk_hash_window<1, KC>(wl, A.k, A.seed);
        h[2] = sk_hash_window<2, KC>(wl, A.k, A.seed);
        h[3] = sk_hash_window<3, KC>(wl, A.k, A.seed);
        h[4] = sk_hash_window<4, KC>(wl, A.k, A.seed);
        h[5] = sk_hash_window<5, KC>(wl, A.k, A.seed);
        h[6] = sk_hash_window<6, KC>(wl, A.k, A.seed);
        h[7] = sk_hash_window<7, KC>(wl, A.k, A.seed);
        if (onerun) {
            const u32 srel = q.s - B.s_first;
            const u32 s24 = (srel < 254u ? srel : 254u) << 24;
            const bool thr = A.max_hash != ~0ULL; // (uniform)
            // The eight bucket counters' atomics leave together and are awaited once: a window that is not kept adds 0 to a spare
            // counter word instead of sitting out in a branch of its own (which put a wait behind every atomic: eight LDS round
            // trips in a row per thread, sixteen with the posting ranks below).
            u32 bb[SK_E], rt[SK_E];
            u32 kpm = 0;
#pragma unroll
            for (int i = 0; i < SK_E; i++) {
                bool keep = (q0 + (u32)i - q.ls) < q.nw && h[i] != 0;
                if (thr) keep = keep && h[i] <= A.max_hash;
                kpm |= keep ? (1u << i) : 0u;
                const u32 b = q.ls + __umulhi((u32)(h[i] >> 32), q.mul);
                bb[i] = keep ? b : (u32)SK_TILE + 2u; // (counter word SK_TILE / 2 + 1: never a bucket's)
            }
#pragma unroll
            for (int i = 0; i < SK_E; i++)
                rt[i] = atomicAdd(&cnt[bb[i] >> 1], ((kpm >> i) & 1u) << ((bb[i] << 4) & 31u)); // two 16-bit counters per word
#pragma unroll
            for (int i = 0; i < SK_E; i++)
                bo[i] = ((kpm >> i) & 1u) ? (s24 | (bb[i] << 12) | __builtin_amdgcn_ubfe(rt[i], bb[i] << 4, 16u)) : 0xffffffffu;
            kept_mask = kpm; srel_one = s24 >> 24;
        } else {
#pragma unroll
            for (int i = 0; i < SK_E; i++) bo[i] = sk_place_window(A, q0 + i, h[i], q, B, s_end, cnt);
        }
        // Postings (query side): a kept hash's rank inside its partition digit is one more LDS atomic — taken HERE, while the
        // vector ALU is what the workgroup waits for and the LDS idles.  A tile without repeats (nearly all) then emits its
        // postings straight from these window-order registers: no sequence lookup, no counting phase of its own.
        if (early) {
            if (A.part_kshift) { // (uniform; decided once, not per window: the digit is a shift at scaled = 1)
                u32 rr[SK_E];
#pragma unroll
                for (int i = 0; i < SK_E; i++) {
                    const bool kept = bo[i] != 0xffffffffu;
                    rr[i] = atomicAdd(&bins[kept ? (((u32)(h[i] >> 32) >> A.part_kshift) & A.part_mask) : 256u], kept ? 1u : 0u); // < SK_TILE
                }
#pragma unroll
                for (int i = 0; i < SK_E; i++) rkp[i >> 1] |= (bo[i] != 0xffffffffu ? rr[i] : 0u) << ((i & 1) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < SK_E; i++)
                    if (bo[i] != 0xffffffffu)
                        rkp[i >> 1] |= atomicAdd(&bins[__umulhi((u32)(h[i] >> 32), A.part_K) & A.part_mask], 1u) << ((i & 1) * 16);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < SK_E; i++) { h[i] = 0; bo[i] = 0xffffffffu; }
    }
    __syncthreads();

    SK_STAMP_AT(2);
    if (MODE == 0) { // a deferred (medium / long) sequence that starts inside this tile (at most one: the last): the length of its
                     // CSR slot — its kept count — is known from the earlier launches (visible behind phase 3's barrier)
        for (u32 s = s_first + tid; s < s_end; s += SK_THREADS) {
            const u32 ls = B.at(s), le = B.at(s + 1);
            if (le > A.le_cap || le - ls > A.max_len_tile) {
                const u32 e = atomicAdd(&ext_n, 1u);
                if (e < 4) { ext_seq[e] = s; ext_cnt[e] = A.kept[s]; ext_d[e] = 0; }
            }
        }
    }
    // ---- phase 3: bucket counts -> bucket starts (exclusive scan over the tile).  The thread that scans a bucket knows its
    // size: buckets that hold more than TWO hashes are LISTED here and put in order in place by phase 5 — nothing is ranked
    // element by element (that cost as much as hashing: 812 VALU instructions per wave against 796, profiles/r02_sq_counters.md).
    // The lists live in the residue buffer, dead since the hash phase (the compacting variant keeps its bucket table in
    // the first 8 bytes per sequence of it): 16-bit entries; qb = buckets of more than SK_C3MAX (never more than
    // SK_TILE / (SK_C3MAX + 1) of them), q3 = buckets of 3 .. SK_C3MAX.  Pairs stay with the thread that scanned them (`pairs`).
    const u32 q_skip = CMP ? 2u * (ns < SK_SEQ_CAP + 1u ? ns : SK_SEQ_CAP + 1u) : 0u; // (8 bytes of bucket table per sequence)
    u16 *qb = (u16 *)((u32 *)res_w + q_skip);
    u16 *q3 = qb + SK_QB_CAP;
    const u32 q_ents = ((SK_TILE + SK_PAD) / 4 - q_skip) * 2 - SK_QB_CAP;
    u32 q3cap = q_ents;
    if (A.debug_qcap) q3cap = q3cap < A.debug_qcap ? q3cap : A.debug_qcap; // (tests: full lists)
    u32 ovf = 0; // my buckets that found no room in a list (bit layout of the class masks below): I put them in order myself
    u32 pairs = 0; // my buckets that hold exactly two hashes (same bit layout): put in order by me in phase 5
    {
        const uint4 w4 = *(const uint4 *)&cnt[q0 >> 1]; // 8 consecutive 16-bit counters
        const u32 w[4] = {w4.x, w4.y, w4.z, w4.w};
        const u32 hs = w[0] + w[1] + w[2] + w[3]; // (both halves at once: a tile's counts sum to <= SK_TILE)
        const u32 s = (hs & 0xffffu) + (hs >> 16);
        // size classes of my 8 buckets, two counters per word at a time (counts <= SK_TILE, so no carry leaves a half):
        // bit j = counter 2j, bit 16 + j = counter 2j + 1
        u32 ge2 = 0, ge3 = 0, big = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ge2 |= (((w[j] + 0x7ffe7ffeu) & 0x80008000u) >> 15) << j;
            ge3 |= (((w[j] + 0x7ffd7ffdu) & 0x80008000u) >> 15) << j;
        }
        if (s > SK_C3MAX) { // (a bucket of more than SK_C3MAX hashes needs that many among my eight: rare)
#pragma unroll
            for (int j = 0; j < 4; j++) big |= (((w[j] + (0x8000u - SK_C3MAX - 1u) * 0x00010001u) & 0x80008000u) >> 15) << j;
        }
        u32 m2 = ge2 & ~ge3, m3 = ge3 & ~big;
        // Pairs (18 % of the buckets, 1.5 per thread) are NOT listed: listing them cost more than it balanced — a loop of ~18 vector
        // instructions per pair that every wave runs as often as its fullest lane (4 - 5 times), plus the list's own pass in
        // phase 5 —; their owner keeps this mask and swaps them in place in phase 5, 8 predicated steps without a loop.
        pairs = m2;
        const u32 code = ((u32)__popc(m3) << 12) | ((u32)__popc(big) << 24); // (tile totals fit the fields)
        u32 total, at;
        u32 ex = sk_block_excl_scan2(s, code, scan_smem, &total, &at);
        u32 i3 = (at >> 12) & 0xfffu, ib = at >> 24;
        u32 o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { // starts <= 4096 fit 16 bits
            const u32 lo = w[j] & 0xffffu;
            o[j] = ex | ((ex + lo) << 16);
            ex += lo + (w[j] >> 16);
        }
        *(uint4 *)&cnt[q0 >> 1] = make_uint4(o[0], o[1], o[2], o[3]);
        if (tid == SK_THREADS - 1) cnt[SK_TILE / 2] = total | (total << 16);
        // list entries: bucket numbers
        while (m3) {
            const u32 bit = (u32)__builtin_ctz(m3);
            m3 &= m3 - 1u;
            if (i3 < q3cap) q3[i3] = (u16)(q0 + 2u * (bit & 15u) + (bit >> 4)); else ovf |= 1u << bit;
            i3++;
        }
        while (big) {
            const u32 bit = (u32)__builtin_ctz(big);
            big &= big - 1u;
            qb[ib++] = (u16)(q0 + 2u * (bit & 15u) + (bit >> 4));
        }
    }
    __syncthreads();
    // (the packed 16-bit starts read as a u16 array: one ds_read_u16 instead of read + shift + mask)
    auto bstart = [&](u32 b) -> u32 { return ((const u16 *)cnt)[b]; };
    const u32 n_kept = bstart(SK_TILE);
    // ---- decoupled look-back, step 1 (MODE 0): publish this tile's aggregate — HERE, half a tile's life before its own look-back.
    // The aggregate is the tile's KEPT count (repeats included), not its distinct count: the CSR gives every sequence a slot as
    // long as its kept hashes and the distinct ones fill its head (ks_sketches::d_counts says how many; slots only differ from
    // runs where a sequence repeats a k-mer).  The distinct count is known after the sort (phase 5), the kept count after the
    // hash: a successor that arrives at its look-back finds aggregates published microseconds ago instead of waiting for
    // its neighbours' sorts (round 3: -DSK_NO_LOOKBACK took 0.59 ms off a 2.11 ms launch, 0.17 ms off one that emits postings).
    const u32 ne = MODE == 0 ? (ext_n < 4 ? ext_n : 4) : 0;
    u64 agg = n_kept;
    for (u32 e = 0; e < ne; e++) agg += ext_cnt[e];
    if (MODE == 0 && tid == 0 && tile != A.debug_skip_tile)
        __hip_atomic_store(&A.tile_status[tile], (tile == 0 ? SK_FLAG_PRE : SK_FLAG_AGG) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    SK_STAMP_AT(3);
    // ---- phase 4: scatter kept hashes into bucket order; sorted position at which every sequence's run starts
    {
        u32 st8[SK_E]; // (the eight bucket starts are read together — the bucket field of "no window" is a bucket like any other —, then stored behind)
#pragma unroll
        for (int i = 0; i < SK_E; i++) st8[i] = bstart(SK_BO_B(bo[i]));
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (bo[i] != 0xffffffffu) tmp[st8[i] + SK_BO_O(bo[i])] = h[i];
    }
    if (B.in_lds) {
        for (u32 i = tid; i <= ns; i += SK_THREADS) // (first bucket of a sequence: its local start, or the compacted base of the bucket table)
            dseq[i] = (u16)(i == ns ? n_kept : bstart(CMP ? ((const uint2 *)res_w)[i].x : loff[i]));
    } else { // (rare: more sequences than the LDS tables hold) the positions at which a run starts, as a bitmap for phase 6
        for (u32 s = s_first + tid; s < s_end; s += SK_THREADS) {
            const u32 ls = B.at(s);
            const u32 d = ls <= SK_TILE ? bstart(ls) : n_kept;
            if (d < n_kept) atomicOr(&flagbits[d >> 5], 1u << (d & 31));
        }
    }
    __syncthreads();

    SK_STAMP_AT(4);
    // ---- phase 5: put the listed buckets in order, in place.  tmp then holds the tile's kept hashes sorted by
    // (sequence, hash): equal hashes of a sequence are neighbours.
    // Equal hashes of a sequence share a bucket, so whoever orders a bucket also sees every repeat: their count makes the
    // tile's number of DISTINCT hashes known here, before anything is read back.
    u32 nd = 0; // repeats seen by me
    auto sort_small = [&](u32 sb, u32 c) { // insertion sort of tmp[sb, sb + c)
        for (u32 a = 1; a < c; a++) {
            const u64 x = tmp[sb + a];
            u32 j = a;
            while (j > 0) {
                const u64 y = tmp[sb + j - 1];
                if (y <= x) { nd += y == x ? 1u : 0u; break; }
                tmp[sb + j] = y;
                j--;
            }
            if (j != a) tmp[sb + j] = x;
        }
    };
    {
        const u32 qc = scan_smem[SK_THREADS / 64]; // list totals, left there by phase 3's scan
        const u32 n3 = ((qc >> 12) & 0xfffu) < q3cap ? ((qc >> 12) & 0xfffu) : q3cap;
        const u32 nb = qc >> 24;
        { // my pairs: the starts of my 8 buckets are where phase 3 left them (16 bits each)
            const uint4 st4 = *(const uint4 *)&cnt[q0 >> 1];
            const u32 st[4] = {st4.x, st4.y, st4.z, st4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 stj = st[j];
#pragma unroll
                for (int hh = 0; hh < 2; hh++)
                    if (pairs & (1u << (j + 16 * hh))) {
                        const u32 sb = hh ? stj >> 16 : stj & 0xffffu;
                        const u64 x = tmp[sb], y = tmp[sb + 1];
                        if (x > y) { tmp[sb] = y; tmp[sb + 1] = x; }
                        nd += x == y ? 1u : 0u;
                    }
            }
        }
        for (u32 e = SK_THREADS - 1u - tid; e < n3; e += SK_THREADS) {
            const u32 b = q3[e], sb = bstart(b);
            sort_small(sb, bstart(b + 1) - sb);
        }
        while (ovf) { // (rare: a list was full)
            const u32 bit = (u32)__builtin_ctz(ovf);
            ovf &= ovf - 1u;
            const u32 b = q0 + 2u * (bit & 15u) + (bit >> 4), sb = bstart(b);
            sort_small(sb, bstart(b + 1) - sb);
        }
        if (nb) { // (uniform; rare) buckets of more than SK_C3MAX hashes — repeats inside a low-complexity sequence, mostly:
                  // all threads rank one bucket's elements by counting (equal hashes keep their order), then move them
            for (u32 e = 0; e < nb; e++) {
                const u32 b = qb[e], sb = bstart(b), c = bstart(b + 1) - sb;
                u64 xs[SK_E];
                u32 rkb[SK_E];
#pragma unroll
                for (int i = 0; i < SK_E; i++) {
                    const u32 j = (u32)i * SK_THREADS + tid; // c <= SK_TILE = SK_E * SK_THREADS
                    rkb[i] = 0xffffffffu;
                    if (j < c) {
                        const u64 x = tmp[sb + j];
                        u32 r = 0, seen = 0;
                        for (u32 m = 0; m < c; m++) {
                            const u64 y = tmp[sb + m];
                            const u32 eqb = (y == x) & (m < j);
                            r += (y < x) | eqb;
                            seen |= eqb;
                        }
                        xs[i] = x; rkb[i] = r;
                        nd += seen;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < SK_E; i++)
                    if (rkb[i] != 0xffffffffu) tmp[sb + rkb[i]] = xs[i];
            }
        }
    }
    if (nd) atomicAdd(&ndup_s, nd);
    __syncthreads();
    const u32 n_rep = ndup_s;          // hashes of the tile that repeat an earlier one of their sequence
    const bool any_dup = n_rep != 0;   // (uniform)
    const u32 n_distinct = n_kept - n_rep;
    const bool posts = A.part_keys != nullptr && B.in_lds; // (uniform)
    if (tid == 0 && n_rep) atomicAdd((unsigned long long *)A.drops_out, (unsigned long long)n_rep); // slots the CSR leaves empty
    SK_STAMP_AT(5);
    // ---- phase 6 (tiles with repeats, tiles that emit postings): every thread takes 8 consecutive SORTED positions back into
    // registers (the LDS buffers are reused below).  A hash that equals its left neighbour inside the same sequence is a
    // repeat: the first of a run is the representative, the run's length its abundance.  A tile without repeats that emits
    // no postings — the index side, nearly always — skips all of this: its sorted run in tmp is what leaves.
    const u32 p0 = tid * SK_E;
    u32 rep = 0;          // bit i: position p0 + i holds a representative
    u32 srl[2] = {0, 0};  // sequence (relative to the tile's first) of each position, 8 bits each (tiles of <= SK_SEQ_CAP sequences)
    u32 rk[SK_E];         // postings: rank of my element inside (tile, digit)
    u32 ab[SK_E / 2];     // abundance of my representatives, 16 bits each (<= SK_TILE)
#pragma unroll
    for (int i = 0; i < SK_E / 2; i++) ab[i] = 0x00010001u;
    const bool fastp = early && !any_dup; // (uniform)
    if (early && any_dup) { // the ranks taken while hashing counted the repeats too: count again below
        if (tid < 256) bins[tid] = 0;
        __syncthreads();
    }
    if (fastp) {
        if (onerun) { // (uniform) the kept mask and the one sequence of the thread are what phase 2 left: nothing to unpack per window
            rep = kept_mask;
            srl[0] = srl[1] = srel_one * 0x01010101u;
#pragma unroll
            for (int i = 0; i < SK_E; i++) rk[i] = (rkp[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
        } else {
#pragma unroll
            for (int i = 0; i < SK_E; i++) {
                const bool kept = bo[i] != 0xffffffffu;
                rep |= (kept ? 1u : 0u) << i;
                srl[i >> 2] |= (kept ? (bo[i] >> 24) : 0u) << (8 * (i & 3));
                rk[i] = (rkp[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
            }
        }
    } else
    if (any_dup || posts) { // (uniform)
    if ((tid & ~63u) * SK_E < n_kept) { // (whole waves: the sequence lookup below works with all lanes of a wave)
        {
            const uint4 a = *(const uint4 *)&tmp[p0], b = *(const uint4 *)&tmp[p0 + 2], c = *(const uint4 *)&tmp[p0 + 4],
                        d = *(const uint4 *)&tmp[p0 + 6];
            h[0] = (u64)a.x | ((u64)a.y << 32); h[1] = (u64)a.z | ((u64)a.w << 32);
            h[2] = (u64)b.x | ((u64)b.y << 32); h[3] = (u64)b.z | ((u64)b.w << 32);
            h[4] = (u64)c.x | ((u64)c.y << 32); h[5] = (u64)c.z | ((u64)c.w << 32);
            h[6] = (u64)d.x | ((u64)d.y << 32); h[7] = (u64)d.z | ((u64)d.w << 32);
        }
        const u32 nl = p0 >= n_kept ? 0u : (n_kept - p0 < SK_E ? n_kept - p0 : SK_E);
        const u32 live = (1u << nl) - 1u; // bit i: position p0 + i < n_kept
        u32 heads = 0; // bit i: position p0 + i is the first of its sequence
        if (B.in_lds) {
            // Which sequence holds each of my positions: sequence s owns sorted positions [dseq[s], dseq[s + 1]) (empty runs
            // included), so the sequence of position p is the number of boundaries dseq[1 .. ns] at or before p.  Per WAVE
            // (512 consecutive positions from pw): the boundaries before pw are counted by all lanes together (4 table
            // entries per lane, one ballot each); the next 8 boundaries are read once (the same address in every lane)
            // and kept in registers, and a boundary at offset e from my first position adds 1 to the nibble of every position
            // from e on: 0x11111111 << 4e.  No chain of dependent LDS reads.  A wave whose 512 positions hold more than 8
            // boundaries (many tiny sequences) walks the table instead.
            const u32 lane = tid & 63u, pw = (tid & ~63u) * SK_E;
            u32 nb4 = 0;
            {
                const uint2 dd = *(const uint2 *)&dseq[4u * lane];
                const u32 dv[4] = {dd.x & 0xffffu, dd.x >> 16, dd.y & 0xffffu, dd.y >> 16};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const u32 j = 4u * lane + (u32)i;
                    nb4 += (u32)__popcll(__ballot(j >= 1u && j <= ns && dv[i] < pw));
                }
            }
            const u32 Bw = __builtin_amdgcn_readfirstlane(nb4); // boundaries before pw = the sequence that holds position pw - 1
            u32 acc = 0;
            u32 dlast = 0xffffu;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 j = Bw + 1u + (u32)k;
                const u32 dk = j <= ns ? (u32)dseq[j] : 0xffffu;
                const i32 e = (i32)dk - (i32)p0;
                const u32 ec = (u32)(e < 0 ? 0 : (e > 8 ? 8 : e));
                acc += (u32)(0x11111111ULL << (4u * ec));
                dlast = dk;
            }
            if (dlast >= pw + 64u * SK_E) { // (uniform) the common case: no further boundary inside this wave's positions
                const u32 below = ks_lane_below(acc >> 28); // boundaries (from pw) at or before the position left of mine
                u32 x = acc ^ ((acc << 4) | below);         // nibble i != 0: position p0 + i starts a sequence
                x |= x >> 1; x |= x >> 2; x &= 0x11111111u;
                x = (x | (x >> 3)) & 0x03030303u;
                x = (x | (x >> 6)) & 0x000f000fu;
                heads = (x | (x >> 12)) & 0xffu;
                u32 lo = acc & 0xffffu, hi = acc >> 16;
                lo = (lo | (lo << 8)) & 0x00ff00ffu; lo = (lo | (lo << 4)) & 0x0f0f0f0fu;
                hi = (hi | (hi << 8)) & 0x00ff00ffu; hi = (hi | (hi << 4)) & 0x0f0f0f0fu;
                srl[0] = lo + Bw * 0x01010101u;
                srl[1] = hi + Bw * 0x01010101u;
            } else {
                // the sequence that holds position p0: the last one whose run starts at or before it (empty runs skipped)
                u32 lo = 0, hi = ns; // dseq[lo] <= p0 < dseq[hi]
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if ((u32)dseq[mid] <= p0) lo = mid; else hi = mid;
                }
                u32 s = lo, nxt = dseq[s + 1];
#pragma unroll
                for (int i = 0; i < SK_E; i++) {
                    const u32 p = p0 + i;
                    while (p >= nxt && s + 1 < ns) { s++; nxt = dseq[s + 1]; }
                    heads |= (p == (u32)dseq[s] ? 1u : 0u) << i;
                    srl[i >> 2] |= s << (8 * (i & 3));
                }
            }
        } else {
            heads = ((const u8 *)flagbits)[tid]; // (set in phase 4 for tiles whose boundaries are not in LDS)
        }
        rep = live;
        if (any_dup) {
            u64 prev = p0 ? tmp[p0 - 1] : 0;
            rep = 0;
#pragma unroll
            for (int i = 0; i < SK_E; i++) {
                const bool r = h[i] != prev || ((heads >> i) & 1u);
                rep |= (r ? 1u : 0u) << i;
                prev = h[i];
            }
            rep = (rep | (p0 == 0 ? 1u : 0u)) & live;
        }
    }
    // The tile's postings (hash, sequence), partitioned on one hash digit, into the regions of that digit: digit-sort the
    // representatives through tmp so that each digit leaves as one run.  Step 1 (here, from registers: the counting touches
    // nothing but the digit bins, so the barrier that closes this phase closes it too): rank of every element inside its digit.
    if (posts) {
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (rep & (1u << i)) rk[i] = atomicAdd(&bins[sk_digit(h[i], A.part_K, A.part_kshift, A.part_mask)], 1u);
    }
    __syncthreads(); // (tmp is read, the LDS buffers may be reused; the digit bins are counted)
    }

    // Tiles without a repeat — for protein k-mers nearly all of them — are done: every kept hash is its own representative
    // with abundance 1 and its distinct rank IS its sorted position.  The others: bit-prefix over the representative flags.
    if (any_dup) { // (uniform)
        ((u8 *)flagbits)[tid] = (u8)rep; // bit p of the tile = position p holds a representative (threads behind n_kept: 0)
        __syncthreads();
        u32 v = tid < SK_NFLAG ? (u32)__popc(flagbits[tid]) : 0;
        u32 total;
        u32 ex = ks_block_excl_scan(v, scan_smem, &total);
        if (tid < SK_NFLAG) flagpre[tid] = ex;
        if (tid == 0) flagpre[SK_NFLAG] = total;
        // run length of each representative = distance to the next one (or to the end of the kept hashes)
        if (rep) {
            u32 after = n_kept; // first representative at or behind position p0 + 8
            for (u32 w = (p0 + SK_E) >> 5; w < SK_NFLAG; w++) {
                u32 m = flagbits[w];
                if (w == (p0 + SK_E) >> 5) m &= ~0u << ((p0 + SK_E) & 31u);
                if (m) { after = w * 32u + (u32)__builtin_ctz(m); break; }
            }
#pragma unroll
            for (int i = 0; i < SK_E; i++)
                if (rep & (1u << i)) {
                    const u32 later = rep >> (i + 1);
                    const u32 len = later ? (u32)__builtin_ctz(later) + 1u : after - (p0 + i);
                    ab[i >> 1] = (ab[i >> 1] & ~(0xffffu << ((i & 1) * 16))) | (len << ((i & 1) * 16));
                }
        }
        __syncthreads();
    }

    auto drank = [&](u32 x) -> u32 { // representatives among sorted positions < x
        if (!any_dup) return x < n_kept ? x : n_kept;
        return x >= SK_TILE ? n_distinct : flagpre[x >> 5] + (u32)__popc(flagbits[x >> 5] & ((1u << (x & 31)) - 1u));
    };
    u16 *abund_s = (u16 *)cnt; // abundance staging reuses the bucket-start words once they are dead (<= 4096 fits 16 bits)
    auto stage_reps = [&]() { // (tiles with repeats) representatives -> LDS in distinct-rank order, over the sorted run
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (rep & (1u << i)) {
                const u32 d = drank(p0 + i);
                tmp[d] = h[i];
                abund_s[d] = (u16)(ab[i >> 1] >> ((i & 1) * 16));
            }
    };
    // Postings, step 2: a slice of every digit's region for this tile — one device atomic per digit, REQUESTED by post_reserve
    // as soon as the digit counts stand and awaited by post_slices only when the slices are about to be stored (a shared tile
    // does its look-back and writes its CSR run in between: the atomics' round trip hides behind the look-back's) —, and the
    // digit starts inside the tile.
    u32 p_cnt = 0, p_off = 0;
    const u32 p_seg = (tid << A.part_sub_shift) | (blockIdx.x & ((1u << A.part_sub_shift) - 1u)); // (blockIdx.x & 7 names the XCD, i.e. the L2, these writes go through)
    auto post_reserve = [&]() {
        p_cnt = tid < 256 ? bins[tid] : 0;
        if (p_cnt) p_off = atomicAdd(&A.part_cursor[p_seg], p_cnt);
    };
    auto post_slices = [&]() {
        const u32 c = p_cnt, off = p_off, seg = p_seg;
        const u32 ex = sk_block_excl_scan1(c, scan_smem);
        if (tid < 256) {
            // element i of the digit-ordered tile goes to slot gaddr[digit] + i (the digit's start inside the tile taken off,
            // modulo 2^64).  A digit whose slice does not fit its region (skewed hashes: the host is told and partitions
            // this batch another way) lands in the SK_TILE spare slots behind the last region instead.
            const bool fits = (u64)off + c <= A.part_cap;
            if (c && !fits) atomicOr(&A.ticket[1], 2u);
            gaddr[tid] = (fits ? (u64)seg * A.part_cap + off : A.part_cap * ((u64)(A.part_mask + 1u) << A.part_sub_shift)) - ex;
            bins[tid] = ex;
        }
    };
    // Postings, step 3 (after a barrier that follows post_offsets and the last reader of tmp / cnt): digit order through tmp,
    // sequence codes through the counter words, then one run per digit.
    auto post_emit = [&]() {
        u16 *qrel = (u16 *)cnt; // sequence (relative) of the element staged at tmp[pos]
        u32 ds8[SK_E]; // (the eight digit starts are read together; a slot without a representative reads some digit's start and drops it)
#pragma unroll
        for (int i = 0; i < SK_E; i++) ds8[i] = bins[sk_digit(h[i], A.part_K, A.part_kshift, A.part_mask)];
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (rep & (1u << i)) {
                const u32 pos = ds8[i] + rk[i];
                tmp[pos] = h[i];
                qrel[pos] = (u16)((srl[i >> 2] >> (8 * (i & 3))) & 0xffu);
            }
        __syncthreads();
        // (Measured, round 4: the 16-bit value column four entries per store — groups of four consecutive elements inside one digit's
        // run have four consecutive slots: one 8-byte store at a 2-byte aligned address instead of four 2-byte stores — made the
        // query launch 3 % SLOWER (2.22 -> 2.29 ms): the unaligned stores split, and the column needs a loop of its own.)
        for (u32 i = tid; i < n_distinct; i += SK_THREADS) {
            const u64 hh = tmp[i];
            const u64 at = gaddr[sk_digit(hh, A.part_K, A.part_kshift, A.part_mask)] + i;
#ifndef SK_NO_POST_STORES // (diagnostic builds only: the kernel's time without the posting stores)
            const u32 qid = s_first + qrel[i];
            if (A.part_s) { // (uniform) 10-byte postings; the field lies in the high word (part_s >= 48): one shift + one bit-field insert
                const u32 sh = A.part_s - 32u, fm = 0xffu << sh;
                const u32 hi = (((u32)(hh >> 32)) & ~fm) | ((qid << sh) & fm);
#ifdef SK_NT_POST
                __builtin_nontemporal_store(((u64)hi << 32) | (u32)hh, &A.part_keys[at]);
                __builtin_nontemporal_store((u16)(qid >> 8), &((u16 *)A.part_vals)[at]);
#else
                A.part_keys[at] = ((u64)hi << 32) | (u32)hh;
                ((u16 *)A.part_vals)[at] = (u16)(qid >> 8);
#endif
            } else {
                A.part_keys[at] = hh;
                A.part_vals[at] = qid;
            }
#else
            if (at == 0xffffffffffffULL) A.part_vals[0] = (u32)hh + qrel[i];
#endif
        }
    };

    if (MODE == 1) {
        if (tid == 0) { A.counts[s_first] = n_distinct; A.kept[s_first] = n_kept; }
        SK_STAMP_AT(6);
        if (any_dup) { // (the sorted run itself otherwise)
            stage_reps();
            __syncthreads();
        }
        SK_STAMP_AT(7);
        // into the side buffer at the sequence's own offset; k_place_long moves it once its CSR slot is known
        const u64 r0 = A.offs[s_first];
        for (u32 d = tid; d < n_distinct; d += SK_THREADS) {
            A.out_hash[r0 + d] = tmp[d];
            A.out_abund[r0 + d] = any_dup ? (u32)abund_s[d] : 1u;
        }
        if (posts) { // a medium tile emits its own postings
            post_reserve();
            post_slices();
            __syncthreads();
            post_emit();
        }
    } else {
        // The CSR run leaves first, the postings behind it.  (Round 3 had it the other way round — the postings filled the
        // wait for the predecessors' aggregates, at the price of reading the sorted run back into registers and staging it
        // again behind the postings' digit order; now that the aggregates are published before the sort, the look-back is one
        // round trip, and the postings' own round trip — the slice reservations — is what hides behind it.)
        if (posts) post_reserve();
        if (B.in_lds && any_dup) // distinct rank at every sequence start (dseq keeps the kept ranks: they place the slots)
            for (u32 i = tid; i <= ns; i += SK_THREADS) dsd[i] = (u16)drank(dseq[i]);
        if (tid < ne) ext_d[tid] = B.in_lds ? (u32)dseq[ext_seq[tid] - s_first] : bstart(B.at(ext_seq[tid])); // kept rank at which the deferred sequence's slot opens
        if (B.in_lds && any_dup) stage_reps(); // (bucket starts are dead: dseq holds what the CSR needs; phase 6 has read the sorted run)
        SK_STAMP_AT(6);
        // ---- step 2: the predecessors' aggregates are summed back to the nearest inclusive prefix, SK_THREADS predecessors at a
        // time: wave w looks at the 64 tiles behind tile - 64 w.  (One wave looking back 64 at a time cost 0.5 ms of a 2.5 ms
        // launch: with ~770 tiles in flight, started ~35 ns apart, the nearest tile that has finished ITS look-back is
        // (look-back time / 35 ns) tiles away, so every round trip of the walk lengthens the next tile's walk — with tiles
        // 3/4 or half the size, 1000+ in flight, that feedback made the kernel 1.4x / 1.8x slower.)
        u64 excl = 0;
#ifdef SK_NO_LOOKBACK // diagnostic build only: wrong CSR, the kernel's time without the wait for the predecessors
        excl = (u64)tile * 3600u;
        if (false) {
#else
        if (tile > 0) { // (uniform)
#endif
            const u32 lane = tid & 63u, wave = tid >> 6;
            i64 idx = (i64)tile - 1;
            u32 spins = 0;
            const long long spin_t0 = wall_clock64();
            for (;;) {
                if (wave < SK_LB_WAVES) { // (uniform per wave: the other waves only meet the barrier — they used to reduce a wave of dummies)
                    const i64 mine = idx - (i64)tid;
                    u64 v = SK_FLAG_PRE; // before tile 0: inclusive prefix 0
                    if (mine >= 0) {
                        v = __hip_atomic_load(&A.tile_status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while ((v >> 62) == 0 && !ks_spin_expired(spin_t0, spins)) {
                            __builtin_amdgcn_s_sleep(1);
                            v = __hip_atomic_load(&A.tile_status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    if ((v >> 62) == 0) { atomicOr(&A.ticket[1], 1u); v = SK_FLAG_PRE; } // spin bound expired: flag the error, do not hang
                    const u64 is_pre = __ballot((v >> 62) == 2);
                    // lanes at or before the wave's first inclusive prefix contribute
                    const u32 first = is_pre ? (u32)__ffsll((long long)is_pre) - 1u : 64u;
                    u64 contrib = lane <= first ? (v & SK_VAL_MASK) : 0;
                    contrib = ks_wave_sum64(contrib);
                    if (lane == 0) { lb_sum[wave] = contrib; lb_pre[wave] = is_pre ? 1u : 0u; }
                }
                __syncthreads(); // (the first one also orders everything placed between publication and look-back)
                bool found = false;
#pragma unroll
                for (u32 w = 0; w < SK_LB_WAVES; w++)
                    if (!found) { excl += lb_sum[w]; found = lb_pre[w] != 0; }
#ifdef SK_LB_STATS // (diagnostic builds only: look-back rounds per launch, printed by the host)
                if (tid == 0) atomicAdd((unsigned long long *)(A.drops_out + 1), 1ULL);
#endif
                if (found) break;
                idx -= 64 * SK_LB_WAVES;
                __syncthreads(); // (lb_sum / lb_pre are written again)
            }
            if (tid == 0 && tile != A.debug_skip_tile)
                __hip_atomic_store(&A.tile_status[tile], SK_FLAG_PRE | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
        }
        const u64 base = excl;
        // final CSR offsets of every sequence that starts in this tile
        // kept rank at the local position p (sequences of a tile that does not hold its boundaries in LDS)
        auto krank = [&](u32 p) -> u32 { return bstart(p < SK_TILE ? p : SK_TILE); };
        for (u32 s = s_first + tid; s < s_end; s += SK_THREADS) {
            const u32 i = s - s_first;
            const u32 ls = B.at(s), le = B.at(s + 1);
            u64 pos = base + (B.in_lds ? (u32)dseq[i] : krank(ls));
            for (u32 e = 0; e < ne; e++) pos += ext_seq[e] < s ? ext_cnt[e] : 0;
            A.csr[s] = pos; // where the sequence's slot starts; its first counts[s] entries are its sketch
            if (!(le > A.le_cap || le - ls > A.max_len_tile)) { // (a deferred sequence's count came from its own launch)
                u32 c;
                if (B.in_lds) c = any_dup ? (u32)dsd[i + 1] - (u32)dsd[i] : (u32)dseq[i + 1] - (u32)dseq[i];
                else c = any_dup ? drank(krank(le)) - drank(krank(ls)) : krank(le) - krank(ls);
                A.counts[s] = c;
            }
        }
        if (s_end == A.n_seqs && tid == 0) { A.csr[A.n_seqs] = base + agg; *A.total_out = base + agg; }
        if (!B.in_lds && any_dup) {
            // Rare twice over: more sequences than the LDS tables hold (peptides) AND repeats.  No tables to look a representative's
            // sequence up in, so the run leaves sequence by sequence: one thread walks a sequence's kept positions in tmp (sorted,
            // repeats still in place; flagbits = representative flags, bucket starts still alive) and writes the representatives to
            // the head of the sequence's slot, each with the length of its run as its abundance.
            for (u32 s = s_first + tid; s < s_end; s += SK_THREADS) {
                const u32 ls = B.at(s), le = B.at(s + 1);
                if (le > A.le_cap || le - ls > A.max_len_tile) continue;
                u64 pos = base + krank(ls);
                for (u32 e = 0; e < ne; e++) pos += ext_seq[e] < s ? ext_cnt[e] : 0;
                const u32 b = krank(ls), e2 = krank(le);
                for (u32 p = b; p < e2;) { // (p holds a representative: the first position of a sequence always does)
                    u32 q = p + 1;
                    while (q < e2 && !((flagbits[q >> 5] >> (q & 31u)) & 1u)) q++;
                    if (pos < A.out_cap) { A.out_hash[pos] = tmp[p]; A.out_abund[pos] = q - p; }
                    pos++;
                    p = q;
                }
            }
        }
        SK_STAMP_AT(7);
        // ---- phase 8: coalesced write-out straight into the final CSR arrays (runs of medium / long neighbours
        // leave gaps that k_place_long fills)
        // (Measured, round 4: four of a thread's rounds at once — four LDS reads in flight, then eight stores back to back — made
        // the query launch 4 % SLOWER, and the same unrolling of the posting stores 1.5 %: bursts of stores delay the loads and
        // atomics other workgroups of the CU are waiting for.  One element per round it stays.)
        // (Measured, round 4: a separate loop for the common tile — no repeats, no deferred sequence: scalar base pointers, 32-bit
        // offsets, ~45 instead of ~106 vector instructions per wave — made the query launch 1.5 % SLOWER: the stores leave in a
        // tighter burst, and two more scalar registers spilled.)
        for (u32 d = tid; d < ((!B.in_lds && any_dup) ? 0u : n_distinct); d += SK_THREADS) {
            u64 pos = base + d;
            if (!any_dup) { // (uniform) distinct rank == kept rank: the run leaves as it lies
                for (u32 e = 0; e < ne; e++) pos += ext_d[e] <= d ? ext_cnt[e] : 0;
            } else {
                // the representative with distinct rank d sits in the slot of its sequence: the last one whose distinct start is <= d
                u32 lo = 0, hi = ns; // dsd[lo] <= d < dsd[hi]   (B.in_lds: tiles without the tables left above)
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if ((u32)dsd[mid] <= d) lo = mid; else hi = mid;
                }
                pos = base + (u32)dseq[lo] + (d - (u32)dsd[lo]);
                for (u32 e = 0; e < ne; e++) pos += ext_seq[e] < s_first + lo ? ext_cnt[e] : 0;
            }
#ifdef SK_NO_CSR_STORES // (diagnostic builds only: the kernel's time without the CSR stores)
            if (pos == 0xffffffffffffULL) A.out_abund[0] = (u32)tmp[d];
#else
            if (pos < A.out_cap) { // (capacity-bounded output: the host sees the true total in csr[n_seqs] and repeats larger)
                // (non-temporal: the run is written once and read by a later kernel, if at all; as ordinary stores its 3.5 GB per
                // launch washed through the L2 that the postings' partial lines are merged in — query launch 2.38 -> 2.24 ms,
                // index-side launch 1.69 -> 1.63 ms; -DSK_T_CSR keeps the ordinary stores)
#ifndef SK_T_CSR
                __builtin_nontemporal_store(tmp[d], &A.out_hash[pos]);
                __builtin_nontemporal_store(any_dup ? (u32)abund_s[d] : 1u, &A.out_abund[pos]);
#else
                A.out_hash[pos] = tmp[d];
                A.out_abund[pos] = any_dup ? (u32)abund_s[d] : 1u;
#endif
            }
#endif
        }
    }
    if (MODE == 0 && posts) { // the postings: digit order through tmp and the counter words, which the CSR run has just left
        __syncthreads();
        post_slices();
        __syncthreads();
        post_emit();
    }
    if (A.part_keys && !B.in_lds) { // sequence ids do not fit the element code: let the host repartition this batch
        if (tid == 0) atomicOr(&A.ticket[1], 2u);
    }
    SK_STAMP_AT(8);
}

// MODE 0: one shared tile per workgroup, ids in dispatch order (the look-back relies on it).  MODE 1: the medium
// sequences — their number is only known on the device (*A.n_list), so a fixed grid strides over the list.
template <int MODE, int CMP, int KC>
// (medium tiles — a side launch of few workgroups — may take more registers instead of spilling: 2 workgroups per CU)
__global__ __launch_bounds__(SK_THREADS, MODE == 1 ? 4 : SK_MINW) void k_sketch_tiles(sk_args A) {
    if (MODE == 1) {
        const u32 n = *A.n_list < A.n_list_cap ? *A.n_list : A.n_list_cap;
        for (u32 t = blockIdx.x; t < n; t += gridDim.x) {
            sk_tile_body<MODE, CMP, KC>(A, t);
            __syncthreads(); // the next sequence re-initialises the LDS state
        }
    } else {
        if (A.n_tiles_dev && blockIdx.x >= *A.n_tiles_dev) return; // (uniform)
        sk_tile_body<MODE, CMP, KC>(A, blockIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------
// long sequences: same algorithm, arrays in a global scratch slab, one workgroup per sequence
// ---------------------------------------------------------------------------------------------
struct sk_long_args {
    sk_args a;
    const u32 *long_ids;
    const u32 *n_long;
    u32 long_cap;  // allocated entries of long_ids
    u32 max_len;   // slab sizing
    u64 *slab_keys; // [grid][max_len]   window-order hashes (0 = dropped)
    u64 *slab_tmp;  // [grid][max_len]   bucket-ordered hashes
    u64 *slab_sorted; // [grid][max_len]
    u32 *slab_cnt;  // [grid][max_len+1] bucket counts -> starts
    u32 *slab_ord;  // [grid][max_len]   arrival slot per window
    u32 *slab_flag; // [grid][max_len+1] representative flags -> distinct ranks
    u32 *slab_ab;   // [grid][max_len]
    u64 *lg_hash;   // [n_res] output of long sequences (own buffer: a tile-packed run may overlap a long span)
    u32 *lg_abund;  // [n_res]
};

// n_cls[0] = medium sequences (own tile), n_cls[1] = long sequences (global-slab path)
__global__ __launch_bounds__(256) void k_find_long(const u64 *offs, u32 n_seqs, u32 R, u32 span, u32 *med_ids, u32 *long_ids, u32 *n_cls,
                                                   u32 med_cap, u32 long_cap) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    u32 cls = 2; // 0 medium, 1 long, 2 neither
    if (s < n_seqs) {
        const u64 len = offs[s + 1] - offs[s];
        // a sequence that does not end inside its shared tile's span is deferred: to a tile of its own if it fits one,
        // else to the global-slab path (with the plain span every sequence longer than SK_MED_MAX is deferred)
        if (R == 0) { if (len > PK_MAX_LEN) cls = 1; } // packed tiles: nothing is deferred for its position
        else if (len + 16 > span || sk_deferred(offs[s], len, R, span)) cls = len > SK_MED_MAX ? 1 : 0;
    }
    // one atomic per wave and class (the lists are short but every thread would hit the same two counters)
#pragma unroll
    for (u32 c = 0; c < 2; c++) {
        const u64 m = __ballot(cls == c);
        if (m == 0) continue;
        const u32 leader = (u32)__ffsll((long long)m) - 1u;
        u32 base = 0;
        if ((threadIdx.x & 63) == leader) base = atomicAdd(&n_cls[c], (u32)__popcll(m));
        base = __shfl(base, (int)leader, 64);
        // (the lists are sized from the caller's max_seq_len when it gave one: an entry beyond them means the hint was too
        // small — dropped here, reported by the host, which compares the hint with the measured maximum)
        if (cls == c && base + ks_lane_lt_count(m) < (c == 0 ? med_cap : long_cap)) (c == 0 ? med_ids : long_ids)[base + ks_lane_lt_count(m)] = s;
    }
}

// block-wide exclusive scan of a global u32 array in place; returns the total (uniform)
KS_DEV u32 sk_block_scan_global(u32 *a, u32 n, u32 *scan_smem) {
    u32 carry = 0;
    for (u32 base = 0; base < n; base += SK_THREADS) {
        u32 i = base + threadIdx.x;
        u32 v = i < n ? a[i] : 0;
        u32 total;
        u32 ex = ks_block_excl_scan(v, scan_smem, &total);
        if (i < n) a[i] = carry + ex;
        carry += total;
    }
    return carry;
}

// All cross-thread traffic goes through global memory inside ONE workgroup: barriers carry
// agent-scope fences so L1-resident lines written by atomics / other waves are re-read (rare path).
#define SK_LONG_SYNC() do { __threadfence(); __syncthreads(); } while (0)

__global__ __launch_bounds__(SK_THREADS) void k_sketch_long(sk_long_args L) {
    __shared__ __attribute__((aligned(16))) u64 res_w[(SK_TILE + SK_PAD) / 8];
    __shared__ u32 scan_smem[SK_THREADS / 64 + 1];
    __shared__ u8 lut_s[256];
    const sk_args &A = L.a;
    const u32 tid = threadIdx.x;
    u8 *res_b = (u8 *)res_w;
    if (tid < 256) lut_s[tid] = A.lut[tid];
    const u32 n_long = L.n_long[1] < L.long_cap ? L.n_long[1] : L.long_cap;
    const u64 slab = (u64)blockIdx.x * ((u64)L.max_len + 1);
    u64 *keys = L.slab_keys + slab, *tmp = L.slab_tmp + slab, *sorted = L.slab_sorted + slab;
    u32 *cnt = L.slab_cnt + slab, *ord = L.slab_ord + slab, *flag = L.slab_flag + slab, *abd = L.slab_ab + slab;

    for (u32 li = blockIdx.x; li < n_long; li += gridDim.x) {
        const u32 s = L.long_ids[li];
        const u64 b = A.offs[s], e = A.offs[s + 1];
        if (e - b > L.max_len) { // the caller's max_seq_len hint was too small: the host reports it (real maximum != hint)
            if (tid == 0) { A.counts[s] = 0; A.kept[s] = 0; }
            continue;
        }
        const u32 len = (u32)(e - b);
        const u32 nw = len >= A.k ? len - A.k + 1 : 0;
        const u32 mul = sk_bucket_mul(nw, A.sfix);
        for (u32 i = tid; i <= nw; i += SK_THREADS) { cnt[i] = 0; flag[i] = 0; }
        SK_LONG_SYNC();
        // hash in chunks of SK_TILE windows staged through LDS
        for (u32 w0 = 0; w0 < nw; w0 += SK_TILE) {
            const u64 gbase = b + w0;
            const u64 g0 = gbase & ~15ULL;
            const u32 shift = (u32)(gbase - g0);
            for (u32 c = tid; c < (SK_TILE + SK_PAD) / 16; c += SK_THREADS) {
                u64 g = g0 + (u64)c * 16;
                u32 t[4] = {0, 0, 0, 0};
                if (g < e) {
                    if (g + 16 <= A.n_res) {
                        uint4 v = *(const uint4 *)(A.res + g);
                        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                    } else {
                        for (u32 bb = 0; bb < 16 && g + bb < A.n_res; bb++) t[bb >> 2] |= (u32)A.res[g + bb] << (8 * (bb & 3));
                    }
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        t[d] = (u32)lut_s[t[d] & 255u] | ((u32)lut_s[(t[d] >> 8) & 255u] << 8) |
                               ((u32)lut_s[(t[d] >> 16) & 255u] << 16) | ((u32)lut_s[t[d] >> 24] << 24);
                }
                *(uint4 *)(res_b + (size_t)c * 16) = make_uint4(t[0], t[1], t[2], t[3]);
            }
            __syncthreads();
            // window w = w0 + j lives at LDS byte shift + j; j strided over threads
            for (u32 j = tid; j < SK_TILE && w0 + j < nw; j += SK_THREADS) {
                const u32 pos = shift + j;
                const u64 *w = res_w + (pos >> 3);
                const u32 bs = pos & 7;
                ks_murmur m;
                m.init(A.seed);
                const u32 nb = A.k >> 4, t = A.k & 15;
                u32 jj = 0;
                for (u32 bl = 0; bl < nb; bl++, jj += 2)
                    m.block(ks_funnel_rt(w[jj], w[jj + 1], bs), ks_funnel_rt(w[jj + 1], w[jj + 2], bs));
                if (t) {
                    u64 k1 = ks_funnel_rt(w[jj], w[jj + 1], bs), k2 = ks_funnel_rt(w[jj + 1], w[jj + 2], bs);
                    if (t > 8) k2 &= ks_mask_bytes(t - 8); else { k1 &= ks_mask_bytes(t); k2 = 0; }
                    m.tail(k1, k2, t);
                }
                u64 h = m.finish((u64)A.k);
                bool keep = h != 0 && h <= A.max_hash;
                keys[w0 + j] = keep ? h : 0;
                if (keep) ord[w0 + j] = atomicAdd(&cnt[__umulhi((u32)(h >> 32), mul)], 1u);
            }
            __syncthreads();
        }
        SK_LONG_SYNC();
        const u32 n_kept = sk_block_scan_global(cnt, nw + 1, scan_smem);
        SK_LONG_SYNC();
        for (u32 w = tid; w < nw; w += SK_THREADS) {
            u64 h = keys[w];
            if (h) tmp[cnt[__umulhi((u32)(h >> 32), mul)] + ord[w]] = h;
        }
        SK_LONG_SYNC();
        for (u32 w = tid; w < nw; w += SK_THREADS) {
            u64 h = keys[w];
            if (!h) continue;
            const u32 bk = __umulhi((u32)(h >> 32), mul), o = ord[w];
            const u32 sb = cnt[bk], c = cnt[bk + 1] - sb;
            u32 less = 0, eq = 0, eqb = 0;
            for (u32 j = 0; j < c; j++) {
                u64 x = tmp[sb + j];
                less += x < h;
                eq += x == h;
                eqb += (x == h) & (j < o);
            }
            if (eqb == 0) {
                const u32 p = sb + less;
                sorted[p] = h;
                abd[p] = eq;
                flag[p] = 1;
            }
        }
        SK_LONG_SYNC();
        const u32 n_distinct = sk_block_scan_global(flag, n_kept + 1, scan_smem);
        SK_LONG_SYNC();
        for (u32 p = tid; p < n_kept; p += SK_THREADS) {
            if (flag[p + 1] != flag[p]) {
                L.lg_hash[b + flag[p]] = sorted[p];
                L.lg_abund[b + flag[p]] = abd[p];
            }
        }
        if (tid == 0) {
            A.counts[s] = n_distinct; A.kept[s] = n_kept;
            if (n_kept != n_distinct) atomicAdd((unsigned long long *)A.drops_out, (unsigned long long)(n_kept - n_distinct));
        }
        SK_LONG_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// CSR assembly: only medium / long sequences need a copy (their runs were produced in side buffers
// before the tile kernel fixed their CSR positions); one workgroup per such sequence.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_place_long(const u32 *ids, const u32 *n_ids_dev, u32 ids_cap, const u64 *offs, const u64 *csr, const u32 *counts, const u64 *lg_hash,
                                                    const u32 *lg_abund, u64 *hashes, u32 *abunds, u64 out_cap, u64 *part_keys,
                                                    u32 *part_vals, u32 *part_cursor, u64 part_cap, u32 part_K,
                                                    u32 part_mask, u32 part_sub_shift, u32 *status, u32 part_s) {
    // a compacting tile that overflowed wrote no CSR offsets for its sequences (the host repeats the batch): nothing
    // here may be trusted then
    if (__hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u) return;
    const u32 n_ids = *n_ids_dev < ids_cap ? *n_ids_dev : ids_cap; // (only known on the device: a fixed grid strides over the list)
    for (u32 li = blockIdx.x; li < n_ids; li += gridDim.x) {
    const u32 s = ids[li];
    const u64 dst = csr[s], src = offs[s];
    u64 n = counts[s]; // (distinct hashes: the head of the sequence's slot)
    if (n > offs[s + 1] - src) n = offs[s + 1] - src; // (a run is never longer than its sequence)
    for (u64 i = threadIdx.x; i < n; i += 256) {
        const u64 h = lg_hash[src + i];
        if (dst + i < out_cap) {
            hashes[dst + i] = h;
            abunds[dst + i] = lg_abund[src + i];
        }
        if (part_keys) { // long sequences are rare: one device atomic per posting is fine here
            const u32 dg = ((ks_join_prefix(h, part_K) & part_mask) << part_sub_shift) | (blockIdx.x & ((1u << part_sub_shift) - 1u));
            const u64 slot = atomicAdd(&part_cursor[dg], 1u);
            if (slot < part_cap) {
                if (part_s) {
                    part_keys[(u64)dg * part_cap + slot] = (h & ~(0xffULL << part_s)) | ((u64)(s & 0xffu) << part_s);
                    ((u16 *)part_vals)[(u64)dg * part_cap + slot] = (u16)(s >> 8);
                } else {
                    part_keys[(u64)dg * part_cap + slot] = h;
                    part_vals[(u64)dg * part_cap + slot] = s;
                }
            } else {
                atomicOr(&status[1], 2u);
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// k-mer position table (ProteomeIndex::process_kmers, src/rust/index.rs:749-786; KmerInfo of kmer.rs:6-12):
// (sequence, start, hash) of every kept window, ordered by (sequence, start), in ONE pass.
// A tile is a fixed range of KP_R residue positions (windows of any sequence: there is no per-sequence sort here, so
// no deferral): residues staged through the LUT like the sketch kernel, 8 windows hashed per thread from LDS, kept
// windows compacted in position order, and the tile's slice of the output found by the same decoupled look-back
// (ticket-ordered tiles, 8-byte {flag, value} status words) the sketch kernel uses for its CSR.
// Not fused into k_sketch_tiles on purpose: that kernel is instruction-bound at its register limit (78 of 80 VGPRs),
// and this one re-hashes at the rate the 16 B per window of output allow anyway.
// ---------------------------------------------------------------------------------------------
#define KP_R SK_TILE
struct kp_args {
    const u8 *res;
    const u64 *offs;
    const u8 *lut;
    const u32 *tile_first;          // first sequence whose END lies beyond the tile's first position
    unsigned long long *tile_status;
    u32 *ticket;                    // [0] tile ids, [1] look-back gave up
    u64 *total;                     // kept windows of the whole batch (written by the last tile)
    u32 *out_seq, *out_start;
    u64 *out_hash;
    u64 n_res, max_hash, seed;
    u32 n_seqs, k, n_tiles;
    u32 use_ticket;                 // tile ids from the atomic ticket (1) or from blockIdx.x (0), as in k_sketch_tiles
};

__global__ __launch_bounds__(256) void k_kmerpos_plan(const u64 *offs, u32 n_seqs, u32 n_tiles, u32 *tile_first) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    // first s with offs[s + 1] > t * KP_R
    tile_first[t] = sk_lower_bound(offs + 1, 0, n_seqs, (u64)t * KP_R + 1);
}

__global__ __launch_bounds__(SK_THREADS) void k_kmerpos_tiles(kp_args A) {
    __shared__ __attribute__((aligned(16))) u64 res_w[(SK_TILE + SK_PAD) / 8];
    __shared__ __attribute__((aligned(16))) u64 stage[SK_TILE]; // compacted output staging: hashes, then (seq, start)
    __shared__ u32 lend[SK_SEQ_CAP + 2]; // local END of the tile's sequences (clamped)
    __shared__ u8 lut_s[256];
    __shared__ u32 scan_smem[SK_THREADS / 64 + 1];
    __shared__ u32 tile_s;
    __shared__ unsigned long long base_s;
    const u32 tid = threadIdx.x;
    constexpr u32 NCH = (SK_TILE + SK_PAD) / 16;
    // (tile ids in dispatch order; a launch whose look-back gave up is repeated with ticket ids: 73k tickets on one address
    // were 0.9 ms of queueing for a 1M-protein batch)
    u32 tile = blockIdx.x;
    if (A.use_ticket) { // (uniform; the repeat launch only)
        if (tid == 0) tile_s = atomicAdd(&A.ticket[0], 1u);
        __syncthreads();
        tile = tile_s;
    }
    // the encode table's byte, the tile's residues and its sequence range are requested together (the table byte used to be
    // stored — i.e. waited for — before anything else was asked for: see k_sketch_tiles)
    u32 lut_v = 0;
    if (tid < 256) lut_v = A.lut[tid];
    const u64 g0 = (u64)tile * KP_R;
    uint4 rv = make_uint4(0, 0, 0, 0);
    if (tid < NCH) {
        const u64 g = g0 + (u64)tid * 16;
        if (g + 16 <= A.n_res) {
            rv = *(const uint4 *)(A.res + g);
        } else {
            u32 t[4] = {0, 0, 0, 0};
            for (u32 b = 0; b < 16 && g + b < A.n_res; b++) t[b >> 2] |= (u32)A.res[g + b] << (8 * (b & 3));
            rv = make_uint4(t[0], t[1], t[2], t[3]);
        }
    }
    const u32 s_first = A.tile_first[tile];
    u32 s_last = A.tile_first[tile + 1]; // the sequence that holds the next tile's first position also ends here or later
    if (s_last >= A.n_seqs) s_last = A.n_seqs ? A.n_seqs - 1 : 0;
    const u32 ns = s_first < A.n_seqs ? s_last - s_first + 1 : 0;
    const bool in_lds = ns <= SK_SEQ_CAP;
    if (in_lds)
        for (u32 i = tid; i < ns; i += SK_THREADS) {
            const u64 v = A.offs[s_first + i + 1] - g0; // ends beyond the tile's first position: never negative
            lend[i] = v > 0x7fffffffULL ? 0x7fffffffu : (u32)v;
        }
    if (tid < 256) lut_s[tid] = (u8)lut_v;
    __syncthreads(); // the table
    if (tid < NCH) {
        const u32 in[4] = {rv.x, rv.y, rv.z, rv.w};
        u32 o[4];
#pragma unroll
        for (int d = 0; d < 4; d++)
            o[d] = (u32)lut_s[in[d] & 255u] | ((u32)lut_s[(in[d] >> 8) & 255u] << 8) |
                   ((u32)lut_s[(in[d] >> 16) & 255u] << 16) | ((u32)lut_s[in[d] >> 24] << 24);
        *(uint4 *)((u8 *)res_w + (size_t)tid * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    auto end_of = [&](u32 s) -> u32 { // local end of sequence s (s_first <= s <= s_last)
        if (in_lds) return lend[s - s_first];
        const u64 v = A.offs[s + 1] - g0;
        return v > 0x7fffffffULL ? 0x7fffffffu : (u32)v;
    };

    const u32 q0 = tid * SK_E;
    u64 h[SK_E];
    u32 sq[SK_E]; // sequence of a kept window, ~0 = not kept
    u32 n_keep = 0;
#pragma unroll
    for (int i = 0; i < SK_E; i++) { h[i] = 0; sq[i] = 0xffffffffu; }
    if (ns && g0 + q0 < A.n_res) {
        // the sequence that holds position q0: first one (from s_first) whose end lies beyond q0
        u32 lo = s_first, hi = s_last;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (end_of(mid) > q0) hi = mid; else lo = mid + 1;
        }
        u32 s = lo, e = end_of(s);
        const u64 *wl = res_w + tid;
        h[0] = sk_hash_window<0>(wl, A.k, A.seed);
        h[1] = sk_hash_window<1>(wl, A.k, A.seed);
        h[2] = sk_hash_window<2>(wl, A.k, A.seed);
        h[3] = sk_hash_window<3>(wl, A.k, A.seed);
        h[4] = sk_hash_window<4>(wl, A.k, A.seed);
        h[5] = sk_hash_window<5>(wl, A.k, A.seed);
        h[6] = sk_hash_window<6>(wl, A.k, A.seed);
        h[7] = sk_hash_window<7>(wl, A.k, A.seed);
#pragma unroll
        for (int i = 0; i < SK_E; i++) {
            const u32 p = q0 + i;
            while (s < s_last && p >= e) { s++; e = end_of(s); }
            // offsets are cumulative, so p lies inside sequence s as soon as p < e; the window must fit before e
            const bool keep = p < e && p + A.k <= e && h[i] != 0 && h[i] <= A.max_hash;
            if (keep) { sq[i] = s; n_keep++; }
        }
    }
    u32 total;
    const u32 ex = ks_block_excl_scan(n_keep, scan_smem, &total);
    // ---- decoupled look-back over the tiles' kept counts (see k_sketch_tiles)
    if (tid == 0)
        __hip_atomic_store(&A.tile_status[tile], (tile == 0 ? SK_FLAG_PRE : SK_FLAG_AGG) | (u64)total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        u64 excl = 0;
        if (tile > 0) {
            i64 idx = (i64)tile - 1;
            bool done = false;
            u32 spins = 0;
            const long long spin_t0 = wall_clock64();
            while (!done) {
                const i64 mine = idx - (i64)tid;
                u64 v = SK_FLAG_PRE;
                if (mine >= 0) {
                    v = __hip_atomic_load(&A.tile_status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while ((v >> 62) == 0 && !ks_spin_expired(spin_t0, spins)) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(&A.tile_status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if ((v >> 62) == 0) { atomicOr(&A.ticket[1], 1u); v = SK_FLAG_PRE; }
                const u64 is_pre = __ballot((v >> 62) == 2);
                const u32 first = is_pre ? (u32)__ffsll((long long)is_pre) - 1u : 64u;
                u64 contrib = tid <= first ? (v & SK_VAL_MASK) : 0;
                contrib = ks_wave_sum64(contrib);
                excl += contrib;
                if (is_pre) done = true; else idx -= 64;
            }
            if (tid == 0)
                __hip_atomic_store(&A.tile_status[tile], SK_FLAG_PRE | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            base_s = excl;
            if (tile == A.n_tiles - 1) *A.total = excl + total;
        }
    }
    __syncthreads();
    // kept windows leave through LDS in compacted order, so the three output streams are written as whole cache lines
    // (per-thread runs of <= 8 entries would touch 64 different 32-byte sectors per store instruction)
    const u64 base = base_s;
    {
        u32 o = ex;
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (sq[i] != 0xffffffffu) stage[o++] = h[i];
    }
    __syncthreads();
    for (u32 i = tid; i < total; i += SK_THREADS) A.out_hash[base + i] = stage[i];
    __syncthreads();
    {
        u32 *st_seq = (u32 *)stage, *st_start = st_seq + SK_TILE;
        u32 o = ex;
#pragma unroll
        for (int i = 0; i < SK_E; i++)
            if (sq[i] != 0xffffffffu) {
                st_seq[o] = sq[i];
                st_start[o] = (u32)(g0 + q0 + i - A.offs[sq[i]]);
                o++;
            }
        __syncthreads();
        for (u32 i = tid; i < total; i += SK_THREADS) {
            A.out_seq[base + i] = st_seq[i];
            A.out_start[base + i] = st_start[i];
        }
    }
}

// d_seq / d_start / d_hash are sized by the batch's window count (an upper bound on the kept windows)
int ks_kmerpos_tiles_launch(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, const ks_params *p, u32 *d_seq,
                            u32 *d_start, u64 *d_hash, u64 *n_out) {
    const u64 n_tiles64 = (n_res + KP_R - 1) / KP_R;
    if (n_tiles64 > 0x7ffffff0ULL) return ks_fail(ctx, KS_ERR_INVALID_ARG, "batch too large");
    const u32 n_tiles = (u32)n_tiles64;
    u32 *tile_first = nullptr, *ticket = nullptr;
    unsigned long long *status = nullptr;
    u64 *total = nullptr;
    int st = ks_alloc(ctx, &tile_first, (size_t)n_tiles + 1);
    if (st == KS_OK) st = ks_alloc(ctx, (u64 **)&status, (size_t)n_tiles);
    if (st == KS_OK) st = ks_alloc(ctx, &ticket, 2);
    if (st == KS_OK) st = ks_alloc(ctx, &total, 1);
    if (st == KS_OK) {
        ks_timer_begin(ctx, "kmerpos_plan");
        hipLaunchKernelGGL(k_kmerpos_plan, dim3((n_tiles + 256) / 256), dim3(256), 0, ctx->stream, d_offs, n_seqs, n_tiles, tile_first);
        ks_timer_end(ctx);
    }
    for (int attempt = 0; st == KS_OK && attempt < 2; attempt++) {
        const bool use_ticket = ctx->sketch_use_ticket || attempt == 1;
        (void)hipMemsetAsync(status, 0, (size_t)n_tiles * sizeof(u64), ctx->stream);
        (void)hipMemsetAsync(ticket, 0, 2 * sizeof(u32), ctx->stream);
        (void)hipMemsetAsync(total, 0, sizeof(u64), ctx->stream);
        kp_args A;
        memset(&A, 0, sizeof A);
        A.res = d_res; A.offs = d_offs; A.lut = ctx->d_lut + 256 * p->moltype; A.tile_first = tile_first;
        A.tile_status = status; A.ticket = ticket; A.total = total; A.out_seq = d_seq; A.out_start = d_start; A.out_hash = d_hash;
        A.n_res = n_res; A.max_hash = ks_max_hash(p->scaled); A.seed = p->seed; A.n_seqs = n_seqs; A.k = p->ksize; A.n_tiles = n_tiles;
        A.use_ticket = use_ticket ? 1u : 0u;
        ks_timer_begin(ctx, "kmerpos_tiles");
        hipLaunchKernelGGL(k_kmerpos_tiles, dim3(n_tiles), dim3(SK_THREADS), 0, ctx->stream, A);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "k-mer position launch failed"); break; }
        if (hipMemcpyAsync(ctx->h_pin, total, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ctx->h_pin + 1, ticket, 2 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            st = ks_fail(ctx, KS_ERR_HIP, "k-mer position status read failed");
            break;
        }
        bool gave_up = ((u32 *)(ctx->h_pin + 1))[1] != 0;
        if (!use_ticket && ks_dbg(ctx, KS_DBG_FORCE_TICKET_RETRY)) gave_up = true; // exercises the repeat
        if (!gave_up) { *n_out = ctx->h_pin[0]; break; }
        if (use_ticket) { st = ks_fail(ctx, KS_ERR_HIP, "k-mer positions: look-back gave up waiting for a predecessor tile"); break; }
        ctx->sketch_use_ticket = true; // dispatch order did not hold here: tickets from now on (shared with the sketch tiles)
        ctx->sketch_ticket_fallbacks++;
    }
    ks_pool_free(ctx, tile_first); ks_pool_free(ctx, status); ks_pool_free(ctx, ticket); ks_pool_free(ctx, total);
    return st;
}

// out[0] = k-mer windows, out[1] = longest sequence; under tile stride cand[c] (a tile spans `span` residues from its
// start): out[4 + c] = medium sequences (deferred, fit a tile of their own), out[4 + SK_NR + c] = long ones (deferred, do not)
struct sk_cands { u32 r[SK_NR]; };
__global__ __launch_bounds__(256) void k_seq_stats(const u64 *offs, u32 n_seqs, u32 k, u32 span, sk_cands cand, u64 *out) {
    u64 w = 0, mx = 0;
    u32 npk = 0; // sequences longer than PK_MAX_LEN (the long ones of a packed plan)
    u32 nd[SK_NR], nl[SK_NR];
#pragma unroll
    for (int c = 0; c < SK_NR; c++) { nd[c] = 0; nl[c] = 0; }
    for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seqs; s += gridDim.x * blockDim.x) {
        const u64 st = offs[s], len = offs[s + 1] - st;
        w += len >= k ? len - k + 1 : 0;
        mx = len > mx ? len : mx;
        npk += len > PK_MAX_LEN ? 1u : 0u;
        // (a sequence no longer than span - 15 - R fits wherever it starts: no 64-bit modulo for those; the two widest
        // strides are only ever taken when EVERY sequence is that short, so they are not counted)
#pragma unroll
        for (int c = 0; c < SK_NR - 2; c++)
            if (len + cand.r[c] + 15 > span) {
                const bool def = len + 16 > span || sk_deferred(st, len, cand.r[c], span);
                nd[c] += (def && len <= SK_MED_MAX) ? 1u : 0u;
                nl[c] += (def && len > SK_MED_MAX) ? 1u : 0u;
            }
    }
    for (int d = 32; d > 0; d >>= 1) {
        w += __shfl_down(w, d, 64);
        npk += __shfl_down(npk, d, 64);
        u64 o = __shfl_down(mx, d, 64);
        mx = o > mx ? o : mx;
#pragma unroll
        for (int c = 0; c < SK_NR; c++) { nd[c] += __shfl_down(nd[c], d, 64); nl[c] += __shfl_down(nl[c], d, 64); }
    }
    if ((threadIdx.x & 63) == 0 && npk) atomicAdd((unsigned long long *)&out[2], (unsigned long long)npk);
    // one set of device atomics per WORKGROUP (they all hit the same few words: per wave they were the whole run time)
    __shared__ u64 red[4][2 + 2 * SK_NR];
    const u32 wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = w; red[wave][1] = mx;
#pragma unroll
        for (int c = 0; c < SK_NR; c++) { red[wave][2 + c] = nd[c]; red[wave][2 + SK_NR + c] = nl[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 + 2 * SK_NR) {
        const u32 j = threadIdx.x;
        u64 v = red[0][j];
        for (u32 q = 1; q < 4; q++) v = j == 1 ? (red[q][j] > v ? red[q][j] : v) : v + red[q][j];
        if (v) {
            if (j == 1) atomicMax((unsigned long long *)&out[1], (unsigned long long)v);
            else atomicAdd((unsigned long long *)&out[j == 0 ? 0 : 4 + (j - 2)], (unsigned long long)v);
        }
    }
}

// One attempt at a batch.  variant bit 0: the compacting tile kernel may be used (scaled > 1); bit 1: outputs sized by the
// window count (always enough) instead of by the expected number of kept hashes.  *redo comes back non-zero (with KS_OK and
// nothing produced) when the attempt has to be repeated without the corresponding economy: 1 = a compacting tile
// overflowed its LDS lists, 2 = the batch kept more hashes than the bounded output arrays hold.
static int sketch_attempt(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, u32 max_seq_len, const ks_params *p,
                          int part_pbits, int part_fmt10, int variant, int allow_defer, int *redo, ks_sketches **out) {
    ks_sketches *S = new ks_sketches();
    memset(S, 0, sizeof *S);
    S->ctx = ctx;
    S->params = *p;
    S->n_seqs = n_seqs;
    *out = nullptr;
    *redo = 0;

    int st = KS_OK;
    u64 *d_stats = nullptr;
    u32 *kept = nullptr;
    u32 *med_ids = nullptr, *long_ids = nullptr, *n_cls = nullptr, *tile_first = nullptr, *ticket = nullptr, *part_snap = nullptr;
    unsigned long long *tile_status = nullptr, *tile_status_free = nullptr;
    u64 *slab64 = nullptr, *lg_hash = nullptr;
    u32 *slab32 = nullptr, *lg_abund = nullptr;
    u64 n_med = 0, n_long = 0, out_cap = 0; // (with a plan from max_seq_len: upper bounds, the true counts stay on the device)
    u64 win_bound = 0;                      // k-mer windows of the batch, or an upper bound (n_res) until the final read
    u32 real_max = 0, tile_R = sk_r_cand_host[0];
    const bool planned = max_seq_len > 0 && !ks_dbg(ctx, KS_DBG_NO_PLAN);
    u32 *pk_tiles = nullptr, *pk_cnt = nullptr, *d_ntiles = nullptr;
    u64 *tile_g0 = nullptr;
    u64 pk_n_tiles = 0; // packed plan: tiles of the batch (read back with the statistics, or an upper bound: pk_bound)
    // Packed plan without the round trip: with every sequence <= L = max_seq_len <= PK_MAX_LEN a tile is closed by a sequence
    // that does not fit, so it holds more than SK_MED_MAX - 15 - L residues (one short tile per chunk besides): when that bound
    // is within 2x of the typical count, the launch takes it as its grid and the tiles beyond the device-side count return at
    // once.  (Worth ~20 us per call: small batches; a 1M-protein launch would not notice either way.)
    u64 pk_bound = 0;
    u64 tiles_hint = 0; // tile status words that lie in the control block (zeroed with it)
    // compacting variant: bucket space = positions / c_div, span = residues per shared tile (see k_sketch_tiles<0, 1>)
    const bool compact = (variant & 1) && p->scaled >= 2 && !ks_dbg(ctx, KS_DBG_NO_COMPACT);
    const u32 c_div = compact ? (p->scaled < 64 ? p->scaled : 64u) : 1u;
    u32 span = SK_TILE;
    if (compact) {
        u64 sp = (u64)c_div * 3840; // span / c_div + SK_SEQ_CAP + 1 < SK_TILE buckets
        if (sp > 8ull * SK_TILE) sp = 8ull * SK_TILE;
        span = (u32)(sp / 512 * 512);
        if (const char *f = ks_dbg(ctx, KS_DBG_SPAN)) { // tuning aid
            const u32 v = (u32)atoi(f) / 512 * 512;
            if (v >= SK_TILE && v <= sp) span = v;
        }
    }
    // packed tiles (whole sequences packed greedily into each tile, see k_pack_walk) for the plain variant
    const bool packed = !compact && !ks_dbg(ctx, KS_DBG_NO_PACK);
    // (the walk of a chunk is a serial chain, ~0.5 us per tile: small batches take shorter chunks — more waves, shorter
    // chains — at the price of one partly filled tile per chunk)
    const u32 pk_chunk = n_seqs >= 262144 ? PK_CHUNK : (n_seqs >= 32768 ? 256u : 64u);
    const u32 pk_chunks = (n_seqs + pk_chunk - 1) / pk_chunk;
    sk_cands cand;
    for (int c = 0; c < SK_NR; c++) cand.r[c] = sk_r_cand_host[c] + (span - SK_TILE);
#define SK_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
#define SK_HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

    SK_CHECK(ks_alloc(ctx, &S->d_offsets, (size_t)n_seqs + 1));
    SK_CHECK(ks_alloc(ctx, &S->d_counts, (size_t)n_seqs + 1));
    if (n_seqs == 0) {
        SK_HIPCHECK(hipMemsetAsync(S->d_offsets, 0, sizeof(u64), ctx->stream));
        SK_CHECK(ks_alloc(ctx, &S->d_hashes, 1));
        SK_CHECK(ks_alloc(ctx, &S->d_abunds, 1));
        SK_CHECK(ks_stream_wait(ctx));
        *out = S;
        return KS_OK;
    }
    {
        // windows, longest sequence, medium / long counts per candidate stride
        // every small word the host zeroes before / reads after the launches lives in ONE control block — one memset, one
        // device -> host copy per read-back instead of one per word (each is a dispatch of its own on this runtime):
        // [0, 20) statistics, [20] ticket + status bits, [21] medium / long counts, [22] tiles of a packed plan, [23] kept hashes
        // ... and the posting cursors (2048 x u32) and the tiles' status words sit right behind them: ONE allocation that the
        // sketches object owns, ONE memset per call.  The tile count is known here when the plan needs no round trip (an
        // upper bound otherwise: a launch with more tiles takes a status array of its own).
        if (packed && planned && max_seq_len <= PK_MAX_LEN && !ks_dbg(ctx, KS_DBG_PLAN_SYNC)) {
            const u64 b = n_res / (SK_MED_MAX - 15 - max_seq_len + 1) + pk_chunks + 1;
            const u64 typical = n_res / 3800 + pk_chunks + 1;
            if (b <= 2 * typical + 256 && b <= n_seqs) pk_bound = b;
            else if (n_seqs <= 2 * typical + 256) pk_bound = n_seqs; // (a tile holds at least one sequence)
        }
        tiles_hint = packed ? pk_bound : n_res / cand.r[0] + 2;
        SK_CHECK(ks_alloc(ctx, &S->ctl_block, (size_t)SK_CTL_WORDS + 1024 + tiles_hint + 1));
        d_stats = S->ctl_block;
        SK_HIPCHECK(hipMemsetAsync(d_stats, 0, ((size_t)SK_CTL_WORDS + 1024 + tiles_hint + 1) * sizeof(u64), ctx->stream));
        ticket = (u32 *)(d_stats + 20); n_cls = (u32 *)(d_stats + 21); d_ntiles = (u32 *)(d_stats + 22);
        u32 g = (n_seqs + 1023) / 1024;
        if (g > 512) g = 512;
        ks_timer_begin(ctx, "seq_stats");
        hipLaunchKernelGGL(k_seq_stats, dim3(g), dim3(256), 0, ctx->stream, d_offs, n_seqs, p->ksize, span, cand, d_stats);
        ks_timer_end(ctx);
        if (packed) {
            // The tile count of a packed plan is only known on the device and sizes the launch: it is read back here, with
            // the statistics (one round trip whether or not the caller gave max_seq_len).
            SK_CHECK(ks_alloc(ctx, &pk_tiles, (size_t)n_seqs));
            SK_CHECK(ks_alloc(ctx, &pk_cnt, (size_t)pk_chunks));
            SK_CHECK(ks_alloc(ctx, &tile_first, (size_t)n_seqs + 1)); // (a tile holds at least one sequence)
            SK_CHECK(ks_alloc(ctx, &tile_g0, (size_t)n_seqs + 1));
            ks_timer_begin(ctx, "tile_plan");
            hipLaunchKernelGGL(k_pack_walk, dim3(pk_chunks), dim3(64), 0, ctx->stream, d_offs, n_seqs, pk_chunk, pk_tiles, pk_cnt);
            hipLaunchKernelGGL(k_pack_fill, dim3(pk_chunks), dim3(256), 0, ctx->stream, d_offs, n_seqs, pk_chunk, (const u32 *)pk_tiles,
                               (const u32 *)pk_cnt, pk_chunks, tile_first, tile_g0, d_ntiles);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }
        if (pk_bound) {
            real_max = max_seq_len;
            win_bound = n_res;
            tile_R = 0; n_med = 0; n_long = 0; // nothing is deferred: every sequence fits a tile
            pk_n_tiles = pk_bound;
        } else if (!planned || packed) {
            // no upper bound on the sequence length from the caller: the plan (tile stride, deferred sequences, slab size)
            // comes from the batch itself, at the price of one device -> host round trip before the tiles are launched
            {
                const ks_fetch_seg f = ks_fetch_words(d_stats, ctx->h_pin, SK_CTL_WORDS * 2);
                SK_CHECK(ks_stream_wait_fetch(ctx, &f, 1));
            }
            win_bound = ctx->h_pin[0];
            if (packed) pk_n_tiles = *(u32 *)(ctx->h_pin + 22);
            if (ctx->h_pin[1] > 0xfffffff0ULL) { st = ks_fail(ctx, KS_ERR_INVALID_ARG, "sequence longer than 2^32 residues"); goto done; }
            real_max = (u32)ctx->h_pin[1];
            // tile stride: fewest (tiles x sub-tiles + 1.75 x medium + 20 x long sequences), see sk_r_cand
            double best = 0;
            const double per_tile = (double)span / SK_TILE;
            for (int c = 0; c < SK_NR; c++) {
                if (c >= SK_NR - 2 && (u64)real_max + cand.r[c] + 15 > span) continue; // uncounted strides
                const double cost = (double)(n_res / cand.r[c] + 1) * per_tile + 1.75 * (double)ctx->h_pin[4 + c] +
                                    20.0 * (double)ctx->h_pin[4 + SK_NR + c];
                if (c == 0 || cost < best) { best = cost; tile_R = cand.r[c]; n_med = ctx->h_pin[4 + c]; n_long = ctx->h_pin[4 + SK_NR + c]; }
            }
            if (const char *force = ks_dbg(ctx, KS_DBG_TILE_R)) // tuning aid: one of sk_r_cand (counted strides only)
                for (int c = 0; c < SK_NR - 2; c++)
                    if (atoi(force) == (int)sk_r_cand_host[c]) { tile_R = cand.r[c]; n_med = ctx->h_pin[4 + c]; n_long = ctx->h_pin[4 + SK_NR + c]; }
            if (packed) { tile_R = 0; n_med = 0; n_long = ctx->h_pin[2]; } // nothing is deferred for its position
        } else {
            // max_seq_len given: everything the launches need follows from it, and what is only known on the device (how
            // many sequences are deferred) is read there by kernels with fixed grids.  The widest stride whose tiles hold
            // every sequence wherever it starts, else the stride the cost model picks for proteome-like batches.
            real_max = max_seq_len;
            win_bound = n_res;
            tile_R = cand.r[3];
            for (int c = SK_NR - 1; c >= SK_NR - 2; c--)
                if ((u64)real_max + cand.r[c] + 15 <= span) { tile_R = cand.r[c]; break; }
            const bool may_defer = (u64)real_max + tile_R + 15 > span;
            const u64 n_t = n_res / tile_R + 1; // at most one deferred sequence per tile (the last that starts in it)
            n_med = may_defer ? n_t : 0;
            n_long = (may_defer && real_max > SK_MED_MAX) ? n_t : 0;
        }
        S->n_windows = win_bound;
    }
    {
        // The tile kernel writes the final arrays in place (no compaction pass).  At scaled = 1 every window may be kept;
        // at scaled > 1 the arrays are sized by the expected 1 / scaled of the windows plus a quarter, and a batch that keeps
        // more (repeats whose hash falls under the threshold) is repeated with the full size.
        out_cap = S->n_windows;
        if (!(variant & 2) && p->scaled >= 2) {
            const u64 want = S->n_windows / p->scaled + S->n_windows / (4ull * p->scaled) + 65536;
            if (want < out_cap) out_cap = want;
        }
        if (const char *f = ks_dbg(ctx, KS_DBG_OUT_CAP)) // exercises the repeat on small inputs
            if (!(variant & 2) && strtoull(f, nullptr, 10) < out_cap) out_cap = strtoull(f, nullptr, 10);
        SK_CHECK(ks_alloc(ctx, &S->d_hashes, (size_t)out_cap));
        SK_CHECK(ks_alloc(ctx, &S->d_abunds, (size_t)out_cap));

        sk_args A;
        memset(&A, 0, sizeof A);
        A.res = d_res; A.offs = d_offs; A.n_seqs = n_seqs; A.n_res = n_res; A.k = p->ksize; A.seed = p->seed;
        A.max_hash = ks_max_hash(p->scaled);
        {
            u64 sf = (1ULL << 48) / ((A.max_hash >> 32) + 1ULL);
            A.sfix = sf > 0x7fffffffULL ? 0x7fffffffu : (u32)sf; // smaller only coarsens the buckets
        }
        A.lut = ctx->d_lut + 256 * p->moltype;
        A.upper_only = p->moltype == KS_PROTEIN ? 1u : 0u;
        if (const char *f = ks_dbg(ctx, KS_DBG_QCAP)) A.debug_qcap = (u32)atoi(f);
        A.counts = S->d_counts;
        A.drops_out = d_stats + 24;
        A.span = SK_TILE; A.c_div = 1; A.c_rcp = 0; A.out_cap = out_cap; A.max_len_tile = 0xffffffffu; A.R = 1;
        A.ticket = ticket; A.total_out = d_stats + 23;
        if (part_pbits > 0 && S->n_windows > 0 && S->n_windows < 0xffff0000ULL) {
            // first partition digit: the low 8 bits of the join's hash prefix (the whole prefix if it is <= 8 bits)
            const int dbits = part_pbits < 8 ? part_pbits : 8;
            S->part_pbits = part_pbits;
            S->part_K = ks_join_prefix_mul(part_pbits, A.max_hash);
            S->part_regions = 1u << dbits;
            // one sub-region per XCD when a second partition pass follows (its tiles are per-segment anyway); when the
            // regions ARE the join buckets (<= 8 prefix bits) they must stay contiguous
            // (... as many as leave a sub-region >= 64k postings: the bucket scatter works on tiles of 8,192 postings per sub-region, and
            // a 125k-query shard of the 1M workload cut 2,048 ways fills 2.2 of them — 0.25 ms against 0.17 with 512 sub-regions;
            // 10k queries: 44 -> 22 us with 256 — while the 1M batch wants all 2,048: fewer posting cursors cost its sketch kernel
            // 2.49 -> 2.61 ms)
            S->part_sub_shift = 0u;
            if (part_pbits > 8)
                while (S->part_sub_shift < 3u && (S->n_windows / p->scaled) >> (8u + S->part_sub_shift + 1u) >= 65536u) S->part_sub_shift++;
            if (const char *f = ks_dbg(ctx, KS_DBG_SUBSHIFT)) { const int v = atoi(f); if (part_pbits > 8 && v >= 0 && v <= 3) S->part_sub_shift = (u32)v; }
            const u32 n_segs = S->part_regions << S->part_sub_shift;
            const u64 per = S->n_windows / p->scaled / n_segs + 1; // FracMinHash keeps ~1/scaled of the windows
            u64 cap = per + per / 4 + 8192;                 // uniform hashes fill regions evenly; skew -> fallback
            cap = (cap + 8191) / 8192 * 8192;
            S->part_cap = cap;
            // (+ SK_TILE spare slots behind the last region: where a tile drops a digit that does not fit its region)
            SK_CHECK(ks_alloc(ctx, &S->part_keys, (size_t)(cap * n_segs) + SK_TILE));
            SK_CHECK(ks_alloc(ctx, &S->part_vals, (size_t)(cap * n_segs) + SK_TILE));
            S->part_len = (u32 *)(S->ctl_block + SK_CTL_WORDS); // (zeroed with the control block)
            A.part_keys = S->part_keys; A.part_vals = S->part_vals; A.part_cursor = S->part_len; A.part_cap = cap;
            A.part_K = S->part_K; A.part_mask = S->part_regions - 1; A.part_sub_shift = S->part_sub_shift;
            A.part_kshift = (S->part_K & (S->part_K - 1)) == 0 && S->part_K > 1 ? 32u - (u32)__builtin_ctz(S->part_K) : 0u;
            // 10-byte postings: when the caller's join reads them (big indexes: the fingerprint joins), the digit is a bit field
            // of the hash (scaled = 1) below a second partition level, and the sequence ids fit 24 bits
            if (part_fmt10 && A.part_kshift && part_pbits > 8 && n_seqs < (1u << 24) && !ks_dbg(ctx, KS_DBG_POSTINGS12))
                A.part_s = S->part_s = 32u + A.part_kshift;
        }

        // ---- medium / long sequences first: their unique counts feed the tile kernel's CSR prefix
        if (n_med + n_long > 0) {
            SK_CHECK(ks_alloc(ctx, &kept, (size_t)n_seqs));
            A.kept = kept;
            SK_CHECK(ks_alloc(ctx, &lg_hash, (size_t)n_res + 1));
            SK_CHECK(ks_alloc(ctx, &lg_abund, (size_t)n_res + 1));
            SK_CHECK(ks_alloc(ctx, &med_ids, (size_t)n_med + 1));
            SK_CHECK(ks_alloc(ctx, &long_ids, (size_t)n_long + 1));
            ks_timer_begin(ctx, "find_long");
            hipLaunchKernelGGL(k_find_long, dim3((n_seqs + 255) / 256), dim3(256), 0, ctx->stream, d_offs, n_seqs, tile_R, span, med_ids, long_ids, n_cls,
                               (u32)n_med, (u32)n_long);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }
        if (n_med > 0) {
            sk_args M = A; // (keeps the posting arguments: a medium tile emits its own postings)
            M.out_hash = lg_hash; M.out_abund = lg_abund; M.le_cap = SK_TILE; M.seq_list = med_ids; // local start <= 15, length <= SK_MED_MAX
            M.out_cap = ~0ULL;
            M.n_list = n_cls; M.n_list_cap = (u32)n_med;
            ks_timer_begin(ctx, "sketch_medium");
            hipLaunchKernelGGL((k_sketch_tiles<1, 0, 0>), dim3((u32)(n_med < 2048 ? n_med : 2048)), dim3(SK_THREADS), 0, ctx->stream, M);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }
        if (n_long > 0) {
            // slab: 3 u64 + 4 u32 arrays of (max_len + 1) per workgroup, capped at ~2 GiB total
            const u64 stride = (u64)real_max + 1;
            const u64 per_wg = stride * (3 * 8 + 4 * 4);
            u64 grid = (2ULL << 30) / per_wg;
            if (grid < 1) grid = 1;
            if (grid > n_long) grid = n_long;
            if (grid > 512) grid = 512;
            SK_CHECK(ks_alloc(ctx, &slab64, (size_t)(grid * stride * 3)));
            SK_CHECK(ks_alloc(ctx, &slab32, (size_t)(grid * stride * 4)));
            sk_long_args L;
            L.a = A; L.long_ids = long_ids; L.n_long = n_cls; L.long_cap = (u32)n_long; L.max_len = real_max;
            L.slab_keys = slab64; L.slab_tmp = slab64 + grid * stride; L.slab_sorted = slab64 + 2 * grid * stride;
            L.slab_cnt = slab32; L.slab_ord = slab32 + grid * stride; L.slab_flag = slab32 + 2 * grid * stride;
            L.slab_ab = slab32 + 3 * grid * stride;
            L.lg_hash = lg_hash; L.lg_abund = lg_abund;
            ks_timer_begin(ctx, "sketch_long");
            hipLaunchKernelGGL(k_sketch_long, dim3((u32)grid), dim3(SK_THREADS), 0, ctx->stream, L);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }

        // ---- shared tiles: hash + sort/unique + CSR placement in one kernel (decoupled look-back across tiles)
        const u64 n_tiles = packed ? pk_n_tiles : n_res / tile_R + 1;
        if (n_tiles > 0x7ffffff0ULL) { st = ks_fail(ctx, KS_ERR_INVALID_ARG, "batch too large"); goto done; }
        unsigned long long *tile_status_own = nullptr;
        if (n_tiles <= tiles_hint) tile_status = (unsigned long long *)(S->ctl_block + SK_CTL_WORDS + 1024); // (zeroed with the control block)
        else {
            SK_CHECK(ks_alloc(ctx, &tile_status_own, (size_t)n_tiles + 1));
            tile_status = tile_status_own;
            SK_HIPCHECK(hipMemsetAsync(tile_status, 0, (size_t)n_tiles * sizeof(unsigned long long), ctx->stream));
        }
        tile_status_free = tile_status_own;
        if (!packed) {
            SK_CHECK(ks_alloc(ctx, &tile_first, (size_t)n_tiles + 1));
            ks_timer_begin(ctx, "tile_plan");
            hipLaunchKernelGGL(k_tile_plan, dim3((u32)((n_tiles + 256) / 256)), dim3(256), 0, ctx->stream, d_offs, n_seqs, (u32)n_tiles, tile_R, tile_first);
            ks_timer_end(ctx);
        }
        A.seq_list = tile_first; A.le_cap = span - 16; A.R = tile_R; A.span = span; A.tile_g0 = tile_g0;
        if (packed) A.max_len_tile = PK_MAX_LEN;
        if (compact) { A.c_div = c_div; A.c_rcp = (u32)(((1ULL << 32) + c_div - 1) / c_div); }
        A.out_hash = S->d_hashes; A.out_abund = S->d_abunds; A.csr = S->d_offsets;
        A.tile_status = tile_status; A.n_tiles = (u32)n_tiles;
        A.n_tiles_dev = pk_bound ? d_ntiles : nullptr;
        // posting cursors as the medium tiles left them (a repeated launch starts from here)
        if (A.part_cursor && n_med > 0) { // (without medium tiles the cursors are still zero: a repeat just clears them)
            SK_CHECK(ks_alloc(ctx, &part_snap, 2048));
            SK_HIPCHECK(hipMemcpyAsync(part_snap, A.part_cursor, 2048 * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
        }
        for (int attempt = 0; attempt < 2; attempt++) {
            A.use_ticket = (ctx->sketch_use_ticket || attempt == 1) ? 1u : 0u;
            A.debug_skip_tile = 0xffffffffu;
            if (!A.use_ticket && ks_dbg(ctx, KS_DBG_LOOKBACK_SKIP)) A.debug_skip_tile = (u32)atoi(ks_dbg(ctx, KS_DBG_LOOKBACK_SKIP)); // (tests: a real expired spin)
            if (attempt == 1) { // the dispatch-order launch gave up a look-back: start the tiles over, ids by ticket
                ctx->sketch_use_ticket = true;
                ctx->sketch_ticket_fallbacks++;
                SK_HIPCHECK(hipMemsetAsync(tile_status, 0, (size_t)n_tiles * sizeof(unsigned long long), ctx->stream));
                // (keeps the status bits the medium tiles set before the loop — "postings not emitted" — and drops only
                // the look-back flag of the first attempt)
                const u32 keep_bits = ((u32 *)(ctx->h_pin + 20))[1] & 2u;
                SK_HIPCHECK(hipMemsetAsync(ticket, 0, 2 * sizeof(u32), ctx->stream));
                if (keep_bits) {
                    ((u32 *)(ctx->h_pin + 30))[0] = keep_bits;
                    SK_HIPCHECK(hipMemcpyAsync(ticket + 1, ctx->h_pin + 30, sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
                }
                if (part_snap) SK_HIPCHECK(hipMemcpyAsync(A.part_cursor, part_snap, 2048 * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
                else if (A.part_cursor) SK_HIPCHECK(hipMemsetAsync(A.part_cursor, 0, 2048 * sizeof(u32), ctx->stream));
            }
            ks_timer_begin(ctx, "sketch_tiles");
            // (the k-mer sizes of the reference's defaults and of BASELINE's configs run kernels with k folded in:
            // src/rust/main.rs:28 k = 10, src/python/kmerseek/index.py:79-81 k = 24; any other size: the generic kernel)
#define SK_LAUNCH_TILES(CMP_, KC_) hipLaunchKernelGGL((k_sketch_tiles<0, CMP_, KC_>), dim3((u32)n_tiles), dim3(SK_THREADS), 0, ctx->stream, A)
            if (compact) {
                if (p->ksize == 16) SK_LAUNCH_TILES(1, 16);
                else if (p->ksize == 24) SK_LAUNCH_TILES(1, 24);
                else SK_LAUNCH_TILES(1, 0);
            } else {
                if (p->ksize == 10) SK_LAUNCH_TILES(0, 10);
                else if (p->ksize == 7) SK_LAUNCH_TILES(0, 7);
                else SK_LAUNCH_TILES(0, 0);
            }
#undef SK_LAUNCH_TILES
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());

            // ---- runs of medium / long sequences into their CSR slots
            if (n_med > 0) {
                ks_timer_begin(ctx, "place_long");
                // medium runs: copy only (their tiles emitted their own postings)
                hipLaunchKernelGGL(k_place_long, dim3((u32)(n_med < 1024 ? n_med : 1024)), dim3(256), 0, ctx->stream, (const u32 *)med_ids,
                                   (const u32 *)n_cls, (u32)n_med, d_offs,
                                   (const u64 *)S->d_offsets, (const u32 *)S->d_counts, (const u64 *)lg_hash, (const u32 *)lg_abund, S->d_hashes, S->d_abunds, out_cap,
                                   (u64 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u64)0, 0u, 0u, 0u, ticket, 0u);
                ks_timer_end(ctx);
            }
            if (n_long > 0) {
                ks_timer_begin(ctx, "place_long");
                hipLaunchKernelGGL(k_place_long, dim3((u32)(n_long < 1024 ? n_long : 1024)), dim3(256), 0, ctx->stream, (const u32 *)long_ids,
                                   (const u32 *)(n_cls + 1), (u32)n_long, d_offs,
                                   (const u64 *)S->d_offsets, (const u32 *)S->d_counts, (const u64 *)lg_hash, (const u32 *)lg_abund, S->d_hashes, S->d_abunds, out_cap,
                                   A.part_keys, A.part_vals, A.part_cursor, A.part_cap, A.part_K, A.part_mask, A.part_sub_shift, ticket, A.part_s);
                ks_timer_end(ctx);
            }
            SK_HIPCHECK(hipGetLastError());
            // total + look-back error flag to the host
            if (allow_defer && attempt == 0 && (pk_bound || (planned && !packed)) && !ks_dbg(ctx, KS_DBG_FORCE_TICKET_RETRY)) {
                // the caller's next wait on this stream (the search's) stands in for this one and brings the control block along
                // (ks_sketch_pending_seg); everything else of this call that is freed below is reused in stream order
                S->pend_stats = d_stats;
                S->pending = 1; S->pend_out_cap = out_cap; S->pend_max_seq_len = max_seq_len; S->pend_planned = planned ? 1 : 0;
                S->n_hashes = S->n_slots = out_cap; // (upper bounds until ks_sketch_finish_pending)
                break;
            }
            {
                const ks_fetch_seg f = ks_fetch_words(d_stats, ctx->h_pin, SK_CTL_WORDS * 2);
                SK_CHECK(ks_stream_wait_fetch(ctx, &f, 1));
            }
            u32 &status_w = ((u32 *)(ctx->h_pin + 20))[1];
            if (attempt == 0 && !A.use_ticket && ks_dbg(ctx, KS_DBG_FORCE_TICKET_RETRY)) status_w |= 1u; // exercises the repeat
            if (!(status_w & 1u) || A.use_ticket) break;
        }
        if (S->pending) goto done;
        // the CSR's slots hold the kept hashes; the distinct ones (the sketches) are fewer by the repeats
        S->n_slots = ctx->h_pin[23];
#ifdef SK_LB_STATS
        fprintf(stderr, "[SK_LB_STATS] tiles %llu look-back rounds %llu\n", (unsigned long long)n_tiles, (unsigned long long)ctx->h_pin[25]);
#endif
        S->n_hashes = S->n_slots - (ctx->h_pin[24] < S->n_slots ? ctx->h_pin[24] : S->n_slots);
        S->gapped = S->n_hashes != S->n_slots;
        if (planned) {
            S->n_windows = ctx->h_pin[0];
            if (ctx->h_pin[1] > (u64)max_seq_len) { // the plan was made for shorter sequences: nothing of this launch can be trusted
                st = ks_fail(ctx, KS_ERR_INVALID_ARG, "max_seq_len = %u, but the batch holds a sequence of %llu residues", max_seq_len,
                             (unsigned long long)ctx->h_pin[1]);
                goto done;
            }
        }
        {
            const u32 status = ((u32 *)(ctx->h_pin + 20))[1];
            if (status & 1u) { st = ks_fail(ctx, KS_ERR_HIP, "sketch: tile look-back timed out"); goto done; }
            if (status & 4u) { *redo = 1; goto done; }          // a compacting tile overflowed: the plain variant always fits
            if (S->n_slots > out_cap) { *redo = 2; goto done; } // more kept hashes than the bounded outputs hold
            if (status & 2u) { // a region overflowed (skewed hashes) or a tile could not code its sequences: no postings,
                               // ks_search repartitions from the CSR instead
                ks_pool_free(ctx, S->part_keys); ks_pool_free(ctx, S->part_vals); // (part_len lies in the control block)
                S->part_keys = nullptr; S->part_vals = nullptr; S->part_len = nullptr; S->part_pbits = 0;
            }
        }
    }

done:
    ks_pool_free(ctx, kept); ks_pool_free(ctx, tile_first); ks_pool_free(ctx, tile_status_free); ks_pool_free(ctx, part_snap);
    ks_pool_free(ctx, med_ids); ks_pool_free(ctx, long_ids);
    ks_pool_free(ctx, slab64); ks_pool_free(ctx, slab32); ks_pool_free(ctx, lg_hash); ks_pool_free(ctx, lg_abund);
    ks_pool_free(ctx, pk_tiles); ks_pool_free(ctx, pk_cnt); ks_pool_free(ctx, tile_g0);
    if (st != KS_OK || *redo) {
        (void)hipStreamSynchronize(ctx->stream);
        ks_sketches_free(S);
        return st;
    }
    *out = S;
    return KS_OK;
#undef SK_CHECK
#undef SK_HIPCHECK
}

ks_fetch_seg ks_sketch_pending_seg(const ks_sketches *S) {
    return ks_fetch_words(S->pend_stats, S->ctx->h_pin + KS_PIN_SKETCH, SK_CTL_WORDS * 2);
}

int ks_sketch_finish_pending(ks_sketches *S, int *redo) {
    *redo = 0;
    if (!S || !S->pending) return KS_OK;
    ks_ctx *ctx = S->ctx;
    const u64 *stats = ctx->h_pin + KS_PIN_SKETCH;
    S->pending = 0;
    S->pend_stats = nullptr; // (lies in the control block, which the object keeps)
    S->n_slots = stats[23];
    S->n_hashes = S->n_slots - (stats[24] < S->n_slots ? stats[24] : S->n_slots);
    S->gapped = S->n_hashes != S->n_slots;
    if (S->pend_planned) {
        S->n_windows = stats[0];
        if (stats[1] > (u64)S->pend_max_seq_len)
            return ks_fail(ctx, KS_ERR_INVALID_ARG, "max_seq_len = %u, but the batch holds a sequence of %llu residues", S->pend_max_seq_len,
                           (unsigned long long)stats[1]);
    }
    const u32 status = ((const u32 *)(stats + 20))[1];
    if (status & 1u) *redo = 3;                       // a look-back gave up: the plain call repeats the launch with tickets
    else if (status & 4u) *redo = 1;                  // a compacting tile overflowed
    else if (S->n_slots > S->pend_out_cap) *redo = 2;  // more kept hashes than the bounded outputs hold
    else if (status & 2u) *redo = 4;                  // postings dropped (skewed hashes): whoever read them read garbage
    return KS_OK;
}

int ks_sketch_device_impl(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, u32 max_seq_len,
                          const ks_params *p, int part_pbits, int part_fmt10, int allow_defer, ks_sketches **out) {
    KS_TRY(ks_check_params(ctx, p));
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "out is NULL");
    if (((uintptr_t)d_res & 15) != 0) return ks_fail(ctx, KS_ERR_INVALID_ARG, "d_residues must be 16-byte aligned");
    KS_HIP(ctx, hipSetDevice(ctx->device));
    // economies first (compacting tiles, outputs sized by the expected kept count); a batch that defeats one — repeats
    // whose hash falls under the threshold — is repeated without it (at most twice; the last variant always fits)
    int variant = 1;
    for (int round = 0; round < 3; round++) {
        int redo = 0;
        const int st = sketch_attempt(ctx, d_res, d_offs, n_seqs, n_res, max_seq_len, p, part_pbits, part_fmt10, variant,
                                      (allow_defer && round == 0) ? 1 : 0, &redo, out);
        if (st != KS_OK || !redo) return st;
        if (redo == 1) { variant &= ~1; ctx->sketch_compact_fallbacks++; }
        else { variant |= 2; ctx->sketch_cap_fallbacks++; }
    }
    return ks_fail(ctx, KS_ERR_HIP, "sketch: no variant fitted the batch");
}
