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
//     starts, elements are ranked inside their (tiny) bucket by comparison.  Equal hashes meet in one
//     bucket: the first arrival is the representative and carries the abundance.
//   * Representatives are compacted through LDS and leave as one contiguous run per tile; a per-sequence
//     count + scan + gather builds the final CSR.
//   * Sequences longer than LS_MAX but shorter than a tile ("medium") get a tile of their own in a second
//     launch of the same kernel; longer ones take the same algorithm with its arrays in a global scratch
//     slab (k_sketch_long), one workgroup per sequence.
#include "ks_device.h"

#define SK_THREADS 512
#define SK_E 8
#define SK_TILE (SK_THREADS * SK_E) // 4096 LDS positions
#define SK_LS_MAX 1536              // longest sequence a shared (multi-sequence) tile takes
#define SK_MED_MAX (SK_TILE - 16)   // longest sequence that still fits one tile on its own ("medium")
#define SK_R (SK_TILE - SK_LS_MAX - 16)
#define SK_PAD 160                  // >= KS_MAX_KSIZE + 24: slack behind the last residue for word reads
#define SK_NFLAG (SK_TILE / 32)

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
    u32 len_cap;   // sequences longer than this are not this launch's business
    const u32 *seq_list; // MODE 0: tile_first[n_tiles + 1] (tile -> first sequence); MODE 1: one medium sequence per workgroup
    u64 start_flag;      // OR-ed into sp_start (marks runs that live in the lg_* buffers)
    u64 *sp_hash;  // [n_res]   tile-packed unique hashes
    u32 *sp_abund; // [n_res]
    u32 *counts;   // [n_seqs]  unique hashes per sequence
    u64 *sp_start; // [n_seqs]  where the sequence's run starts in sp_hash / sp_abund
};

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

// hash of the window that starts at LDS byte `pos8 + I` where pos8 is 8-byte aligned
template <int I>
KS_DEV u64 sk_hash_window(const u64 *w /* LDS words starting at pos8 */, u32 k, u64 seed) {
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

#define SK_SEQ_CAP 510 // sequence boundaries of a tile staged in LDS (tiles with more fall back to global reads)

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
    q.nw = (len >= A.k && len <= A.len_cap) ? (len - A.k + 1) : 0;
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
        const u32 o = atomicAdd(&cnt[b], 1u);
        bo = (b << 16) | o; // b < 4096, o < 4096
    }
    return bo;
}

// tile_first[t] = first sequence whose start offset is >= t * SK_R (n_tiles + 1 entries): one parallel
// binary search per tile here instead of a serial, latency-bound one at the head of every workgroup
__global__ __launch_bounds__(256) void k_tile_plan(const u64 *offs, u32 n_seqs, u32 n_tiles, u32 *tile_first) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    tile_first[t] = t == n_tiles ? n_seqs : sk_lower_bound(offs, 0, n_seqs, (u64)t * SK_R);
}

// MODE 0: shared tiles cut by residue range; MODE 1: one listed medium sequence per workgroup
template <int MODE>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_tiles(sk_args A) {
    __shared__ __attribute__((aligned(16))) u64 res_w[(SK_TILE + SK_PAD) / 8];
    __shared__ __attribute__((aligned(16))) u32 cnt[SK_TILE + 8];
    __shared__ __attribute__((aligned(16))) u64 tmp[SK_TILE];
    __shared__ u32 flagbits[SK_NFLAG];
    __shared__ u32 flagpre[SK_NFLAG + 1];
    __shared__ u32 scan_smem[SK_THREADS / 64 + 1];
    __shared__ u32 loff[SK_SEQ_CAP + 2];
    __shared__ u8 lut_s[256];

    const u32 tid = threadIdx.x;
    u8 *res_b = (u8 *)res_w;
#ifdef SK_STAMP
    unsigned long long sk_t_prev = clock64();
#endif

    // ---- phase 0: tile -> sequence range (planned ahead), zero LDS state, stage LUT + sequence boundaries
    u32 s_first, s_end;
    if (MODE == 1) { s_first = A.seq_list[blockIdx.x]; s_end = s_first + 1; }
    else { s_first = A.seq_list[blockIdx.x]; s_end = A.seq_list[blockIdx.x + 1]; } // seq_list = tile_first here
    if (s_first >= s_end) return;
    const u64 r0 = A.offs[s_first];
    const u64 g0 = r0 & ~15ULL; // A.res is 16-byte aligned (checked on the host)
    const u32 ns = s_end - s_first;
    sk_bounds B;
    B.loff = loff; B.goff = A.offs; B.g0 = g0; B.s_first = s_first; B.in_lds = ns <= SK_SEQ_CAP;
    if (B.in_lds)
        for (u32 i = tid; i <= ns; i += SK_THREADS) {
            u64 v = A.offs[s_first + i] - g0;
            loff[i] = v > 0x7fffffffULL ? 0x7fffffffu : (u32)v;
        }
    if (tid < 256) lut_s[tid] = A.lut[tid];
    for (u32 i = tid; i < SK_TILE + 8; i += SK_THREADS) cnt[i] = 0;
    if (tid < SK_NFLAG) flagbits[tid] = 0;
    u64 span_end = A.offs[s_end];
    if (span_end > g0 + SK_TILE) span_end = g0 + SK_TILE;
    __syncthreads();

    SK_STAMP_AT(0);
    // ---- phase 1: residues -> LDS through the encode LUT, 16 B per lane
    for (u32 c = tid; c < (SK_TILE + SK_PAD) / 16; c += SK_THREADS) {
        u64 g = g0 + (u64)c * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < span_end) {
            if (g + 16 <= A.n_res) {
                v = *(const uint4 *)(A.res + g);
            } else {
                u32 t[4] = {0, 0, 0, 0};
                for (u32 b = 0; b < 16 && g + b < A.n_res; b++) t[b >> 2] |= (u32)A.res[g + b] << (8 * (b & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            u32 in[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
            for (int d = 0; d < 4; d++)
                o[d] = (u32)lut_s[in[d] & 255u] | ((u32)lut_s[(in[d] >> 8) & 255u] << 8) |
                       ((u32)lut_s[(in[d] >> 16) & 255u] << 16) | ((u32)lut_s[in[d] >> 24] << 24);
            v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *(uint4 *)(res_b + (size_t)c * 16) = v;
    }
    __syncthreads();

    SK_STAMP_AT(1);
    // ---- phase 2: hash 8 consecutive windows per thread, bucket + arrival slot via LDS atomics
    const u32 q0 = tid * SK_E;
    sk_seq q;
    {
        // first sequence of the tile whose end lies beyond q0
        u32 lo = s_first, hi = s_end;
        while (lo < hi) {
            u32 mid = lo + ((hi - lo) >> 1);
            if (B.at(mid + 1) > q0) hi = mid; else lo = mid + 1;
        }
        q.s = lo;
        sk_load_seq(q, A, B, s_end);
    }
    const u64 *wl = res_w + tid; // word at byte q0
    u64 h[SK_E];
    u32 bo[SK_E];
    h[0] = sk_hash_window<0>(wl, A.k, A.seed);
    h[1] = sk_hash_window<1>(wl, A.k, A.seed);
    h[2] = sk_hash_window<2>(wl, A.k, A.seed);
    h[3] = sk_hash_window<3>(wl, A.k, A.seed);
    h[4] = sk_hash_window<4>(wl, A.k, A.seed);
    h[5] = sk_hash_window<5>(wl, A.k, A.seed);
    h[6] = sk_hash_window<6>(wl, A.k, A.seed);
    h[7] = sk_hash_window<7>(wl, A.k, A.seed);
#pragma unroll
    for (int i = 0; i < SK_E; i++) bo[i] = sk_place_window(A, q0 + i, h[i], q, B, s_end, cnt);
    __syncthreads();

    SK_STAMP_AT(2);
    // ---- phase 3: bucket counts -> bucket starts (exclusive scan over the tile)
    {
        u32 c[SK_E], s = 0;
#pragma unroll
        for (int i = 0; i < SK_E; i++) { c[i] = cnt[q0 + i]; s += c[i]; }
        u32 total;
        u32 ex = ks_block_excl_scan(s, scan_smem, &total);
#pragma unroll
        for (int i = 0; i < SK_E; i++) { cnt[q0 + i] = ex; ex += c[i]; }
        if (tid == SK_THREADS - 1) {
            for (int i = 0; i < 8; i++) cnt[SK_TILE + i] = total;
        }
    }
    __syncthreads();

    SK_STAMP_AT(3);
    // ---- phase 4: scatter kept hashes into bucket order
#pragma unroll
    for (int i = 0; i < SK_E; i++)
        if (bo[i] != 0xffffffffu) tmp[cnt[bo[i] >> 16] + (bo[i] & 0xffffu)] = h[i];
    __syncthreads();

    SK_STAMP_AT(4);
    // ---- phase 5: rank inside the bucket; first arrival of each distinct hash is its representative
    u32 pr[SK_E]; // (sorted position << 1) | is_representative, or ~0
    u32 ab[SK_E];
    {
        u32 sb[SK_E], c[SK_E], less[SK_E], eqb[SK_E];
        u32 maxc = 0;
#pragma unroll
        for (int i = 0; i < SK_E; i++) {
            sb[i] = 0; c[i] = 0; less[i] = 0; eqb[i] = 0; ab[i] = 0;
            if (bo[i] != 0xffffffffu) {
                const u32 b = bo[i] >> 16;
                sb[i] = cnt[b];
                c[i] = cnt[b + 1] - sb[i];
            }
            maxc = c[i] > maxc ? c[i] : maxc;
        }
        for (u32 j = 0; j < maxc; j++) {
#pragma unroll
            for (int i = 0; i < SK_E; i++) {
                if (j < c[i]) {
                    const u64 x = tmp[sb[i] + j];
                    less[i] += x < h[i];
                    ab[i] += x == h[i];
                    eqb[i] += (x == h[i]) & (j < (bo[i] & 0xffffu));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SK_E; i++) {
            pr[i] = 0xffffffffu;
            if (bo[i] != 0xffffffffu) {
                const u32 p = sb[i] + less[i] + eqb[i];
                pr[i] = (p << 1) | (eqb[i] == 0);
                if (eqb[i] == 0) atomicOr(&flagbits[p >> 5], 1u << (p & 31));
            }
        }
    }
    __syncthreads();

    SK_STAMP_AT(5);
    // ---- phase 6: prefix over representative flags -> distinct rank
    {
        u32 v = tid < SK_NFLAG ? (u32)__popc(flagbits[tid]) : 0;
        u32 total;
        u32 ex = ks_block_excl_scan(v, scan_smem, &total);
        if (tid < SK_NFLAG) flagpre[tid] = ex;
        if (tid == 0) flagpre[SK_NFLAG] = total;
    }
    __syncthreads();
    const u32 n_distinct = flagpre[SK_NFLAG];

    // per-sequence unique counts and run starts (tile-packed layout starting at offs[s_first])
    for (u32 s = s_first + tid; s < s_end; s += SK_THREADS) {
        const u32 ls = B.at(s), le = B.at(s + 1);
        if (le - ls > A.len_cap) continue; // a later launch owns it
        u32 x0 = cnt[ls], x1 = cnt[le];
        u32 d0 = x0 >= SK_TILE ? n_distinct : flagpre[x0 >> 5] + (u32)__popc(flagbits[x0 >> 5] & ((1u << (x0 & 31)) - 1u));
        u32 d1 = x1 >= SK_TILE ? n_distinct : flagpre[x1 >> 5] + (u32)__popc(flagbits[x1 >> 5] & ((1u << (x1 & 31)) - 1u));
        A.counts[s] = d1 - d0;
        A.sp_start[s] = (r0 + d0) | A.start_flag;
    }
    __syncthreads();

    SK_STAMP_AT(6);
    // ---- phase 7: representatives -> LDS staging in distinct-rank order (tmp / cnt are free now)
    u32 *abund_s = cnt;
#pragma unroll
    for (int i = 0; i < SK_E; i++) {
        if (pr[i] != 0xffffffffu && (pr[i] & 1u)) {
            const u32 p = pr[i] >> 1;
            const u32 d = flagpre[p >> 5] + (u32)__popc(flagbits[p >> 5] & ((1u << (p & 31)) - 1u));
            tmp[d] = h[i];
            abund_s[d] = ab[i];
        }
    }
    __syncthreads();

    SK_STAMP_AT(7);
    // ---- phase 8: one contiguous, coalesced run per tile
    for (u32 d = tid; d < n_distinct; d += SK_THREADS) {
        A.sp_hash[r0 + d] = tmp[d];
        A.sp_abund[r0 + d] = abund_s[d];
    }
    SK_STAMP_AT(8);
}

// ---------------------------------------------------------------------------------------------
// long sequences: same algorithm, arrays in a global scratch slab, one workgroup per sequence
// ---------------------------------------------------------------------------------------------
struct sk_long_args {
    sk_args a;
    const u32 *long_ids;
    const u32 *n_long;
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
#define SK_LONG_FLAG (1ULL << 63)

// n_cls[0] = medium sequences (own tile), n_cls[1] = long sequences (global-slab path)
__global__ __launch_bounds__(256) void k_find_long(const u64 *offs, u32 n_seqs, u32 *med_ids, u32 *long_ids, u32 *n_cls) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    u64 len = offs[s + 1] - offs[s];
    if (len > SK_MED_MAX) long_ids[atomicAdd(&n_cls[1], 1u)] = s;
    else if (len > SK_LS_MAX) med_ids[atomicAdd(&n_cls[0], 1u)] = s;
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
    const u32 n_long = L.n_long[1];
    const u64 slab = (u64)blockIdx.x * ((u64)L.max_len + 1);
    u64 *keys = L.slab_keys + slab, *tmp = L.slab_tmp + slab, *sorted = L.slab_sorted + slab;
    u32 *cnt = L.slab_cnt + slab, *ord = L.slab_ord + slab, *flag = L.slab_flag + slab, *abd = L.slab_ab + slab;

    for (u32 li = blockIdx.x; li < n_long; li += gridDim.x) {
        const u32 s = L.long_ids[li];
        const u64 b = A.offs[s], e = A.offs[s + 1];
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
        if (tid == 0) { A.counts[s] = n_distinct; A.sp_start[s] = b | SK_LONG_FLAG; }
        SK_LONG_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// CSR assembly
// ---------------------------------------------------------------------------------------------
// one wave per sequence: gather its run from the tile-packed buffers into the final CSR
__global__ __launch_bounds__(256) void k_sketch_gather(const u64 *sp_hash, const u32 *sp_abund, const u64 *lg_hash,
                                                       const u32 *lg_abund, const u64 *sp_start, const u64 *csr,
                                                       u32 n_seqs, u64 *hashes, u32 *abunds) {
    const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seqs) return;
    const u32 lane = threadIdx.x & 63;
    const u64 dst = csr[s], n = csr[s + 1] - dst;
    if (n == 0) return;
    u64 src = sp_start[s];
    const u64 *sh = sp_hash;
    const u32 *sa = sp_abund;
    if (src & SK_LONG_FLAG) { src &= ~SK_LONG_FLAG; sh = lg_hash; sa = lg_abund; }
    for (u64 i = lane; i < n; i += 64) {
        hashes[dst + i] = sh[src + i];
        abunds[dst + i] = sa[src + i];
    }
}

// out[0] = k-mer windows, out[1] = longest sequence, out[2] = medium sequences, out[3] = long sequences
__global__ __launch_bounds__(256) void k_seq_stats(const u64 *offs, u32 n_seqs, u32 k, u64 *out) {
    u64 w = 0, mx = 0, nm = 0, nl = 0;
    for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seqs; s += gridDim.x * blockDim.x) {
        u64 len = offs[s + 1] - offs[s];
        w += len >= k ? len - k + 1 : 0;
        mx = len > mx ? len : mx;
        nl += len > SK_MED_MAX;
        nm += (len > SK_LS_MAX) & (len <= SK_MED_MAX);
    }
    for (int d = 32; d > 0; d >>= 1) {
        w += __shfl_down(w, d, 64);
        nm += __shfl_down(nm, d, 64);
        nl += __shfl_down(nl, d, 64);
        u64 o = __shfl_down(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long *)&out[0], (unsigned long long)w);
        atomicMax((unsigned long long *)&out[1], (unsigned long long)mx);
        if (nm) atomicAdd((unsigned long long *)&out[2], (unsigned long long)nm);
        if (nl) atomicAdd((unsigned long long *)&out[3], (unsigned long long)nl);
    }
}

int ks_sketch_device_impl(ks_ctx *ctx, const u8 *d_res, const u64 *d_offs, u32 n_seqs, u64 n_res, u32 max_seq_len,
                          const ks_params *p, ks_sketches **out) {
    KS_TRY(ks_check_params(ctx, p));
    if (!out) return ks_fail(ctx, KS_ERR_INVALID_ARG, "out is NULL");
    if (((uintptr_t)d_res & 15) != 0) return ks_fail(ctx, KS_ERR_INVALID_ARG, "d_residues must be 16-byte aligned");
    (void)max_seq_len; // hint only: the real maximum is measured on the device below
    KS_HIP(ctx, hipSetDevice(ctx->device));

    ks_sketches *S = new ks_sketches();
    memset(S, 0, sizeof *S);
    S->ctx = ctx;
    S->params = *p;
    S->n_seqs = n_seqs;
    *out = nullptr;

    int st = KS_OK;
    u64 *sp_hash = nullptr, *sp_start = nullptr, *d_stats = nullptr;
    u32 *sp_abund = nullptr, *counts = nullptr;
    u32 *med_ids = nullptr, *long_ids = nullptr, *n_cls = nullptr, *tile_first = nullptr;
    u64 *slab64 = nullptr, *lg_hash = nullptr;
    u32 *slab32 = nullptr, *lg_abund = nullptr;
    u64 n_med = 0, n_long = 0;
    u32 real_max = 0;
#define SK_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
#define SK_HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

    SK_CHECK(ks_alloc(ctx, &S->d_offsets, (size_t)n_seqs + 1));
    if (n_seqs == 0) {
        SK_HIPCHECK(hipMemsetAsync(S->d_offsets, 0, sizeof(u64), ctx->stream));
        SK_CHECK(ks_alloc(ctx, &S->d_hashes, 1));
        SK_CHECK(ks_alloc(ctx, &S->d_abunds, 1));
        SK_HIPCHECK(hipStreamSynchronize(ctx->stream));
        *out = S;
        return KS_OK;
    }
    {
        // windows, longest sequence, medium / long counts: one small D2H
        SK_CHECK(ks_alloc(ctx, &d_stats, 4));
        SK_HIPCHECK(hipMemsetAsync(d_stats, 0, 4 * sizeof(u64), ctx->stream));
        u32 g = (n_seqs + 1023) / 1024;
        if (g > 2048) g = 2048;
        ks_timer_begin(ctx, "seq_stats");
        hipLaunchKernelGGL(k_seq_stats, dim3(g), dim3(256), 0, ctx->stream, d_offs, n_seqs, p->ksize, d_stats);
        ks_timer_end(ctx);
        SK_HIPCHECK(hipMemcpyAsync(ctx->h_pin, d_stats, 4 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        SK_HIPCHECK(hipStreamSynchronize(ctx->stream));
        S->n_windows = ctx->h_pin[0];
        if (ctx->h_pin[1] > 0xfffffff0ULL) { st = ks_fail(ctx, KS_ERR_INVALID_ARG, "sequence longer than 2^32 residues"); goto done; }
        real_max = (u32)ctx->h_pin[1];
        n_med = ctx->h_pin[2];
        n_long = ctx->h_pin[3];
    }

    SK_CHECK(ks_alloc(ctx, &sp_hash, (size_t)n_res + 1));
    SK_CHECK(ks_alloc(ctx, &sp_abund, (size_t)n_res + 1));
    SK_CHECK(ks_alloc(ctx, &counts, (size_t)n_seqs));
    SK_CHECK(ks_alloc(ctx, &sp_start, (size_t)n_seqs));
    SK_HIPCHECK(hipMemsetAsync(counts, 0, (size_t)n_seqs * sizeof(u32), ctx->stream));
    {
        sk_args A;
        A.res = d_res; A.offs = d_offs; A.n_seqs = n_seqs; A.n_res = n_res; A.k = p->ksize; A.seed = p->seed;
        A.max_hash = ks_max_hash(p->scaled);
        {
            u64 sf = (1ULL << 48) / ((A.max_hash >> 32) + 1ULL);
            A.sfix = sf > 0x7fffffffULL ? 0x7fffffffu : (u32)sf; // smaller only coarsens the buckets
        }
        A.lut = ctx->d_lut + 256 * p->moltype;
        A.sp_hash = sp_hash; A.sp_abund = sp_abund; A.counts = counts; A.sp_start = sp_start;
        A.len_cap = SK_LS_MAX; A.start_flag = 0;
        const u64 n_tiles = n_res / SK_R + 1;
        if (n_tiles > 0x7ffffff0ULL) { st = ks_fail(ctx, KS_ERR_INVALID_ARG, "batch too large"); goto done; }
        SK_CHECK(ks_alloc(ctx, &tile_first, (size_t)n_tiles + 1));
        ks_timer_begin(ctx, "tile_plan");
        hipLaunchKernelGGL(k_tile_plan, dim3((u32)((n_tiles + 256) / 256)), dim3(256), 0, ctx->stream, d_offs, n_seqs, (u32)n_tiles, tile_first);
        ks_timer_end(ctx);
        A.seq_list = tile_first;
        ks_timer_begin(ctx, "sketch_tiles");
        hipLaunchKernelGGL(k_sketch_tiles<0>, dim3((u32)n_tiles), dim3(SK_THREADS), 0, ctx->stream, A);
        ks_timer_end(ctx);
        SK_HIPCHECK(hipGetLastError());

        if (n_med + n_long > 0) {
            // runs of medium / long sequences live in their own buffers: a shared tile's packed run may
            // extend over the residue span of a long neighbour
            SK_CHECK(ks_alloc(ctx, &lg_hash, (size_t)n_res + 1));
            SK_CHECK(ks_alloc(ctx, &lg_abund, (size_t)n_res + 1));
            SK_CHECK(ks_alloc(ctx, &med_ids, (size_t)n_med + 1));
            SK_CHECK(ks_alloc(ctx, &long_ids, (size_t)n_long + 1));
            SK_CHECK(ks_alloc(ctx, &n_cls, 2));
            SK_HIPCHECK(hipMemsetAsync(n_cls, 0, 2 * sizeof(u32), ctx->stream));
            ks_timer_begin(ctx, "find_long");
            hipLaunchKernelGGL(k_find_long, dim3((n_seqs + 255) / 256), dim3(256), 0, ctx->stream, d_offs, n_seqs, med_ids, long_ids, n_cls);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }
        if (n_med > 0) {
            sk_args M = A;
            M.sp_hash = lg_hash; M.sp_abund = lg_abund; M.len_cap = SK_MED_MAX; M.seq_list = med_ids;
            M.start_flag = SK_LONG_FLAG;
            ks_timer_begin(ctx, "sketch_medium");
            hipLaunchKernelGGL(k_sketch_tiles<1>, dim3((u32)n_med), dim3(SK_THREADS), 0, ctx->stream, M);
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
            L.a = A; L.long_ids = long_ids; L.n_long = n_cls; L.max_len = real_max;
            L.slab_keys = slab64; L.slab_tmp = slab64 + grid * stride; L.slab_sorted = slab64 + 2 * grid * stride;
            L.slab_cnt = slab32; L.slab_ord = slab32 + grid * stride; L.slab_flag = slab32 + 2 * grid * stride;
            L.slab_ab = slab32 + 3 * grid * stride;
            L.lg_hash = lg_hash; L.lg_abund = lg_abund;
            ks_timer_begin(ctx, "sketch_long");
            hipLaunchKernelGGL(k_sketch_long, dim3((u32)grid), dim3(SK_THREADS), 0, ctx->stream, L);
            ks_timer_end(ctx);
            SK_HIPCHECK(hipGetLastError());
        }
    }
    // counts -> CSR offsets; total to the host to size the final arrays
    SK_CHECK(ks_scan_u32_to_u64(ctx, counts, S->d_offsets, n_seqs));
    SK_HIPCHECK(hipMemcpyAsync(ctx->h_pin, S->d_offsets + n_seqs, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    SK_HIPCHECK(hipStreamSynchronize(ctx->stream));
    S->n_hashes = ctx->h_pin[0];
    SK_CHECK(ks_alloc(ctx, &S->d_hashes, (size_t)S->n_hashes));
    SK_CHECK(ks_alloc(ctx, &S->d_abunds, (size_t)S->n_hashes));
    ks_timer_begin(ctx, "sketch_gather");
    hipLaunchKernelGGL(k_sketch_gather, dim3((n_seqs + 3) / 4), dim3(256), 0, ctx->stream, (const u64 *)sp_hash,
                       (const u32 *)sp_abund, (const u64 *)lg_hash, (const u32 *)lg_abund, (const u64 *)sp_start,
                       (const u64 *)S->d_offsets, n_seqs, S->d_hashes, S->d_abunds);
    ks_timer_end(ctx);
    SK_HIPCHECK(hipGetLastError());
    SK_HIPCHECK(hipStreamSynchronize(ctx->stream));

done:
    ks_pool_free(ctx, sp_hash); ks_pool_free(ctx, sp_abund); ks_pool_free(ctx, counts); ks_pool_free(ctx, sp_start);
    ks_pool_free(ctx, tile_first);
    ks_pool_free(ctx, d_stats); ks_pool_free(ctx, med_ids); ks_pool_free(ctx, long_ids); ks_pool_free(ctx, n_cls);
    ks_pool_free(ctx, slab64); ks_pool_free(ctx, slab32); ks_pool_free(ctx, lg_hash); ks_pool_free(ctx, lg_abund);
    if (st != KS_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        ks_sketches_free(S);
        return st;
    }
    *out = S;
    return KS_OK;
#undef SK_CHECK
#undef SK_HIPCHECK
}
