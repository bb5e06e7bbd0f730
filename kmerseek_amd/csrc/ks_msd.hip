// ks_msd.hip — sort of the match list of a search: MSD partition + in-LDS bucket sort.
//
// ks_search (src/python/kmerseek/search.py:125-141 -> branchwater manysearch) turns the matched (query posting, index
// posting) pairs into COO rows sorted by (qid, tid): the match list — one packed 8-byte record ((qid << tbits | tid) <<
// abits) | abundance per pair — has to be ordered on its id bits before the run-length reduce.  An LSD radix sort moves
// every record ceil(bits / 8) times (5 passes for 1M x 1M: 40 id bits).  Here the list is moved three times whatever the
// key width:
//   level 1  exact partition on the top 8 id bits            (histogram + one scatter pass)
//   level 2  exact partition on the next 8 bits, per region  (histogram over 65,536 (d1, d2) bins + one scatter pass)
//   level 3  every level-2 bucket (n / 65,536 records on average) is sorted on its remaining bits inside LDS, in place
// The partition passes need no order inside a bucket, so a record's rank inside its (tile, bin) group is the return
// value of one LDS atomic and a tile reserves its slice of a bin with ONE global atomic — the bins are sized exactly by
// the histogram, so skew (a query that matches every target) cannot overflow anything.  Skew only decides WHERE a
// bucket is sorted: one that does not fit LDS is sorted by its workgroup with serial stable LSD passes over global
// memory (ping-pong with the scratch list) — slower per record, still exact.
#include <type_traits>

#include "ks_device.h"

#define MS_THREADS 512
#define MS_IPT 16
#define MS_TILE (MS_THREADS * MS_IPT) // 8192 records per tile of a partition pass
#define MS_WAVES (MS_THREADS / 64)
static_assert(MS_TILE == 8192, "ks_search.hip counts the level-1 tiles of a segmented list in units of 8192 records");
#define MS_SUB 8u // level-1 sub-histograms

// Digit of a record at a level: NB = 256 -> 8 bits at `shift`; NB = 512 -> the 16 bits at `shift` RELATIVE to the tile's
// base (the level-1 digit of the tile's first record << 8): a tile of the level-1 output lies inside one level-1 region or
// straddles two, so its 16-bit digits fall in a window of 512 — the rare record beyond it (tiny regions) is an "outlier"
// and goes through a global atomic of its own.
template <int NB>
KS_DEV u32 ms_digit(u64 key, int shift, u32 base, u32 mask) {
    if constexpr (NB == 256) return (u32)(key >> shift) & 255u;
    else return ((u32)(key >> shift) & mask) - base; // >= NB: outlier
}
// base of a level-2 tile: the (level-1 digit << bits2) of its first record
template <int NB>
KS_DEV u32 ms_base(const u64 *keys, u64 tile_base, int shift, u32 mask, int bits2) {
    if constexpr (NB == 256) return 0u;
    else return (((u32)(keys[tile_base] >> shift) & mask) >> bits2) << bits2;
}

// where tile `t` of a level-1 pass reads: records [base, base + nvalid) of the list — dense, or in the join's segments
template <int SEG>
KS_DEV void ms_tile_range(u32 t, u64 n, const ks_msd_segs &S, u64 *base, u32 *nvalid) {
    if constexpr (SEG == 0) {
        *base = (u64)t * MS_TILE;
        *nvalid = (n - *base) < MS_TILE ? (u32)(n - *base) : MS_TILE;
    } else {
        u32 lo = 0, hi = S.n; // tile_start[lo] <= t < tile_start[hi] (uniform: scalar loads of the kernel arguments)
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (S.tile_start[mid] <= t) lo = mid; else hi = mid;
        }
        const u64 off = (u64)(t - S.tile_start[lo]) * MS_TILE;
        *base = (u64)lo * S.seg_cap + off;
        *nvalid = ((u64)S.count[lo] - off) < MS_TILE ? (u32)((u64)S.count[lo] - off) : MS_TILE;
    }
}

// hist[bin] += records of this tile per bin
// (`sub_stride`: level 1 keeps MS_SUB histograms, tile t counting in number t mod MS_SUB — atomics on ONE address are served
// one at a time, ~12 ns each, and a level-1 bin would take one from every tile; 0 = one histogram)
template <int NB, int SEG = 0>
__global__ __launch_bounds__(MS_THREADS) void k_msd_hist(const u64 *keys, u64 n, int shift, u32 *hist, u32 mask, int bits2, u32 sub_stride,
                                                         ks_msd_segs segs) {
    __shared__ u32 bins[NB];
    const u32 tid = threadIdx.x;
    hist += (blockIdx.x & (MS_SUB - 1u)) * sub_stride;
    for (u32 i = tid; i < NB; i += MS_THREADS) bins[i] = 0;
    u64 tile_base;
    u32 nvalid;
    ms_tile_range<SEG>(blockIdx.x, n, segs, &tile_base, &nvalid);
    const u32 base = ms_base<NB>(keys, tile_base, shift, mask, bits2);
    __syncthreads();
    // (all of a thread's records are requested before the first is counted: written as a load inside the guarded loop the
    // compiler waited for every one of them in turn — 16 dependent memory latencies per tile, 2.2 TB/s for a read-only pass)
    u64 key[MS_IPT];
#pragma unroll
    for (int r = 0; r < MS_IPT; r++) {
        const u32 li = (u32)r * MS_THREADS + tid;
        key[r] = li < nvalid ? keys[tile_base + li] : 0ULL;
    }
#pragma unroll
    for (int r = 0; r < MS_IPT; r++) {
        const u32 li = (u32)r * MS_THREADS + tid;
        if (li < nvalid) {
            const u32 d = ms_digit<NB>(key[r], shift, base, mask);
            if (d < NB) atomicAdd(&bins[d], 1u);
            else atomicAdd(&hist[base + d], 1u);
        }
    }
    __syncthreads();
    for (u32 i = tid; i < NB; i += MS_THREADS)
        if (bins[i]) atomicAdd(&hist[base + i], bins[i]);
}

// 256 counts (in n_sub histograms, sub-major) -> exclusive offsets in place (one workgroup): the cursor of (sub, bin) starts
// where the records of the lower subs of that bin end
__global__ __launch_bounds__(256) void k_msd_scan256(u32 *hist, u32 n_sub) {
    __shared__ u32 smem[256 / 64 + 1];
    u32 total, v = 0;
    for (u32 s = 0; s < n_sub; s++) v += hist[s * 256u + threadIdx.x];
    u32 run = ks_block_excl_scan(v, smem, &total);
    for (u32 s = 0; s < n_sub; s++) {
        const u32 c = hist[s * 256u + threadIdx.x];
        hist[s * 256u + threadIdx.x] = run;
        run += c;
    }
}

// One partition pass: record -> bin cursor (exact sizes).  Ranks inside (tile, bin) are LDS atomic returns, one global
// atomic per (tile, non-empty bin) reserves the slice, records leave through LDS in bin order so that every bin is
// written as one contiguous run.
template <int NB, int SEG = 0>
__global__ __launch_bounds__(MS_THREADS, 2) void k_msd_scatter(const u64 *kin, u64 *kout, u64 n, int shift, u32 *cur, u32 mask, int bits2,
                                                               u32 sub_stride, ks_msd_segs segs) {
    __shared__ u32 cnt[NB];
    __shared__ u32 dstart[NB];
    __shared__ u32 gbase[NB];
    __shared__ u32 scan_smem[MS_WAVES + 1];
    __shared__ __attribute__((aligned(16))) u64 stage[MS_TILE];
    const u32 tid = threadIdx.x;
    const u32 bid = ks_xcd_block(); // neighbouring tiles (whose runs meet in the same cache lines) share one L2
    u64 tile_base;
    u32 nvalid;
    ms_tile_range<SEG>(bid, n, segs, &tile_base, &nvalid);
    cur += (bid & (MS_SUB - 1u)) * sub_stride; // (the cursors of this tile's sub-histogram, see k_msd_hist)
    for (u32 i = tid; i < NB; i += MS_THREADS) cnt[i] = 0;
    const u32 base = ms_base<NB>(kin, tile_base, shift, mask, bits2);
    u64 key[MS_IPT];
    u32 rank[MS_IPT]; // (bin << 16) | rank within (tile, bin); 0xffffffff = no record (or an outlier, already written)
#pragma unroll
    for (int r = 0; r < MS_IPT; r++) {
        const u32 li = (u32)r * MS_THREADS + tid;
        key[r] = li < nvalid ? kin[tile_base + li] : 0ULL;
    }
    __syncthreads();
    u32 n_out = 0; // outliers of this thread
#pragma unroll
    for (int r = 0; r < MS_IPT; r++) {
        const u32 li = (u32)r * MS_THREADS + tid;
        rank[r] = 0xffffffffu;
        if (li < nvalid) {
            const u32 d = ms_digit<NB>(key[r], shift, base, mask);
            if (d < NB) rank[r] = (d << 16) | atomicAdd(&cnt[d], 1u);
            else { kout[atomicAdd(&cur[base + d], 1u)] = key[r]; n_out++; }
        }
    }
    __syncthreads();
    {
        // thread t owns bins t * BPT .. t * BPT + BPT - 1 (BPT = 1 up to NB = 512 bins, 2 for the 1024-bin window)
        constexpr u32 BPT = (NB + MS_THREADS - 1) / MS_THREADS;
        u32 c[BPT], sum = 0;
#pragma unroll
        for (u32 j = 0; j < BPT; j++) { const u32 b = tid * BPT + j; c[j] = b < NB ? cnt[b] : 0u; sum += c[j]; }
        u32 total;
        u32 ds = ks_block_excl_scan(sum, scan_smem, &total);
#pragma unroll
        for (u32 j = 0; j < BPT; j++) {
            const u32 b = tid * BPT + j;
            if (b < NB) {
                dstart[b] = ds;
                gbase[b] = c[j] ? atomicAdd(&cur[base + b], c[j]) : 0u;
                ds += c[j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MS_IPT; r++)
        if (rank[r] != 0xffffffffu) stage[dstart[rank[r] >> 16] + (rank[r] & 0xffffu)] = key[r];
    __syncthreads();
    const u32 n_staged = dstart[NB - 1] + cnt[NB - 1];
#pragma unroll
    for (int i = 0; i < MS_IPT; i++) {
        const u32 p = (u32)i * MS_THREADS + tid;
        if (p < n_staged) {
            const u64 k = stage[p];
            const u32 d = ms_digit<NB>(k, shift, base, mask);
            kout[(u64)gbase[d] + (p - dstart[d])] = k;
        }
    }
    (void)n_out;
}

// ---- level 3: every level-2 bucket sorted on its remaining bits [lo_bit, lo_bit + rem_bits) -----------------------------
#define ML_THREADS 256
#define ML_IPT 16
#define ML_CAP (ML_THREADS * ML_IPT) // 4096 records: buckets up to this size are sorted inside LDS
#define ML_WAVES (ML_THREADS / 64)

// stable ranks of one wave-ordered tile on an 8-bit digit (the scheme of k_radix_scatter): item order is (wave, round,
// lane); rank[r] = (digit << 16) | rank within (wave, digit); wcnt[w][d] = count, turned into tile-wide starts by ml_offsets
KS_DEV void ml_rank(const u64 *key, u32 *rank, u32 nvalid, u32 wloc, u32 lane, u32 wave, int shift, u32 mask, u32 (*wcnt)[256]) {
#pragma unroll
    for (int r = 0; r < ML_IPT; r++) {
        const u32 li = wloc + (u32)r * 64 + lane;
        const u32 d = li < nvalid ? ((u32)(key[r] >> shift) & mask) : 255u;
        rank[r] = ks_match8_rank(d, wcnt[wave]);
    }
}

// wcnt[w][d] (counts) -> start of (wave w, digit d) inside the digit-ordered tile; dcount[d] = records of digit d
KS_DEV void ml_offsets(u32 (*wcnt)[256], u32 *dcount, u32 *dstart, u32 *scan_smem) {
    const u32 tid = threadIdx.x;
    u32 c[ML_WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < ML_WAVES; w++) { c[w] = wcnt[w][tid]; tot += c[w]; } // ML_THREADS == 256: thread d owns digit d
    u32 total;
    u32 ds = ks_block_excl_scan(tot, scan_smem, &total);
    dcount[tid] = tot;
    dstart[tid] = ds;
#pragma unroll
    for (int w = 0; w < ML_WAVES; w++) { wcnt[w][tid] = ds; ds += c[w]; }
}

#define ML_WCAP (64 * ML_IPT) // 1024 records: a bucket up to this size is sorted by ONE wave, four buckets per workgroup

// One wave sorts one bucket of <= ML_WCAP records: keys in registers (item order = (round, lane)), LSD passes through the
// wave's own slice of LDS.  No workgroup barrier anywhere: the LDS operations of one wave execute in order.
KS_DEV void ml_sort_wave(u64 *keys, u64 s, u32 nb, int lo_bit, int rem_bits, u32 *wc /* [256] */, u64 *wstage /* [ML_WCAP] */, u32 lane) {
    u64 key[ML_IPT];
    u32 rank[ML_IPT];
    const int nr = (int)((nb + 63) >> 6); // rounds that hold records (uniform): a 300-record bucket takes 5 of the 16
#pragma unroll
    for (int r = 0; r < ML_IPT; r++) {
        const u32 li = (u32)r * 64 + lane;
        key[r] = (r < nr && li < nb) ? keys[s + li] : ~0ULL;
    }
    const int n_pass = (rem_bits + 7) / 8;
    for (int p = 0; p < n_pass; p++) {
        const int shift = lo_bit + 8 * p;
        const u32 mask = (rem_bits - 8 * p) >= 8 ? 255u : ((1u << (rem_bits - 8 * p)) - 1u);
#pragma unroll
        for (int j = 0; j < 4; j++) wc[j * 64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < ML_IPT; r++) {
            if (r < nr) { // (a guard, not a break: the loop must unroll or key[] / rank[] leave the registers)
                const u32 li = (u32)r * 64 + lane;
                const u32 d = li < nb ? ((u32)(key[r] >> shift) & mask) : 255u;
                rank[r] = ks_match8_rank(d, wc);
            }
        }
        __builtin_amdgcn_wave_barrier();
        { // counts -> starts: lane l owns digits 4l .. 4l + 3
            const uint4 c = *(const uint4 *)&wc[4 * lane];
            const u32 tot = c.x + c.y + c.z + c.w;
            u32 ex = ks_wave_incl_scan(tot) - tot;
            uint4 o;
            o.x = ex; ex += c.x; o.y = ex; ex += c.y; o.z = ex; ex += c.z; o.w = ex;
            *(uint4 *)&wc[4 * lane] = o;
        }
        __builtin_amdgcn_wave_barrier();
        // (the padding slots of the last round carry digit 255 and entered last: a stable pass keeps them behind every
        // record of the bucket, so wstage[0, nb) is the bucket)
#pragma unroll
        for (int r = 0; r < ML_IPT; r++)
            if (r < nr) wstage[wc[rank[r] >> 16] + (rank[r] & 0xffffu)] = key[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < ML_IPT; r++) {
            const u32 li = (u32)r * 64 + lane;
            if (r < nr) key[r] = li < nb ? wstage[li] : ~0ULL;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < ML_IPT; r++) {
        const u32 li = (u32)r * 64 + lane;
        if (r < nr && li < nb) keys[s + li] = key[r];
    }
}

// Workgroup b sorts buckets 4b .. 4b + 3, each wave its own bucket — when it holds <= ML_WCAP records (the common case:
// the partition levels are sized for a few hundred per bucket); larger ones are left to k_msd_local_big.
__global__ __launch_bounds__(ML_THREADS) void k_msd_local(u64 *keys, const u32 *off, u32 n_buckets, int lo_bit, int rem_bits, u32 lds_cap,
                                                          u32 *big /* [0] = count, then bucket ids */) {
    __shared__ u32 wcnt[ML_WAVES][256];
    __shared__ __attribute__((aligned(16))) u64 stage[ML_CAP];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (rem_bits <= 0) return;
    const u32 wave_cap = lds_cap < ML_WCAP ? lds_cap : ML_WCAP;
    const u32 bw = blockIdx.x * ML_WAVES + wave;
    if (bw < n_buckets) {
        const u64 s = bw ? off[bw - 1] : 0, nb64 = (u64)off[bw] - s; // off[b] = END of bucket b (the scatter's cursors, spent)
        if (nb64 > 1 && nb64 <= wave_cap) ml_sort_wave(keys, s, (u32)nb64, lo_bit, rem_bits, wcnt[wave], stage + wave * ML_WCAP, lane);
        else if (nb64 > wave_cap && lane == 0) big[1 + atomicAdd(&big[0], 1u)] = bw; // for k_msd_local_big
    }
}

// The buckets k_msd_local listed (more than ML_WCAP records: skewed lists), a fixed grid striding over the list, one
// workgroup per bucket: inside LDS up to lds_cap records, else by serial stable LSD passes over global memory.
__global__ __launch_bounds__(ML_THREADS) void k_msd_local_big(u64 *keys, u64 *scratch, const u32 *off, const u32 *big, int lo_bit, int rem_bits,
                                                              u32 lds_cap) {
    __shared__ u32 wcnt[ML_WAVES][256];
    __shared__ u32 dcount[256], dstart[256], cursor[256];
    __shared__ u32 scan_smem[ML_WAVES + 1];
    __shared__ __attribute__((aligned(16))) u64 stage[ML_CAP];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rem_bits <= 0) return;
    const int n_pass = (rem_bits + 7) / 8;
    const u32 wloc = wave * (64 * ML_IPT);
    const u32 n_big = big[0];
    for (u32 w = blockIdx.x; w < n_big; w += gridDim.x) {
        const u32 bb = big[1 + w];
        const u64 s = bb ? off[bb - 1] : 0, e = off[bb];
        const u64 nb64 = e - s;
        __syncthreads(); // the previous large bucket is done with LDS
        u64 key[ML_IPT];
        u32 rank[ML_IPT];
        if (nb64 <= lds_cap) {
            // ---- the bucket fits: LSD passes over the remaining bits without leaving LDS
            const u32 nb = (u32)nb64;
#pragma unroll
            for (int r = 0; r < ML_IPT; r++) {
                const u32 li = wloc + (u32)r * 64 + lane;
                key[r] = li < nb ? keys[s + li] : ~0ULL;
            }
            for (int p = 0; p < n_pass; p++) {
                const int shift = lo_bit + 8 * p;
                const u32 mask = (rem_bits - 8 * p) >= 8 ? 255u : ((1u << (rem_bits - 8 * p)) - 1u);
                for (int ww = 0; ww < ML_WAVES; ww++) wcnt[ww][tid] = 0;
                __syncthreads();
                ml_rank(key, rank, nb, wloc, lane, wave, shift, mask, wcnt);
                __syncthreads();
                ml_offsets(wcnt, dcount, dstart, scan_smem);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < ML_IPT; r++) stage[wcnt[wave][rank[r] >> 16] + (rank[r] & 0xffffu)] = key[r];
                __syncthreads();
                // (slots beyond the bucket carry digit 255 in every pass and entered last: a stable pass keeps them behind
                // every record of the bucket, so stage[0, nb) is the bucket)
#pragma unroll
                for (int r = 0; r < ML_IPT; r++) {
                    const u32 li = wloc + (u32)r * 64 + lane;
                    key[r] = li < nb ? stage[li] : ~0ULL;
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < ML_IPT; r++) {
                const u32 li = wloc + (u32)r * 64 + lane;
                if (li < nb) keys[s + li] = key[r];
            }
            continue;
        }
        // ---- the bucket does not fit LDS (a query / target that matches nearly everything): this workgroup sorts it with
        // serial stable LSD passes over global memory, tile by tile, ping-pong between the list and the scratch list
        u64 *src = keys + s, *dst = scratch + s;
        for (int p = 0; p < n_pass; p++) {
            const int shift = lo_bit + 8 * p;
            const u32 mask = (rem_bits - 8 * p) >= 8 ? 255u : ((1u << (rem_bits - 8 * p)) - 1u);
            cursor[tid] = 0;
            __syncthreads();
            for (u64 t0 = 0; t0 < nb64; t0 += ML_CAP) // histogram of the pass
                for (u32 i = tid; i < ML_CAP && t0 + i < nb64; i += ML_THREADS) atomicAdd(&cursor[(u32)(src[t0 + i] >> shift) & mask], 1u);
            __syncthreads();
            {
                u32 total;
                const u32 ex = ks_block_excl_scan(cursor[tid], scan_smem, &total);
                __syncthreads();
                cursor[tid] = ex; // first free slot of digit d in dst (a bucket holds < 2^32 records: the whole list does)
            }
            __syncthreads();
            for (u64 t0 = 0; t0 < nb64; t0 += ML_CAP) {
                const u32 nv = (nb64 - t0) < ML_CAP ? (u32)(nb64 - t0) : ML_CAP;
#pragma unroll
                for (int r = 0; r < ML_IPT; r++) {
                    const u32 li = wloc + (u32)r * 64 + lane;
                    key[r] = li < nv ? src[t0 + li] : ~0ULL;
                }
                for (int ww = 0; ww < ML_WAVES; ww++) wcnt[ww][tid] = 0;
                __syncthreads();
                ml_rank(key, rank, nv, wloc, lane, wave, shift, mask, wcnt);
                __syncthreads();
                ml_offsets(wcnt, dcount, dstart, scan_smem);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < ML_IPT; r++) {
                    const u32 li = wloc + (u32)r * 64 + lane;
                    if (li < nv) {
                        const u32 d = rank[r] >> 16;
                        dst[(u64)cursor[d] + (wcnt[wave][d] - dstart[d]) + (rank[r] & 0xffffu)] = key[r];
                    }
                }
                __syncthreads();
                // the padding slots (~0 keys) of a partial tile were counted under the pass's LARGEST digit — `mask`, which is
                // 255 only when the pass has all 8 bits: only real records advance the cursors
                if (tid == mask && nv < ML_CAP) dcount[mask] -= ML_CAP - nv;
                __syncthreads();
                cursor[tid] += dcount[tid];
                __syncthreads();
            }
            __threadfence(); // the next pass reads what other waves of this workgroup wrote to global memory
            __syncthreads();
            u64 *t = src; src = dst; dst = t;
        }
        if (src != keys + s) { // an odd number of passes left the bucket in the scratch list
            for (u64 i = tid; i < nb64; i += ML_THREADS) dst[i] = src[i];
        }
    }
}

// Sorts the n packed match records of `ka` on their key bits [lo_bit, lo_bit + nbits) (kb: scratch of the same size).
// *done = 0 when the list is too small / the key too narrow for this path to pay (the caller takes the LSD sort).
int ks_sort_pairs_msd(ks_ctx *ctx, u64 *ka, u64 *kb, u64 n, int lo_bit, int nbits, int *done, const ks_msd_segs *segs) {
    *done = 0;
    if (nbits <= 16 || n < 65536 || n >= 0xffffffffULL || ks_dbg(ctx, KS_DBG_PAIRS_LSD)) return KS_OK;
    // level 2 is as wide as it takes for ~768 records per bucket (0 .. 8 bits): a short list does not pay 65,536 buckets
    // ... and then as wide as it takes to save a local pass (remaining bits a multiple of 8): passes cost more than buckets
    // (up to 9 bits: 131,072 buckets through a 1024-bin window — lists beyond ~40 M records, where 65,536 buckets would leave
    // an average of ~750 and the heavy rows of an all-vs-all push half of them past what one wave sorts: 0.75 of 3.6 ms
    // went to k_msd_local_big at 200k x 200k hp)
    int bits2 = 0;
    while (bits2 < 9 && (n >> (8 + bits2)) > (bits2 < 8 ? 768u : 560u)) bits2++;
    for (int b = bits2 + 1; b <= 8 && nbits - 8 - b > 0; b++)
        if ((nbits - 8 - b + 7) / 8 < (nbits - 8 - bits2 + 7) / 8) { bits2 = b; break; }
    if (nbits - 8 - bits2 <= 0) bits2 = nbits - 9 > 0 ? nbits - 9 : 0; // (at least one bit left for the local sort)
    const int shift1 = lo_bit + nbits - 8, shift2 = shift1 - bits2;
    const u32 mask2 = (1u << (8 + bits2)) - 1u, n_buckets = 1u << (8 + bits2);
    const u32 n_tiles = (u32)((n + MS_TILE - 1) / MS_TILE);             // of a dense list (level 2 always reads one)
    const u32 n_tiles1 = segs ? segs->tile_start[segs->n] : n_tiles;    // of level 1
    ks_msd_segs no_segs;
    no_segs.n = 0;
    u32 lds_cap = ML_CAP;
    if (const char *f = ks_dbg(ctx, KS_DBG_MSD_LDS_CAP)) { // exercises the large-bucket paths on small inputs
        const u32 v = (u32)atoi(f);
        if (v >= 2 && v < ML_CAP) lds_cap = v;
    }
    // The exclusive offsets double as the scatter's cursors: once spent, cursor[b] is the END of bucket b, which is all the
    // local sort needs (no copy of the offsets is kept).
    // Layout of the one block: off1 (256 x MS_SUB) | off2 (n_buckets + 1, when there is a level 2) | big-list counter | big list
    // — everything that starts at zero comes first, so ONE memset clears both histograms and the counter.
    u32 *blk = nullptr;
    // (level 1 followed by level 2: MS_SUB histograms; level 1 alone — short lists — one, whose spent cursors are the ends)
    const u32 n_sub = bits2 > 0 ? MS_SUB : 1u, sub_stride = bits2 > 0 ? 256u : 0u;
    const size_t off2_words = bits2 > 0 ? (size_t)n_buckets + 1 : 0, n_zero = 256 * MS_SUB + off2_words + 1;
    int st = ks_alloc(ctx, &blk, n_zero + (size_t)(n_buckets > 65536u ? n_buckets : 65536u) + 1);
    u32 *off1 = blk, *off2 = blk ? blk + 256 * MS_SUB : nullptr, *big = blk ? blk + 256 * MS_SUB + off2_words : nullptr;
    u64 *sorted_in = ka; // where the partitioned list ends up
    if (st == KS_OK) {
        (void)hipMemsetAsync(blk, 0, n_zero * sizeof(u32), ctx->stream);
        ks_timer_begin(ctx, "msd_hist");
        if (segs) hipLaunchKernelGGL((k_msd_hist<256, 1>), dim3(n_tiles1), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)ka, n, shift1, off1, 255u, 0, sub_stride, *segs);
        else hipLaunchKernelGGL((k_msd_hist<256, 0>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)ka, n, shift1, off1, 255u, 0, sub_stride, no_segs);
        ks_timer_end(ctx);
        ks_timer_begin(ctx, "msd_scan");
        hipLaunchKernelGGL(k_msd_scan256, dim3(1), dim3(256), 0, ctx->stream, off1, n_sub);
        ks_timer_end(ctx);
        ks_timer_begin(ctx, "msd_scatter");
        if (segs) hipLaunchKernelGGL((k_msd_scatter<256, 1>), dim3(n_tiles1), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)ka, kb, n, shift1, off1, 255u, 0,
                                     sub_stride, *segs);
        else hipLaunchKernelGGL((k_msd_scatter<256, 0>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)ka, kb, n, shift1, off1, 255u, 0,
                                sub_stride, no_segs);
        ks_timer_end(ctx);
        sorted_in = kb;
    }
    const u32 *off = off1;
    if (st == KS_OK && bits2 > 0) {
        ks_timer_begin(ctx, "msd_hist");
        if (bits2 <= 8) hipLaunchKernelGGL((k_msd_hist<512>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)kb, n, shift2, off2, mask2, bits2, 0u, no_segs);
        else hipLaunchKernelGGL((k_msd_hist<1024>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)kb, n, shift2, off2, mask2, bits2, 0u, no_segs);
        ks_timer_end(ctx);
        st = ks_scan_u32_inplace(ctx, off2, n_buckets, nullptr);
        if (st == KS_OK) {
            ks_timer_begin(ctx, "msd_scatter");
            if (bits2 <= 8) hipLaunchKernelGGL((k_msd_scatter<512>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)kb, ka, n, shift2, off2, mask2, bits2, 0u, no_segs);
            else hipLaunchKernelGGL((k_msd_scatter<1024>), dim3(n_tiles), dim3(MS_THREADS), 0, ctx->stream, (const u64 *)kb, ka, n, shift2, off2, mask2, bits2, 0u, no_segs);
            ks_timer_end(ctx);
            sorted_in = ka;
            off = off2;
        }
    }
    if (st == KS_OK) {
        u64 *other = sorted_in == ka ? kb : ka;
        ks_timer_begin(ctx, "msd_local");
        hipLaunchKernelGGL(k_msd_local, dim3((n_buckets + ML_WAVES - 1) / ML_WAVES), dim3(ML_THREADS), 0, ctx->stream, sorted_in, off,
                           n_buckets, lo_bit, shift2 - lo_bit, lds_cap, big);
        ks_timer_end(ctx);
        ks_timer_begin(ctx, "msd_local_big");
        hipLaunchKernelGGL(k_msd_local_big, dim3(n_buckets < 2048 ? n_buckets : 2048), dim3(ML_THREADS), 0, ctx->stream, sorted_in, other, off,
                           (const u32 *)big, lo_bit, shift2 - lo_bit, lds_cap);
        ks_timer_end(ctx);
        if (sorted_in != ka) // (one partition level only: the sorted list sits in the scratch buffer)
            if (hipMemcpyAsync(ka, kb, (size_t)n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                st = ks_fail(ctx, KS_ERR_HIP, "msd sort: copy back failed");
        if (st == KS_OK && hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "msd sort launch failed");
        if (st == KS_OK) *done = 1;
    }
    ks_pool_free(ctx, blk);
    return st;
}
