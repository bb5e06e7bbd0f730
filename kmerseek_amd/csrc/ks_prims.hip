// ks_prims.hip — device-wide primitives written for gfx950: exclusive scans and a stable LSD
// radix sort (8-bit digits) for (u64 key, u32|u64 value) records.  Integer / HBM-bound work:
// coalesced 8..16-B per-lane accesses, LDS-staged scatter so each digit leaves as a contiguous run.
#include "ks_device.h"
#include <type_traits>

// =============================================================================================
// scans
// =============================================================================================
template <typename T>
KS_DEV T wave_incl_scan_t(T v) {
    const u32 lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = __shfl_up(v, d, 64);
        if (lane >= (u32)d) v += t;
    }
    return v;
}

template <typename T>
KS_DEV T block_excl_scan_t(T v, T *smem /* blockDim/64 + 1 */, T *total) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T incl = wave_incl_scan_t<T>(v);
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        T w = lane < nw ? smem[lane] : (T)0;
        T wi = wave_incl_scan_t<T>(w);
        if (lane < nw) smem[lane] = wi - w;
        if (lane == nw - 1) smem[nw] = wi;
    }
    __syncthreads();
    T base = smem[wave];
    *total = smem[nw];
    __syncthreads();
    return base + incl - v;
}

#define SCAN_THREADS 256
#define SCAN_IPT 8
#define SCAN_TILE (SCAN_THREADS * SCAN_IPT)

template <typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const TIn *in, TOut *block_sums, u64 n) {
    __shared__ TOut smem[SCAN_THREADS / 64 + 1];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        u64 idx = base + (u64)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += (TOut)in[idx];
    }
    TOut total;
    (void)block_excl_scan_t<TOut>(s, smem, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// out[i] = block_offsets[block] + exclusive prefix within the tile.  in may alias out.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const TIn *in, TOut *out, const TOut *block_offsets,
                                                             u64 n, TOut *total_out) {
    __shared__ TOut smem[SCAN_THREADS / 64 + 1];
    const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_IPT;
    TOut v[SCAN_IPT];
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        u64 idx = base + i;
        v[i] = idx < n ? (TOut)in[idx] : (TOut)0;
        s += v[i];
    }
    TOut total;
    TOut ex = block_excl_scan_t<TOut>(s, smem, &total);
    TOut off = (block_offsets ? block_offsets[blockIdx.x] : (TOut)0) + ex;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        u64 idx = base + i;
        if (idx < n) out[idx] = off;
        off += v[i];
    }
    // the thread that owns the last element also publishes the grand total
    if (total_out && base <= n - 1 && n - 1 < base + SCAN_IPT) *total_out = off;
}

// single-block scan of a small array in place (n <= SCAN_TILE)
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_small(T *data, u64 n) {
    __shared__ T smem[SCAN_THREADS / 64 + 1];
    const u64 base = (u64)threadIdx.x * SCAN_IPT;
    T v[SCAN_IPT];
    T s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        v[i] = base + i < n ? data[base + i] : (T)0;
        s += v[i];
    }
    T total;
    T off = block_excl_scan_t<T>(s, smem, &total);
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        if (base + i < n) data[base + i] = off;
        off += v[i];
    }
}

// One-launch exclusive scan: tiles in ticket order, the tile's offset by decoupled look-back (as the sketch kernel
// places its CSR).  The status words live in a ring owned by the context that is never cleared: word =
// flag (2) | tag (24) | value (38), tag = low bits of the GLOBAL tile number (ticket counter value) that wrote it, so a
// left-over of an earlier scan never matches the tile number a reader expects (the slot was rewritten RING tiles ago
// at the latest, and RING < 2^24).  Saves two launches (reduce + scan of the sums) per scan and the memset a fresh
// status array would need: a step of the search runs seven scans.
#define SCAN_RING (1u << 20)
#define SCAN_FLAG_AGG 1ULL
#define SCAN_FLAG_PRE 2ULL
#define SCAN_VAL_BITS 38
KS_DEV unsigned long long scan_word(u64 flag, u32 gtile, u64 value) {
    return (flag << 62) | ((u64)(gtile & 0xffffffu) << SCAN_VAL_BITS) | value;
}
template <typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(const TIn *in, TOut *out, u64 n, TOut *total_out,
                                                                unsigned long long *ring, u32 *ticket, u32 ticket_base,
                                                                u32 n_tiles) {
    __shared__ TOut smem[SCAN_THREADS / 64 + 1];
    __shared__ u32 tile_s;
    __shared__ unsigned long long base_s;
    const u32 tid = threadIdx.x;
    if (tid == 0) tile_s = atomicAdd(&ticket[0], 1u) - ticket_base;
    __syncthreads();
    const u32 tile = tile_s;
    const u64 base = (u64)tile * SCAN_TILE + (u64)tid * SCAN_IPT;
    TOut v[SCAN_IPT];
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        const u64 idx = base + i;
        v[i] = idx < n ? (TOut)in[idx] : (TOut)0;
        s += v[i];
    }
    TOut total;
    const TOut ex = block_excl_scan_t<TOut>(s, smem, &total);
    const u32 g = ticket_base + tile; // global tile number
    if (tid == 0)
        __hip_atomic_store(&ring[g % SCAN_RING], scan_word(tile == 0 ? SCAN_FLAG_PRE : SCAN_FLAG_AGG, g, (u64)total),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        u64 excl = 0;
        if (tile > 0) {
            i64 idx = (i64)tile - 1;
            bool done = false;
            u32 spins = 0;
            const long long spin_t0 = wall_clock64();
            while (!done) {
                const i64 mine = idx - (i64)tid;
                u64 flag = SCAN_FLAG_PRE, val = 0; // before tile 0: inclusive prefix 0
                if (mine >= 0) {
                    const u32 gp = ticket_base + (u32)mine;
                    const u64 want = (u64)(gp & 0xffffffu);
                    for (;;) {
                        const u64 w = __hip_atomic_load(&ring[gp % SCAN_RING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag = w >> 62;
                        if (flag != 0 && ((w >> SCAN_VAL_BITS) & 0xffffffu) == want) { val = w & ((1ULL << SCAN_VAL_BITS) - 1ULL); break; }
                        flag = 0;
                        if (ks_spin_expired(spin_t0, spins)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (flag == 0) { atomicOr(&ticket[1], 1u); flag = SCAN_FLAG_PRE; val = 0; } // gave up: the host reports it
                const u64 is_pre = __ballot(flag == SCAN_FLAG_PRE);
                const u32 first = is_pre ? (u32)__ffsll((long long)is_pre) - 1u : 64u;
                u64 contrib = tid <= first ? val : 0;
                contrib = ks_wave_sum64(contrib);
                excl += contrib;
                if (is_pre) done = true; else idx -= 64;
            }
            if (tid == 0)
                __hip_atomic_store(&ring[g % SCAN_RING], scan_word(SCAN_FLAG_PRE, g, excl + (u64)total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) base_s = excl;
    }
    __syncthreads();
    TOut off = (TOut)base_s + ex;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        const u64 idx = base + i;
        if (idx < n) out[idx] = off;
        off += v[i];
    }
    if (total_out && tile == n_tiles - 1 && tid == 0) *total_out = (TOut)(base_s + (u64)total);
}

template <typename TIn, typename TOut>
static int scan_generic(ks_ctx *ctx, const TIn *in, TOut *out, u64 n, TOut *d_total) {
    if (n == 0) {
        if (d_total) KS_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(TOut), ctx->stream));
        return KS_OK;
    }
    u64 nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nblocks == 1) {
        KS_LAUNCH(ctx, "scan_apply", (k_scan_apply<TIn, TOut>), 1, SCAN_THREADS, in, out, (const TOut *)nullptr, n, d_total);
        return KS_OK;
    }
    if (nblocks <= SCAN_RING / 2 && n < (1ULL << SCAN_VAL_BITS) && !ks_dbg(ctx, KS_DBG_SCAN_3PASS)) {
        if (!ctx->scan_ring) {
            KS_HIP(ctx, hipMalloc((void **)&ctx->scan_ring, (size_t)SCAN_RING * sizeof(unsigned long long)));
            KS_HIP(ctx, hipMalloc((void **)&ctx->scan_ticket, 2 * sizeof(u32)));
            KS_HIP(ctx, hipMemsetAsync(ctx->scan_ring, 0, (size_t)SCAN_RING * sizeof(unsigned long long), ctx->stream));
            KS_HIP(ctx, hipMemsetAsync(ctx->scan_ticket, 0, 2 * sizeof(u32), ctx->stream));
            ctx->scan_ticket_base = 0;
        }
        ks_timer_begin(ctx, "scan_lookback");
        hipLaunchKernelGGL((k_scan_lookback<TIn, TOut>), dim3((u32)nblocks), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, d_total,
                           ctx->scan_ring, ctx->scan_ticket, ctx->scan_ticket_base, (u32)nblocks);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) return ks_fail(ctx, KS_ERR_HIP, "scan launch failed");
        ctx->scan_ticket_base += (u32)nblocks;
        return KS_OK;
    }
    TOut *sums = nullptr;
    KS_TRY(ks_alloc(ctx, &sums, nblocks));
    KS_LAUNCH(ctx, "scan_reduce", (k_scan_reduce<TIn, TOut>), (u32)nblocks, SCAN_THREADS, in, sums, n);
    int st;
    if (nblocks <= SCAN_TILE) {
        KS_LAUNCH(ctx, "scan_small", (k_scan_small<TOut>), 1, SCAN_THREADS, sums, nblocks);
        st = KS_OK;
    } else {
        st = scan_generic<TOut, TOut>(ctx, sums, sums, nblocks, (TOut *)nullptr);
    }
    if (st == KS_OK) {
        ks_timer_begin(ctx, "scan_apply");
        hipLaunchKernelGGL((k_scan_apply<TIn, TOut>), dim3((u32)nblocks), dim3(SCAN_THREADS), 0, ctx->stream, in, out,
                           (const TOut *)sums, n, d_total);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "scan_apply launch failed");
    }
    ks_pool_free(ctx, sums); // stream-ordered reuse: later kernels on the same stream run after this one
    return st;
}

int ks_scan_status_fetch(ks_ctx *ctx) {
    if (!ctx->scan_ticket) return KS_OK;
    KS_HIP(ctx, hipMemcpyAsync(ctx->h_pin + 40, ctx->scan_ticket + 1, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    return KS_OK;
}
bool ks_scan_status_seg(ks_ctx *ctx, ks_fetch_seg *out) {
    if (!ctx->scan_ticket) return false;
    *out = ks_fetch_words(ctx->scan_ticket + 1, ctx->h_pin + 40, 1);
    return true;
}
int ks_scan_status_check(ks_ctx *ctx) {
    if (!ctx->scan_ticket || *(u32 *)(ctx->h_pin + 40) == 0) return KS_OK;
    *(u32 *)(ctx->h_pin + 40) = 0;
    (void)hipMemsetAsync(ctx->scan_ticket + 1, 0, sizeof(u32), ctx->stream);
    return ks_fail(ctx, KS_ERR_HIP, "scan: look-back gave up waiting for a predecessor tile");
}

int ks_scan_u32_to_u64(ks_ctx *ctx, const u32 *in, u64 *out, u64 n) {
    // out[n] receives the total
    return scan_generic<u32, u64>(ctx, in, out, n, out + n);
}

int ks_scan_u32_inplace(ks_ctx *ctx, u32 *data, u64 n, u32 *d_total) {
    return scan_generic<u32, u32>(ctx, data, data, n, d_total);
}

// =============================================================================================
// radix sort
// =============================================================================================
#ifndef RS_THREADS
#define RS_THREADS 512
#endif
#ifndef RS_IPT
#define RS_IPT 16
#endif
#ifndef RS_MINW
#define RS_MINW 2 // waves per SIMD the scatter kernel is compiled for (register budget)
#endif
#define RS_TILE (RS_THREADS * RS_IPT) // 8192 records: ~32 per digit, so digits leave the tile as >= 128-B runs
#define RS_WAVES (RS_THREADS / 64)
#define BKS_NB 512 // most high digits of the histogram-free bucket scatter (join prefixes of up to 8 + 9 bits)

// Input geometry of a pass.  Dense: tile b covers records [b*RS_TILE, ...) of n.  Segmented (seg_len != NULL): the
// input is a set of fixed-capacity regions (region r holds seg_len[r] records from r*seg_cap on) and tile b is the
// (b % tiles_per_seg)-th tile of region b / tiles_per_seg — how the sketch kernel's pre-partitioned postings enter.
KS_DEV void rs_tile_geom(u32 bid, u64 n, const u32 *seg_len, u64 seg_cap, u32 tiles_per_seg, u64 &in_base, u32 &nvalid) {
    if (seg_len) {
        const u32 r = bid / tiles_per_seg, j = bid % tiles_per_seg;
        const u64 len = seg_len[r], off = (u64)j * RS_TILE;
        nvalid = off < len ? (u32)((len - off) < RS_TILE ? (len - off) : RS_TILE) : 0u;
        in_base = (u64)r * seg_cap + off;
    } else {
        in_base = (u64)bid * RS_TILE;
        const u64 remain = n - in_base;
        nvalid = remain < RS_TILE ? (u32)remain : RS_TILE;
    }
}

// TAG only gives the instantiations of one kernel distinct names per use (index build / query
// partition / match sort), so profiler rows and the library's own HIP-event table line up.
// hist[d * nblocks + block] = number of keys of this block's tile with digit d
template <int TAG>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const u64 *keys, u32 *hist, u64 n, int shift, u32 nblocks,
                                                           const u32 *seg_len, u64 seg_cap, u32 tiles_per_seg, u32 pfxK) {
    __shared__ u32 bins[256];
    if (threadIdx.x < 256) bins[threadIdx.x] = 0;
    __syncthreads();
    u64 base;
    u32 nvalid;
    rs_tile_geom(blockIdx.x, n, seg_len, seg_cap, tiles_per_seg, base, nvalid);
    u64 key[RS_IPT]; // (requested together: a load inside the guarded loop is waited for in every turn)
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const u32 li = (u32)i * RS_THREADS + threadIdx.x;
        key[i] = li < nvalid ? keys[base + li] : 0ULL;
    }
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const u32 li = (u32)i * RS_THREADS + threadIdx.x;
        if (li < nvalid) atomicAdd(&bins[ks_rs_digit(key[i], shift, pfxK)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(u64)threadIdx.x * nblocks + blockIdx.x] = bins[threadIdx.x];
}

struct ks_noval {}; // value type of a keys-only sort: every value move below compiles away

// Stable scatter.  Item order inside a tile is (wave, round, lane) = ascending global index, so
// ranks computed per wave with ballot-matching + per-wave digit counters preserve input order.
// Records are staged through ONE LDS buffer in local digit order (keys first, then values), so each
// digit leaves the tile as a contiguous run of full cache lines.
template <typename V, int TAG>
__global__ __launch_bounds__(RS_THREADS, RS_MINW) void k_radix_scatter(const u64 *kin, const V *vin, u64 *kout, V *vout,
                                                              const u32 *goffs, u64 n, int shift, u32 nblocks,
                                                              const u32 *seg_len, u64 seg_cap, u32 tiles_per_seg, u32 pfxK) {
    __shared__ u32 wcnt[RS_WAVES][256];
    __shared__ u32 dstart[256];
    __shared__ u32 gbase[256];
    __shared__ u32 scan_smem[RS_WAVES + 1];
    __shared__ __attribute__((aligned(16))) u64 stage[RS_TILE];

    constexpr bool HAS_V = !std::is_same<V, ks_noval>::value;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) {
        for (int w = 0; w < RS_WAVES; w++) wcnt[w][tid] = 0;
        gbase[tid] = goffs[(u64)tid * nblocks + blockIdx.x];
    }
    __syncthreads();

    u64 tile_base;
    u32 nvalid;
    rs_tile_geom(blockIdx.x, n, seg_len, seg_cap, tiles_per_seg, tile_base, nvalid);
    const u32 wloc = wave * (64 * RS_IPT); // this wave's first record inside the tile
    u64 key[RS_IPT];
    V val[RS_IPT];
    u32 rank[RS_IPT]; // (digit << 16) | rank within (wave, digit)
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        const u32 li = wloc + (u32)r * 64 + lane;
        const bool valid = li < nvalid;
        key[r] = valid ? kin[tile_base + li] : ~0ULL;
        if constexpr (HAS_V) val[r] = valid ? vin[tile_base + li] : (V)0;
    }
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        const u32 li = wloc + (u32)r * 64 + lane;
        u32 d = li < nvalid ? ks_rs_digit(key[r], shift, pfxK) : 255u;
        rank[r] = ks_match8_rank(d, wcnt[wave]); // lanes holding the same digit, in lane order
    }
    __syncthreads();
    // digit-major exclusive offsets: thread d < 256 owns digit d
    {
        u32 c[RS_WAVES];
        u32 tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { c[w] = wcnt[w][tid]; tot += c[w]; }
        }
        u32 total;
        u32 ds = ks_block_excl_scan(tot, scan_smem, &total);
        if (tid < 256) {
            dstart[tid] = ds;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { wcnt[w][tid] = ds; ds += c[w]; }
        }
    }
    __syncthreads();
    u32 pos[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        pos[r] = wcnt[wave][rank[r] >> 16] + (rank[r] & 0xffffu);
        stage[pos[r]] = key[r];
    }
    __syncthreads();
    u64 gdst[RS_IPT]; // global destination of local slot i*RS_THREADS + tid
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        u32 p = (u32)i * RS_THREADS + tid;
        gdst[i] = ~0ULL;
        if (p < nvalid) {
            u64 k = stage[p];
            u32 d = ks_rs_digit(k, shift, pfxK);
            gdst[i] = (u64)gbase[d] + (p - dstart[d]);
            kout[gdst[i]] = k;
        }
    }
    if constexpr (HAS_V) {
        __syncthreads();
        V *vstage = (V *)stage;
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) vstage[pos[r]] = val[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RS_IPT; i++) {
            u32 p = (u32)i * RS_THREADS + tid;
            if (gdst[i] != ~0ULL) vout[gdst[i]] = vstage[p];
        }
    }
}

// The last partition pass of a search: segmented postings (region r = low digit of the join prefix) -> 2^pbits
// fixed-capacity buckets (bucket (d << 8 | r) in storage slot KS_BSLOT(d, r) * bcap, d = high digit).  Each tile reserves its slice of a
// bucket with one global atomic per digit, so there is no histogram pass and no scan; order inside a bucket is
// irrelevant to the join, so the rank of a record inside its (tile, digit) group is just the return value of one LDS
// atomic — no ballot matching, no per-wave counters.  A bucket that would overflow raises status[1] and the host
// redoes the pass the dense (stable, histogrammed) way.
// V16 = 1: the value column is 16 bits wide, in and out (10-byte query postings: ks_sketches::part_s).
// V16 = 2: 16 bits in, 8 bits out (9-byte postings; join prefixes of 16 bits only): the bucket implies the key's top byte too —
// it is this pass's digit —, so the low byte of the value moves there and its high byte is the 8-bit column that leaves.
template <int V16>
// (two workgroups of 8 waves per CU by the 70 KB of LDS: 4 waves per SIMD, i.e. a budget of 128 registers)
__global__ __launch_bounds__(RS_THREADS, 4) void k_bucket_scatter(const u64 *kin, const u32 *vin, u64 *kout, u32 *vout,
                                                                        int shift, const u32 *seg_len, u64 seg_cap,
                                                                        u32 tiles_per_seg, u32 *bcur, u32 bcap,
                                                                        unsigned long long *status, u32 pfxK, u32 n_hi, u32 sub_shift) {
    __shared__ u32 cnt[BKS_NB];     // the high digit takes up to BKS_NB values (n_hi of them in use: a power of two)
    __shared__ u32 dstart[BKS_NB];
    __shared__ u32 gbase[BKS_NB];
    __shared__ u32 scan_smem[RS_WAVES + 1];
    __shared__ __attribute__((aligned(16))) u64 stage[RS_TILE];

    const u32 tid = threadIdx.x;
    u64 tile_base;
    u32 nvalid;
    const u32 bid = ks_xcd_block(); // neighbouring tiles (whose runs meet in the same cache lines) share one L2
    {
        const u32 r = bid / tiles_per_seg, j = bid % tiles_per_seg;
        const u64 len = seg_len[r], off = (u64)j * RS_TILE;
        nvalid = off < len ? (u32)((len - off) < RS_TILE ? (len - off) : RS_TILE) : 0u;
        tile_base = (u64)r * seg_cap + off;
    }
    if (nvalid == 0) return; // tile beyond the region's fill (uniform per block)
    const u32 region = (bid / tiles_per_seg) >> sub_shift;
    if (tid < BKS_NB) cnt[tid] = 0;
    u64 key[RS_IPT];
    u32 val[RS_IPT];
    u32 rank[RS_IPT]; // (digit << 16) | rank within (tile, digit); 0xffffffff = no record
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        const u32 li = (u32)r * RS_THREADS + tid;
        const bool valid = li < nvalid;
        key[r] = valid ? kin[tile_base + li] : 0ULL;
        val[r] = valid ? (V16 ? (u32)((const u16 *)vin)[tile_base + li] : vin[tile_base + li]) : 0u;
    }
    __syncthreads();
    // A full tile — all but the last of a sub-region — ranks its records without a guard: sixteen LDS atomics in flight, one wait.
    // (Behind `if (li < nvalid)` every atomic sat in a branch of its own and was waited for in turn: sixteen LDS round trips in a
    // row per thread, with four waves per SIMD to hide them.)
    const bool full = nvalid == RS_TILE; // (uniform)
    if (full) {
        u32 dg[RS_IPT];
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) dg[r] = (ks_join_prefix(key[r], pfxK) >> shift) & (n_hi - 1u);
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) rank[r] = atomicAdd(&cnt[dg[r]], 1u);
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) rank[r] |= dg[r] << 16;
    } else {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) {
            const u32 li = (u32)r * RS_THREADS + tid;
            rank[r] = 0xffffffffu;
            if (li < nvalid) {
                const u32 d = (ks_join_prefix(key[r], pfxK) >> shift) & (n_hi - 1u);
                rank[r] = (d << 16) | atomicAdd(&cnt[d], 1u);
            }
        }
    }
    __syncthreads();
    {
        const u32 c = tid < BKS_NB ? cnt[tid] : 0u;
        u32 total;
        const u32 ds = ks_block_excl_scan(c, scan_smem, &total);
        if (tid < BKS_NB) {
            dstart[tid] = ds;
            u32 base = 0;
            if (c) {
                base = atomicAdd(&bcur[KS_BSLOT(tid, region, n_hi)], c);
                if (base + c > bcap) atomicOr(&status[1], 1ULL);
            }
            gbase[tid] = base;
        }
    }
    __syncthreads();
    if (full) {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) rank[r] = dstart[rank[r] >> 16] + (rank[r] & 0xffffu);
    } else {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++)
            if (rank[r] != 0xffffffffu) rank[r] = dstart[rank[r] >> 16] + (rank[r] & 0xffffu);
    }
    u32 vo[V16 == 2 ? RS_IPT : 1]; // V16 = 2: the values in output order, before the keys go through the same buffer
    if (V16 == 2) {
        u32 *vstage0 = (u32 *)stage;
#pragma unroll
        for (int r = 0; r < RS_IPT; r++)
            if (rank[r] != 0xffffffffu) vstage0[rank[r]] = val[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RS_IPT; i++) vo[V16 == 2 ? i : 0] = vstage0[(u32)i * RS_THREADS + tid];
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RS_IPT; r++)
        if (rank[r] != 0xffffffffu) stage[rank[r]] = key[r];
    __syncthreads();
    // The staged records and their digits' table entries are READ together (48 LDS reads in flight), then stored: one read, two
    // dependent table reads and the stores per record in turn were 32 LDS latencies in a row per thread.
    u32 dd[RS_IPT], sl[RS_IPT]; // digit and slot inside the bucket of local slot i * RS_THREADS + tid (sl = ~0: nothing to store)
    {
        u64 ks_[RS_IPT];
#pragma unroll
        for (int i = 0; i < RS_IPT; i++) ks_[i] = stage[(u32)i * RS_THREADS + tid]; // (slots behind nvalid: stale, masked below)
#pragma unroll
        for (int i = 0; i < RS_IPT; i++) {
            const u32 p = (u32)i * RS_THREADS + tid;
            dd[i] = (ks_join_prefix(ks_[i], pfxK) >> shift) & (n_hi - 1u);
            sl[i] = gbase[dd[i]] + (p - dstart[dd[i]]);
        }
#pragma unroll
        for (int i = 0; i < RS_IPT; i++) {
            const u32 p = (u32)i * RS_THREADS + tid;
            if (!(p < nvalid && sl[i] < bcap)) sl[i] = 0xffffffffu;
            if (sl[i] != 0xffffffffu) {
                const u64 k = ks_[i];
                const u64 g = (u64)KS_BSLOT(dd[i], region, n_hi) * bcap + sl[i];
                if (V16 == 2) {
                    const u32 v = vo[V16 == 2 ? i : 0];
                    kout[g] = (k & ~(0xffULL << 56)) | ((u64)(v & 0xffu) << 56);
                    ((u8 *)vout)[g] = (u8)(v >> 8);
                } else {
                    kout[g] = k;
                }
            }
        }
    }
    if (V16 == 2) return;
    __syncthreads();
    u32 *vstage = (u32 *)stage;
    if (full) {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) vstage[rank[r]] = val[r];
    } else {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++)
            if (rank[r] != 0xffffffffu) vstage[rank[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_IPT; i++)
        if (sl[i] != 0xffffffffu) {
            const u64 g = (u64)KS_BSLOT(dd[i], region, n_hi) * bcap + sl[i];
            if (V16) ((u16 *)vout)[g] = (u16)vstage[(u32)i * RS_THREADS + tid];
            else vout[g] = vstage[(u32)i * RS_THREADS + tid];
        }
}

// ---------------------------------------------------------------------------------------------
// Index build: (hash, abund << 32 | tid) postings -> dense arrays sorted by (hash, tid) in THREE passes over the data
// instead of the eight of a 64-bit LSD sort: two fixed-capacity partition passes on the bits of a SORT prefix
// (floor(hash * 2^S / (max_hash + 1)), S <= 18, ~1100 postings per sort bucket) and one pass that sorts every
// bucket inside LDS.  Murmur output is uniform, so fixed capacities hold; a pass that would overflow raises a flag
// and the host falls back to the LSD sort (skewed inputs: thousands of copies of one protein).
// ---------------------------------------------------------------------------------------------
// One partition pass, up to 512 digits: record -> bucket (region * rmul + digit), digit = (prefix >> shift) & dmask,
// region = the input segment of the tile (0 for a dense input).  Ranks inside (tile, digit) are LDS atomic returns
// (order inside a bucket does not matter to the bucket sort); one global atomic per (tile, digit) reserves the slice.
__global__ __launch_bounds__(RS_THREADS, RS_MINW) void k_part_scatter64(const u64 *kin, const u64 *vin, u64 *kout, u64 *vout, u64 n,
                                                                        const u32 *seg_len, u64 seg_cap, u32 tiles_per_seg,
                                                                        int shift, u32 dmask, u32 pfxK, u32 rmul, u32 *bcur,
                                                                        u64 bcap, u32 *overflow) {
    __shared__ u32 cnt[512];
    __shared__ u32 dstart[512];
    __shared__ u32 gbase[512];
    __shared__ u32 scan_smem[RS_WAVES + 1];
    __shared__ __attribute__((aligned(16))) u64 stage[RS_TILE];
    static_assert(RS_THREADS == 512, "one thread per digit");
    const u32 tid = threadIdx.x;
    u64 tile_base;
    u32 nvalid;
    // second pass (segmented input): all tiles of a region go to one XCD, so the slices that meet in a bucket's cache
    // lines are assembled in one L2 (2.99 -> 2.44 ms); the first pass reserves slices in arrival order: nothing to gain
    const u32 bid = seg_len ? ks_xcd_block() : blockIdx.x;
    rs_tile_geom(bid, n, seg_len, seg_cap, tiles_per_seg, tile_base, nvalid);
    if (nvalid == 0) return;
    const u32 region = seg_len ? bid / tiles_per_seg : 0u;
    cnt[tid] = 0;
    u64 key[RS_IPT];
    u64 val[RS_IPT];
    u32 rank[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        const u32 li = (u32)r * RS_THREADS + tid;
        const bool valid = li < nvalid;
        key[r] = valid ? kin[tile_base + li] : 0ULL;
        val[r] = valid ? vin[tile_base + li] : 0ULL;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        const u32 li = (u32)r * RS_THREADS + tid;
        rank[r] = 0xffffffffu;
        if (li < nvalid) {
            const u32 d = (ks_join_prefix(key[r], pfxK) >> shift) & dmask;
            rank[r] = (d << 16) | atomicAdd(&cnt[d], 1u);
        }
    }
    __syncthreads();
    {
        const u32 c = cnt[tid];
        u32 total;
        const u32 ds = ks_block_excl_scan(c, scan_smem, &total);
        dstart[tid] = ds;
        u32 base = 0;
        if (c) {
            base = atomicAdd(&bcur[region * rmul + tid], c);
            if ((u64)base + c > bcap) atomicOr(overflow, 1u);
        }
        gbase[tid] = base;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; r++)
        if (rank[r] != 0xffffffffu) {
            rank[r] = dstart[rank[r] >> 16] + (rank[r] & 0xffffu);
            stage[rank[r]] = key[r];
        }
    __syncthreads();
    u64 gdst[RS_IPT];
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const u32 p = (u32)i * RS_THREADS + tid;
        gdst[i] = ~0ULL;
        if (p < nvalid) {
            const u64 k = stage[p];
            const u32 d = (ks_join_prefix(k, pfxK) >> shift) & dmask;
            const u64 slot = (u64)gbase[d] + (p - dstart[d]);
            if (slot < bcap) {
                gdst[i] = (u64)(region * rmul + d) * bcap + slot;
                kout[gdst[i]] = k;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; r++)
        if (rank[r] != 0xffffffffu) stage[rank[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_IPT; i++)
        if (gdst[i] != ~0ULL) vout[gdst[i]] = stage[(u32)i * RS_THREADS + tid];
}

// Sort one bucket (<= BS_CAP postings) inside LDS and write it to its place in the dense index.  Hashes are uniform
// inside a bucket too, so posting (h, tid) goes to sub-bucket floor(frac(h) * n) (frac = position of h inside the
// bucket's hash range, 32 bits; monotone in h) — about one posting per sub-bucket — and its final rank is the
// sub-bucket's start plus the number of smaller (hash, tid, arrival) triples in it.  O(n) LDS work, no comparison sort.
#define BS_THREADS 256
#define BS_E 8
#define BS_CAP (BS_THREADS * BS_E)
__global__ __launch_bounds__(BS_THREADS) void k_bucket_sort(const u64 *bkeys, const u64 *bvals, const u32 *bcount, u64 bcap,
                                                            const u64 *dense_start, u32 pfxK, u64 *okeys, u32 *otids,
                                                            u32 *oabunds, u32 *max_abund) {
    __shared__ u64 tk[BS_CAP];
    __shared__ u32 tt[BS_CAP];
    __shared__ u32 cnt[BS_CAP + 1];
    __shared__ u32 scan_smem[BS_THREADS / 64 + 1];
    const u32 tid = threadIdx.x, b = blockIdx.x;
    const u32 n = bcount[b];
    if (n == 0) return;
    const u64 src = (u64)b * bcap, dst = dense_start[b];
    u64 key[BS_E], val[BS_E];
    u32 so[BS_E]; // sub-bucket << 12 | arrival slot (n <= 2048: 11 bits each)
#pragma unroll
    for (int e = 0; e < BS_E; e++) cnt[(u32)e * BS_THREADS + tid] = 0;
    if (tid == 0) cnt[BS_CAP] = 0;
#pragma unroll
    for (int e = 0; e < BS_E; e++) {
        const u32 i = (u32)e * BS_THREADS + tid;
        key[e] = i < n ? bkeys[src + i] : 0ULL;
        val[e] = i < n ? bvals[src + i] : 0ULL;
    }
    __syncthreads();
    u32 amax = 0;
#pragma unroll
    for (int e = 0; e < BS_E; e++) {
        const u32 i = (u32)e * BS_THREADS + tid;
        so[e] = 0xffffffffu;
        if (i < n) {
            const u32 frac = (u32)((u64)(u32)(key[e] >> 32) * pfxK); // low word of the product whose high word is the bucket
            const u32 sb = __umulhi(frac, n);
            so[e] = (sb << 12) | atomicAdd(&cnt[sb], 1u);
            const u32 a = (u32)(val[e] >> 32);
            amax = a > amax ? a : amax;
        }
    }
    __syncthreads();
    { // counts -> starts (8 consecutive sub-buckets per thread)
        u32 c[BS_E], s = 0;
#pragma unroll
        for (int e = 0; e < BS_E; e++) { c[e] = cnt[tid * BS_E + e]; s += c[e]; }
        u32 total;
        u32 ex = ks_block_excl_scan(s, scan_smem, &total);
#pragma unroll
        for (int e = 0; e < BS_E; e++) { cnt[tid * BS_E + e] = ex; ex += c[e]; }
        if (tid == BS_THREADS - 1) cnt[BS_CAP] = total;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < BS_E; e++)
        if (so[e] != 0xffffffffu) {
            const u32 p = cnt[so[e] >> 12] + (so[e] & 0xfffu);
            tk[p] = key[e];
            tt[p] = (u32)val[e];
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < BS_E; e++)
        if (so[e] != 0xffffffffu) {
            const u32 sb = so[e] >> 12, o = so[e] & 0xfffu;
            const u32 s0 = cnt[sb], c = cnt[sb + 1] - s0; // (sub-buckets >= n are empty: their start is the total)
            const u32 t = (u32)val[e];
            u32 less = 0;
            for (u32 j = 0; j < c; j++) {
                const u64 kj = tk[s0 + j];
                const u32 tj = tt[s0 + j];
                less += (kj < key[e]) || (kj == key[e] && (tj < t || (tj == t && j < o)));
            }
            const u64 p = dst + s0 + less;
            okeys[p] = key[e];
            otids[p] = t;
            oabunds[p] = (u32)(val[e] >> 32);
        }
    for (int d = 32; d > 0; d >>= 1) {
        const u32 o = __shfl_down(amax, d, 64);
        amax = o > amax ? o : amax;
    }
    if ((tid & 63) == 0 && amax > *max_abund) atomicMax(max_abund, amax);
}

// returns KS_OK with *overflowed = 1 when a fixed capacity did not hold (nothing usable was written)
int ks_index_sort_partitioned(ks_ctx *ctx, const u64 *keys_in, const u64 *vals_in, u64 n, u64 max_hash, u64 *okeys, u32 *otids,
                              u32 *oabunds, u32 *d_max_abund, int *overflowed) {
    *overflowed = 0;
    // sort-prefix bits: ~1100 postings per bucket, at most 18 (two 9-bit partition passes)
    int S = 0;
    while (S < 18 && (n >> S) > 1100) S++;
    const int lo = S < 9 ? S : 9, hi = S - lo;
    const u32 n_buckets = 1u << S;
    const u64 per = n >> S;
    u64 bcap = per + per / 4 + 64;
    { u64 r = 8; while (r * r < per * 64) r++; bcap += r; } // + 8 sigma of a Poisson bucket
    if (bcap > BS_CAP) bcap = BS_CAP;
    const u32 pfxK = ks_join_prefix_mul(S, max_hash); // hi32(hash) * K: high word = sort bucket, low word = place inside it
    u64 *ak = nullptr, *av = nullptr, *bk = nullptr, *bv = nullptr, *dpos = nullptr;
    u32 *acur = nullptr, *bcur = nullptr, *oflow = nullptr;
    int st = KS_OK;
    const u32 nA = 1u << hi;
    u64 capA = 0;
#define PS_CHECK(x) do { st = (x); if (st != KS_OK) goto done; } while (0)
    PS_CHECK(ks_alloc(ctx, &bcur, (size_t)n_buckets + 1));
    PS_CHECK(ks_alloc(ctx, &oflow, 1));
    PS_CHECK(ks_alloc(ctx, &dpos, (size_t)n_buckets + 1));
    (void)hipMemsetAsync(bcur, 0, ((size_t)n_buckets + 1) * sizeof(u32), ctx->stream);
    (void)hipMemsetAsync(oflow, 0, sizeof(u32), ctx->stream);
    if (S > 0) {
        PS_CHECK(ks_alloc(ctx, &bk, (size_t)n_buckets * bcap));
        PS_CHECK(ks_alloc(ctx, &bv, (size_t)n_buckets * bcap));
    }
    if (hi > 0) { // pass A: dense input -> 2^hi regions on the high bits of the sort prefix
        const u64 perA = n >> hi;
        capA = perA + perA / 8 + 8192;
        capA = (capA + RS_TILE - 1) / RS_TILE * RS_TILE;
        PS_CHECK(ks_alloc(ctx, &ak, (size_t)nA * capA));
        PS_CHECK(ks_alloc(ctx, &av, (size_t)nA * capA));
        PS_CHECK(ks_alloc(ctx, &acur, (size_t)nA));
        (void)hipMemsetAsync(acur, 0, (size_t)nA * sizeof(u32), ctx->stream);
        const u32 nblocks = (u32)((n + RS_TILE - 1) / RS_TILE);
        ks_timer_begin(ctx, "index_part_a");
        hipLaunchKernelGGL(k_part_scatter64, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, keys_in, vals_in, ak, av, n,
                           (const u32 *)nullptr, (u64)0, 0u, lo, nA - 1, pfxK, 0u, acur, capA, oflow);
        ks_timer_end(ctx);
        // pass B: regions -> sort buckets (region << lo | low digit)
        const u32 tps = (u32)(capA / RS_TILE);
        ks_timer_begin(ctx, "index_part_b");
        hipLaunchKernelGGL(k_part_scatter64, dim3(nA * tps), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)ak, (const u64 *)av, bk, bv,
                           (u64)0, (const u32 *)acur, capA, tps, 0, (1u << lo) - 1, pfxK, 1u << lo, bcur, bcap, oflow);
        ks_timer_end(ctx);
    } else if (S > 0) { // one pass is enough
        const u32 nblocks = (u32)((n + RS_TILE - 1) / RS_TILE);
        ks_timer_begin(ctx, "index_part_b");
        hipLaunchKernelGGL(k_part_scatter64, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, keys_in, vals_in, bk, bv, n,
                           (const u32 *)nullptr, (u64)0, 0u, 0, n_buckets - 1, pfxK, 0u, bcur, bcap, oflow);
        ks_timer_end(ctx);
    } else { // a single bucket: the input is the bucket
        if (n > BS_CAP) { *overflowed = 1; goto done; }
        const u32 cnt1 = (u32)n;
        (void)hipMemcpyAsync(bcur, &cnt1, sizeof(u32), hipMemcpyHostToDevice, ctx->stream);
    }
    if (hipGetLastError() != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "index partition launch failed"); goto done; }
    // overflow flags decide before anything is sorted
    if (hipMemcpyAsync(ctx->h_pin + 41, oflow, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { st = ks_fail(ctx, KS_ERR_HIP, "index partition status read failed"); goto done; }
    if (*(u32 *)(ctx->h_pin + 41) != 0) { *overflowed = 1; goto done; }
    PS_CHECK(ks_scan_u32_to_u64(ctx, bcur, dpos, n_buckets));
    ks_timer_begin(ctx, "index_bucket_sort");
    hipLaunchKernelGGL(k_bucket_sort, dim3(n_buckets), dim3(BS_THREADS), 0, ctx->stream, S ? (const u64 *)bk : keys_in,
                       S ? (const u64 *)bv : vals_in, (const u32 *)bcur, S ? bcap : (u64)0, (const u64 *)dpos, pfxK, okeys, otids, oabunds,
                       d_max_abund);
    ks_timer_end(ctx);
    if (hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "index bucket sort launch failed");
done:
    ks_pool_free(ctx, ak); ks_pool_free(ctx, av); ks_pool_free(ctx, bk); ks_pool_free(ctx, bv); ks_pool_free(ctx, dpos);
    ks_pool_free(ctx, acur); ks_pool_free(ctx, bcur); ks_pool_free(ctx, oflow);
    return st;
#undef PS_CHECK
}

static const char *const rs_tag_names[3] = {"index", "qpart", "pairs"};

template <typename V, int TAG>
static int radix_sort_tagged(ks_ctx *ctx, const u64 *keys_in, const V *vals_in, u64 *ka, V *va, u64 *kb, V *vb, u64 n,
                             const int *shifts, int n_shifts, u64 **keys_out, V **vals_out, const ks_rs_segments *seg,
                             u32 pfxK) {
    *keys_out = (u64 *)keys_in;
    *vals_out = (V *)vals_in;
    // nothing to sort: the caller gets its own arrays back (and must not take them for one of the scratch pairs).  A SEGMENTED
    // input of one record still goes through its passes: the record sits at its region's base, not at index 0.
    if (n == 0 || n_shifts <= 0 || (n == 1 && !seg)) return KS_OK;
    if (n >= 0xffffffffULL) return ks_fail(ctx, KS_ERR_CAPACITY, "radix sort: %llu records exceed the 32-bit offset range", (unsigned long long)n);
    const u32 nblocks_dense = (u32)((n + RS_TILE - 1) / RS_TILE);
    const u32 tiles_per_seg = seg ? (u32)((seg->cap + RS_TILE - 1) / RS_TILE) : 0;
    const u32 nblocks_seg = seg ? seg->regions * tiles_per_seg : 0;
    u32 *hist = nullptr;
    KS_TRY(ks_alloc(ctx, &hist, (size_t)256 * (nblocks_seg > nblocks_dense ? nblocks_seg : nblocks_dense)));
    const std::string nm_hist = std::string("radix_hist.") + rs_tag_names[TAG], nm_scat = std::string("radix_scatter.") + rs_tag_names[TAG];
    const u64 *kin = keys_in;
    const V *vin = vals_in;
    // the caller's input is only ever read; passes ping-pong between the two scratch pairs
    bool to_a = (keys_in != ka);
    int st = KS_OK;
    for (int i = 0; i < n_shifts && st == KS_OK; i++) {
        u64 *kout = to_a ? ka : kb;
        V *vout = to_a ? va : vb;
        // only the first pass can read a segmented input; its output (and every later pass) is dense
        const bool segmented = seg && i == 0;
        const u32 nblocks = segmented ? nblocks_seg : nblocks_dense;
        const u32 *s_len = segmented ? seg->len : nullptr;
        const u64 s_cap = segmented ? seg->cap : 0;
        ks_timer_begin(ctx, nm_hist.c_str());
        hipLaunchKernelGGL((k_radix_hist<TAG>), dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, kin, hist, n, shifts[i], nblocks,
                           s_len, s_cap, tiles_per_seg, pfxK);
        ks_timer_end(ctx);
        st = ks_scan_u32_inplace(ctx, hist, (u64)256 * nblocks, nullptr);
        if (st != KS_OK) break;
        ks_timer_begin(ctx, nm_scat.c_str());
        hipLaunchKernelGGL((k_radix_scatter<V, TAG>), dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, kin, vin, kout, vout,
                           (const u32 *)hist, n, shifts[i], nblocks, s_len, s_cap, tiles_per_seg, pfxK);
        ks_timer_end(ctx);
        if (hipGetLastError() != hipSuccess) st = ks_fail(ctx, KS_ERR_HIP, "radix sort launch failed");
        kin = kout;
        vin = vout;
        to_a = !to_a;
    }
    ks_pool_free(ctx, hist);
    *keys_out = (u64 *)kin;
    *vals_out = (V *)vin;
    return st;
}

int ks_radix_sort_u32(ks_ctx *ctx, int tag, const u64 *keys_in, const u32 *vals_in, u64 *ka, u32 *va, u64 *kb, u32 *vb, u64 n,
                      const int *shifts, int n_shifts, u64 **keys_out, u32 **vals_out, const ks_rs_segments *seg, u32 pfxK) {
    if (tag == KS_SORT_QPART)
        return radix_sort_tagged<u32, KS_SORT_QPART>(ctx, keys_in, vals_in, ka, va, kb, vb, n, shifts, n_shifts, keys_out, vals_out, seg, pfxK);
    return radix_sort_tagged<u32, KS_SORT_PAIRS>(ctx, keys_in, vals_in, ka, va, kb, vb, n, shifts, n_shifts, keys_out, vals_out, seg, pfxK);
}
int ks_radix_sort_keys(ks_ctx *ctx, int tag, const u64 *keys_in, u64 *ka, u64 *kb, u64 n, const int *shifts, int n_shifts,
                       u64 **keys_out) {
    (void)tag;
    ks_noval *vo = nullptr;
    return radix_sort_tagged<ks_noval, KS_SORT_PAIRS>(ctx, keys_in, (const ks_noval *)nullptr, ka, (ks_noval *)nullptr, kb,
                                                      (ks_noval *)nullptr, n, shifts, n_shifts, keys_out, &vo, nullptr, 0u);
}
int ks_radix_sort_u64(ks_ctx *ctx, int tag, const u64 *keys_in, const u64 *vals_in, u64 *ka, u64 *va, u64 *kb, u64 *vb, u64 n,
                      const int *shifts, int n_shifts, u64 **keys_out, u64 **vals_out) {
    (void)tag;
    return radix_sort_tagged<u64, KS_SORT_INDEX>(ctx, keys_in, vals_in, ka, va, kb, vb, n, shifts, n_shifts, keys_out, vals_out, nullptr, 0u);
}

// One pass, no histogram: segmented postings (regions by the low digit) -> 2^pbits fixed-capacity buckets.
int ks_bucket_scatter_u32(ks_ctx *ctx, const u64 *keys_in, const u32 *vals_in, const ks_rs_segments *seg, int shift,
                          u32 pfxK, u64 *bkeys, u32 *bvals, u32 *bcur, u32 bcap, unsigned long long *status, u32 n_hi, int vfmt) {
    const u32 tiles_per_seg = (u32)((seg->cap + RS_TILE - 1) / RS_TILE);
    const u32 nblocks = seg->regions * tiles_per_seg;
    ks_timer_begin(ctx, "bucket_scatter");
#define BKS_LAUNCH(V_) hipLaunchKernelGGL((k_bucket_scatter<V_>), dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, keys_in, vals_in, bkeys, bvals, \
                                          shift, seg->len, seg->cap, tiles_per_seg, bcur, bcap, status, pfxK, n_hi, seg->sub_shift)
    if (vfmt == 2) BKS_LAUNCH(2); // 16-bit values in, key byte + 8-bit values out (9-byte postings)
    else if (vfmt == 1) BKS_LAUNCH(1);
    else BKS_LAUNCH(0);
#undef BKS_LAUNCH
    ks_timer_end(ctx);
    KS_HIP(ctx, hipGetLastError());
    return KS_OK;
}
