// ks_device.h — device-side building blocks: MurmurHash3_x64_128.h1, wave/block scans.
// Written for gfx950: 64-lane waves, DPP/shuffle wave scans, LDS cross-wave combine.
#pragma once
#include "ks_common.h"

#define KS_DEV __device__ __forceinline__

// rotl by a compile-time count, as the two v_alignbit_b32 it is (the generic shift / or form costs four instructions here)
KS_DEV u64 ks_rotl64(u64 x, int r) {
#ifndef KS_NO_OPAQUE
    if (__builtin_constant_p(r) && r > 0 && r < 64 && r != 32) {
        const u32 lo = (u32)x, hi = (u32)(x >> 32);
        const u32 s = (u32)(r < 32 ? 32 - r : 64 - r);
        const u32 a = __builtin_amdgcn_alignbit(lo, hi, s), b = __builtin_amdgcn_alignbit(hi, lo, s);
        return r < 32 ? (((u64)b << 32) | a) : (((u64)a << 32) | b);
    }
#endif
    return (x << r) | (x >> (64 - r));
}

KS_DEV u64 ks_fmix64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

#define KS_C1 0x87c37b91114253d5ULL
#define KS_C2 0x4cf5ad432745937fULL

// A product that is rotated next: the compiler folds the rotation's left shift into the multiplication — rotl(x * c, 31) becomes
// (x * (c << 31)) | ((x * c) >> 33), five 32-bit multiplies (half-rate instructions on gfx950: tools/gpu/valu_rates.hip) where
// three and two v_alignbit do.  An empty asm statement hides the product from that rewrite; it costs no instruction.
#ifndef KS_NO_OPAQUE
#define KS_OPAQUE64(x) asm("" : "+v"(x))
#else
#define KS_OPAQUE64(x) do { } while (0)
#endif

// Streaming state of MurmurHash3_x64_128 (both lanes seeded), as sourmash::_hash_murmur uses it
// (reference call site src/rust/index.rs:766; add_protein via src/rust/signature.rs:274).
struct ks_murmur {
    u64 h1, h2;
    KS_DEV void init(u64 seed) { h1 = seed; h2 = seed; }
    KS_DEV void block(u64 k1, u64 k2) {
        k1 *= KS_C1; KS_OPAQUE64(k1); k1 = ks_rotl64(k1, 31); k1 *= KS_C2; h1 ^= k1;
        h1 = ks_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
        k2 *= KS_C2; KS_OPAQUE64(k2); k2 = ks_rotl64(k2, 33); k2 *= KS_C1; h2 ^= k2;
        h2 = ks_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    }
    // tail: k1 = bytes 0..7 (already masked), k2 = bytes 8..14 (already masked); t = len & 15
    KS_DEV void tail(u64 k1, u64 k2, u32 t) {
        if (t > 8) {
#ifndef KS_NO_OPAQUE
            if (__builtin_constant_p(t) && t <= 12) { // a tail of <= 4 bytes behind the first word: 32 x 64 bits = one v_mad_u64_u32, one v_mul_lo_u32, one add
                const u32 x = (u32)k2;
                u64 p = (u64)x * (u32)KS_C2;
                u32 xh = x * (u32)(KS_C2 >> 32);
                asm("" : "+v"(xh)); // (keeps the compiler from re-fusing the two into a chain of two v_mad_u64_u32 with moves between them)
                k2 = p + ((u64)xh << 32);
            } else
#endif
            k2 *= KS_C2;
            KS_OPAQUE64(k2); k2 = ks_rotl64(k2, 33); k2 *= KS_C1; h2 ^= k2;
        }
        if (t > 0) { k1 *= KS_C1; KS_OPAQUE64(k1); k1 = ks_rotl64(k1, 31); k1 *= KS_C2; h1 ^= k1; }
    }
    KS_DEV u64 finish(u64 len) {
        h1 ^= len; h2 ^= len;
        h1 += h2; h2 += h1;
        h1 = ks_fmix64(h1); h2 = ks_fmix64(h2);
        return h1 + h2;
    }
};

// bytes [8*I, 8*I+8) of the 16-byte little-endian pair (a, b): a funnel shift with a
// compile-time byte count, which hipcc lowers to v_alignbit/v_perm.
template <int I>
KS_DEV u64 ks_funnel(u64 a, u64 b) {
    if constexpr (I == 0) return a;
#ifndef KS_NO_OPAQUE
    else { // on the 32-bit words: two v_alignbyte_b32 at most (the 64-bit shift / or form takes three or four instructions)
        const u32 w0 = (u32)a, w1 = (u32)(a >> 32), w2 = (u32)b, w3 = (u32)(b >> 32);
        u32 lo, hi;
        if constexpr (I < 4) { lo = __builtin_amdgcn_alignbyte(w1, w0, (u32)I); hi = __builtin_amdgcn_alignbyte(w2, w1, (u32)I); }
        else if constexpr (I == 4) { lo = w1; hi = w2; }
        else { lo = __builtin_amdgcn_alignbyte(w2, w1, (u32)(I - 4)); hi = __builtin_amdgcn_alignbyte(w3, w2, (u32)(I - 4)); }
        return ((u64)hi << 32) | lo;
    }
#else
    else return (a >> (8 * I)) | (b << (64 - 8 * I));
#endif
}

KS_DEV u64 ks_funnel_rt(u64 a, u64 b, u32 byte_shift) {
    u32 s = byte_shift * 8;
    return s ? ((a >> s) | (b << (64 - s))) : a;
}

KS_DEV u64 ks_mask_bytes(u32 n) { return n >= 8 ? ~0ULL : ((1ULL << (8 * n)) - 1ULL); }

// Workgroups of a launch go round-robin to the 8 XCDs of the MI355X (each with its own L2).  Kernels whose neighbouring
// tiles touch the same cache lines (partition scatters: a tile's digit runs end where the next tile's begin) take their
// tile id from this instead of blockIdx.x: XCD x works through a contiguous eighth of the tiles, so the lines two tiles
// share are assembled in ONE L2 (measured on the bucket scatter: 1.58 -> 1.29 ms, and the join that reads its output
// 2.17 -> 1.80 ms).  A bijection on [0, gridDim.x) for any grid size: XCD x = b & 7 owns g/8 tiles (one more for the
// first g%8 XCDs), its j-th workgroup (j = b >> 3) takes the j-th of them.
KS_DEV u32 ks_xcd_block() {
    const u32 b = blockIdx.x, g = gridDim.x, x = b & 7u, q = g >> 3, rem = g & 7u;
    return x * q + (x < rem ? x : rem) + (b >> 3);
}

// Decoupled look-back spins are bounded by TIME — the constant-rate wall clock (100 MHz on gfx950), read only every 1024
// polls — not by an iteration count: a predecessor that is merely slow (a shared GPU, several ranks rehearsing on one
// device) must not look like a protocol violation.  ~2 s.
#define KS_SPIN_TICKS 200000000LL
KS_DEV bool ks_spin_expired(long long t0, u32 &polls) {
    return ((++polls) & 1023u) == 0 && wall_clock64() - t0 > KS_SPIN_TICKS;
}

// ---- wave / block exclusive scans (u32) ----
// Inclusive scan over the 64 lanes of a wave with DPP moves only (no LDS crossbar round trips: __shfl_up lowers to
// ds_bpermute_b32): four row_shr steps scan each row of 16, row_bcast:15 carries row 0 -> 1 and row 2 -> 3, row_bcast:31
// carries the first half into the second.  Lanes without a source add 0.
KS_DEV u32 ks_wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
// Sum of a 64-bit value over the 64 lanes of a wave (uniform result), DPP moves only: the look-backs' reduce used to be
// `v += __shfl_xor(v, d)` — twelve ds_bpermute in a row on the critical path of every tile.
KS_DEV u64 ks_wave_sum64(u64 v) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
#define KS_SUM64_STEP(CTRL, RMASK, BC) do { \
        const u32 ol_ = (u32)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, RMASK, 0xf, BC); \
        const u32 oh_ = (u32)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, RMASK, 0xf, BC); \
        const u64 s_ = (((u64)hi << 32) | lo) + (((u64)oh_ << 32) | ol_); \
        lo = (u32)s_; hi = (u32)(s_ >> 32); } while (0)
    KS_SUM64_STEP(0x111, 0xf, true);  // row_shr:1
    KS_SUM64_STEP(0x112, 0xf, true);  // row_shr:2
    KS_SUM64_STEP(0x114, 0xf, true);  // row_shr:4
    KS_SUM64_STEP(0x118, 0xf, true);  // row_shr:8
    KS_SUM64_STEP(0x142, 0xa, false); // row_bcast:15 into rows 1 and 3
    KS_SUM64_STEP(0x143, 0xc, false); // row_bcast:31 into rows 2 and 3
#undef KS_SUM64_STEP
    return ((u64)(u32)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (u32)__builtin_amdgcn_readlane((int)lo, 63);
}
// value of the lane below (lane 0: 0)
KS_DEV u32 ks_lane_below(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); } // wave_shr:1

// Block-wide exclusive scan; `smem` must hold (blockDim.x/64 + 1) u32.  Returns the exclusive
// prefix of v; *total receives the block sum.  Contains three __syncthreads().
KS_DEV u32 ks_block_excl_scan(u32 v, u32 *smem, u32 *total) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32 incl = ks_wave_incl_scan(v);
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        u32 w = lane < nw ? smem[lane] : 0;
        u32 wi = ks_wave_incl_scan(w);
        if (lane < nw) smem[lane] = wi - w;
        if (lane == nw - 1) smem[nw] = wi;
    }
    __syncthreads();
    u32 base = smem[wave];
    *total = smem[nw];
    __syncthreads(); // smem may be reused by the caller right away
    return base + incl - v;
}

KS_DEV u64 ks_ballot(bool p) { return __ballot(p); }
KS_DEV u32 ks_lane_lt_count(u64 mask) {
    // number of set bits of mask in lanes below this one
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0));
}

// Stable rank of an 8-bit digit inside one wave ("match-any" by eight ballots): returns d << 16 | (records of digit d the wave
// counted before this round + lanes below mine that hold d), and adds the round's count to the wave's counter wc[d].
// Per bit: the lane's bit as a mask (v_bfe_i32: 0 / -1), one ballot, and peers &= ~(ballot ^ mask) on each half — a three-input
// boolean, one v_bitop3_b32 (the generic select form `bit ? m : ~m` compiled to nine vector instructions per bit).  The lowest
// lane of a group of peers (no peer below it) publishes the new count; EVERY lane reads the old one first — the LDS operations
// of a wave execute in order, so no lane-to-lane exchange (ds_bpermute) is needed for it.
KS_DEV u32 ks_match8_rank(u32 d, u32 *wc) {
    u32 plo = ~0u, phi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const u32 bm = (u32)__builtin_amdgcn_sbfe((int)d, (u32)b, 1u);
        const u64 m = __ballot(bm != 0u);
        plo &= ~((u32)m ^ bm);
        phi &= ~((u32)(m >> 32) ^ bm);
    }
    const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0));
    const u32 pre = wc[d];
    if (below == 0) wc[d] = pre + (u32)__popc(plo) + (u32)__popc(phi);
    return (d << 16) | (pre + below);
}

// Join prefix of a kept hash: floor(h * 2^pbits / (max_hash + 1)) computed on the top 32 bits — uniform over
// [0, 2^pbits) for every `scaled` (kept hashes only span [0, max_hash], so plain top bits are NOT uniform for scaled > 1).
// K = floor(2^(pbits+32) / ((max_hash >> 32) + 1)); for scaled = 1 this is exactly h >> (64 - pbits).  Monotone in h.
KS_DEV u32 ks_join_prefix(u64 h, u32 K) { return __umulhi((u32)(h >> 32), K); }
// digit of a radix pass: plain key bits (K == 0) or bits [shift, shift + 8) of the join prefix
KS_DEV u32 ks_rs_digit(u64 key, int shift, u32 K) {
    return K ? ((ks_join_prefix(key, K) >> shift) & 255u) : ((u32)(key >> shift) & 255u);
}
