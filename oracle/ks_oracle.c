/*
 * ks_oracle.c — CPU restatement of the kmerseek sketch-and-search hot path.
 * TEST INFRASTRUCTURE ONLY — see ks_oracle.h for the rules and the parity status (PINNED).
 */
#include "ks_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * MurmurHash3_x64_128 (public-domain algorithm by A. Appleby), first output word only.
 * Reference call sites: sourmash::_hash_murmur at src/rust/index.rs:766 and, inside
 * sourmash 0.20.0, KmerMinHash::add_protein reached from src/rust/signature.rs:274.
 * The murmurhash3 0.0.5 crate seeds BOTH lanes with the u64 seed and reads blocks
 * little-endian.  KATs: tests/golden/hash_kats.json (index.rs:1084-1103,1187-1205,1309-1326).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p, size_t n) {
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

uint64_t kso_hash_murmur(const uint8_t *data, size_t len, uint64_t seed) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    size_t nblocks = len / 16;
    for (size_t i = 0; i < nblocks; i++) {
        uint64_t k1 = load_le64(data + 16 * i, 8);
        uint64_t k2 = load_le64(data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
    }
    const uint8_t *tail = data + 16 * nblocks;
    size_t t = len & 15;
    if (t > 8) {
        uint64_t k2 = load_le64(tail + 8, t - 8);
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (t > 0) {
        uint64_t k1 = load_le64(tail, t > 8 ? 8 : t);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

/* sourmash max_hash_for_scaled: 0 -> 0, 1 -> u64::MAX, else (u64::MAX as f64 / scaled as f64) as u64.
 * (u64::MAX as f64) rounds to 2^64; the final cast saturates.  Pinned by the max_hash field of
 * every golden signature (3689348814741910528 for scaled=5). */
uint64_t kso_max_hash(uint32_t scaled) {
    if (scaled == 0) return 0;
    if (scaled == 1) return UINT64_MAX;
    double v = 18446744073709551616.0 / (double)scaled;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

/* sourmash::encodings::aa_to_dayhoff / aa_to_hp (selected at src/rust/encoding.rs:43-53).
 * Pinned by LIVINGALIVE -> eeeecbbeeec / hhhhphhhhhp (encoding.rs:195,209) and the `encoded`
 * column of the two golden k-mer tables. */
uint8_t kso_encode_residue(uint8_t aa, int moltype) {
    if (moltype == KSO_PROTEIN) return aa;
    if (moltype == KSO_DAYHOFF) {
        switch (aa) {
        case 'C': return 'a';
        case 'A': case 'G': case 'P': case 'S': case 'T': return 'b';
        case 'D': case 'E': case 'N': case 'Q': return 'c';
        case 'H': case 'K': case 'R': return 'd';
        case 'I': case 'L': case 'M': case 'V': return 'e';
        case 'F': case 'W': case 'Y': return 'f';
        default: return 'X';
        }
    }
    switch (aa) { /* hp */
    case 'A': case 'F': case 'G': case 'I': case 'L': case 'M': case 'P': case 'V': case 'W':
    case 'Y': return 'h';
    case 'N': case 'C': case 'S': case 'T': case 'D': case 'E': case 'R': case 'H': case 'K':
    case 'Q': return 'p';
    default: return 'X';
    }
}

void kso_encode_kmer(const uint8_t *kmer, size_t k, int moltype, uint8_t *out) {
    for (size_t i = 0; i < k; i++) out[i] = kso_encode_residue(kmer[i], moltype);
}

static inline uint8_t upper_ascii(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

/* KmerMinHash::add_protein as reached from src/rust/signature.rs:273-282.
 * sourmash's SeqToHashes upper-cases the sequence before hashing (unobservable on the Rust
 * path, which upper-cases / validates first: src/rust/index.rs:1000, aminoacid.rs:74-105).
 * The insert is the reference's own algorithm (binary search + Vec::insert memmove) so this
 * function doubles as the CPU-baseline "port". */
size_t kso_sketch_protein(const uint8_t *seq, size_t len, uint32_t k, uint32_t scaled,
                          int moltype, uint64_t seed, uint64_t *mins, uint64_t *abunds) {
    if (k == 0 || len < k) return 0; /* unpinned edge: sourmash may raise; Rust path loop is empty */
    const uint64_t max_hash = kso_max_hash(scaled);
    uint8_t buf[256];
    size_t n = 0;
    for (size_t i = 0; i + k <= len; i++) {
        for (uint32_t j = 0; j < k; j++) buf[j] = kso_encode_residue(upper_ascii(seq[i + j]), moltype);
        uint64_t h = kso_hash_murmur(buf, k, seed);
        if (h == 0) continue;        /* add_protein: Ok(0) => continue */
        if (h > max_hash) continue;  /* add_hash_with_abundance: hash > max_hash => return */
        size_t lo = 0, hi = n;       /* mins.binary_search(&hash) */
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (mins[mid] < h) lo = mid + 1; else hi = mid;
        }
        if (lo < n && mins[lo] == h) {
            abunds[lo] += 1;
        } else {
            memmove(mins + lo + 1, mins + lo, (n - lo) * sizeof(uint64_t));
            memmove(abunds + lo + 1, abunds + lo, (n - lo) * sizeof(uint64_t));
            mins[lo] = h;
            abunds[lo] = 1;
            n++;
        }
    }
    return n;
}

typedef struct {
    const uint8_t *residues; const uint64_t *seq_offsets; uint32_t s0, s1;
    uint32_t k, scaled; int moltype; uint64_t seed;
    uint64_t *sp_mins; uint64_t *sp_abunds; uint64_t *counts;
} sketch_job;

static void *sketch_worker(void *arg) {
    sketch_job *j = (sketch_job *)arg;
    for (uint32_t s = j->s0; s < j->s1; s++) {
        uint64_t b = j->seq_offsets[s], e = j->seq_offsets[s + 1];
        j->counts[s] = kso_sketch_protein(j->residues + b, (size_t)(e - b), j->k, j->scaled,
                                          j->moltype, j->seed, j->sp_mins + b, j->sp_abunds + b);
    }
    return NULL;
}

uint64_t kso_sketch_batch(const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_seqs,
                          uint32_t k, uint32_t scaled, int moltype, uint64_t seed,
                          uint64_t *out_offsets, uint64_t *out_mins, uint32_t *out_abunds,
                          int n_threads) {
    uint64_t total_res = n_seqs ? seq_offsets[n_seqs] : 0;
    uint64_t *sp_mins = (uint64_t *)malloc((total_res + 1) * sizeof(uint64_t));
    uint64_t *sp_abunds = (uint64_t *)malloc((total_res + 1) * sizeof(uint64_t));
    uint64_t *counts = (uint64_t *)calloc((size_t)n_seqs + 1, sizeof(uint64_t));
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_seqs) n_threads = n_seqs ? (int)n_seqs : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    sketch_job *jobs = (sketch_job *)malloc(sizeof(sketch_job) * (size_t)n_threads);
    /* balance by residue count: contiguous ranges with ~equal residues */
    uint32_t s = 0;
    for (int t = 0; t < n_threads; t++) {
        uint64_t target = total_res * (uint64_t)(t + 1) / (uint64_t)n_threads;
        uint32_t s1 = s;
        while (s1 < n_seqs && (seq_offsets[s1 + 1] <= target || t == n_threads - 1)) s1++;
        if (t == n_threads - 1) s1 = n_seqs;
        jobs[t] = (sketch_job){residues, seq_offsets, s, s1, k, scaled, moltype, seed,
                               sp_mins, sp_abunds, counts};
        s = s1;
    }
    if (n_threads == 1) {
        sketch_worker(&jobs[0]);
    } else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, sketch_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    uint64_t pos = 0;
    for (uint32_t q = 0; q < n_seqs; q++) {
        out_offsets[q] = pos;
        uint64_t b = seq_offsets[q];
        for (uint64_t i = 0; i < counts[q]; i++) {
            out_mins[pos + i] = sp_mins[b + i];
            out_abunds[pos + i] = (uint32_t)sp_abunds[b + i];
        }
        pos += counts[q];
    }
    out_offsets[n_seqs] = pos;
    free(sp_mins); free(sp_abunds); free(counts); free(th); free(jobs);
    return pos;
}

/* ProteomeIndex::process_kmers, src/rust/index.rs:749-786. */
size_t kso_kmer_positions(const uint8_t *seq, size_t len, uint32_t k, int moltype, uint64_t seed,
                          const uint64_t *mins, size_t n_mins, int faithful,
                          uint32_t *starts, uint64_t *hashes) {
    if (k == 0 || len < k) return 0;
    uint8_t buf[256];
    size_t n = 0;
    for (size_t i = 0; i + k <= len; i++) { /* 0..sequence.len().saturating_sub(ksize-1) */
        for (uint32_t j = 0; j < k; j++) buf[j] = kso_encode_residue(seq[i + j], moltype);
        uint64_t h = kso_hash_murmur(buf, k, seed);
        int found = 0;
        if (faithful) { /* hashvals.contains(&hashval): linear scan, index.rs:769 */
            for (size_t m = 0; m < n_mins; m++) if (mins[m] == h) { found = 1; break; }
        } else {
            size_t lo = 0, hi = n_mins;
            while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (mins[mid] < h) lo = mid + 1; else hi = mid; }
            found = (lo < n_mins && mins[lo] == h);
        }
        if (found) { starts[n] = (uint32_t)i; hashes[n] = h; n++; }
    }
    return n;
}

/* Batch form of process_kmers for the CPU baseline: the second pass create_protein_signature makes over every
 * record (src/rust/index.rs:737, 749-786) under the rayon loop of process_batch_parallel (:984-1016), with the
 * reference's linear `contains` scan.  Returns the number of (start, hash) rows found; the rows are not kept. */
typedef struct {
    const uint8_t *residues; const uint64_t *seq_offsets; const uint64_t *sk_off; const uint64_t *sk_mins;
    uint32_t s0, s1, k; int moltype; uint64_t seed; int faithful; uint64_t found;
} kmerpos_job;

static void *kmerpos_worker(void *arg) {
    kmerpos_job *j = (kmerpos_job *)arg;
    uint64_t max_len = 0;
    for (uint32_t s = j->s0; s < j->s1; s++) {
        uint64_t l = j->seq_offsets[s + 1] - j->seq_offsets[s];
        if (l > max_len) max_len = l;
    }
    uint32_t *starts = (uint32_t *)malloc((max_len + 1) * sizeof(uint32_t));
    uint64_t *hashes = (uint64_t *)malloc((max_len + 1) * sizeof(uint64_t));
    for (uint32_t s = j->s0; s < j->s1; s++) {
        uint64_t b = j->seq_offsets[s], e = j->seq_offsets[s + 1];
        j->found += kso_kmer_positions(j->residues + b, (size_t)(e - b), j->k, j->moltype, j->seed,
                                       j->sk_mins + j->sk_off[s], (size_t)(j->sk_off[s + 1] - j->sk_off[s]),
                                       j->faithful, starts, hashes);
    }
    free(starts); free(hashes);
    return NULL;
}

uint64_t kso_kmer_positions_batch(const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_seqs,
                                  uint32_t k, int moltype, uint64_t seed, const uint64_t *sk_offsets,
                                  const uint64_t *sk_mins, int faithful, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_seqs) n_threads = n_seqs ? (int)n_seqs : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    kmerpos_job *jobs = (kmerpos_job *)malloc(sizeof(kmerpos_job) * (size_t)n_threads);
    uint64_t total_res = n_seqs ? seq_offsets[n_seqs] : 0;
    uint32_t s = 0;
    for (int t = 0; t < n_threads; t++) {
        uint64_t target = total_res * (uint64_t)(t + 1) / (uint64_t)n_threads;
        uint32_t s1 = s;
        while (s1 < n_seqs && (seq_offsets[s1 + 1] <= target || t == n_threads - 1)) s1++;
        if (t == n_threads - 1) s1 = n_seqs;
        jobs[t] = (kmerpos_job){residues, seq_offsets, sk_offsets, sk_mins, s, s1, k, moltype, seed, faithful, 0};
        s = s1;
    }
    if (n_threads == 1) kmerpos_worker(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, kmerpos_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    uint64_t found = 0;
    for (int t = 0; t < n_threads; t++) found += jobs[t].found;
    free(th); free(jobs);
    return found;
}

/* AminoAcidAmbiguity::validate_and_resolve, src/rust/aminoacid.rs:74-105. */
int kso_validate_and_resolve(const uint8_t *seq, size_t len, const uint8_t *choices,
                             size_t n_choices, uint8_t *out, size_t *out_len, uint8_t *bad_char,
                             size_t *bad_pos) {
    static const char *standard = "ACDEFGHIKLMNPQRSTVWY"; /* aminoacid.rs:8-11 */
    static const char *special = "XUO*";                   /* aminoacid.rs:14 */
    size_t n = 0, amb = 0;
    for (size_t i = 0; i < len; i++) {
        uint8_t c = seq[i];
        if (c == '*') { out[n++] = c; break; } /* aminoacid.rs:79-83 */
        int ambiguous = (c == 'B' || c == 'Z' || c == 'J');
        int valid = c != 0 && (strchr(standard, c) || strchr(special, c) || ambiguous);
        if (!valid) { /* aminoacid.rs:85-87: position = result.len() + 1 */
            if (bad_char) *bad_char = c;
            if (bad_pos) *bad_pos = n + 1;
            *out_len = n;
            return 1;
        }
        if (ambiguous) { /* aminoacid.rs:32-36: B->{D,N}, Z->{E,Q}, J->{I,L} */
            int pick = (amb < n_choices && choices) ? (choices[amb] & 1) : 0;
            amb++;
            const char *cand = c == 'B' ? "DN" : c == 'Z' ? "EQ" : "IL";
            out[n++] = (uint8_t)cand[pick];
        } else {
            out[n++] = c;
        }
    }
    *out_len = n;
    return 0;
}

uint64_t kso_wrapping_sum(const uint64_t *mins, size_t n) {
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s += mins[i];
    return s;
}

/* manysearch pair overlap: sorted-merge intersection; weighted count uses TARGET abundances
 * (pinned by total_weighted_hashes in tests/test_search.py:33-39). */
uint64_t kso_intersect(const uint64_t *q, size_t nq, const uint64_t *t, const uint32_t *t_abund,
                       size_t nt, uint64_t *n_weighted, uint32_t *isect_abunds) {
    size_t i = 0, j = 0;
    uint64_t c = 0, w = 0;
    while (i < nq && j < nt) {
        if (q[i] < t[j]) i++;
        else if (q[i] > t[j]) j++;
        else {
            uint32_t a = t_abund ? t_abund[j] : 1;
            if (isect_abunds) isect_abunds[c] = a;
            w += a; c++; i++; j++;
        }
    }
    if (n_weighted) *n_weighted = w;
    return c;
}

typedef struct {
    const uint64_t *q_off, *q_mins, *t_off, *t_mins; const uint32_t *t_abund;
    uint32_t q0, q1, n_t;
    uint32_t *qid, *tid, *isect; uint64_t *nw; uint64_t count, cap; int fill;
} search_job;

static void *search_worker(void *arg) {
    search_job *j = (search_job *)arg;
    uint64_t c = 0;
    for (uint32_t q = j->q0; q < j->q1; q++) {
        const uint64_t *qm = j->q_mins + j->q_off[q];
        size_t nq = (size_t)(j->q_off[q + 1] - j->q_off[q]);
        if (nq == 0) continue;
        for (uint32_t t = 0; t < j->n_t; t++) {
            size_t nt = (size_t)(j->t_off[t + 1] - j->t_off[t]);
            if (nt == 0) continue;
            uint64_t w = 0;
            uint64_t is = kso_intersect(qm, nq, j->t_mins + j->t_off[t],
                                        j->t_abund ? j->t_abund + j->t_off[t] : NULL, nt, &w, NULL);
            if (is > 0) {
                if (j->fill && c < j->cap) {
                    j->qid[c] = q; j->tid[c] = t; j->isect[c] = (uint32_t)is; j->nw[c] = w;
                }
                c++;
            }
        }
    }
    j->count = c;
    return NULL;
}

uint64_t kso_manysearch(const uint64_t *q_off, const uint64_t *q_mins, uint32_t q_begin,
                        uint32_t q_end, const uint64_t *t_off, const uint64_t *t_mins,
                        const uint32_t *t_abund, uint32_t n_t, uint32_t *out_qid,
                        uint32_t *out_tid, uint32_t *out_isect, uint64_t *out_nw, uint64_t cap,
                        int n_threads) {
    uint32_t nq = q_end - q_begin;
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > nq) n_threads = nq ? (int)nq : 1;
    int fill = out_qid != NULL;
    search_job *jobs = (search_job *)calloc((size_t)n_threads, sizeof(search_job));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    /* pass 1: count per thread */
    for (int t = 0; t < n_threads; t++) {
        uint32_t a = q_begin + (uint32_t)((uint64_t)nq * (uint64_t)t / (uint64_t)n_threads);
        uint32_t b = q_begin + (uint32_t)((uint64_t)nq * (uint64_t)(t + 1) / (uint64_t)n_threads);
        jobs[t] = (search_job){q_off, q_mins, t_off, t_mins, t_abund, a, b, n_t,
                               NULL, NULL, NULL, NULL, 0, 0, 0};
    }
    for (int pass = 0; pass < (fill ? 2 : 1); pass++) {
        if (pass == 1) {
            uint64_t base = 0;
            for (int t = 0; t < n_threads; t++) {
                uint64_t c = jobs[t].count;
                jobs[t].qid = out_qid + base; jobs[t].tid = out_tid + base;
                jobs[t].isect = out_isect + base; jobs[t].nw = out_nw + base;
                jobs[t].cap = cap > base ? cap - base : 0; jobs[t].fill = 1;
                base += c;
            }
        }
        if (n_threads == 1) search_worker(&jobs[0]);
        else {
            for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, search_worker, &jobs[t]);
            for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        }
    }
    uint64_t total = 0;
    for (int t = 0; t < n_threads; t++) total += jobs[t].count;
    free(jobs); free(th);
    return total;
}
