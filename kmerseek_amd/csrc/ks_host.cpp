// ks_host.cpp — C++ host mirror of the reference's ProteomeIndex (include/kmerseek_host.hpp) and its flat C
// shim (include/kmerseek_host_c.h).  Pure host code above the compute ABI: all hashing / sorting happens in
// the HIP library through ks_sketch_batch / ks_kmer_positions / ks_sketches_union.
#include "../../include/kmerseek_host.hpp"


#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <thread>

#include "../../include/kmerseek_amd.h"
#include "../../include/kmerseek_host_c.h"
#include "ks_input.h"

namespace kmerseek {

size_t KmerInfo::total_occurrences() const {
    size_t n = 0;
    for (auto &kv : original_kmer_to_position) n += kv.second.size();
    return n;
}

static IndexError gpu_error(ks_ctx *ctx, int st, const char *what) {
    std::string msg = std::string(what) + ": " + (ctx ? ks_last_error(ctx) : ks_status_string(st));
    if (st == KS_ERR_INVALID_MOLTYPE) return IndexError(IndexError::SourmashError, "Sourmash error: " + std::string(ctx ? ks_last_error(ctx) : ""));
    return IndexError(IndexError::Gpu, msg);
}

static std::string hex64(uint64_t v) { // format!("{:x}", v), signature.rs:279
    char b[32];
    snprintf(b, sizeof b, "%llx", (unsigned long long)v);
    return b;
}

// sourmash aa_to_dayhoff / aa_to_hp (selected at encoding.rs:43-53) — only used to spell `encoded_kmer` strings
static char encode_residue(char c, uint32_t moltype) {
    if (moltype == KS_PROTEIN) return c;
    if (moltype == KS_DAYHOFF) {
        switch (c) {
        case 'C': return 'a';
        case 'A': case 'G': case 'P': case 'S': case 'T': return 'b';
        case 'D': case 'E': case 'N': case 'Q': return 'c';
        case 'H': case 'K': case 'R': return 'd';
        case 'I': case 'L': case 'M': case 'V': return 'e';
        case 'F': case 'W': case 'Y': return 'f';
        default: return 'X';
        }
    }
    switch (c) {
    case 'A': case 'F': case 'G': case 'I': case 'L': case 'M': case 'P': case 'V': case 'W': case 'Y': return 'h';
    case 'N': case 'C': case 'S': case 'T': case 'D': case 'E': case 'R': case 'H': case 'K': case 'Q': return 'p';
    default: return 'X';
    }
}

ProteomeIndex::ProteomeIndex(const std::string &path, uint32_t ksize, uint32_t scaled, const std::string &moltype,
                             bool store_raw_sequences, int device)
    : path_(path), ksize_(ksize), scaled_(scaled), moltype_(moltype), store_raw_(store_raw_sequences) {
    // get_hash_function_from_moltype failure is wrapped as IndexError::SourmashError (index.rs:168-172)
    if (ks_moltype_from_string(moltype.c_str(), &moltype_id_) != KS_OK)
        throw IndexError(IndexError::SourmashError, "Sourmash error: Invalid moltype: " + moltype +
                                                        ", only 'protein', 'hp', or 'dayhoff' are supported");
    if (ksize < 1 || ksize > KS_MAX_KSIZE)
        throw IndexError(IndexError::InvalidKsize, "Invalid k-mer size: " + std::to_string(ksize));
    if (scaled < 1) throw IndexError(IndexError::ValidationError, "Validation error: scaled must be >= 1");
    int st = ks_ctx_create(device, nullptr, &ctx_);
    if (st != KS_OK) throw IndexError(IndexError::Gpu, std::string("ks_ctx_create failed: ") + ks_status_string(st) + " (no CPU fallback)");
}

ProteomeIndex::~ProteomeIndex() {
    drop_device_index();
    if (ctx_) ks_ctx_destroy(ctx_);
}

ProteomeIndexBuilder ProteomeIndex::builder() { return ProteomeIndexBuilder(); }

static void split_path(const std::string &p, std::string &parent, std::string &file) {
    size_t pos = p.find_last_of('/');
    if (pos == std::string::npos) { parent = ""; file = p; }
    else { parent = p.substr(0, pos); file = p.substr(pos + 1); }
}

std::unique_ptr<ProteomeIndex> ProteomeIndex::new_with_auto_filename(const std::string &base_path, uint32_t ksize,
                                                                      uint32_t scaled, const std::string &moltype,
                                                                      bool store_raw_sequences, int device) {
    std::string parent, file;
    split_path(base_path, parent, file);
    std::string name = file + "." + moltype + ".k" + std::to_string(ksize) + ".scaled" + std::to_string(scaled) + ".kmerseek.rocksdb";
    std::string full = parent.empty() ? name : parent + "/" + name;
    return std::make_unique<ProteomeIndex>(full, ksize, scaled, moltype, store_raw_sequences, device);
}

std::string ProteomeIndex::generate_filename(const std::string &base_name) const {
    return base_name + "." + moltype_ + ".k" + std::to_string(ksize_) + ".scaled" + std::to_string(scaled_) + ".kmerseek.rocksdb";
}

ProteinSignature ProteomeIndex::create_protein_signature(const std::string &sequence, const std::string &name) {
    // called directly, the sequence is NOT upper-cased (only the FASTA path is: index.rs:1000)
    auto v = create_protein_signatures({{sequence, name}}, false);
    return std::move(v[0]);
}

void ProteomeIndex::prepare_records(const std::vector<std::pair<std::string, std::string>> &records, bool upper,
                                    std::vector<std::string> &processed, std::vector<uint64_t> &offs, std::vector<uint8_t> &res) {
    const size_t n = records.size();
    // ---- host pre-step, parallel over records: validate / resolve (aminoacid.rs:74-105)
    processed.assign(n, std::string());
    std::vector<ks_residue_error> errs(n);
    std::vector<int> status(n, KS_OK);
    unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16));
    if (n < 64) nt = 1;
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const std::string &s = records[i].first;
            processed[i].resize(s.size());
            uint64_t olen = 0;
            status[i] = ks_validate_and_resolve((const uint8_t *)s.data(), s.size(), upper ? 1 : 0,
                                                rng_seed_ + 0x9e3779b97f4a7c15ULL * (i + 1), (uint8_t *)&processed[i][0], &olen, &errs[i]);
            processed[i].resize(olen);
        }
    };
    if (nt == 1) work(0, n);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto &t : th) t.join();
    }
    rng_seed_ += n;
    for (size_t i = 0; i < n; i++)
        if (status[i] != KS_OK) { // first failing record aborts the batch (index.rs:993-1008)
            char msg[128];
            snprintf(msg, sizeof msg, "Invalid amino acid '%c' found at position %u", (char)errs[i].residue, errs[i].position);
            IndexError e(IndexError::InvalidAminoAcid, msg);
            e.residue = (char)errs[i].residue; e.position = errs[i].position; e.seq_index = i;
            throw e;
        }
    // ---- pack: the batch layout of the C ABI
    offs.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) offs[i + 1] = offs[i] + processed[i].size();
    res.assign(offs[n] + 1, 0);
    for (size_t i = 0; i < n; i++) memcpy(res.data() + offs[i], processed[i].data(), processed[i].size());
}

std::vector<ProteinSignature> ProteomeIndex::create_protein_signatures(
    const std::vector<std::pair<std::string, std::string>> &records, bool upper) {
    const size_t n = records.size();
    std::vector<ProteinSignature> out(n);
    if (n == 0) return out;
    std::vector<std::string> processed;
    std::vector<uint64_t> offs;
    std::vector<uint8_t> res;
    prepare_records(records, upper, processed, offs, res);
    // ---- the two batched GPU calls
    ks_params p{ksize_, scaled_, moltype_id_, 0, SEED};
    ks_sketches *S = nullptr;
    int st = ks_sketch_batch(ctx_, res.data(), offs.data(), (uint32_t)n, &p, &S);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketch_batch");
    std::vector<uint64_t> so(n + 1), sh(ks_sketches_n_hashes(S) + 1);
    std::vector<uint32_t> sa(sh.size());
    st = ks_sketches_copy_to_host(ctx_, S, so.data(), sh.data(), sa.data());
    ks_sketches_free(S);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketches_copy_to_host");
    ks_kmerpos *K = nullptr;
    st = ks_kmer_positions(ctx_, res.data(), offs.data(), (uint32_t)n, &p, &K);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_kmer_positions");
    const uint64_t np = ks_kmerpos_count(K);
    std::vector<uint32_t> pseq(np + 1), pstart(np + 1);
    std::vector<uint64_t> phash(np + 1);
    st = ks_kmerpos_copy_to_host(ctx_, K, pseq.data(), pstart.data(), phash.data());
    ks_kmerpos_free(K);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_kmerpos_copy_to_host");
    // ---- build the signatures
    for (size_t i = 0; i < n; i++) {
        ProteinSignature &g = out[i];
        g.name = records[i].second;
        g.moltype = moltype_;
        g.protein_ksize = ksize_;
        g.scaled = scaled_;
        g.mins.assign(sh.begin() + so[i], sh.begin() + so[i + 1]);
        g.abunds.assign(sa.begin() + so[i], sa.begin() + so[i + 1]);
        uint64_t sum = 0;
        for (uint64_t m : g.mins) sum += m; // wrapping
        g.md5sum = hex64(sum);
        if (store_raw_) g.raw_sequence = processed[i];
    }
    for (uint64_t j = 0; j < np; j++) { // triples are ordered by (seq, start): positions come out ascending
        ProteinSignature &g = out[pseq[j]];
        const std::string &seq = processed[pseq[j]];
        KmerInfo &ki = g.kmer_infos[phash[j]];
        if (ki.encoded_kmer.empty()) {
            ki.ksize = ksize_;
            ki.hashval = phash[j];
            ki.encoded_kmer.resize(ksize_);
            for (uint32_t c = 0; c < ksize_; c++) ki.encoded_kmer[c] = encode_residue(seq[pstart[j] + c], moltype_id_);
        }
        ki.original_kmer_to_position[seq.substr(pstart[j], ksize_)].push_back(pstart[j]);
    }
    return out;
}

void ProteomeIndex::process_kmers(const std::string &sequence, ProteinSignature &sig) {
    std::vector<uint64_t> offs{0, sequence.size()};
    ks_params p{ksize_, scaled_, moltype_id_, 0, SEED};
    ks_kmerpos *K = nullptr;
    int st = ks_kmer_positions(ctx_, (const uint8_t *)sequence.data(), offs.data(), 1, &p, &K);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_kmer_positions");
    const uint64_t np = ks_kmerpos_count(K);
    std::vector<uint32_t> pseq(np + 1), pstart(np + 1);
    std::vector<uint64_t> phash(np + 1);
    st = ks_kmerpos_copy_to_host(ctx_, K, pseq.data(), pstart.data(), phash.data());
    ks_kmerpos_free(K);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_kmerpos_copy_to_host");
    for (uint64_t j = 0; j < np; j++) {
        // "if this hashval is in the minhash" (index.rs:769): FracMinHash membership == kept by the threshold,
        // checked against the signature anyway so a foreign signature behaves like the reference
        if (!std::binary_search(sig.mins.begin(), sig.mins.end(), phash[j])) continue;
        KmerInfo &ki = sig.kmer_infos[phash[j]];
        if (ki.encoded_kmer.empty()) {
            ki.ksize = ksize_;
            ki.hashval = phash[j];
            ki.encoded_kmer.resize(ksize_);
            for (uint32_t c = 0; c < ksize_; c++) ki.encoded_kmer[c] = encode_residue(sequence[pstart[j] + c], moltype_id_);
        }
        ki.original_kmer_to_position[sequence.substr(pstart[j], ksize_)].push_back(pstart[j]);
    }
}

void ProteomeIndex::store_signatures(std::vector<ProteinSignature> sigs) {
    drop_device_index(); // the search index is laid out over signatures_: it is rebuilt on the next search
    // combined sketch: union with summed abundances (index.rs:803-827), done on the GPU for the batch and
    // merged into the running (sorted) combined sketch on the host
    std::vector<uint64_t> offs(sigs.size() + 1, 0);
    for (size_t i = 0; i < sigs.size(); i++) offs[i + 1] = offs[i] + sigs[i].mins.size();
    if (offs.back() > 0) {
        std::vector<uint64_t> h(offs.back());
        std::vector<uint32_t> a(offs.back());
        for (size_t i = 0; i < sigs.size(); i++)
            for (size_t j = 0; j < sigs[i].mins.size(); j++) {
                h[offs[i] + j] = sigs[i].mins[j];
                a[offs[i] + j] = (uint32_t)std::min<uint64_t>(sigs[i].abunds[j], 0xffffffffu);
            }
        ks_params p{ksize_, scaled_, moltype_id_, 0, SEED};
        ks_sketches *S = nullptr, *U = nullptr;
        int st = ks_sketches_from_host(ctx_, offs.data(), h.data(), a.data(), (uint32_t)sigs.size(), &p, &S);
        if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketches_from_host");
        st = ks_sketches_union(ctx_, S, &U);
        ks_sketches_free(S);
        if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketches_union");
        const uint64_t nu = ks_sketches_n_hashes(U);
        std::vector<uint64_t> uo(2), uh(nu + 1);
        std::vector<uint32_t> ua(nu + 1);
        st = ks_sketches_copy_to_host(ctx_, U, uo.data(), uh.data(), ua.data());
        ks_sketches_free(U);
        if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketches_copy_to_host");
        std::vector<uint64_t> mm, ma;
        mm.reserve(combined_mins_.size() + nu);
        ma.reserve(combined_mins_.size() + nu);
        size_t i = 0, j = 0;
        while (i < combined_mins_.size() || j < nu) {
            if (j >= nu || (i < combined_mins_.size() && combined_mins_[i] < uh[j])) { mm.push_back(combined_mins_[i]); ma.push_back(combined_abunds_[i]); i++; }
            else if (i >= combined_mins_.size() || uh[j] < combined_mins_[i]) { mm.push_back(uh[j]); ma.push_back(ua[j]); j++; }
            else { mm.push_back(uh[j]); ma.push_back(combined_abunds_[i] + ua[j]); i++; j++; }
        }
        combined_mins_.swap(mm);
        combined_abunds_.swap(ma);
    }
    for (auto &g : sigs) { // same pseudo-md5 overwrites (DashMap::insert, index.rs:817-820)
        std::string key = g.md5sum;
        signatures_[key] = std::move(g);
    }
}

void ProteomeIndex::store_signatures_batch(const std::vector<ProteinSignature> &sigs) {
    store_signatures(std::vector<ProteinSignature>(sigs));
}

// FASTA records of a plain / gzip / zstd / bzip2 / xz file (ks_input.h: by magic number, as needletail's parse_fastx_file does,
// index.rs:907-961), handed to `fn` in batches of `batch_size` (sequence, full header line) pairs.
void ProteomeIndex::for_each_fasta_batch(const std::string &fasta_path, size_t batch_size, uint32_t progress_interval,
                                         const std::function<void(std::vector<std::pair<std::string, std::string>> &)> &fn,
                                         size_t *n_records) {
    std::string oerr;
    std::unique_ptr<KsInput> in(KsInput::open(fasta_path.c_str(), oerr));
    if (!in) throw IndexError(IndexError::ParseError, "Parse error: " + oerr);
    if (batch_size == 0) batch_size = 1000;
    std::vector<std::pair<std::string, std::string>> batch; // (sequence, id)
    size_t record_count = 0;
    std::string line, id, seq;
    bool have = false;
    auto flush_record = [&]() {
        if (!have) return;
        batch.emplace_back(std::move(seq), std::move(id));
        seq.clear(); id.clear();
        record_count++;
        if (batch.size() >= batch_size) {
            fn(batch);
            batch.clear();
        }
        if (progress_interval > 0 && record_count % progress_interval == 0) printf("Read %zu sequences...\n", record_count);
    };
    auto take_line = [&]() { // `line` holds one line without its terminator
        while (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) return;
        if (line[0] == '>') {
            flush_record();
            id = line.substr(1);
            have = true;
        } else if (have) {
            seq += line;
        } else {
            throw IndexError(IndexError::ParseError, "Parse error: FASTA record does not start with '>'");
        }
    };
    std::vector<char> buf(1 << 20);
    for (;;) {
        const long n = in->read(buf.data(), buf.size());
        if (n < 0) throw IndexError(IndexError::ParseError, "Parse error: " + in->error());
        if (n == 0) break;
        const char *p = buf.data(), *end = p + n;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!nl) { line.append(p, (size_t)(end - p)); break; } // the line goes on in the next chunk
            line.append(p, (size_t)(nl - p));
            take_line();
            line.clear();
            p = nl + 1;
        }
    }
    if (!line.empty()) { take_line(); line.clear(); } // last line without a terminator
    flush_record();
    if (!batch.empty()) fn(batch);
    if (n_records) *n_records = record_count;
}

void ProteomeIndex::process_fasta(const std::string &fasta_path, uint32_t progress_interval, size_t batch_size) {
    if (progress_interval > 0) printf("Reading FASTA file with automatic compression detection and parallel processing...\n");
    size_t record_count = 0;
    for_each_fasta_batch(fasta_path, batch_size, progress_interval,
                         [&](std::vector<std::pair<std::string, std::string>> &batch) { store_signatures(create_protein_signatures(batch, true)); },
                         &record_count);
    save_state();
    if (progress_interval > 0) printf("Successfully processed and stored %zu sequences.\n", record_count);
}

// ---- search (added: SURVEY 8(b); rows of branchwater manysearch, src/python/kmerseek/search.py:125-141) ------------------
namespace {
// MD5 (RFC 1321), for sourmash's md5sum of a sketch: MD5(ascii(3k) || ascii(min) || ...)
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0;
    uint8_t buf[64];
    size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; i++) {
            uint32_t f;
            int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D;
            D = C; C = B;
            B = B + rol(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const void *data, size_t n) {
        const uint8_t *p = (const uint8_t *)data;
        len += n;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        char out[33];
        const uint32_t w[4] = {a, b, c, d};
        for (int i = 0; i < 16; i++) snprintf(out + 2 * i, 3, "%02x", (w[i / 4] >> (8 * (i % 4))) & 0xffu);
        return std::string(out, 32);
    }
};

std::string sourmash_md5(const uint64_t *mins, size_t n, uint32_t protein_ksize) {
    Md5 m;
    char b[32];
    int l = snprintf(b, sizeof b, "%u", protein_ksize * PROTEIN_TO_MINHASH_RATIO);
    m.update(b, (size_t)l);
    for (size_t i = 0; i < n; i++) {
        l = snprintf(b, sizeof b, "%llu", (unsigned long long)mins[i]);
        m.update(b, (size_t)l);
    }
    return m.hex();
}
} // namespace

const std::vector<std::string> &ProteomeIndex::search_columns() {
    static const std::vector<std::string> cols = {
        "query_name", "query_md5", "match_name", "containment", "intersect_hashes", "ksize", "scaled", "moltype", "match_md5", "jaccard",
        "max_containment", "average_abund", "median_abund", "std_abund", "query_containment_ani", "match_containment_ani",
        "average_containment_ani", "max_containment_ani", "n_weighted_found", "total_weighted_hashes", "containment_target_in_query",
        "f_weighted_target_in_query"};
    return cols;
}

void ProteomeIndex::drop_device_index() {
    if (dev_index_) { ks_index_free(dev_index_); dev_index_ = nullptr; }
    if (dev_targets_) { ks_sketches_free(dev_targets_); dev_targets_ = nullptr; }
    dev_order_.clear(); dev_md5_.clear(); dev_total_abund_.clear();
}

void ProteomeIndex::ensure_device_index() {
    if (dev_index_) return;
    // targets = the stored signatures in key order (get_signatures, index.rs:642-652), as one CSR
    const size_t n = signatures_.size();
    std::vector<uint64_t> offs(n + 1, 0);
    dev_order_.clear(); dev_order_.reserve(n);
    size_t i = 0;
    for (auto &kv : signatures_) { dev_order_.push_back(&kv.second); offs[i + 1] = offs[i] + kv.second.mins.size(); i++; }
    std::vector<uint64_t> h(offs[n] + 1);
    std::vector<uint32_t> a(offs[n] + 1);
    dev_total_abund_.assign(n, 0);
    for (i = 0; i < n; i++) {
        const ProteinSignature &g = *dev_order_[i];
        for (size_t j = 0; j < g.mins.size(); j++) {
            h[offs[i] + j] = g.mins[j];
            a[offs[i] + j] = (uint32_t)std::min<uint64_t>(g.abunds[j], 0xffffffffu);
            dev_total_abund_[i] += g.abunds[j];
        }
    }
    dev_md5_.assign(n, std::string());
    ks_params p{ksize_, scaled_, moltype_id_, 0, SEED};
    int st = ks_sketches_from_host(ctx_, offs.data(), h.data(), a.data(), (uint32_t)n, &p, &dev_targets_);
    if (st != KS_OK) { dev_targets_ = nullptr; throw gpu_error(ctx_, st, "ks_sketches_from_host"); }
    st = ks_index_build(ctx_, dev_targets_, &dev_index_);
    if (st != KS_OK) { dev_index_ = nullptr; drop_device_index(); throw gpu_error(ctx_, st, "ks_index_build"); }
}

std::vector<SearchResult> ProteomeIndex::search(const std::vector<std::pair<std::string, std::string>> &queries, bool upper) {
    std::vector<SearchResult> rows;
    const size_t nq = queries.size();
    if (nq == 0 || signatures_.empty()) return rows;
    if (nq > 0xfffffff0ULL) throw IndexError(IndexError::ValidationError, "Validation error: too many query records in one batch");
    std::vector<std::string> processed;
    std::vector<uint64_t> offs;
    std::vector<uint8_t> res;
    prepare_records(queries, upper, processed, offs, res);
    ensure_device_index();
    // ---- one call: sketch the query batch and join it against the resident index
    ks_sketches *S = nullptr;
    ks_hits *H = nullptr;
    int st = ks_sketch_search(ctx_, dev_index_, res.data(), offs.data(), (uint32_t)nq, &S, &H);
    if (st != KS_OK) throw gpu_error(ctx_, st, "ks_sketch_search");
    const uint64_t nh = ks_hits_count(H);
    std::vector<uint32_t> qid(nh + 1), tid(nh + 1), isect(nh + 1);
    std::vector<uint64_t> nw(nh + 1);
    std::vector<uint64_t> qo(nq + 1), qm(ks_sketches_n_hashes(S) + 1);
    st = ks_hits_copy_to_host(ctx_, H, qid.data(), tid.data(), isect.data(), nw.data());
    if (st == KS_OK) st = ks_sketches_copy_to_host(ctx_, S, qo.data(), qm.data(), nullptr);
    ks_hits_free(H);
    ks_sketches_free(S);
    if (st != KS_OK) throw gpu_error(ctx_, st, "copy of the search result");
    // ---- rows: f64 ratios of the integer results (formulas: SURVEY.md 8(a) row a10)
    rows.reserve(nh);
    const double k3 = (double)(ksize_ * PROTEIN_TO_MINHASH_RATIO);
    std::vector<std::string> q_md5(nq);
    std::vector<double> shared;
    for (uint64_t r = 0; r < nh; r++) {
        const uint32_t q = qid[r], t = tid[r];
        const ProteinSignature &g = *dev_order_[t];
        const uint64_t *qmins = qm.data() + qo[q];
        const size_t n_q = (size_t)(qo[q + 1] - qo[q]), n_t = g.mins.size();
        // the match's abundances of the shared hashes: a sorted merge of the two sketches
        shared.clear();
        for (size_t i = 0, j = 0; i < n_q && j < n_t;) {
            if (qmins[i] < g.mins[j]) i++;
            else if (g.mins[j] < qmins[i]) j++;
            else { shared.push_back((double)g.abunds[j]); i++; j++; }
        }
        if (shared.size() != isect[r])
            throw IndexError(IndexError::Gpu, "search: a row's intersect does not match the sketches it was computed from");
        std::sort(shared.begin(), shared.end());
        const size_t n = shared.size();
        double sum = 0;
        for (double x : shared) sum += x;
        const double mean = sum / (double)n;
        double ss = 0;
        for (double x : shared) ss += (x - mean) * (x - mean);
        SearchResult o;
        o.query_name = queries[q].second;
        if (q_md5[q].empty()) q_md5[q] = sourmash_md5(qmins, n_q, ksize_);
        o.query_md5 = q_md5[q];
        o.match_name = g.name;
        if (dev_md5_[t].empty()) dev_md5_[t] = sourmash_md5(g.mins.data(), n_t, ksize_);
        o.match_md5 = dev_md5_[t];
        o.moltype = moltype_;
        o.intersect_hashes = isect[r];
        o.ksize = ksize_ * PROTEIN_TO_MINHASH_RATIO;
        o.scaled = scaled_;
        const double I = (double)isect[r];
        const double cq = I / (double)n_q, ct = I / (double)n_t;
        o.containment = cq;
        o.jaccard = I / (double)(n_q + n_t - isect[r]);
        o.max_containment = std::max(cq, ct);
        o.average_abund = mean;
        o.median_abund = (n % 2) ? shared[n / 2] : (shared[n / 2 - 1] + shared[n / 2]) / 2.0;
        o.std_abund = std::sqrt(ss / (double)n);
        o.query_containment_ani = std::pow(cq, 1.0 / k3);
        o.match_containment_ani = std::pow(ct, 1.0 / k3);
        o.average_containment_ani = (o.query_containment_ani + o.match_containment_ani) / 2.0;
        o.max_containment_ani = std::max(o.query_containment_ani, o.match_containment_ani);
        o.n_weighted_found = nw[r];
        o.total_weighted_hashes = dev_total_abund_[t];
        o.containment_target_in_query = ct;
        o.f_weighted_target_in_query = (double)nw[r] / (double)dev_total_abund_[t];
        rows.push_back(std::move(o));
    }
    return rows;
}

std::vector<SearchResult> ProteomeIndex::search_fasta(const std::string &fasta_path, size_t batch_size) {
    std::vector<SearchResult> rows;
    for_each_fasta_batch(fasta_path, batch_size ? batch_size : 100000, 0,
                         [&](std::vector<std::pair<std::string, std::string>> &batch) {
                             std::vector<SearchResult> part = search(batch, true); // the FASTA path upper-cases (index.rs:1000)
                             for (auto &r : part) rows.push_back(std::move(r));
                         },
                         nullptr);
    return rows;
}

// ---- persistence: flat little-endian file (own format; RocksDB layout is out of scope) --------------------
static void put_u64(std::ostream &o, uint64_t v) { o.write((const char *)&v, 8); }
static void put_str(std::ostream &o, const std::string &s) { put_u64(o, s.size()); o.write(s.data(), (std::streamsize)s.size()); }

void ProteomeIndex::save_state() {
    std::ofstream o(path_, std::ios::binary | std::ios::trunc);
    if (!o) throw IndexError(IndexError::Io, "IO error: cannot write " + path_);
    o.write("KSIDX001", 8);
    put_str(o, moltype_); put_u64(o, ksize_); put_u64(o, scaled_); put_u64(o, store_raw_ ? 1 : 0);
    put_u64(o, combined_mins_.size());
    o.write((const char *)combined_mins_.data(), (std::streamsize)(combined_mins_.size() * 8));
    o.write((const char *)combined_abunds_.data(), (std::streamsize)(combined_abunds_.size() * 8));
    put_u64(o, signatures_.size());
    for (auto &kv : signatures_) {
        const ProteinSignature &g = kv.second;
        put_str(o, g.name); put_str(o, g.md5sum);
        put_u64(o, g.mins.size());
        o.write((const char *)g.mins.data(), (std::streamsize)(g.mins.size() * 8));
        o.write((const char *)g.abunds.data(), (std::streamsize)(g.abunds.size() * 8));
        put_u64(o, g.raw_sequence ? 1 : 0);
        if (g.raw_sequence) put_str(o, *g.raw_sequence);
        put_u64(o, g.kmer_infos.size());
        for (auto &ki : g.kmer_infos) {
            put_u64(o, ki.first); put_str(o, ki.second.encoded_kmer);
            put_u64(o, ki.second.original_kmer_to_position.size());
            for (auto &op : ki.second.original_kmer_to_position) {
                put_str(o, op.first); put_u64(o, op.second.size());
                for (size_t x : op.second) put_u64(o, x);
            }
        }
    }
    if (!o) throw IndexError(IndexError::Io, "IO error: short write to " + path_);
}

// Bounded reader of the flat index file: every length field is checked against the bytes that remain BEFORE anything is
// allocated, and the stream state after every block, so a corrupt or foreign file that happens to start with the magic
// fails with an IndexError instead of a multi-GB allocation.
namespace {
struct IndexReader {
    std::istream &i;
    uint64_t left;
    const std::string &path;
    [[noreturn]] void corrupt(const char *what) const {
        throw IndexError(IndexError::Io, "IO error: corrupt index file " + path + " (" + what + ")");
    }
    void need(uint64_t n, const char *what) const { if (n > left) corrupt(what); }
    void raw(void *dst, uint64_t n, const char *what) {
        need(n, what);
        i.read((char *)dst, (std::streamsize)n);
        if (!i) corrupt(what);
        left -= n;
    }
    uint64_t u64(const char *what) { uint64_t v = 0; raw(&v, 8, what); return v; }
    // a count of records that each take at least `min_each` bytes of the file
    uint64_t count(uint64_t min_each, const char *what) {
        const uint64_t n = u64(what);
        if (min_each && n > left / min_each) corrupt(what);
        return n;
    }
    std::string str(const char *what) {
        const uint64_t n = count(1, what);
        std::string s((size_t)n, '\0');
        if (n) raw(&s[0], n, what);
        return s;
    }
};
} // namespace

std::unique_ptr<ProteomeIndex> ProteomeIndex::load(const std::string &path, int device) {
    std::ifstream i(path, std::ios::binary);
    char magic[8] = {0};
    if (!i || !i.read(magic, 8) || memcmp(magic, "KSIDX001", 8) != 0)
        throw IndexError(IndexError::NoSavedState, "No saved state found in database"); // errors.rs:23-24
    i.seekg(0, std::ios::end);
    const uint64_t size = (uint64_t)i.tellg();
    i.seekg(8, std::ios::beg);
    IndexReader r{i, size - 8, path};
    std::string moltype = r.str("moltype");
    const uint64_t k64 = r.u64("ksize"), sc64 = r.u64("scaled");
    if (k64 == 0 || k64 > 0xffffffffULL || sc64 == 0 || sc64 > 0xffffffffULL) r.corrupt("ksize / scaled");
    const uint32_t k = (uint32_t)k64, sc = (uint32_t)sc64;
    const bool raw = r.u64("store_raw") != 0;
    uint32_t mt_id = 0;
    if (ks_moltype_from_string(moltype.c_str(), &mt_id) != KS_OK || k > KS_MAX_KSIZE) r.corrupt("moltype / ksize");
    auto ix = std::make_unique<ProteomeIndex>(path, k, sc, moltype, raw, device);
    const uint64_t nc = r.count(16, "combined sketch size");
    ix->combined_mins_.resize(nc); ix->combined_abunds_.resize(nc);
    r.raw(ix->combined_mins_.data(), nc * 8, "combined mins");
    r.raw(ix->combined_abunds_.data(), nc * 8, "combined abundances");
    const uint64_t ns = r.count(40, "signature count"); // name + md5 + mins + raw flag + k-mer count: >= 5 length words
    for (uint64_t s = 0; s < ns; s++) {
        ProteinSignature g;
        g.name = r.str("name"); g.md5sum = r.str("md5sum");
        g.moltype = moltype; g.protein_ksize = k; g.scaled = sc;
        const uint64_t nm = r.count(16, "mins size");
        g.mins.resize(nm); g.abunds.resize(nm);
        r.raw(g.mins.data(), nm * 8, "mins");
        r.raw(g.abunds.data(), nm * 8, "abundances");
        if (r.u64("raw flag")) g.raw_sequence = r.str("raw sequence");
        const uint64_t nk = r.count(24, "k-mer count");
        for (uint64_t q = 0; q < nk; q++) {
            const uint64_t h = r.u64("k-mer hash");
            KmerInfo &ki = g.kmer_infos[h];
            ki.ksize = k; ki.hashval = h; ki.encoded_kmer = r.str("encoded k-mer");
            const uint64_t no = r.count(16, "original k-mer count");
            for (uint64_t o = 0; o < no; o++) {
                std::string orig = r.str("original k-mer");
                const uint64_t np = r.count(8, "position count");
                auto &v = ki.original_kmer_to_position[orig];
                v.reserve((size_t)np);
                for (uint64_t x = 0; x < np; x++) v.push_back((size_t)r.u64("position"));
            }
        }
        std::string key = g.md5sum;
        ix->signatures_[key] = std::move(g);
    }
    if (r.left != 0) r.corrupt("trailing bytes");
    return ix;
}

bool ProteomeIndex::is_equivalent_to(const ProteomeIndex &o) const { // index.rs:524-625
    if (ksize_ != o.ksize_ || scaled_ != o.scaled_ || moltype_ != o.moltype_) return false;
    if (signature_count() != o.signature_count()) return false;
    if (combined_minhash_size() != o.combined_minhash_size()) return false;
    for (auto &kv : signatures_) {
        auto it = o.signatures_.find(kv.first);
        if (it == o.signatures_.end()) return false;
        const ProteinSignature &a = kv.second, &b = it->second;
        if (a.mins != b.mins) return false;
        if (a.kmer_infos.size() != b.kmer_infos.size()) return false;
        for (auto &ki : a.kmer_infos) {
            auto jt = b.kmer_infos.find(ki.first);
            if (jt == b.kmer_infos.end()) return false;
            if (ki.second.ksize != jt->second.ksize || ki.second.hashval != jt->second.hashval ||
                ki.second.encoded_kmer != jt->second.encoded_kmer ||
                ki.second.original_kmer_to_position != jt->second.original_kmer_to_position)
                return false;
        }
    }
    return combined_mins_ == o.combined_mins_;
}

void ProteomeIndex::print_stats() const { // index.rs:628-639
    printf("ProteomeIndex Statistics:\n  K-mer size: %u\n  Scaled: %u\n  Molecular type: %s\n  Combined minhash size: %zu\n"
           "  Raw sequence storage: %s\n",
           ksize_, scaled_, moltype_.c_str(), combined_minhash_size(), store_raw_ ? "enabled" : "disabled");
}

std::unique_ptr<ProteomeIndex> ProteomeIndexBuilder::build() const { // index.rs:3021-3036
    if (!path_) throw IndexError(IndexError::BuilderError, "Builder error: Database path is required");
    if (!ksize_) throw IndexError(IndexError::BuilderError, "Builder error: K-mer size is required");
    if (!scaled_) throw IndexError(IndexError::BuilderError, "Builder error: Scaled value is required");
    if (!moltype_) throw IndexError(IndexError::BuilderError, "Builder error: Molecular type is required");
    return std::make_unique<ProteomeIndex>(*path_, *ksize_, *scaled_, *moltype_, store_raw_, device_);
}

std::unique_ptr<ProteomeIndex> ProteomeIndexBuilder::build_with_auto_filename() const { // index.rs:3039-3060
    if (!path_) throw IndexError(IndexError::BuilderError, "Builder error: Base path is required");
    if (!ksize_) throw IndexError(IndexError::BuilderError, "Builder error: K-mer size is required");
    if (!scaled_) throw IndexError(IndexError::BuilderError, "Builder error: Scaled value is required");
    if (!moltype_) throw IndexError(IndexError::BuilderError, "Builder error: Molecular type is required");
    return ProteomeIndex::new_with_auto_filename(*path_, *ksize_, *scaled_, *moltype_, store_raw_, device_);
}

} // namespace kmerseek

// =============================================================================================
// flat C shim
// =============================================================================================
using kmerseek::IndexError;
using kmerseek::ProteinSignature;
using kmerseek::ProteomeIndex;

struct ksh_index {
    std::unique_ptr<ProteomeIndex> ix;
};

static int fail(const IndexError &e, char *err, size_t cap) {
    if (err && cap) { strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return (int)e.kind + 1;
}
static int fail_std(const std::exception &e, char *err, size_t cap) {
    if (err && cap) { strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return 1000;
}
#define KSH_GUARD(...)                                             \
    try { __VA_ARGS__; return 0; }                                 \
    catch (const IndexError &e) { return fail(e, err, err_cap); }  \
    catch (const std::exception &e) { return fail_std(e, err, err_cap); } \
    catch (...) { if (err && err_cap) { strncpy(err, "internal error (unknown exception)", err_cap - 1); err[err_cap - 1] = 0; } return 1000; }

static void json_str(std::ostringstream &o, const std::string &s) {
    o << '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') o << '\\' << c;
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o << b; }
        else o << c;
    }
    o << '"';
}

static void json_sig(std::ostringstream &o, const ProteinSignature &g, bool with_kmers) {
    o << "{\"name\":"; json_str(o, g.name);
    o << ",\"md5sum\":"; json_str(o, g.md5sum);
    o << ",\"moltype\":"; json_str(o, g.moltype);
    o << ",\"protein_ksize\":" << g.protein_ksize << ",\"minhash_ksize\":" << g.minhash_ksize() << ",\"scaled\":" << g.scaled;
    o << ",\"mins\":[";
    for (size_t i = 0; i < g.mins.size(); i++) o << (i ? "," : "") << g.mins[i];
    o << "],\"abunds\":[";
    for (size_t i = 0; i < g.abunds.size(); i++) o << (i ? "," : "") << g.abunds[i];
    o << "],\"n_kmer_infos\":" << g.kmer_infos.size();
    o << ",\"raw_sequence\":";
    if (g.raw_sequence) json_str(o, *g.raw_sequence); else o << "null";
    if (with_kmers) {
        o << ",\"kmer_infos\":{";
        bool first = true;
        for (auto &ki : g.kmer_infos) {
            o << (first ? "" : ",") << '"' << ki.first << "\":{\"ksize\":" << ki.second.ksize << ",\"encoded_kmer\":";
            json_str(o, ki.second.encoded_kmer);
            o << ",\"original_kmer_to_position\":{";
            bool f2 = true;
            for (auto &op : ki.second.original_kmer_to_position) {
                o << (f2 ? "" : ","); json_str(o, op.first); o << ":[";
                for (size_t x = 0; x < op.second.size(); x++) o << (x ? "," : "") << op.second[x];
                o << "]";
                f2 = false;
            }
            o << "}}";
            first = false;
        }
        o << "}";
    }
    o << "}";
}

static char *dup_string(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    if (p) memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

extern "C" {

int ksh_index_new(const char *path, uint32_t ksize, uint32_t scaled, const char *moltype, int store_raw, int device,
                  int auto_filename, ksh_index **out, char *err, size_t err_cap) {
    KSH_GUARD({
        auto h = new ksh_index();
        try {
            if (auto_filename) h->ix = ProteomeIndex::new_with_auto_filename(path, ksize, scaled, moltype, store_raw != 0, device);
            else h->ix = std::make_unique<ProteomeIndex>(path, ksize, scaled, moltype, store_raw != 0, device);
        } catch (...) { delete h; throw; }
        *out = h;
    })
}

int ksh_index_build(const char *path, int has_ksize, uint32_t ksize, int has_scaled, uint32_t scaled, const char *moltype,
                    int store_raw, int auto_filename, int device, ksh_index **out, char *err, size_t err_cap) {
    KSH_GUARD({
        kmerseek::ProteomeIndexBuilder b = ProteomeIndex::builder();
        if (path) b.path(path);
        if (has_ksize) b.ksize(ksize);
        if (has_scaled) b.scaled(scaled);
        if (moltype) b.moltype(moltype);
        b.store_raw_sequences(store_raw != 0).device(device);
        auto h = new ksh_index();
        try { h->ix = auto_filename ? b.build_with_auto_filename() : b.build(); } catch (...) { delete h; throw; }
        *out = h;
    })
}

void ksh_index_free(ksh_index *ix) { delete ix; }

int ksh_index_create_signature(ksh_index *ix, const char *sequence, const char *name, int store, char **json_out,
                               char *err, size_t err_cap) {
    KSH_GUARD({
        ProteinSignature g = ix->ix->create_protein_signature(sequence, name);
        if (json_out) { std::ostringstream o; json_sig(o, g, true); *json_out = dup_string(o.str()); }
        if (store) { std::vector<ProteinSignature> v; v.push_back(std::move(g)); ix->ix->store_signatures(std::move(v)); }
    })
}

int ksh_index_add_records(ksh_index *ix, const char *const *sequences, const char *const *names, uint32_t n, int upper,
                          char *err, size_t err_cap) {
    KSH_GUARD({
        std::vector<std::pair<std::string, std::string>> recs(n);
        for (uint32_t i = 0; i < n; i++) recs[i] = {sequences[i], names[i]};
        ix->ix->store_signatures(ix->ix->create_protein_signatures(recs, upper != 0));
    })
}

int ksh_index_process_fasta(ksh_index *ix, const char *fasta_path, uint32_t progress_interval, uint64_t batch_size,
                            char *err, size_t err_cap) {
    KSH_GUARD(ix->ix->process_fasta(fasta_path, progress_interval, (size_t)batch_size))
}

static std::string json_rows(const std::vector<kmerseek::SearchResult> &rows) {
    std::ostringstream o;
    auto num = [&](double v) { char b[40]; snprintf(b, sizeof b, "%.17g", v); o << b; }; // (17 digits: the f64 round-trips)
    o << "[";
    for (size_t i = 0; i < rows.size(); i++) {
        const kmerseek::SearchResult &r = rows[i];
        o << (i ? "," : "") << "{\"query_name\":"; json_str(o, r.query_name);
        o << ",\"query_md5\":"; json_str(o, r.query_md5);
        o << ",\"match_name\":"; json_str(o, r.match_name);
        o << ",\"containment\":"; num(r.containment);
        o << ",\"intersect_hashes\":" << r.intersect_hashes << ",\"ksize\":" << r.ksize << ",\"scaled\":" << r.scaled;
        o << ",\"moltype\":"; json_str(o, r.moltype);
        o << ",\"match_md5\":"; json_str(o, r.match_md5);
        o << ",\"jaccard\":"; num(r.jaccard);
        o << ",\"max_containment\":"; num(r.max_containment);
        o << ",\"average_abund\":"; num(r.average_abund);
        o << ",\"median_abund\":"; num(r.median_abund);
        o << ",\"std_abund\":"; num(r.std_abund);
        o << ",\"query_containment_ani\":"; num(r.query_containment_ani);
        o << ",\"match_containment_ani\":"; num(r.match_containment_ani);
        o << ",\"average_containment_ani\":"; num(r.average_containment_ani);
        o << ",\"max_containment_ani\":"; num(r.max_containment_ani);
        o << ",\"n_weighted_found\":" << r.n_weighted_found << ",\"total_weighted_hashes\":" << r.total_weighted_hashes;
        o << ",\"containment_target_in_query\":"; num(r.containment_target_in_query);
        o << ",\"f_weighted_target_in_query\":"; num(r.f_weighted_target_in_query);
        o << "}";
    }
    o << "]";
    return o.str();
}

int ksh_index_search(ksh_index *ix, const char *const *sequences, const char *const *names, uint32_t n, int upper, char **json_out,
                     char *err, size_t err_cap) {
    KSH_GUARD({
        if (!ix || !json_out || (n && (!sequences || !names))) throw IndexError(IndexError::ValidationError, "Validation error: NULL argument");
        std::vector<std::pair<std::string, std::string>> recs(n);
        for (uint32_t i = 0; i < n; i++) recs[i] = {sequences[i], names[i]};
        *json_out = dup_string(json_rows(ix->ix->search(recs, upper != 0)));
        if (!*json_out) throw std::bad_alloc();
    })
}

// sourmash md5sum of a sketch (what the search rows' query_md5 / match_md5 columns hold); needs no GPU
int ksh_sourmash_md5(const uint64_t *mins, uint64_t n, uint32_t protein_ksize, char *out33) {
    if ((!mins && n) || !out33) return 1;
    try {
        const std::string h = kmerseek::sourmash_md5(mins, (size_t)n, protein_ksize);
        memcpy(out33, h.c_str(), 33);
        return 0;
    } catch (...) { return 1000; }
}

int ksh_index_search_fasta(ksh_index *ix, const char *fasta_path, uint64_t batch_size, char **json_out, char *err, size_t err_cap) {
    KSH_GUARD({
        if (!ix || !json_out || !fasta_path) throw IndexError(IndexError::ValidationError, "Validation error: NULL argument");
        *json_out = dup_string(json_rows(ix->ix->search_fasta(fasta_path, (size_t)batch_size)));
        if (!*json_out) throw std::bad_alloc();
    })
}

uint64_t ksh_index_signature_count(const ksh_index *ix) { return ix->ix->signature_count(); }
uint64_t ksh_index_combined_minhash_size(const ksh_index *ix) { return ix->ix->combined_minhash_size(); }
uint32_t ksh_index_ksize(const ksh_index *ix) { return ix->ix->ksize(); }
uint32_t ksh_index_scaled(const ksh_index *ix) { return ix->ix->scaled(); }
int ksh_index_store_raw_sequences(const ksh_index *ix) { return ix->ix->store_raw_sequences() ? 1 : 0; }

static int copy_out(const std::string &s, char *out, size_t cap) {
    if (!out || !cap) return 1;
    strncpy(out, s.c_str(), cap - 1);
    out[cap - 1] = 0;
    return 0;
}
int ksh_index_moltype(const ksh_index *ix, char *out, size_t cap) { return copy_out(ix->ix->moltype(), out, cap); }
int ksh_index_path(const ksh_index *ix, char *out, size_t cap) { return copy_out(ix->ix->path(), out, cap); }
int ksh_index_generate_filename(const ksh_index *ix, const char *base_name, char *out, size_t cap) {
    return copy_out(ix->ix->generate_filename(base_name), out, cap);
}

int ksh_index_dump_json(const ksh_index *ix, int with_kmers, char **json_out) {
    std::ostringstream o;
    o << "{\"ksize\":" << ix->ix->ksize() << ",\"scaled\":" << ix->ix->scaled() << ",\"moltype\":";
    json_str(o, ix->ix->moltype());
    o << ",\"combined_mins\":[";
    auto &cm = ix->ix->combined_mins();
    for (size_t i = 0; i < cm.size(); i++) o << (i ? "," : "") << cm[i];
    o << "],\"combined_abunds\":[";
    auto &ca = ix->ix->combined_abunds();
    for (size_t i = 0; i < ca.size(); i++) o << (i ? "," : "") << ca[i];
    o << "],\"signatures\":{";
    bool first = true;
    for (auto &kv : ix->ix->get_signatures()) {
        o << (first ? "" : ","); json_str(o, kv.first); o << ":";
        json_sig(o, kv.second, with_kmers != 0);
        first = false;
    }
    o << "}}";
    *json_out = dup_string(o.str());
    return *json_out ? 0 : 1;
}

int ksh_index_is_equivalent_to(const ksh_index *a, const ksh_index *b, int *equal, char *err, size_t err_cap) {
    KSH_GUARD(*equal = a->ix->is_equivalent_to(*b->ix) ? 1 : 0)
}

int ksh_index_save_state(ksh_index *ix, char *err, size_t err_cap) { KSH_GUARD(ix->ix->save_state()) }

int ksh_index_load(const char *path, int device, ksh_index **out, char *err, size_t err_cap) {
    KSH_GUARD({
        auto h = new ksh_index();
        try { h->ix = ProteomeIndex::load(path, device); } catch (...) { delete h; throw; }
        *out = h;
    })
}

void ksh_string_free(char *s) { free(s); }

} // extern "C"
