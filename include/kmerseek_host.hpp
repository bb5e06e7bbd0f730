// kmerseek_host.hpp — C++ host-side mirror of the reference's Rust `ProteomeIndex` API
// (src/rust/index.rs:58-1017, 2975-3061; signature.rs; kmer.rs; errors.rs), written ABOVE the C ABI of
// include/kmerseek_amd.h: every sketch / k-mer-position / union computation is one batched call into the
// HIP library; this layer only validates, packs, groups and stores.  Same method names, argument meaning and
// error text as the reference so its tests translate line by line (tests/test_host_index.py).
//
// Differences that are deliberate:
//   * persistence is a flat file in this library's own format, not RocksDB (storage is out of
//     scope; `save_state` / `load_state` keep the API shape and the `NoSavedState` error);
//   * B/Z/J resolution draws from a seeded SplitMix64 stream instead of rand::rng() (aminoacid.rs:48);
//   * FASTA input: plain or gzip (zlib).  bz2 / xz / zstd inputs raise ParseError (needletail's niffler is absent).
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

struct ks_ctx;

namespace kmerseek {

constexpr uint64_t SEED = 42;                    // src/rust/signature.rs:12
constexpr uint32_t PROTEIN_TO_MINHASH_RATIO = 3; // src/rust/signature.rs:13

// src/rust/errors.rs:4-56 — what() carries the reference's Display text
class IndexError : public std::runtime_error {
  public:
    enum Kind { Database, InvalidMoltype, InvalidAminoAcid, InvalidKsize, NoSavedState, Io, Utf8, FastaParsing,
                BuilderError, SourmashError, ParseError, ValidationError, Gpu };
    IndexError(Kind k, const std::string &msg) : std::runtime_error(msg), kind(k) {}
    Kind kind;
    char residue = 0;     // InvalidAminoAcid
    size_t position = 0;  // InvalidAminoAcid: 1-based position in the output so far (aminoacid.rs:86)
    size_t seq_index = 0; // which record of the batch
};

// src/rust/kmer.rs:6-12
struct KmerInfo {
    size_t ksize = 0;
    uint64_t hashval = 0;
    std::string encoded_kmer;
    std::map<std::string, std::vector<size_t>> original_kmer_to_position;
    size_t unique_kmer_count() const { return original_kmer_to_position.size(); }
    size_t total_occurrences() const;
};

// src/rust/signature.rs:104-318 (the parts the index path uses)
class ProteinSignature {
  public:
    std::string name;
    std::string md5sum; // hex of the wrapping u64 sum of mins (signature.rs:277-279)
    std::string moltype;
    uint32_t protein_ksize = 0;
    uint32_t scaled = 0;
    std::vector<uint64_t> mins;   // ascending
    std::vector<uint64_t> abunds; // same length
    std::unordered_map<uint64_t, KmerInfo> kmer_infos;
    std::optional<std::string> raw_sequence; // store_raw_sequences (index.rs:737-743)

    uint32_t minhash_ksize() const { return protein_ksize * PROTEIN_TO_MINHASH_RATIO; }
    bool has_efficient_data() const { return raw_sequence.has_value(); }
    const std::string *get_raw_sequence() const { return raw_sequence ? &*raw_sequence : nullptr; }
};

class ProteomeIndexBuilder;

class ProteomeIndex {
  public:
    // ProteomeIndex::new (index.rs:130-197).  `device` picks the GPU (the reference has no such argument).
    ProteomeIndex(const std::string &path, uint32_t ksize, uint32_t scaled, const std::string &moltype,
                  bool store_raw_sequences, int device = 0);
    ~ProteomeIndex();
    ProteomeIndex(const ProteomeIndex &) = delete;
    ProteomeIndex &operator=(const ProteomeIndex &) = delete;

    static ProteomeIndexBuilder builder(); // index.rs:126
    // index.rs:655-673: <parent>/<file>.{moltype}.k{k}.scaled{s}.kmerseek.rocksdb
    static std::unique_ptr<ProteomeIndex> new_with_auto_filename(const std::string &base_path, uint32_t ksize,
                                                                  uint32_t scaled, const std::string &moltype,
                                                                  bool store_raw_sequences, int device = 0);

    // index.rs:719-747 — validate/resolve, sketch, k-mer positions (one GPU batch of size 1)
    ProteinSignature create_protein_signature(const std::string &sequence, const std::string &name);
    // the batched form the GPU wants: records = (sequence, name); upper = FASTA-path upper-casing (index.rs:1000).
    // First invalid record aborts the whole batch (index.rs:993-1008).
    std::vector<ProteinSignature> create_protein_signatures(const std::vector<std::pair<std::string, std::string>> &records,
                                                            bool upper);
    // index.rs:749-786 — fills sig.kmer_infos for `sequence`
    void process_kmers(const std::string &sequence, ProteinSignature &sig);
    void store_signatures(std::vector<ProteinSignature> sigs);               // index.rs:800-830
    void store_signatures_batch(const std::vector<ProteinSignature> &sigs);  // index.rs:850-857
    // index.rs:907-961 (plain / gzip FASTA); batch_size = records per GPU batch
    void process_fasta(const std::string &fasta_path, uint32_t progress_interval, size_t batch_size);

    void save_state();                                                   // index.rs:227-269 (own file format)
    static std::unique_ptr<ProteomeIndex> load(const std::string &path, int device = 0); // index.rs:430-511

    size_t signature_count() const { return signatures_.size(); }       // index.rs:514
    size_t combined_minhash_size() const { return combined_mins_.size(); } // index.rs:519
    const std::map<std::string, ProteinSignature> &get_signatures() const { return signatures_; } // index.rs:199
    const std::vector<uint64_t> &combined_mins() const { return combined_mins_; }
    const std::vector<uint64_t> &combined_abunds() const { return combined_abunds_; }
    bool is_equivalent_to(const ProteomeIndex &other) const;            // index.rs:524-625
    void print_stats() const;                                            // index.rs:628-639
    std::string generate_filename(const std::string &base_name) const;  // index.rs:647-652
    uint32_t ksize() const { return ksize_; }
    uint32_t scaled() const { return scaled_; }
    const std::string &moltype() const { return moltype_; }
    bool store_raw_sequences() const { return store_raw_; }
    const std::string &path() const { return path_; }
    void set_rng_seed(uint64_t s) { rng_seed_ = s; }

  private:
    std::string path_;
    uint32_t ksize_, scaled_;
    std::string moltype_;
    uint32_t moltype_id_ = 0;
    bool store_raw_;
    ks_ctx *ctx_ = nullptr;
    uint64_t rng_seed_ = 0x6b6d6572ULL;
    std::map<std::string, ProteinSignature> signatures_; // pseudo-md5 -> signature (same-key records overwrite, index.rs:817-820)
    std::vector<uint64_t> combined_mins_, combined_abunds_;
};

// index.rs:2975-3061
class ProteomeIndexBuilder {
  public:
    ProteomeIndexBuilder &path(const std::string &p) { path_ = p; return *this; }
    ProteomeIndexBuilder &ksize(uint32_t k) { ksize_ = k; return *this; }
    ProteomeIndexBuilder &scaled(uint32_t s) { scaled_ = s; return *this; }
    ProteomeIndexBuilder &moltype(const std::string &m) { moltype_ = m; return *this; }
    ProteomeIndexBuilder &store_raw_sequences(bool b) { store_raw_ = b; return *this; }
    ProteomeIndexBuilder &device(int d) { device_ = d; return *this; }
    std::unique_ptr<ProteomeIndex> build() const;                    // "Database path is required", ...
    std::unique_ptr<ProteomeIndex> build_with_auto_filename() const; // "Base path is required", ...
  private:
    std::optional<std::string> path_, moltype_;
    std::optional<uint32_t> ksize_, scaled_;
    bool store_raw_ = false;
    int device_ = 0;
};

} // namespace kmerseek
