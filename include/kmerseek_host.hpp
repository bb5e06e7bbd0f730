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
//   * FASTA input: plain, gzip, zstd, bzip2 or xz, told apart by magic number as needletail does (ks_input.cpp; libzstd / libbz2 /
//     liblzma are bound at run time); a truncated archive raises ParseError;
//   * `search` / `search_fasta` are ADDED: the reference's crate builds and stores signatures and has no
//     search of its own (index.rs ends at :1017); its only search is branchwater manysearch on the Python side
//     (src/python/kmerseek/search.py:125-141), whose rows these methods return for the signatures the index holds.
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

struct ks_ctx;
struct ks_index;
struct ks_sketches;

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

// One row of a search: the 22 columns branchwater manysearch writes for a (query, match) pair that shares at least one
// hash (src/python/kmerseek/search.py:125-141; values pinned by the reference's tests/test_search.py:33-39).
// Integer columns come from the GPU join (intersect_hashes, n_weighted_found); the ratios are f64 arithmetic on them.
struct SearchResult {
    std::string query_name, query_md5, match_name, match_md5, moltype; // *_md5: sourmash md5sum = MD5(str(3k) || str(min) ...)
    uint64_t intersect_hashes = 0;
    uint32_t ksize = 0;  // 3 * protein k-mer size, as sourmash reports it
    uint32_t scaled = 0;
    double containment = 0, jaccard = 0, max_containment = 0;
    double average_abund = 0, median_abund = 0, std_abund = 0; // over the match's abundances of the shared hashes (population std)
    double query_containment_ani = 0, match_containment_ani = 0, average_containment_ani = 0, max_containment_ani = 0;
    uint64_t n_weighted_found = 0, total_weighted_hashes = 0;
    double containment_target_in_query = 0, f_weighted_target_in_query = 0;
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

    // ---- search (added; see the header comment).  The stored signatures (get_signatures, index.rs:642-652) are the targets:
    // they are laid out once as a device-resident ks_index (rebuilt after the next store_signatures) and every call is one
    // ks_sketch_search of the whole query batch.  Queries go through the same pre-step as create_protein_signature
    // (validate / resolve; `upper` = the FASTA path's upper-casing); rows come out ordered by (query, match key).
    std::vector<SearchResult> search(const std::vector<std::pair<std::string, std::string>> &queries, bool upper = false);
    // every record of a FASTA file (plain / gzip / zstd / bzip2 / xz) as queries, `batch_size` records per GPU batch
    std::vector<SearchResult> search_fasta(const std::string &fasta_path, size_t batch_size = 100000);
    // the two-line form a consumer streams into the reference's CSV: column names in file order
    static const std::vector<std::string> &search_columns();

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
    // device-resident search index over signatures_ (targets in map order), built on first use by ensure_device_index()
    ks_sketches *dev_targets_ = nullptr;
    ks_index *dev_index_ = nullptr;
    std::vector<const ProteinSignature *> dev_order_; // target id -> signature
    std::vector<std::string> dev_md5_;                // target id -> sourmash md5sum (filled on demand)
    std::vector<uint64_t> dev_total_abund_;           // target id -> sum of abundances
    void ensure_device_index();
    void drop_device_index();
    // validate / resolve + pack: the host pre-step shared by create_protein_signatures and search
    void prepare_records(const std::vector<std::pair<std::string, std::string>> &records, bool upper,
                         std::vector<std::string> &processed, std::vector<uint64_t> &offs, std::vector<uint8_t> &res);
    void for_each_fasta_batch(const std::string &fasta_path, size_t batch_size, uint32_t progress_interval,
                              const std::function<void(std::vector<std::pair<std::string, std::string>> &)> &fn, size_t *n_records);
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
