/*
 * kmerseek_host_c.h — flat C shim over the C++ host mirror (include/kmerseek_host.hpp), so Python
 * (kmerseek_amd/host.py: PyProteomeIndex & co, mirroring src/rust/lib.rs:28-103) and tests can drive the
 * `ProteomeIndex` API.  It sits ABOVE the compute ABI of kmerseek_amd.h: nothing here launches a kernel
 * itself.  Every call returns 0 on success or (IndexError::Kind + 1) with the reference's Display text in err.
 */
#ifndef KMERSEEK_HOST_C_H
#define KMERSEEK_HOST_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ksh_index ksh_index;

/* ProteomeIndex::new (src/rust/index.rs:130-136) / new_with_auto_filename (:655-673) */
int ksh_index_new(const char *path, uint32_t ksize, uint32_t scaled, const char *moltype, int store_raw_sequences,
                  int device, int auto_filename, ksh_index **out, char *err, size_t err_cap);
/* ProteomeIndexBuilder (src/rust/index.rs:2975-3061): NULL path / moltype and has_* = 0 mean "not set" */
int ksh_index_build(const char *path, int has_ksize, uint32_t ksize, int has_scaled, uint32_t scaled,
                    const char *moltype, int store_raw_sequences, int auto_filename, int device, ksh_index **out,
                    char *err, size_t err_cap);
void ksh_index_free(ksh_index *ix);

/* create_protein_signature (:719-747); the signature comes back as JSON (ksh_string_free), optionally stored too */
int ksh_index_create_signature(ksh_index *ix, const char *sequence, const char *name, int store, char **json_out,
                               char *err, size_t err_cap);
/* batched create + store_signatures_batch (:850-857); upper = FASTA-path upper-casing (:1000) */
int ksh_index_add_records(ksh_index *ix, const char *const *sequences, const char *const *names, uint32_t n, int upper,
                          char *err, size_t err_cap);
/* process_fasta (:907-961) */
int ksh_index_process_fasta(ksh_index *ix, const char *fasta_path, uint32_t progress_interval, uint64_t batch_size,
                            char *err, size_t err_cap);
/* search (added — SURVEY 8(b); see include/kmerseek_host.hpp): (sequence, name) query records, or every record of a FASTA
 * file, against the signatures the index holds.  One GPU call per batch (ks_sketch_search against a device-resident ks_index
 * built once from the stored signatures and dropped by the next store).  *json_out (ksh_string_free): an array of row objects
 * with the 22 columns of branchwater manysearch (src/python/kmerseek/search.py:125-141), f64 columns printed with 17
 * significant digits.  upper: FASTA-path upper-casing of the queries (src/rust/index.rs:1000); an invalid residue is the same
 * error create_protein_signature raises. */
int ksh_index_search(ksh_index *ix, const char *const *sequences, const char *const *names, uint32_t n, int upper,
                     char **json_out, char *err, size_t err_cap);
int ksh_index_search_fasta(ksh_index *ix, const char *fasta_path, uint64_t batch_size, char **json_out, char *err,
                           size_t err_cap);
/* md5sum field of a sourmash signature — MD5(ascii(3 * protein_ksize) || ascii decimal of every min in order) — as the search rows'
 * query_md5 / match_md5 columns hold it; out33 receives 32 hex digits + NUL.  Needs no GPU. */
int ksh_sourmash_md5(const uint64_t *mins, uint64_t n, uint32_t protein_ksize, char *out33);
uint64_t ksh_index_signature_count(const ksh_index *ix);
uint64_t ksh_index_combined_minhash_size(const ksh_index *ix);
uint32_t ksh_index_ksize(const ksh_index *ix);
uint32_t ksh_index_scaled(const ksh_index *ix);
int ksh_index_store_raw_sequences(const ksh_index *ix);
/* moltype / path / generate_filename(base_name) copied into out (NUL-terminated, truncated to cap) */
int ksh_index_moltype(const ksh_index *ix, char *out, size_t cap);
int ksh_index_path(const ksh_index *ix, char *out, size_t cap);
int ksh_index_generate_filename(const ksh_index *ix, const char *base_name, char *out, size_t cap);
/* all stored signatures (and the combined sketch) as one JSON document; with_kmers adds kmer_infos */
int ksh_index_dump_json(const ksh_index *ix, int with_kmers, char **json_out);
int ksh_index_is_equivalent_to(const ksh_index *a, const ksh_index *b, int *equal, char *err, size_t err_cap);
int ksh_index_save_state(ksh_index *ix, char *err, size_t err_cap);
int ksh_index_load(const char *path, int device, ksh_index **out, char *err, size_t err_cap);
void ksh_string_free(char *s);

/* ---- pipelined FASTA ingest (kmerseek_amd/csrc/ks_ingest.cpp) -------------------------------------
 * FASTA file (plain, gzip or zstd) -> the sketches of all its records as one host CSR, with the stages running concurrently:
 * reader thread -> validate / pack threads into pinned buffers -> H2D on a copy stream -> ks_sketch_batch_device + D2H.
 *   validate = 1: upper-case + AminoAcidAmbiguity::validate_and_resolve per record, first bad residue aborts with the
 *                 reference's message (the Rust index path: src/rust/index.rs:984-1016, aminoacid.rs:74-105);
 *   validate = 0: raw record bytes, as sourmash_plugin_branchwater.do_manysketch takes them (src/python/kmerseek/sketch.py:28-40).
 *   batch_residues: residues per device batch (0 = 16 MiB); pipeline = 0 runs the stages back to back (baseline). */
typedef struct ksh_fasta_sketches ksh_fasta_sketches;
int ksh_sketch_fasta(const char *fasta_path, uint32_t ksize, uint32_t scaled, const char *moltype, int validate, int device,
                     uint64_t batch_residues, int pipeline, ksh_fasta_sketches **out, char *err, size_t err_cap);
uint64_t ksh_fs_n_records(const ksh_fasta_sketches *r);
uint64_t ksh_fs_n_hashes(const ksh_fasta_sketches *r);
const uint64_t *ksh_fs_offsets(const ksh_fasta_sketches *r); /* n_records + 1 */
const uint64_t *ksh_fs_hashes(const ksh_fasta_sketches *r);
const uint32_t *ksh_fs_abunds(const ksh_fasta_sketches *r);
const char *ksh_fs_names(ksh_fasta_sketches *r, uint64_t *len); /* record ids joined by '\n' */
/* seconds[6] = wall, reader busy, validate/pack busy, H2D busy, device (sketch + D2H to pinned) busy, collector busy */
void ksh_fs_stats(const ksh_fasta_sketches *r, uint64_t *n_residues, uint64_t *n_windows, uint64_t *n_batches, double *seconds);
void ksh_fs_free(ksh_fasta_sketches *r);

/* ---- input layer on its own (kmerseek_amd/csrc/ks_input.cpp): plain / gzip / zstd by magic number, as needletail's
 * parse_fastx_file does for the reference (src/rust/index.rs:907-961, tested at :1734-1845).  Whole file decompressed into
 * a malloc'ed buffer (ksh_input_free); format = "plain" | "gzip" | "zstd".  A truncated archive is an error (11), never a
 * shorter file.  Needs no GPU. */
int ksh_input_decompress(const char *path, uint8_t **data, uint64_t *len, char *format, uint32_t format_cap, char *err,
                         uint32_t err_cap);
void ksh_input_free(uint8_t *data);

#ifdef __cplusplus
}
#endif
#endif
