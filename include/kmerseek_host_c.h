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

#ifdef __cplusplus
}
#endif
#endif
