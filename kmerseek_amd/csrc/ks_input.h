// ks_input.h — byte stream over a plain, gzip or zstd file, chosen by magic number.
//
// The reference reads FASTA through needletail's parse_fastx_file, which sniffs the compression and decompresses
// transparently (src/rust/index.rs:907-961; tests at index.rs:1734-1788 for .zst, :1790-1845 for .gz).  Here: plain = fread,
// gzip = zlib, zstd = libzstd's streaming API bound at run time (the image ships libzstd.so.1 without headers, so the
// six entry points used are declared in ks_input.cpp and resolved with dlopen).  bzip2 / xz are refused by name.
// A truncated gzip or zstd stream is an ERROR, not a shorter file.
#pragma once
#include <cstddef>
#include <string>

class KsInput {
  public:
    // nullptr + err on failure (cannot open / unsupported compression / libzstd missing)
    static KsInput *open(const char *path, std::string &err);
    virtual ~KsInput() {}
    // bytes read (> 0), 0 at end of stream, -1 on error (error() says why; a truncated archive is an error)
    virtual long read(void *dst, size_t cap) = 0;
    virtual const char *format() const = 0; // "plain" | "gzip" | "zstd"
    const std::string &error() const { return err_; }

  protected:
    std::string err_;
};
