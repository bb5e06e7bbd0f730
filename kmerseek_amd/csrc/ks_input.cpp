// ks_input.cpp — plain / gzip / zstd byte streams behind one read() (see ks_input.h).
#include "ks_input.h"

#include <dlfcn.h>
#include <zlib.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/kmerseek_host_c.h"

namespace {

class PlainInput : public KsInput {
  public:
    explicit PlainInput(FILE *f) : f_(f) {}
    ~PlainInput() override { fclose(f_); }
    long read(void *dst, size_t cap) override {
        const size_t n = fread(dst, 1, cap, f_);
        if (n == 0 && ferror(f_)) { err_ = std::string("read failed: ") + strerror(errno); return -1; }
        return (long)n;
    }
    const char *format() const override { return "plain"; }

  private:
    FILE *f_;
};

class GzipInput : public KsInput {
  public:
    explicit GzipInput(gzFile g) : g_(g) { gzbuffer(g_, 1 << 20); }
    ~GzipInput() override { gzclose(g_); }
    long read(void *dst, size_t cap) override {
        if (cap > (1u << 30)) cap = 1u << 30;
        const int n = gzread(g_, dst, (unsigned)cap);
        // gzread hands out what it has of a cut-off archive without a negative return: the stream state tells
        int zerr = Z_OK;
        const char *msg = gzerror(g_, &zerr);
        if (n < 0 || zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR) {
            err_ = std::string("gzip stream is truncated or corrupt (") + (msg && *msg ? msg : "unexpected end of file") + ")";
            return -1;
        }
        return n;
    }
    const char *format() const override { return "gzip"; }

  private:
    gzFile g_;
};

// ---- libzstd, bound at run time (streaming decompression API, stable since zstd 1.0) ----
struct zstd_in { const void *src; size_t size; size_t pos; };   // ZSTD_inBuffer
struct zstd_out { void *dst; size_t size; size_t pos; };        // ZSTD_outBuffer
struct ZstdApi {
    void *lib = nullptr;
    void *(*createDStream)() = nullptr;
    size_t (*freeDStream)(void *) = nullptr;
    size_t (*initDStream)(void *) = nullptr;
    size_t (*decompressStream)(void *, zstd_out *, zstd_in *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char *(*getErrorName)(size_t) = nullptr;
    bool ok = false;
};

const ZstdApi &zstd_api() {
    static ZstdApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.createDStream = (void *(*)())dlsym(api.lib, "ZSTD_createDStream");
        api.freeDStream = (size_t(*)(void *))dlsym(api.lib, "ZSTD_freeDStream");
        api.initDStream = (size_t(*)(void *))dlsym(api.lib, "ZSTD_initDStream");
        api.decompressStream = (size_t(*)(void *, zstd_out *, zstd_in *))dlsym(api.lib, "ZSTD_decompressStream");
        api.isError = (unsigned (*)(size_t))dlsym(api.lib, "ZSTD_isError");
        api.getErrorName = (const char *(*)(size_t))dlsym(api.lib, "ZSTD_getErrorName");
        api.ok = api.createDStream && api.freeDStream && api.initDStream && api.decompressStream && api.isError && api.getErrorName;
    });
    return api;
}

class ZstdInput : public KsInput {
  public:
    ZstdInput(FILE *f, void *ds) : f_(f), ds_(ds), in_(1 << 20) {}
    ~ZstdInput() override {
        zstd_api().freeDStream(ds_);
        fclose(f_);
    }
    long read(void *dst, size_t cap) override {
        const ZstdApi &z = zstd_api();
        zstd_out o{dst, cap, 0};
        while (o.pos == 0) {
            if (pos_ == len_ && !file_eof_) {
                len_ = fread(in_.data(), 1, in_.size(), f_);
                pos_ = 0;
                if (len_ == 0) {
                    if (ferror(f_)) { err_ = std::string("read failed: ") + strerror(errno); return -1; }
                    file_eof_ = true;
                }
            }
            if (pos_ == len_ && file_eof_) {
                // no more input: fine only between frames (last return 0 = frame complete and flushed)
                if (last_ != 0) { err_ = "zstd stream is truncated (unexpected end of file inside a frame)"; return -1; }
                return 0;
            }
            zstd_in i{in_.data(), len_, pos_};
            const size_t r = z.decompressStream(ds_, &o, &i);
            pos_ = i.pos;
            if (z.isError(r)) { err_ = std::string("zstd stream is corrupt (") + z.getErrorName(r) + ")"; return -1; }
            last_ = r;
        }
        return (long)o.pos;
    }
    const char *format() const override { return "zstd"; }

  private:
    FILE *f_;
    void *ds_;
    std::vector<unsigned char> in_;
    size_t pos_ = 0, len_ = 0, last_ = 0;
    bool file_eof_ = false;
};

// ---- libbz2 and liblzma, bound at run time like libzstd (needletail's default `compression` feature reads bzip2 and xz
// as well: src/rust/index.rs:920 parse_fastx_file).  Stream structs as bzlib.h / lzma/base.h declare them (stable ABIs).
struct bz_stream_t {
    char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
struct Bz2Api {
    void *lib = nullptr;
    int (*init)(bz_stream_t *, int, int) = nullptr;
    int (*run)(bz_stream_t *) = nullptr;
    int (*end)(bz_stream_t *) = nullptr;
    bool ok = false;
};
const Bz2Api &bz2_api() {
    static Bz2Api api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.init = (int (*)(bz_stream_t *, int, int))dlsym(api.lib, "BZ2_bzDecompressInit");
        api.run = (int (*)(bz_stream_t *))dlsym(api.lib, "BZ2_bzDecompress");
        api.end = (int (*)(bz_stream_t *))dlsym(api.lib, "BZ2_bzDecompressEnd");
        api.ok = api.init && api.run && api.end;
    });
    return api;
}

class Bz2Input : public KsInput {
  public:
    explicit Bz2Input(FILE *f) : f_(f), in_(1 << 20) { memset(&s_, 0, sizeof s_); live_ = bz2_api().init(&s_, 0, 0) == 0; }
    ~Bz2Input() override {
        if (live_) bz2_api().end(&s_);
        fclose(f_);
    }
    long read(void *dst, size_t cap) override {
        const Bz2Api &z = bz2_api();
        if (cap > 0x40000000u) cap = 0x40000000u;
        s_.next_out = (char *)dst; s_.avail_out = (unsigned)cap;
        while (s_.avail_out == cap) {
            if (s_.avail_in == 0 && !file_eof_) {
                const size_t n = fread(in_.data(), 1, in_.size(), f_);
                if (n == 0) {
                    if (ferror(f_)) { err_ = std::string("read failed: ") + strerror(errno); return -1; }
                    file_eof_ = true;
                }
                s_.next_in = (char *)in_.data(); s_.avail_in = (unsigned)n;
            }
            if (!live_) { // between streams: more input starts another one (concatenated archives are one file)
                if (s_.avail_in == 0 && file_eof_) return 0;
                if (z.init(&s_, 0, 0) != 0) { err_ = "BZ2_bzDecompressInit failed"; return -1; }
                live_ = true;
            }
            if (s_.avail_in == 0 && file_eof_) { err_ = "bzip2 stream is truncated (unexpected end of file)"; return -1; }
            const int r = z.run(&s_);
            if (r == 4) { z.end(&s_); live_ = false; } // BZ_STREAM_END
            else if (r != 0) { err_ = "bzip2 stream is corrupt (code " + std::to_string(r) + ")"; return -1; }
        }
        return (long)(cap - s_.avail_out);
    }
    const char *format() const override { return "bzip2"; }

  private:
    FILE *f_;
    bz_stream_t s_;
    std::vector<unsigned char> in_;
    bool live_ = false, file_eof_ = false;
};

struct lzma_stream_t {
    const uint8_t *next_in; size_t avail_in; uint64_t total_in;
    uint8_t *next_out; size_t avail_out; uint64_t total_out;
    const void *allocator; void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
struct XzApi {
    void *lib = nullptr;
    int (*decoder)(lzma_stream_t *, uint64_t, uint32_t) = nullptr;
    int (*code)(lzma_stream_t *, int) = nullptr;
    void (*end)(lzma_stream_t *) = nullptr;
    bool ok = false;
};
const XzApi &xz_api() {
    static XzApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"liblzma.so.5", "liblzma.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.decoder = (int (*)(lzma_stream_t *, uint64_t, uint32_t))dlsym(api.lib, "lzma_stream_decoder");
        api.code = (int (*)(lzma_stream_t *, int))dlsym(api.lib, "lzma_code");
        api.end = (void (*)(lzma_stream_t *))dlsym(api.lib, "lzma_end");
        api.ok = api.decoder && api.code && api.end;
    });
    return api;
}

class XzInput : public KsInput {
  public:
    explicit XzInput(FILE *f) : f_(f), in_(1 << 20) {
        memset(&s_, 0, sizeof s_); // LZMA_STREAM_INIT
        live_ = xz_api().decoder(&s_, ~0ULL, 0x08 /* LZMA_CONCATENATED */) == 0;
    }
    ~XzInput() override {
        if (live_) xz_api().end(&s_);
        fclose(f_);
    }
    long read(void *dst, size_t cap) override {
        const XzApi &z = xz_api();
        if (!live_) { err_ = "lzma_stream_decoder failed"; return -1; }
        if (done_) return 0;
        s_.next_out = (uint8_t *)dst; s_.avail_out = cap;
        while (s_.avail_out == cap) {
            if (s_.avail_in == 0 && !file_eof_) {
                const size_t n = fread(in_.data(), 1, in_.size(), f_);
                if (n == 0) {
                    if (ferror(f_)) { err_ = std::string("read failed: ") + strerror(errno); return -1; }
                    file_eof_ = true;
                }
                s_.next_in = in_.data(); s_.avail_in = n;
            }
            const int r = z.code(&s_, file_eof_ ? 3 /* LZMA_FINISH */ : 0 /* LZMA_RUN */);
            if (r == 1) { done_ = true; break; } // LZMA_STREAM_END (with LZMA_CONCATENATED: only once the input is finished)
            if (r == 10 || (r == 0 && file_eof_ && s_.avail_in == 0 && s_.avail_out == cap)) { // LZMA_BUF_ERROR: no progress possible
                err_ = "xz stream is truncated (unexpected end of file)";
                return -1;
            }
            if (r != 0) { err_ = "xz stream is corrupt (code " + std::to_string(r) + ")"; return -1; }
        }
        return (long)(cap - s_.avail_out);
    }
    const char *format() const override { return "xz"; }

  private:
    FILE *f_;
    lzma_stream_t s_;
    std::vector<unsigned char> in_;
    bool live_ = false, file_eof_ = false, done_ = false;
};

} // namespace

KsInput *KsInput::open(const char *path, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path + ": " + strerror(errno); return nullptr; }
    unsigned char m[6] = {0};
    const size_t got = fread(m, 1, 6, f);
    if (got >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') {
        if (!bz2_api().ok) { fclose(f); err = std::string("bzip2 input, but libbz2.so.1.0 could not be loaded: ") + path; return nullptr; }
        if (fseek(f, 0, SEEK_SET) != 0) { fclose(f); err = std::string("cannot rewind ") + path; return nullptr; }
        return new Bz2Input(f);
    }
    if (got >= 6 && m[0] == 0xfd && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0) {
        if (!xz_api().ok) { fclose(f); err = std::string("xz input, but liblzma.so.5 could not be loaded: ") + path; return nullptr; }
        if (fseek(f, 0, SEEK_SET) != 0) { fclose(f); err = std::string("cannot rewind ") + path; return nullptr; }
        return new XzInput(f);
    }
    if (got >= 2 && m[0] == 0x1f && m[1] == 0x8b) {
        fclose(f);
        gzFile g = gzopen(path, "rb");
        if (!g) { err = std::string("cannot open ") + path; return nullptr; }
        return new GzipInput(g);
    }
    if (fseek(f, 0, SEEK_SET) != 0) { fclose(f); err = std::string("cannot rewind ") + path; return nullptr; }
    if (got >= 4 && m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) {
        const ZstdApi &z = zstd_api();
        if (!z.ok) { fclose(f); err = std::string("zstd input, but libzstd.so.1 could not be loaded: ") + path; return nullptr; }
        void *ds = z.createDStream();
        if (!ds || z.isError(z.initDStream(ds))) { if (ds) z.freeDStream(ds); fclose(f); err = "ZSTD_createDStream failed"; return nullptr; }
        return new ZstdInput(f, ds);
    }
    return new PlainInput(f);
}

// ---- C shim (include/kmerseek_host_c.h): the decompression layer on its own, usable without a GPU ----
extern "C" int ksh_input_decompress(const char *path, uint8_t **data, uint64_t *len, char *format, uint32_t format_cap, char *err,
                                    uint32_t err_cap) {
    auto put = [](char *dst, uint32_t cap, const std::string &s) {
        if (dst && cap) { strncpy(dst, s.c_str(), cap - 1); dst[cap - 1] = 0; }
    };
    if (!path || !data || !len) { if (err && err_cap) { strncpy(err, "NULL argument", err_cap - 1); err[err_cap - 1] = 0; } return 1; }
    *data = nullptr; *len = 0;
    try { // (nothing throws across the boundary: the strings below may allocate)
    std::string e;
    KsInput *in = KsInput::open(path, e);
    if (!in) { put(err, err_cap, "Parse error: " + e); return 11; }
    put(format, format_cap, in->format());
    size_t cap = 1 << 20, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    int rc = 0;
    while (buf) {
        if (n == cap) {
            cap *= 2;
            uint8_t *q = (uint8_t *)realloc(buf, cap);
            if (!q) { free(buf); buf = nullptr; break; }
            buf = q;
        }
        const long r = in->read(buf + n, cap - n);
        if (r < 0) { put(err, err_cap, "Parse error: " + in->error()); rc = 11; break; }
        if (r == 0) break;
        n += (size_t)r;
    }
    delete in;
    if (!buf) { put(err, err_cap, "out of host memory"); return 13; }
    if (rc) { free(buf); return rc; }
    *data = buf; *len = n;
    return 0;
    } catch (...) {
        if (err && err_cap) { strncpy(err, "out of host memory", err_cap - 1); err[err_cap - 1] = 0; }
        return 13;
    }
}

extern "C" void ksh_input_free(uint8_t *data) { free(data); }
