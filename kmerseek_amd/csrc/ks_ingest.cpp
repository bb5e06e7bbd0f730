// ks_ingest.cpp — pipelined FASTA ingest above the compute ABI.
//
// The reference reads a FASTA file record by record (needletail, src/rust/index.rs:907-961), collects 1000 records,
// runs them through rayon and only then reads on.  At GPU sketching rates the parser and the PCIe copies are the
// whole cost, so here the stages run concurrently, each in its own thread, connected by bounded queues:
//
//   reader    plain / gzip / zstd FASTA (ks_input.h) -> raw record batches of >= batch_residues residues
//   packer    validate_and_resolve (aminoacid.rs:74-105; upper-cased first as index.rs:1000 does) on host threads, or
//             raw record bytes as manysketch takes them (sketch.py:28-40) -> residues + offsets in a PINNED slot
//   uploader  hipMemcpyAsync of the slot on its own stream (overlaps the kernels of the previous batch)
//   device    ks_sketch_batch_device on the context's stream, D2H of the batch's CSR into the slot's pinned staging
//   collector appends the staged CSR to the result arrays (page-faulting fresh host memory is the slowest step of
//             all at scaled = 1: 12 B per window against 1 B of input — so it gets a thread of its own)
//
// `pipeline = 0` runs the same five steps one after the other in the calling thread (the baseline the overlap is
// measured against).  Nothing here computes a hash on the CPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kmerseek_amd.h"
#include "../../include/kmerseek_host_c.h"
#include "ks_input.h"

namespace {

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

template <typename T>
class BoundedQueue {
  public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    bool push(T v) { // false once closed
        std::unique_lock<std::mutex> l(m_);
        not_full_.wait(l, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return false;
        q_.push_back(std::move(v));
        not_empty_.notify_one();
        return true;
    }
    bool pop(T &out) { // false when closed and drained
        std::unique_lock<std::mutex> l(m_);
        not_empty_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m_);
        closed_ = true;
        not_empty_.notify_all();
        not_full_.notify_all();
    }

  private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

struct RawBatch { // records as the reader found them: sequence bytes concatenated (line breaks removed)
    std::vector<uint8_t> res;
    std::vector<uint64_t> offs{0};
    std::vector<std::string> names;
};

struct Slot { // one batch in flight: pinned host staging + its device copy
    uint8_t *h_res = nullptr, *d_res = nullptr;
    uint64_t *h_offs = nullptr, *d_offs = nullptr;
    size_t res_cap = 0, offs_cap = 0;
    size_t n_res = 0;
    uint32_t n_seqs = 0, max_len = 0;
    std::vector<std::string> names;
    // pinned landing zone of the batch's sketches (kept hashes <= windows <= residues)
    uint64_t *h_coffs = nullptr, *h_hash = nullptr;
    uint32_t *h_abund = nullptr;
    size_t out_cap = 0, coffs_cap = 0;
    uint64_t n_hashes = 0, n_windows = 0;
};

template <typename T>
struct Grow { // result array: realloc-managed (large blocks move by mremap, nothing is zero-filled)
    T *p = nullptr;
    size_t n = 0, cap = 0;
    bool reserve_more(size_t extra) {
        if (n + extra <= cap) return true;
        size_t nc = cap ? cap : 1024;
        while (nc < n + extra) nc += nc / 2 + 1024;
        T *q = (T *)realloc(p, nc * sizeof(T));
        if (!q) return false;
        p = q; cap = nc;
        return true;
    }
    ~Grow() { free(p); }
};

struct Failure {
    std::mutex m;
    int code = 0;
    std::string msg;
    std::atomic<bool> set{false};
    void raise(int c, const std::string &s) {
        std::lock_guard<std::mutex> l(m);
        if (!set.load()) { code = c; msg = s; set.store(true); }
    }
};

} // namespace

struct ksh_fasta_sketches {
    std::vector<std::string> names;
    Grow<uint64_t> offsets; // CSR over all records of the file (n_records + 1 entries)
    Grow<uint64_t> hashes;
    Grow<uint32_t> abunds;
    std::string names_blob;           // '\n'-joined, built on demand
    uint64_t n_residues = 0, n_windows = 0, n_batches = 0;
    double t_read = 0, t_pack = 0, t_h2d = 0, t_device = 0, t_collect = 0, t_wall = 0;
};

namespace {

class Ingest {
  public:
    Ingest(const char *path, uint32_t ksize, uint32_t scaled, uint32_t moltype, int validate, int device, uint64_t batch_residues,
           int pipeline, ksh_fasta_sketches *out)
        : path_(path), validate_(validate != 0), device_(device), batch_res_(batch_residues ? batch_residues : (16u << 20)),
          pipeline_(pipeline != 0), out_(out) {
        params_.ksize = ksize; params_.scaled = scaled; params_.moltype = moltype; params_.flags = 0; params_.seed = 42;
    }
    ~Ingest() {
        for (auto &s : slots_) {
            if (s.h_res) (void)hipHostFree(s.h_res);
            if (s.h_offs) (void)hipHostFree(s.h_offs);
            if (s.d_res) (void)hipFree(s.d_res);
            if (s.d_offs) (void)hipFree(s.d_offs);
            if (s.h_coffs) (void)hipHostFree(s.h_coffs);
            if (s.h_hash) (void)hipHostFree(s.h_hash);
            if (s.h_abund) (void)hipHostFree(s.h_abund);
        }
        if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
        if (ctx_) ks_ctx_destroy(ctx_);
    }

    int run(std::string &err) {
        const auto t0 = clk::now();
        int st = ks_ctx_create(device_, nullptr, &ctx_);
        if (st != KS_OK) { err = "no HIP device (ks_ctx_create failed)"; return 13; }
        if (hipSetDevice(device_) != hipSuccess || hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking) != hipSuccess) {
            err = "hipStreamCreate failed";
            return 13;
        }
        {
            std::string oerr;
            in_.reset(KsInput::open(path_.c_str(), oerr));
            if (!in_) { err = "Parse error: " + oerr; return 11; }
        }
        slots_.resize(pipeline_ ? 3 : 1);
        if (!out_->offsets.reserve_more(1)) { err = "out of host memory"; return 13; }
        out_->offsets.p[0] = 0;
        out_->offsets.n = 1;
        if (pipeline_) run_pipelined(); else run_serial();
        in_.reset();
        out_->t_wall = secs(t0, clk::now());
        if (fail_.set.load()) { err = fail_.msg; return fail_.code; }
        return 0;
    }

  private:
    // ---- stage 1: reader ------------------------------------------------------------------------------------------
    // Bulk read + memchr over 4 MiB chunks; sequence bytes go straight into the batch's contiguous buffer (no
    // per-line or per-record strings: the line-by-line reader this replaced parsed 0.4 GB/s and WAS the pipeline).
    bool read_batch(RawBatch &b) { // false at end of file with nothing read
        const auto t0 = clk::now();
        b = RawBatch();
        b.res.reserve((size_t)batch_res_ + (batch_res_ >> 3) + (1u << 16));
        cur_ = &b;
        bool cut = false;
        while (!cut && !fail_.set.load()) {
            if (pos_ == len_) {
                if (eof_) break;
                const long n = in_->read(chunk_.data(), chunk_.size());
                if (n < 0) { fail_.raise(11, "Parse error: " + in_->error()); break; }
                if (n == 0) { eof_ = true; break; }
                pos_ = 0;
                len_ = (size_t)n;
            }
            while (pos_ < len_) {
                const uint8_t *p = chunk_.data() + pos_;
                const size_t left = len_ - pos_;
                if (kind_ == LINE_START) {
                    if (*p == '\n' || *p == '\r') { pos_++; continue; } // blank line
                    if (*p == '>') {
                        if (in_record_) {
                            end_record();
                            in_record_ = false;
                            // cut the batch between records once it is full (this '>' is looked at again next time)
                            if (b.res.size() >= batch_res_) { cut = true; break; }
                        }
                        kind_ = HEADER;
                        header_.clear();
                        pos_++;
                        continue;
                    }
                    if (!in_record_) { fail_.raise(11, "Parse error: FASTA record does not start with '>'"); break; }
                    kind_ = SEQ;
                }
                const uint8_t *nl = (const uint8_t *)memchr(p, '\n', left);
                const size_t take = nl ? (size_t)(nl - p) : left;
                if (kind_ == HEADER) header_.append((const char *)p, take);
                else b.res.insert(b.res.end(), p, p + take);
                pos_ += take + (nl ? 1 : 0);
                if (nl) {
                    if (kind_ == HEADER) {
                        while (!header_.empty() && header_.back() == '\r') header_.pop_back();
                        b.names.push_back(header_);
                        in_record_ = true;
                    }
                    kind_ = LINE_START;
                }
            }
            if (fail_.set.load()) break;
        }
        if (!cut && eof_ && !fail_.set.load()) { // end of file closes the last record (and an unterminated header line)
            if (kind_ == HEADER) {
                while (!header_.empty() && header_.back() == '\r') header_.pop_back();
                b.names.push_back(header_);
                in_record_ = true;
                kind_ = LINE_START;
            }
            if (in_record_) { end_record(); in_record_ = false; }
        }
        cur_ = nullptr;
        out_->t_read += secs(t0, clk::now());
        return b.offs.size() > 1;
    }
    void end_record() { // carriage returns of CRLF files are dropped from the record's bytes
        RawBatch &b = *cur_;
        const size_t start = (size_t)b.offs.back();
        uint8_t *q = b.res.data() + start;
        const size_t n = b.res.size() - start;
        if (n && memchr(q, '\r', n)) {
            size_t w = 0;
            for (size_t i = 0; i < n; i++)
                if (q[i] != '\r') q[w++] = q[i];
            b.res.resize(start + w);
        }
        b.offs.push_back(b.res.size());
    }

    // ---- stage 2: packer ------------------------------------------------------------------------------------------
    bool ensure_slot(Slot &s, size_t n_res, size_t n_seqs) {
        if (hipSetDevice(device_) != hipSuccess) return false;
        if (n_res + 16 > s.res_cap) {
            if (s.h_res) (void)hipHostFree(s.h_res);
            if (s.d_res) (void)hipFree(s.d_res);
            s.res_cap = (n_res + 16) + (n_res >> 3);
            if (hipHostMalloc((void **)&s.h_res, s.res_cap) != hipSuccess || hipMalloc((void **)&s.d_res, s.res_cap) != hipSuccess) return false;
        }
        if (n_seqs + 1 > s.offs_cap) {
            if (s.h_offs) (void)hipHostFree(s.h_offs);
            if (s.d_offs) (void)hipFree(s.d_offs);
            s.offs_cap = (n_seqs + 1) + (n_seqs >> 3);
            if (hipHostMalloc((void **)&s.h_offs, s.offs_cap * sizeof(uint64_t)) != hipSuccess ||
                hipMalloc((void **)&s.d_offs, s.offs_cap * sizeof(uint64_t)) != hipSuccess)
                return false;
        }
        if (n_seqs + 1 > s.coffs_cap) {
            if (s.h_coffs) (void)hipHostFree(s.h_coffs);
            s.coffs_cap = (n_seqs + 1) + (n_seqs >> 3);
            if (hipHostMalloc((void **)&s.h_coffs, s.coffs_cap * sizeof(uint64_t)) != hipSuccess) return false;
        }
        if (n_res + 1 > s.out_cap) { // kept hashes <= windows <= residues
            if (s.h_hash) (void)hipHostFree(s.h_hash);
            if (s.h_abund) (void)hipHostFree(s.h_abund);
            s.out_cap = (n_res + 1) + (n_res >> 3);
            if (hipHostMalloc((void **)&s.h_hash, s.out_cap * sizeof(uint64_t)) != hipSuccess ||
                hipHostMalloc((void **)&s.h_abund, s.out_cap * sizeof(uint32_t)) != hipSuccess)
                return false;
        }
        return true;
    }
    bool pack(RawBatch &b, Slot &s) {
        const auto t0 = clk::now();
        const size_t n = b.offs.size() - 1, total = b.res.size();
        if (!ensure_slot(s, total, n)) { fail_.raise(13, "pinned / device staging allocation failed"); return false; }
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16));
        if (total < (1u << 20)) nt = 1;
        if (!validate_) { // raw record bytes: one (split) copy into the pinned slot, offsets as read
            auto part = [&](unsigned t) {
                const size_t lo = total * t / nt, hi = total * (t + 1) / nt;
                memcpy(s.h_res + lo, b.res.data() + lo, hi - lo);
            };
            if (nt == 1) part(0);
            else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++) th.emplace_back(part, t);
                for (auto &t : th) t.join();
            }
            memcpy(s.h_offs, b.offs.data(), (n + 1) * sizeof(uint64_t));
        } else {
            // processed in place at the raw offset ('*' truncation only shortens), compacted below
            std::vector<uint64_t> out_len(n, 0);
            std::atomic<size_t> bad{(size_t)-1};
            std::vector<ks_residue_error> errs(n);
            auto work = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++) {
                    uint64_t ol = 0;
                    const int rc = ks_validate_and_resolve(b.res.data() + b.offs[i], b.offs[i + 1] - b.offs[i], 1,
                                                           0x6b6d6572ULL + 0x9e3779b97f4a7c15ULL * (records_done_ + i + 1),
                                                           s.h_res + b.offs[i], &ol, &errs[i]);
                    if (rc != KS_OK) {
                        size_t cur = bad.load();
                        while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
                        return;
                    }
                    out_len[i] = ol;
                }
            };
            if (nt == 1) work(0, n);
            else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
                for (auto &t : th) t.join();
            }
            if (bad.load() != (size_t)-1) { // first failing record aborts (index.rs:993-1008)
                const size_t i = bad.load();
                char msg[160];
                snprintf(msg, sizeof msg, "Invalid amino acid '%c' found at position %u", (char)errs[i].residue, errs[i].position);
                fail_.raise(3, msg);
                return false;
            }
            uint64_t o = 0;
            s.h_offs[0] = 0;
            for (size_t i = 0; i < n; i++) {
                if (o != b.offs[i]) memmove(s.h_res + o, s.h_res + b.offs[i], out_len[i]);
                o += out_len[i];
                s.h_offs[i + 1] = o;
            }
        }
        s.n_res = s.h_offs[n];
        s.n_seqs = (uint32_t)n;
        uint64_t mx = 0; // upper bound on the sequence length: the device call then plans its launches without a round trip
        for (size_t i = 0; i < n; i++) mx = std::max<uint64_t>(mx, s.h_offs[i + 1] - s.h_offs[i]);
        s.max_len = mx > 0xfffffff0ULL ? 0u : (uint32_t)mx;
        s.names = std::move(b.names);
        records_done_ += n;
        out_->t_pack += secs(t0, clk::now());
        return true;
    }

    // ---- stage 3: uploader ----------------------------------------------------------------------------------------
    bool upload(Slot &s) {
        const auto t0 = clk::now();
        bool ok = hipSetDevice(device_) == hipSuccess;
        if (ok && s.n_res) ok = hipMemcpyAsync(s.d_res, s.h_res, s.n_res, hipMemcpyHostToDevice, copy_stream_) == hipSuccess;
        if (ok) ok = hipMemcpyAsync(s.d_offs, s.h_offs, ((size_t)s.n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, copy_stream_) == hipSuccess;
        if (ok) ok = hipStreamSynchronize(copy_stream_) == hipSuccess;
        if (!ok) fail_.raise(13, "host-to-device copy failed");
        out_->t_h2d += secs(t0, clk::now());
        return ok;
    }

    // ---- stage 4: device ------------------------------------------------------------------------------------------
    bool sketch(Slot &s) {
        const auto t0 = clk::now();
        ks_sketches *S = nullptr;
        int st = ks_sketch_batch_device(ctx_, s.d_res, s.d_offs, s.n_seqs, s.n_res, s.max_len, &params_, &S);
        if (st != KS_OK) { fail_.raise(13, std::string("ks_sketch_batch_device: ") + ks_last_error(ctx_)); return false; }
        s.n_hashes = ks_sketches_n_hashes(S);
        s.n_windows = ks_sketches_n_windows(S);
        if (s.n_hashes > s.out_cap) { ks_sketches_free(S); fail_.raise(13, "sketch larger than its staging buffer"); return false; }
        st = ks_sketches_copy_to_host(ctx_, S, s.h_coffs, s.n_hashes ? s.h_hash : nullptr, s.n_hashes ? s.h_abund : nullptr);
        ks_sketches_free(S);
        if (st != KS_OK) { fail_.raise(13, std::string("ks_sketches_copy_to_host: ") + ks_last_error(ctx_)); return false; }
        out_->t_device += secs(t0, clk::now());
        return true;
    }

    // ---- stage 5: collector ---------------------------------------------------------------------------------------
    bool collect(Slot &s) {
        const auto t0 = clk::now();
        ksh_fasta_sketches &R = *out_;
        if (!R.offsets.reserve_more(s.n_seqs) || !R.hashes.reserve_more(s.n_hashes) || !R.abunds.reserve_more(s.n_hashes)) {
            fail_.raise(13, "out of host memory");
            return false;
        }
        const uint64_t base = R.offsets.p[R.offsets.n - 1];
        for (uint32_t i = 1; i <= s.n_seqs; i++) R.offsets.p[R.offsets.n - 1 + i] = base + s.h_coffs[i];
        R.offsets.n += s.n_seqs;
        // first touch of fresh pages is what costs here: split the copy over a few threads
        const unsigned nt = s.n_hashes > (1u << 22) ? 4 : 1;
        auto part = [&](unsigned t) {
            const size_t lo = s.n_hashes * t / nt, hi = s.n_hashes * (t + 1) / nt;
            memcpy(R.hashes.p + R.hashes.n + lo, s.h_hash + lo, (hi - lo) * sizeof(uint64_t));
            memcpy(R.abunds.p + R.abunds.n + lo, s.h_abund + lo, (hi - lo) * sizeof(uint32_t));
        };
        if (nt == 1) part(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(part, t);
            for (auto &t : th) t.join();
        }
        R.hashes.n += s.n_hashes;
        R.abunds.n += s.n_hashes;
        for (auto &nm : s.names) R.names.push_back(std::move(nm));
        s.names.clear();
        R.n_residues += s.n_res;
        R.n_windows += s.n_windows;
        R.n_batches++;
        R.t_collect += secs(t0, clk::now());
        return true;
    }

    void run_serial() {
        RawBatch b;
        while (!fail_.set.load() && read_batch(b)) {
            if (!pack(b, slots_[0]) || !upload(slots_[0]) || !sketch(slots_[0]) || !collect(slots_[0])) break;
        }
    }

    void run_pipelined() {
        BoundedQueue<RawBatch> q_raw(2);
        BoundedQueue<int> q_free(slots_.size() + 1), q_packed(slots_.size() + 1), q_dev(slots_.size() + 1), q_done(slots_.size() + 1);
        for (size_t i = 0; i < slots_.size(); i++) q_free.push((int)i);
        auto abort_all = [&] { q_raw.close(); q_free.close(); q_packed.close(); q_dev.close(); q_done.close(); };
        // a stage that throws (std::bad_alloc from a growing vector, mostly) must not take the process down from inside a
        // std::thread: the failure is recorded, every queue is closed, the other stages drain and the caller gets a code
        auto stage_failed = [&](int code, const char *what) {
            try { fail_.raise(code, what); } catch (...) { fail_.set.store(true); }
            abort_all();
        };
        auto guarded = [&](auto body) {
            return [&stage_failed, body]() {
                try { body(); }
                catch (const std::bad_alloc &) { stage_failed(13, "out of host memory"); }
                catch (const std::exception &e) { stage_failed(1000, e.what()); }
                catch (...) { stage_failed(1000, "unknown exception in an ingest stage"); }
            };
        };
        std::thread t_read(guarded([&] {
            RawBatch b;
            while (!fail_.set.load() && read_batch(b))
                if (!q_raw.push(std::move(b))) break;
            q_raw.close();
            if (fail_.set.load()) abort_all();
        }));
        std::thread t_pack(guarded([&] {
            RawBatch b;
            int slot;
            while (q_raw.pop(b)) {
                if (!q_free.pop(slot)) break;
                if (!pack(b, slots_[slot])) { abort_all(); return; }
                if (!q_packed.push(slot)) break;
            }
            q_packed.close();
        }));
        std::thread t_up(guarded([&] {
            int slot;
            while (q_packed.pop(slot)) {
                if (!upload(slots_[slot])) { abort_all(); return; }
                if (!q_dev.push(slot)) break;
            }
            q_dev.close();
        }));
        std::thread t_collect(guarded([&] {
            int slot;
            while (q_done.pop(slot)) {
                if (!collect(slots_[slot])) { abort_all(); return; }
                if (!q_free.push(slot)) break;
            }
        }));
        // the device stage stays in the calling thread: ks_ctx is bound to the thread that issues its batches
        guarded([&] {
            int slot;
            while (q_dev.pop(slot)) {
                if (!sketch(slots_[slot])) { abort_all(); break; }
                if (!q_done.push(slot)) break;
            }
        })();
        q_done.close();
        t_collect.join();
        abort_all();
        t_read.join();
        t_pack.join();
        t_up.join();
    }

    std::string path_;
    bool validate_;
    int device_;
    uint64_t batch_res_;
    bool pipeline_;
    ksh_fasta_sketches *out_;
    ks_params params_{};
    ks_ctx *ctx_ = nullptr;
    hipStream_t copy_stream_ = nullptr;
    std::unique_ptr<KsInput> in_;
    enum LineKind { LINE_START, HEADER, SEQ };
    std::vector<uint8_t> chunk_ = std::vector<uint8_t>(4u << 20);
    size_t pos_ = 0, len_ = 0;
    LineKind kind_ = LINE_START;
    bool in_record_ = false, eof_ = false;
    std::string header_;
    RawBatch *cur_ = nullptr;
    size_t records_done_ = 0;
    std::vector<Slot> slots_;
    Failure fail_;
};

} // namespace

extern "C" int ksh_sketch_fasta(const char *fasta_path, uint32_t ksize, uint32_t scaled, const char *moltype, int validate, int device,
                                uint64_t batch_residues, int pipeline, ksh_fasta_sketches **out, char *err, size_t err_cap) {
    auto fail = [&](int code, const std::string &m) {
        if (err && err_cap) { strncpy(err, m.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        return code;
    };
    if (!fasta_path || !moltype || !out) return fail(12, "NULL argument");
    uint32_t mt = 0;
    if (ks_moltype_from_string(moltype, &mt) != KS_OK)
        return fail(2, std::string("Invalid moltype: ") + moltype + ". Must be one of: protein, dayhoff, hp");
    if (ksize == 0 || ksize > 128) return fail(4, "Invalid ksize");
    if (scaled == 0) return fail(12, "scaled must be >= 1");
    // nothing throws across this boundary (include/kmerseek_host_c.h): 13 = out of host memory, 1000 = any other exception
    ksh_fasta_sketches *R = nullptr;
    try {
        R = new ksh_fasta_sketches();
        std::string e;
        int rc;
        {
            Ingest ing(fasta_path, ksize, scaled, mt, validate, device, batch_residues, pipeline, R);
            rc = ing.run(e);
        }
        if (rc != 0) { delete R; return fail(rc, e); }
        *out = R;
        return 0;
    } catch (const std::bad_alloc &) {
        delete R;
        if (err && err_cap) { strncpy(err, "out of host memory", err_cap - 1); err[err_cap - 1] = 0; }
        return 13;
    } catch (...) {
        delete R;
        if (err && err_cap) { strncpy(err, "internal error (exception)", err_cap - 1); err[err_cap - 1] = 0; }
        return 1000;
    }
}

extern "C" uint64_t ksh_fs_n_records(const ksh_fasta_sketches *r) { return r ? r->names.size() : 0; }
extern "C" uint64_t ksh_fs_n_hashes(const ksh_fasta_sketches *r) { return r ? r->hashes.n : 0; }
extern "C" const uint64_t *ksh_fs_offsets(const ksh_fasta_sketches *r) { return r ? r->offsets.p : nullptr; }
extern "C" const uint64_t *ksh_fs_hashes(const ksh_fasta_sketches *r) { return r ? r->hashes.p : nullptr; }
extern "C" const uint32_t *ksh_fs_abunds(const ksh_fasta_sketches *r) { return r ? r->abunds.p : nullptr; }
extern "C" const char *ksh_fs_names(ksh_fasta_sketches *r, uint64_t *len) {
    if (!r) return nullptr;
    if (r->names_blob.empty() && !r->names.empty()) try {
        size_t tot = 0;
        for (auto &n : r->names) tot += n.size() + 1;
        r->names_blob.reserve(tot);
        for (size_t i = 0; i < r->names.size(); i++) {
            if (i) r->names_blob += '\n';
            r->names_blob += r->names[i];
        }
    } catch (...) { // (out of host memory for the joined blob)
        if (len) *len = 0;
        return nullptr;
    }
    if (len) *len = r->names_blob.size();
    return r->names_blob.c_str();
}
extern "C" void ksh_fs_stats(const ksh_fasta_sketches *r, uint64_t *n_residues, uint64_t *n_windows, uint64_t *n_batches, double *seconds) {
    if (!r) return;
    if (n_residues) *n_residues = r->n_residues;
    if (n_windows) *n_windows = r->n_windows;
    if (n_batches) *n_batches = r->n_batches;
    if (seconds) { seconds[0] = r->t_wall; seconds[1] = r->t_read; seconds[2] = r->t_pack; seconds[3] = r->t_h2d; seconds[4] = r->t_device; seconds[5] = r->t_collect; }
}
extern "C" void ksh_fs_free(ksh_fasta_sketches *r) { delete r; }
