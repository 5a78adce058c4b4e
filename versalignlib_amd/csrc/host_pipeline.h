// host_pipeline.h -- the host-only half of libHIPKernel.so's host-pointer path: the persistent worker pool and the
// gather / scatter between the caller's scattered heap blocks (what the plugin ABI hands over: N `char *`, one per
// sequence, src/util/versalignUtil.cpp:24-31; 2N operator new[] result rows, include/AlignmentKernel.h:20-23) and the
// contiguous pair-major staging buffers the device works on.  The reference's precedent is the OpenCL backend's
// gather / collect loops (src/Kernels/OpenCL/OpenCLKernel.cpp:57-66, 613-645: one memcpy per sequence, an OpenMP copy
// back); nothing of them is reused.
//
// No HIP in here on purpose: this header compiles with plain g++ so that the code 16 host threads run concurrently
// into caller-owned arrays is exercised under -fsanitize=thread / address,undefined on the CPU
// (tests/host_pipeline_check.cpp, `make sanitize`); hip_engine.hip.h includes it unchanged.
#pragma once

#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace valign {

// Host worker threads that outlive a call: gather / scatter run on them chunk after chunk (spawning
// num_threads std::threads per chunk cost as much as the copying itself).  run(parts, fn) calls
// fn(part) for every part in [0, parts) on the workers and the calling thread and returns when all
// are done; an exception from fn is rethrown on the caller.
class WorkerPool {
public:
    explicit WorkerPool(int workers) {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &t : threads_) t.join();
    }
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    int workers() const { return (int)threads_.size(); }

    void run(int parts, const std::function<void(int)> &fn) {
        if (parts <= 0) return;
        if (parts == 1 || threads_.empty()) {
            for (int p = 0; p < parts; ++p) fn(p);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            job_ = &fn;
            parts_ = parts;
            next_ = 0;
            left_ = parts;
            error_ = nullptr;
            ++generation_;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this] { return left_ == 0 && busy_ == 0; });
        job_ = nullptr;
        if (error_) std::rethrow_exception(error_);
    }

private:
    // Parts are claimed under the mutex (a part is thousands of pairs: the lock is nothing beside it).  `busy_` counts
    // the threads inside work(): run() only returns -- and the job's std::function only dies -- once every worker has
    // left it, so a worker that woke up late can never look at a job that is already gone.
    void work() {
        std::unique_lock<std::mutex> lock(m_);
        if (!job_) return;
        ++busy_;
        for (;;) {
            if (next_ >= parts_) break;
            const int p = next_++;
            const std::function<void(int)> *job = job_;
            lock.unlock();
            std::exception_ptr err;
            try {
                (*job)(p);
            } catch (...) {
                err = std::current_exception();
            }
            lock.lock();
            if (err && !error_) error_ = err;
            --left_;
        }
        --busy_;
        if (left_ == 0 && busy_ == 0) done_.notify_all();
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            work();
        }
    }

    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int)> *job_ = nullptr;     // everything below: guarded by m_
    int next_ = 0, parts_ = 0, left_ = 0, busy_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;
};

// ---- 4-bit base classes for the trip across PCIe (score path) ----
// score_alignments only ever looks at the CLASS of a base (DefaultKernel.h:43-60: A/a 1, T/t 2, C/c 3, G/g 4, N/n 5,
// anything else 0), so the host-pointer path may send classes instead of ASCII: two bases per byte, first base in
// the low nibble, a sequence of `len` bases in (len + 1) / 2 bytes.  The device expands them back to one canonical
// byte per base (pack_kernels.hip.h) and the score kernels run unchanged -- bit-identical scores, half the H2D bytes.
// (compute_alignments copies the caller's BYTES into its result rows -- lower case stays lower case -- and keeps ASCII.)
inline uint8_t base_class_of(uint8_t ch) {
    switch (ch & 0xDF) {            // fold case; bytes >= 0x80 never match
        case 'A': return 1;
        case 'T': return 2;
        case 'C': return 3;
        case 'G': return 4;
        case 'N': return 5;
        default: return 0;
    }
}

inline size_t packed_length(int len) { return ((size_t)len + 1) / 2; }

inline void pack_classes_scalar(const uint8_t *src, int len, uint8_t *dst) {
    static const struct Table {
        uint8_t cls[256];
        Table() { for (int c = 0; c < 256; ++c) cls[c] = base_class_of((uint8_t)c); }
    } table;
    int i = 0;
    for (; i + 1 < len; i += 2) dst[i >> 1] = (uint8_t)(table.cls[src[i]] | (table.cls[src[i + 1]] << 4));
    if (i < len) dst[i >> 1] = table.cls[src[i]];
}

// 32 bases -> 16 bytes per step: the class comes from a 16-entry table on the low nibble of the byte, valid only where
// the byte's high nibble (case bit cleared) is the one that letter has; vpmaddubsw folds byte pairs into nibble pairs
__attribute__((target("avx2"))) inline void pack_classes_avx2(const uint8_t *src, int len, uint8_t *dst) {
    const __m256i lo_mask = _mm256_set1_epi8(0x0F);
    // low nibble: A = x1, C = x3, T = x4, G = x7, N = xE
    const __m256i cls_of = _mm256_setr_epi8(0, 1, 0, 3, 2, 0, 0, 4, 0, 0, 0, 0, 0, 0, 5, 0, 0, 1, 0, 3, 2, 0, 0, 4, 0, 0, 0, 0, 0, 0, 5, 0);
    const __m256i hi_of = _mm256_setr_epi8(-1, 4, -1, 4, 5, -1, -1, 4, -1, -1, -1, -1, -1, -1, 4, -1, -1, 4, -1, 4, 5, -1, -1, 4, -1, -1, -1, -1,
                                           -1, -1, 4, -1);
    const __m256i fold = _mm256_set1_epi8((char)0xDF), weights = _mm256_set1_epi16(0x1001);
    int i = 0;
    for (; i + 32 <= len; i += 32) {
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i));
        const __m256i lo = _mm256_and_si256(c, lo_mask);
        const __m256i hi = _mm256_and_si256(_mm256_srli_epi16(_mm256_and_si256(c, fold), 4), lo_mask);
        const __m256i ok = _mm256_cmpeq_epi8(hi, _mm256_shuffle_epi8(hi_of, lo));
        const __m256i cls = _mm256_and_si256(_mm256_shuffle_epi8(cls_of, lo), ok);
        const __m256i pairs = _mm256_maddubs_epi16(cls, weights);              // 16 x (even + 16 * odd)
        const __m256i bytes = _mm256_permute4x64_epi64(_mm256_packus_epi16(pairs, pairs), 0x08);
        _mm_storeu_si128((__m128i *)(dst + (i >> 1)), _mm256_castsi256_si128(bytes));
    }
    if (i < len) pack_classes_scalar(src + i, len - i, dst + (i >> 1));
}

inline void pack_classes(const uint8_t *src, int len, uint8_t *dst) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) pack_classes_avx2(src, len, dst);
    else pack_classes_scalar(src, len, dst);
}

// Caller-provided contiguous result buffers (valign_hip_align_host): rows = n * 2 * AL bytes, idx = n * 4 shorts
struct FlatSink {
    uint8_t *rows;
    short *idx;
    size_t AL;
    FlatSink operator+(long long k) const { return FlatSink{rows + (size_t)k * 2 * AL, idx + 4 * k, AL}; }
};

// Gather / scatter of one engine: fixed (read_length, ref_length), a pool that is resized when the caller's
// num_threads changes.  Threads write disjoint pair ranges; nothing else is shared.
class HostPacker {
public:
    HostPacker(int R, int F) : R_(R), F_(F) {}

    // fn(part, lo, hi) over `threads` equal ranges of [0, cnt) on the persistent workers
    template <typename Fn>
    void for_ranges(int threads, long long cnt, long long serial_below, Fn fn) {
        if (threads <= 1 || cnt < serial_below) {
            fn(0, 0ll, cnt);
            return;
        }
        if (!pool_ || pool_->workers() != threads - 1) pool_.reset(new WorkerPool(threads - 1));
        const long long per = (cnt + threads - 1) / threads;
        pool_->run(threads, [&](int t) {
            const long long lo = t * per, hi = std::min(cnt, lo + per);
            if (lo < hi) fn(t, lo, hi);
        });
    }

    // reads[i] (exactly R bytes) -> dst_reads + i * R, refs likewise: the pair-major layout of the device buffers
    void gather(const char *const *reads, const char *const *refs, long long cnt, uint8_t *dst_reads, uint8_t *dst_refs,
                int threads) {
        const int R = R_, F = F_;
        for_ranges(threads, cnt, 4096, [=](int, long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                memcpy(dst_reads + (size_t)i * R, reads[i], (size_t)R);
                memcpy(dst_refs + (size_t)i * F, refs[i], (size_t)F);
            }
        });
    }

    // the same with 4-bit classes: reads[i] -> dst_reads + i * packed_length(R) ...
    void gather_packed(const char *const *reads, const char *const *refs, long long cnt, uint8_t *dst_reads,
                       uint8_t *dst_refs, int threads) {
        const int R = R_, F = F_;
        const size_t PR = packed_length(R), PF = packed_length(F);
        for_ranges(threads, cnt, 4096, [=](int, long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                if (i + 4 < hi) {                      // every sequence is a heap block of its own: keep misses in flight
                    __builtin_prefetch(reads[i + 4]);
                    __builtin_prefetch(refs[i + 4]);
                    __builtin_prefetch(refs[i + 4] + 64);
                }
                pack_classes((const uint8_t *)reads[i], R, dst_reads + (size_t)i * PR);
                pack_classes((const uint8_t *)refs[i], F, dst_refs + (size_t)i * PF);
            }
        });
    }

    // result rows are copied by one thread below 768 KB (waking the pool costs what copying that much does): 2,048 pairs of
    // 64 x 128, 590 pairs of 150 x 500 -- a call of 1,000 pairs of 150 x 500 spent 58 us copying 1.3 MB alone
    static long long serial_below_for_rows(size_t AL) {
        return std::max<long long>(64, (768ll << 10) / (2 * (long long)std::max<size_t>(AL, 1)));
    }

    // staging -> the caller's contiguous buffers.  Like the Alignment scatter below, only the strings are read out of the
    // staging (both rows of a pair start at idx[0]); the zeros in front are written here -- the staging's columns before
    // the chunk's first string are not even copied from the device any more (Engine::align_host).
    // `first_col`: the staging holds every row from that column on only (rows of AL - first_col bytes: the device packed
    // them before the copy, compact_rows_kernel); 0 = whole rows.  No string starts before it.
    void scatter(FlatSink sink, long long cnt, const uint8_t *rows, const short *idx, int threads, size_t first_col = 0) {
        const size_t AL = sink.AL, W = AL - first_col;
        for_ranges(threads, cnt, serial_below_for_rows(AL), [=](int, long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                size_t start = idx[4 * i + 0] > 0 ? (size_t)idx[4 * i + 0] : 0;
                if (start > AL) start = AL;
                if (start < first_col) start = first_col;       // (cannot happen: first_col <= every start)
                const uint8_t *src = rows + (size_t)i * 2 * W;
                uint8_t *dst = sink.rows + (size_t)i * 2 * AL;
                memset(dst, 0, start);
                memcpy(dst + start, src + (start - first_col), AL - start);
                memset(dst + AL, 0, start);
                memcpy(dst + AL + start, src + W + (start - first_col), AL - start);
            }
            memcpy(sink.idx + 4 * lo, idx + 4 * lo, sizeof(short) * 4 * (size_t)(hi - lo));
        });
    }

    // staging -> the ABI's Alignment array: two fresh operator new[] rows per pair (the caller delete[]s them)
    template <typename AlignmentT>
    void scatter(AlignmentT *alignments, long long cnt, const uint8_t *rows, const short *idx, int threads, size_t first_col = 0) {
        const size_t AL = (size_t)R_ + F_, W = AL - first_col;
        for_ranges(threads, cnt, serial_below_for_rows(AL), [=](int, long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                AlignmentT &a = alignments[i];
                a.read = new char[AL ? AL : 1];
                a.ref = new char[AL ? AL : 1];
                // Both rows are right-justified strings that begin at offset idx[0] (= readStart = refStart,
                // DefaultKernel.cpp:430-456) with zeros in front: only the strings are read out of the staging -- a
                // third of its bytes at 150 x 500 -- the zeros are written here.
                size_t start = idx[4 * i + 0] > 0 ? (size_t)idx[4 * i + 0] : 0;
                if (start > AL) start = AL;
                if (start < first_col) start = first_col;       // (cannot happen: first_col <= every start)
                const uint8_t *src = rows + (size_t)i * 2 * W;
                memset(a.read, 0, start);
                memcpy(a.read + start, src + (start - first_col), AL - start);
                memset(a.ref, 0, start);
                memcpy(a.ref + start, src + W + (start - first_col), AL - start);
                a.readStart = idx[4 * i + 0];
                a.readEnd = idx[4 * i + 1];
                a.refStart = idx[4 * i + 2];
                a.refEnd = idx[4 * i + 3];
            }
        });
    }

private:
    int R_, F_;
    std::unique_ptr<WorkerPool> pool_;
};

}  // namespace valign
