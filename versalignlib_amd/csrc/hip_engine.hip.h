// hip_engine.hip.h -- host side of libHIPKernel.so: kernel selection, launches, and the
// host-pointer path (gather -> pinned staging -> H2D -> kernel -> D2H, double buffered).
// The closest reference precedent for the staging loop is the OpenCL backend's
// gather/copy/launch/collect loop (src/Kernels/OpenCL/OpenCLKernel.cpp:57-108); unlike
// it, chunks here are large (tens of MB), asynchronous and overlapped on two streams.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host_pipeline.h"
#include "kernel_instances.hip.h"
#include "band_kernels.hip.h"
#include "long_kernels.hip.h"
#include "pack_kernels.hip.h"
#include "ragged_kernels.hip.h"
#include "strip_kernels.hip.h"

namespace valign {

struct Scoring {
    int match = 2, mismatch = -1, gap_read = -3, gap_ref = -3;
    bool affine = false;
    int open_read = -3, ext_read = -3, open_ref = -3, ext_ref = -3;
};

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// The per-geometry kernels are compiled in kernel_part.hip (one object per part, in parallel)
#define VALIGN_DECLARE(G, K) VALIGN_GEOMETRY_KERNELS(extern template, G, K)
VALIGN_ALL_GEOMETRIES(VALIGN_DECLARE)
#undef VALIGN_DECLARE

// One compiled (G, K) geometry with its kernel variants.
struct Geometry {
    int G, K;
    WaveLds (*lds)(int R, int F);
    const void *kernel[2][7];      // score kernels [alg][linear, symmetric linear, affine, symmetric affine,
                                   //                     symmetric affine / affine on half floats (SW only)]
    const void *fill[2][12];        // alignment fill kernels [alg][linear, symmetric linear, affine, SSE policy,
                                   //                               linear with the pointer tagged into the cell, the same with
                                   //                               one end-cell key per lane (SW), symmetric affine,
                                   //                               affine / symmetric affine with tagged cells,
                                   //                               SSE policy with tagged cells (per-row / lane key),
                                   //                               tagged with the end-cell key in the query profile (SW)]
};

template <int G, int K>
constexpr Geometry make_geometry() {
    return Geometry{G, K, &wave_lds<G, K>,
                    {{(const void *)&score_kernel<G, K, kAlgSW, kGapLinear>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapSym>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffine>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffineSym>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffineSymF16>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffineF16>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapSymF16>},
                     {(const void *)&score_kernel<G, K, kAlgNW, kGapLinear>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapSym>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffine>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffineSym>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffineSymF16>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffineF16>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapSymF16>}},
                    {{(const void *)&align_fill_kernel<G, K, kAlgSW, false>, (const void *)&align_fill_kernel<G, K, kAlgSW, true>,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgSW, false>, (const void *)&align_fill_sse_kernel<G, K, kAlgSW>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgSW, false, false>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, false>,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgSW, true>,
                      (const void *)&align_fill_affine_tag_kernel<G, K, kAlgSW, false>,
                      (const void *)&align_fill_affine_tag_kernel<G, K, kAlgSW, true>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgSW, false, true>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, true>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, false, false, true>},
                     {(const void *)&align_fill_kernel<G, K, kAlgNW, false>, (const void *)&align_fill_kernel<G, K, kAlgNW, true>,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgNW, false>, (const void *)&align_fill_sse_kernel<G, K, kAlgNW>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgNW, false, false>, nullptr,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgNW, true>,
                      (const void *)&align_fill_affine_tag_kernel<G, K, kAlgNW, false>,
                      (const void *)&align_fill_affine_tag_kernel<G, K, kAlgNW, true>,
                      (const void *)&align_fill_tag_kernel<G, K, kAlgNW, false, true>, nullptr, nullptr}}};
}

// Rows covered = G*K.  Ordered by capacity; selection is by estimated cost.
static const Geometry kGeometries[] = {
    make_geometry<8, 4>(),   make_geometry<8, 6>(),   make_geometry<8, 8>(),   make_geometry<16, 4>(),
    make_geometry<8, 10>(),  make_geometry<8, 12>(),  make_geometry<8, 16>(),  make_geometry<16, 8>(),
    make_geometry<8, 20>(),  make_geometry<16, 10>(), make_geometry<16, 12>(), make_geometry<16, 16>(),
    make_geometry<32, 8>(),  make_geometry<32, 10>(), make_geometry<32, 12>(), make_geometry<32, 16>(),
    make_geometry<64, 8>(),  make_geometry<64, 12>(), make_geometry<64, 16>(), make_geometry<64, 24>(),
    make_geometry<64, 32>(),
};
constexpr int kNumGeometries = sizeof(kGeometries) / sizeof(kGeometries[0]);
#define VALIGN_COUNT(G, K) +1
static_assert(kNumGeometries == 0 VALIGN_ALL_GEOMETRIES(VALIGN_COUNT), "kernel_instances.hip.h lists other geometries than this table");
#undef VALIGN_COUNT

constexpr int kMaxBlockLds = 160 * 1024;       // gfx950: 160 KiB per CU, one block may take it all
constexpr int kSlots = 4;                      // staging slots of the host-pointer pipeline
constexpr int kDefaultBlockLds = 64 * 1024;    // above this the kernel attribute must be raised

// Long-read path (row strips + column phases, long_kernels.hip.h): one geometry.
constexpr int kLongG = 16, kLongK = 10;
static const void *const kLongAffineKernels[2][2][2] = {       // [alg][same scores both ways][int32 cells]
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, false, true>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, true, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, true>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, true, true>}},
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, false, true>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, true, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, false, true>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, true, true>}}};
static const void *const kLongKernels[2][2][2] = {     // [alg][gap_read == gap_ref][int32 cells]
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, true>}},
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, true>}}};

// compute_alignments for reads beyond one register sweep (strip_kernels.hip.h): a wave per pair-of-pairs, K rows per lane
struct StripGeometry {
    int K;
    WaveLds (*lds)(int R, int F);
    const void *kernel[2];
    const void *affine_kernel[2];      // nullptr: too many rows per lane for the affine kernel's registers
    const void *sse_kernel[2];         // traceback_policy = 1 (linear gaps)
    const void *wide_kernel;           // NW variant on int32 cells (linear gaps, default tie-breaks); nullptr: too many registers
};
#define VALIGN_STRIP_SSE(K) {(const void *)&align_strip_kernel<K, kAlgSW, false, true>, (const void *)&align_strip_kernel<K, kAlgNW, false, true>}
static const StripGeometry kStripGeometries[] = {
    {32, &wave_lds<64, 32>, {(const void *)&align_strip_kernel<32, kAlgSW>, (const void *)&align_strip_kernel<32, kAlgNW>}, {nullptr, nullptr}, {nullptr, nullptr}, nullptr},
    {24, &wave_lds<64, 24>, {(const void *)&align_strip_kernel<24, kAlgSW>, (const void *)&align_strip_kernel<24, kAlgNW>}, {nullptr, nullptr}, VALIGN_STRIP_SSE(24), nullptr},
    {16, &wave_lds<64, 16>, {(const void *)&align_strip_kernel<16, kAlgSW>, (const void *)&align_strip_kernel<16, kAlgNW>},
     {(const void *)&align_strip_kernel<16, kAlgSW, true>, (const void *)&align_strip_kernel<16, kAlgNW, true>}, VALIGN_STRIP_SSE(16),
     (const void *)&align_strip_wide_kernel<16>},
    {12, &wave_lds<64, 12>, {(const void *)&align_strip_kernel<12, kAlgSW>, (const void *)&align_strip_kernel<12, kAlgNW>},
     {(const void *)&align_strip_kernel<12, kAlgSW, true>, (const void *)&align_strip_kernel<12, kAlgNW, true>}, VALIGN_STRIP_SSE(12),
     (const void *)&align_strip_wide_kernel<12>},
    {8, &wave_lds<64, 8>, {(const void *)&align_strip_kernel<8, kAlgSW>, (const void *)&align_strip_kernel<8, kAlgNW>},
     {(const void *)&align_strip_kernel<8, kAlgSW, true>, (const void *)&align_strip_kernel<8, kAlgNW, true>}, VALIGN_STRIP_SSE(8),
     (const void *)&align_strip_wide_kernel<8>},
};
#undef VALIGN_STRIP_SSE

// Small batches: fill + traceback in one launch, pointer stream in LDS (align_fill_tag_kernel<..., FUSED>)
struct FusedGeometry {
    int G, K;
    WaveLds (*lds)(int R, int F);
    int (*total)(int wave_lds, int R, int F, int blocks8);
    const void *kernel[2];
};
template <int G, int K>
constexpr FusedGeometry make_fused() {
    return FusedGeometry{G, K, &wave_lds<G, K>,
                         [](int wl, int R, int F, int b8) { return fused_lds<G, K>(wl, R, F, b8).total; },
                         {(const void *)&align_fill_tag_kernel<G, K, kAlgSW, false, false, true>,
                          (const void *)&align_fill_tag_kernel<G, K, kAlgNW, false, false, true>}};
}
// (32 x 2 / 32 x 4: few rows per lane -- the shortest dependent chain per step, which is what a single-wave call costs)
// (64 x 4: a whole wave per pair-of-pairs -- the one geometry whose pointer stream fits LDS at 150 x 500, 83 KB: a
// 1,000-pair compute_alignments call of that shape is ONE launch instead of memset + fill + a traceback that chases
// pointers through HBM, 600 -> ~200 us)
static const FusedGeometry kFusedGeometries[] = {make_fused<8, 4>(), make_fused<16, 4>(), make_fused<32, 2>(), make_fused<16, 8>(),
                                                 make_fused<32, 4>(), make_fused<16, 10>(), make_fused<32, 8>(), make_fused<64, 4>()};

struct LaunchPlan {
    bool long_mode = false;        // sequences too long for one register sweep / LDS-resident reference
    const Geometry *geo = nullptr;
    WaveLds lds{};
    int waves_per_block = 4;
    int pairs_per_wave = 0;
};

// Issues the device-to-host result copies of align_host from a thread of its own, each one only after the HOST has seen
// its chunk's kernels finish.  Why not simply hipStreamWaitEvent + hipMemcpyAsync: measured on this stack (rocprofv3,
// profiles/r03_d2h_engine.txt), a D2H copy enqueued behind a still-pending barrier or kernel in its stream is carried out
// by a shader (__amd_rocclr_copyBuffer) instead of the SDMA engine -- and that blit kernel, waiting on PCIe, sits on
// the same CUs as the fill kernel of the next chunk: the fills of a 16-chunk call took 1.75x as long.  A copy issued
// into a stream whose previous command is a finished copy goes to SDMA and costs the kernels nothing.
class CopyIssuer {
public:
    struct Job {
        hipEvent_t ready;           // the chunk's last kernel (waited for on the host)
        void *dst[2];
        const void *src[2];
        size_t bytes[2];
        hipStream_t stream;
        hipEvent_t done;            // recorded behind the copies
        int slot;
    };
    explicit CopyIssuer(int device) : device_(device), thread_([this] { loop(); }) {}
    ~CopyIssuer() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    void submit(const Job &job) {
        {
            std::lock_guard<std::mutex> lock(m_);
            if (error_) std::rethrow_exception(error_);
            jobs_.push_back(job);
            ++submitted_[job.slot];
        }
        cv_.notify_all();
    }
    // every job submitted for `slot` has been issued: its `done` event is recorded and may be waited for
    void wait_issued(int slot) {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return issued_[slot] == submitted_[slot] || error_; });
        if (error_) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            jobs_.clear();
            for (int s = 0; s < 16; ++s) issued_[s] = submitted_[s];
            std::rethrow_exception(e);
        }
    }
    void wait_idle() {
        for (int s = 0; s < 16; ++s) wait_issued(s);
    }

private:
    void loop() {
        (void)hipSetDevice(device_);
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;          // (stop requested and nothing left)
                job = jobs_.front();
                jobs_.erase(jobs_.begin());
            }
            std::exception_ptr err;
            try {
                hip_check(hipEventSynchronize(job.ready), "hipEventSynchronize(kernels of the chunk)");
                for (int k = 0; k < 2; ++k)
                    if (job.bytes[k])
                        hip_check(hipMemcpyAsync(job.dst[k], job.src[k], job.bytes[k], hipMemcpyDeviceToHost, job.stream), "D2H results");
                hip_check(hipEventRecord(job.done, job.stream), "hipEventRecord");
            } catch (...) {
                err = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                if (err && !error_) error_ = err;
                ++issued_[job.slot];
            }
            cv_.notify_all();
        }
    }
    int device_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Job> jobs_;
    long long submitted_[16] = {}, issued_[16] = {};
    bool stop_ = false;
    std::exception_ptr error_;
    std::thread thread_;            // last: starts when everything above exists
};

// Host memory the caller registered with valign_hip_host_register (page-locked and mapped for the device): result
// buffers of valign_hip_align_host that lie inside such a range receive their rows straight from the device's copy
// engine -- no pinned staging, no host-side copy.  Process-wide; ranges do not overlap.
class HostRegistry {
public:
    static HostRegistry &instance() {
        static HostRegistry r;
        return r;
    }
    void add(void *ptr, size_t bytes) {
        if (!ptr || bytes == 0) throw std::runtime_error("valign_hip_host_register: empty range");
        std::lock_guard<std::mutex> lock(m_);
        const uintptr_t lo = (uintptr_t)ptr, hi = lo + bytes;
        for (const auto &r : ranges_)
            if (lo < r.second && r.first < hi) throw std::runtime_error("valign_hip_host_register: overlaps a registered range");
        hip_check(hipHostRegister(ptr, bytes, hipHostRegisterDefault), "hipHostRegister");
        ranges_[lo] = hi;
    }
    void remove(void *ptr) {
        std::lock_guard<std::mutex> lock(m_);
        auto it = ranges_.find((uintptr_t)ptr);
        if (it == ranges_.end()) throw std::runtime_error("valign_hip_host_unregister: not the start of a registered range");
        hip_check(hipHostUnregister(ptr), "hipHostUnregister");
        ranges_.erase(it);
    }
    // [ptr, ptr + bytes) is page-locked: registered here, or by the caller's own hipHostRegister / hipHostMalloc
    bool covers(const void *ptr, size_t bytes) {
        if (!ptr || bytes == 0) return false;
        const uintptr_t lo = (uintptr_t)ptr, hi = lo + bytes;
        {
            std::lock_guard<std::mutex> lock(m_);
            auto it = ranges_.upper_bound(lo);
            if (it != ranges_.begin()) {
                --it;
                if (it->first <= lo && hi <= it->second) return true;
            }
        }
        for (const void *probe : {ptr, (const void *)(hi - 1)}) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, probe) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            if (attr.type != hipMemoryTypeHost) return false;
        }
        return true;
    }

private:
    std::mutex m_;
    std::map<uintptr_t, uintptr_t> ranges_;      // start -> end
};

class Engine {
public:
    struct LengthGroup {
        int R = 0, F = 0;               // strides (= swept shape) of the group
        long long pairs = 0, pair_ofs = 0;
        size_t read_ofs = 0, ref_ofs = 0;
    };
    struct HostStats {             // of the last score_host call
        int launches = 0;
        double cells_swept = 0, cells_padded = 0;
        double gather_ms = 0, wait_ms = 0, drain_ms = 0;     // host time: packing, blocked on the device, copy-out
        double classify_ms = 0;                              // length-sorted batching: host time spent waiting for the device's histograms
        double launch_ms = 0;                                // ... and laying the groups out + launching their sweeps
        int packed = 0;                                      // 1: the sequences crossed PCIe as 4-bit classes
        int direct_out = 0;                                  // 1: results were copied straight into the caller's (registered) buffers
        int direct = 0;                                      // 1: small call, kernels worked on the pinned staging directly; 2: ... in one fused launch
    };

    Engine(int device, int R, int F, const Scoring &sc, int force_g, int force_k)
        : device_(device), R_(R), F_(F), sc_(sc) {
        if (R < 0 || F < 0) throw std::runtime_error("negative sequence length");
        if ((long long)R + F > 32767)
            throw std::runtime_error("read_length + ref_length exceeds the ABI's 16-bit coordinates");
        validate_scoring();
        int count = 0;
        hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
        if (device < 0 || device >= count)
            throw std::runtime_error("HIP device " + std::to_string(device) + " not present (" +
                                     std::to_string(count) + " visible): libHIPKernel.so has no CPU path");
        hip_check(hipSetDevice(device_), "hipSetDevice");
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
        arch_ = prop.gcnArchName;
        cu_count_ = prop.multiProcessorCount;
        if (arch_.find("gfx950") == std::string::npos)
            throw std::runtime_error("device is " + arch_ + "; this library carries gfx950 code only");
        force_g_ = force_g;
        force_k_ = force_k;
        plan_ = choose_plan(R_, F_, force_g, force_k);
        latency_plan_ = (force_g || force_k || plan_.long_mode) ? plan_ : choose_plan(R_, F_, 0, 0, true);
        build_length_classes();
        for (int s = 0; s < kSlots; ++s) hip_check(hipStreamCreateWithFlags(&streams_[s], hipStreamNonBlocking), "hipStreamCreate");
        for (int s = 0; s < kSlots; ++s) {
            hip_check(hipEventCreateWithFlags(&slot_done_[s], hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&in_done_[s], hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&kernels_done_[s], hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate");     // (the copy issuer sleeps on it)
        }
    }

    ~Engine() {
        copy_issuer_.reset();               // (joins its thread; nothing is queued outside a call)
        (void)hipSetDevice(device_);
        if (trace_stream_) (void)hipStreamSynchronize(trace_stream_);
        release_staging();
        release_trace_scratch();
        if (d_brow_) (void)hipFree(d_brow_);
        if (d_band_blocks_) (void)hipFree(d_band_blocks_);
        if (d_band_fill_) (void)hipFree(d_band_fill_);
        release_ragged();
        for (int s = 0; s < kSlots; ++s) {
            if (slot_done_[s]) (void)hipEventDestroy(slot_done_[s]);
            if (in_done_[s]) (void)hipEventDestroy(in_done_[s]);
            if (kernels_done_[s]) (void)hipEventDestroy(kernels_done_[s]);
            if (streams_[s]) (void)hipStreamDestroy(streams_[s]);
        }
        if (trace_stream_) {
            for (int r = 0; r < 2; ++r) {
                (void)hipEventDestroy(fill_done_[r]);
                (void)hipEventDestroy(trace_done_[r]);
            }
            (void)hipEventDestroy(entry_ev_);
            (void)hipStreamDestroy(trace_stream_);
        }
    }

    // Banded Smith-Waterman scores (strip band of long_kernels.hip.h); 0 = every cell.  Takes the
    // long-read path whatever the shape.
    void set_band_width(int diagonals) {
        if (diagonals < 0) throw std::runtime_error("band_width must be >= 0");
        band_width_ = diagonals;
        if (diagonals > 0 && !plan_.long_mode) plan_ = long_plan();
    }
    int band_width() const { return band_width_; }
    // DP cell width of score_alignments: 0 = int16 unless the shape could overflow it (default),
    // 16 = int16 or refuse, 32 = always int32 (strip path, half the throughput)
    void set_score_width(int bits) {
        if (bits != 0 && bits != 16 && bits != 32) throw std::runtime_error("score_width must be 0, 16 or 32");
        score_width_ = bits;
    }
    // 0: Default/OpenCL kernel tie-breaks (default); 1: SSE2/AVX2 kernel tie-breaks
    void set_traceback_policy(int policy) {
        if (policy != 0 && policy != 1) throw std::runtime_error("traceback_policy must be 0 (default) or 1 (sse)");
        sse_policy_ = policy == 1;
    }
    // Length-sorted batching of score calls (both modes), done on the device -- classification, packing by length class,
    // one sweep per read class (ragged_kernels.hip.h): 0 = never (every pair is swept at read_length x ref_length; default),
    // 1 = when the call is ragged enough to skip a third of the cells (host pointers: judged from a sample of the call's
    // tails; device-resident batches: from the device's own histogram, which the call then waits for), 2 = always
    void set_ragged_batching(int mode) {
        if (mode < 0 || mode > 2) throw std::runtime_error("ragged_batching must be 0, 1 or 2");
        ragged_ = mode;
    }
    // 4-bit base classes instead of ASCII on the host-pointer score path (host_pipeline.h / pack_kernels.hip.h): 1 on
    // (default), 0 off.  Identical scores; half the bytes across PCIe.
    void set_host_packing(int mode) {
        if (mode != 0 && mode != 1) throw std::runtime_error("host_packing must be 0 or 1");
        pack_ = mode == 1;
    }
    // Cap of the internal pointer scratch of compute_alignments in MiB (0: 64 GiB / half the free HBM); batches
    // that need more run in chunks.  The environment's VALIGN_HIP_SCRATCH_CAP_MB (test switch) applies when this is 0.
    void set_pointer_scratch_cap_mb(long long mb) {
        if (mb < 0) throw std::runtime_error("pointer_scratch_cap_mb must be >= 0");
        if (mb > 0) scratch_cap_mb_ = mb;
    }
    int device() const { return device_; }
    int read_length() const { return R_; }
    int ref_length() const { return F_; }
    const LaunchPlan &plan() const { return plan_; }
    hipStream_t own_stream() const { return streams_[0]; }

    // Device-resident batch, asynchronous on `stream`.
    // `length_sorted` false: the caller has decided about length-sorted batching itself (the chunk pipeline of score_host)
    void score_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs,
                      int16_t *d_scores, hipStream_t stream, bool length_sorted = true) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;          // reference: unsupported mode is a silent no-op
        hip_check(hipSetDevice(device_), "hipSetDevice");
        if (length_sorted) host_stats_ = HostStats{};
        if (score_width_ == 16) check_int16_range(alg, true);
        const bool wide = score_width_ == 32 || (score_width_ == 0 && !int16_range_ok(alg));
        if (plan_.long_mode || wide) {      // int32 cells exist on the strip path only
            score_long_device(alg, n, d_reads, d_refs, d_scores, stream, wide);
            return;
        }
        // ragged_batching on a device-resident batch: classify, pack and sweep by length class (ragged_kernels.hip.h).  The
        // call then WAITS for the classification (the host lays the groups out); mode 1 sweeps the batch as it stands when
        // the length classes would not skip a third of the cells.
        if (length_sorted && ragged_applies(alg) && ragged_fits(n) && n >= 2 * ragged_min_) {
            ragged_begin(kSlots, n, d_reads, d_refs, stream);           // (a context of its own: the pipeline's slots may be busy on the engine's streams)
            if (ragged_finish(kSlots, alg, n, d_scores, stream, ragged_ == 2)) return;
        }
        // a batch that leaves most SIMDs with at most one wave is over when its slowest wave is: shortest sweep
        const bool few = n <= (long long)latency_plan_.pairs_per_wave * 1024 && band_width_ == 0;
        launch_score(few ? latency_plan_ : plan_, alg, R_, F_, n, d_reads, d_refs, d_scores, stream);
    }

    // One launch of the register-sweep score kernel over n pairs of shape R x F (sequences laid
    // out pair-major at exactly those strides) with the geometry of `plan`.
    // `groups` (length-sorted batches): packed groups sharing the read stride R, each with its own
    // reference stride <= F, swept by one launch; offsets are relative to d_reads / d_refs / d_scores.
    void launch_score(const LaunchPlan &plan, int alg, int R, int F, long long n, const uint8_t *d_reads,
                      const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream,
                      const LengthGroup *groups = nullptr, int n_groups = 0) {
        ScoreArgs a;
        a.reads = d_reads;
        a.refs = d_refs;
        a.scores = d_scores;
        a.n = n;
        a.R = R;
        a.F = F;
        a.n_groups = n_groups;
        const long long pairs_per_block = (long long)plan.pairs_per_wave * plan.waves_per_block;
        long long blocks = (n + pairs_per_block - 1) / pairs_per_block;
        if (n_groups > 0) {
            if (n_groups > kMaxScoreGroups) throw std::runtime_error("too many length groups for one launch");
            blocks = 0;
            for (int g = 0; g < n_groups; ++g) {
                blocks += (groups[g].pairs + pairs_per_block - 1) / pairs_per_block;
                if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
                a.groups[g].block_end = (unsigned)blocks;
                a.groups[g].F = groups[g].F;
                a.groups[g].n = groups[g].pairs;
                a.groups[g].pair_ofs = groups[g].pair_ofs;
                a.groups[g].read_ofs = (long long)groups[g].read_ofs;
                a.groups[g].ref_ofs = (long long)groups[g].ref_ofs;
            }
        }
        a.prof_area = plan.lds.prof_area;
        a.refc_stride = plan.lds.refc_stride;
        a.wave_lds = plan.lds.total;
        a.match = (short)sc_.match;
        a.mismatch = (short)sc_.mismatch;
        a.gap_read = (short)sc_.gap_read;
        a.gap_ref = (short)sc_.gap_ref;
        a.open_read = (short)sc_.open_read;
        a.ext_read = (short)sc_.ext_read;
        a.open_ref = (short)sc_.open_ref;
        a.ext_ref = (short)sc_.ext_ref;
        int gaps;
        if (sc_.affine) {
            gaps = (sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_) ? kGapAffineSym : kGapAffine;
            if (!no_f16_ && half_float_exact(alg, R, F, plan.geo->G * plan.geo->K)) gaps = gaps == kGapAffineSym ? kGapAffineSymF16 : kGapAffineF16;
        } else {
            gaps = (sc_.gap_read == sc_.gap_ref && !no_sym_) ? kGapSym : kGapLinear;
            // (the NW variant's tilted frame has no gap constants left: its half-float kernel serves gap_read != gap_ref too)
            if ((gaps == kGapSym || alg == kAlgNW) && !no_f16_ &&
                (alg == kAlgNW ? half_float_exact(alg, R, F, plan.geo->G * plan.geo->K) : half_float_unit_exact(R, F)))
                gaps = kGapSymF16;
        }
        const void *fn = plan.geo->kernel[alg][gaps];
        const int block_lds = plan.lds.total * plan.waves_per_block;
        if (block_lds > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, block_lds),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
        if (blocks == 0) return;
        void *kargs[] = {&a};
        hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(plan.waves_per_block * kWave), kargs,
                                  (size_t)block_lds, stream),
                  "hipLaunchKernel(score_kernel)");
    }


    // n sequences of `len` 4-bit classes -> n * len canonical bytes (pack_kernels.hip.h)
    void launch_unpack(const uint8_t *d_packed, uint8_t *d_out, long long n, int len, hipStream_t stream) {
        if (n <= 0 || len <= 0) return;
        UnpackArgs a{d_packed, d_out, n, len};
        void *kargs[] = {&a};
        const bool even = (len & 1) == 0;
        const long long items = even ? (n * (long long)(len / 2) + 7) / 8 : n * (long long)((len + 1) / 2);
        const long long blocks = (items + 255) / 256;
        if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
        hip_check(hipLaunchKernel(even ? (const void *)&unpack_even_kernel : (const void *)&unpack_odd_kernel, dim3((unsigned)blocks),
                                  dim3(256), kargs, 0, stream),
                  "hipLaunchKernel(unpack_kernel)");
    }

    // ---- banded Smith-Waterman scores, linear gaps: the cyclic block chain of band_kernels.hip.h ----
    static constexpr int kBandK = 16;              // rows per block = the band definition's block (describe: band_block_rows)

    struct BandPlan {
        bool usable = false, unit_delay = false;
        int nb = 0, pad_rows = 0, d = 0, ring_depth = 0, code_cols = 0, events = 0;
        std::vector<BandBlock> blocks;
        std::vector<int> fill_to;
    };

    // Windows, start distance, delays and ring sizes of the block chain for (R, F, band): what tools/band_schedule_model.py
    // calls plan().  `usable` is false where the chain does not pay or does not fit (then score_long_kernel's strips run).
    BandPlan make_band_plan() const {
        BandPlan p;
        const int R = R_, F = F_, w = band_width_ / 2, G = kBandG, K = kBandK;
        if (band_width_ <= 0 || R <= 0 || F <= 0) return p;
        const int rows = G * K;
        const int strips = std::max(1, (R + rows - 1) / rows);
        p.pad_rows = strips * rows - R;
        p.nb = strips * G;
        p.events = p.nb + G;
        std::vector<int> start((size_t)p.nb), lo((size_t)p.nb), hi((size_t)p.nb);
        int first_real = -1;
        for (int b = 0; b < p.nb; ++b) {
            int r_lo = b * K - p.pad_rows, r_hi = (b + 1) * K - p.pad_rows - 1;
            if (r_hi < 0) {                            // a block of padding rows only
                lo[(size_t)b] = 1;
                hi[(size_t)b] = 0;
                continue;
            }
            if (first_real < 0) first_real = b;
            r_lo = std::max(r_lo, 0);
            r_hi = std::min(r_hi, R - 1);
            const long long a = (long long)r_lo * F / R - w;
            start[(size_t)b] = (int)a - 1;             // the warm-up column: the diagonal neighbour of the window's first cell
            lo[(size_t)b] = (int)std::max<long long>(a, 0);
            hi[(size_t)b] = (int)std::min<long long>((long long)r_hi * F / R + w, F - 1);
        }
        for (int b = 0; b < first_real; ++b) start[(size_t)b] = start[(size_t)first_real];
        int width = 1, dmax = 0, dmin = 1 << 30;
        for (int b = 0; b < p.nb; ++b) {
            width = std::max(width, hi[(size_t)b] - start[(size_t)b] + 1);
            if (b > first_real) {
                dmax = std::max(dmax, start[(size_t)b] - start[(size_t)b - 1]);
                dmin = std::min(dmin, start[(size_t)b] - start[(size_t)b - 1]);
            }
        }
        if (dmin > dmax) dmin = dmax;
        // A block reads its predecessor up to dmax steps late; by then the predecessor may have begun its next block, but only
        // with that block's warm-up step, which writes the 0 the band gives that cell: width + dmax - 1 steps per period
        // suffice -- and a lane finishes its own block first (tools/band_schedule_model.py).
        p.d = std::max((std::max(width, width + dmax - 1) + G - 1) / G, dmax + 1);
        // every block one step behind its predecessor on the same column: the cell travels by DPP, no ring (UNIT kernel);
        // otherwise the ring is read one step ahead, which needs every delay >= 2
        p.unit_delay = dmin == dmax && p.d == dmax + 1 && !getenv("VALIGN_HIP_BAND_RING");
        if (!p.unit_delay) p.d = std::max(p.d, dmax + 2);
        const int delay_max = p.d - dmin;
        p.ring_depth = 4;
        while (p.ring_depth < delay_max + 1) p.ring_depth *= 2;
        p.blocks.assign((size_t)p.events + 2, BandBlock{0, 0x3FFFFFFF, 0, 1});
        for (int b = 0; b < p.nb; ++b) {
            BandBlock &k = p.blocks[(size_t)b];
            k.start = start[(size_t)b];
            if (lo[(size_t)b] <= hi[(size_t)b]) {
                k.lo = lo[(size_t)b];
                k.span = hi[(size_t)b] - lo[(size_t)b];
            }
            // (blocks of padding write zeros whatever they are asked: their successor may read any slot)
            k.delay = b > first_real ? p.d - (start[(size_t)b] - start[(size_t)b - 1]) : 2;
        }
        // reference ring: by event e every column below fill_to[e] is in the ring -- what any running block reaches in the d
        // steps after the event plus the sweep's look-ahead of two; the ring must span from the newest block's column to there
        p.fill_to.assign((size_t)p.events + 2, 0);
        int reach = 0, span = 0;
        for (int e = 0; e <= p.events + 1; ++e) {
            int head = -(1 << 30), tail = 1 << 30;
            for (int b = std::max(0, e - G + 1); b <= std::min(e, p.nb - 1); ++b) {
                head = std::max(head, start[(size_t)b] + (e - b) * p.d);
                tail = std::min(tail, start[(size_t)b] + (e - b) * p.d);
            }
            if (head > -(1 << 30)) reach = std::max(reach, std::min(head + p.d + 3, F));
            p.fill_to[(size_t)e] = reach;
            if (tail < (1 << 30)) span = std::max(span, reach + 2 * G - std::max(tail, 0));     // (+ what one event may commit early)
            if (e > 0 && p.fill_to[(size_t)e] - p.fill_to[(size_t)e - 1] > 2 * G) return p;      // more than two rounds per event: not built
        }
        p.code_cols = 128;
        while (p.code_cols < span + 8) p.code_cols *= 2;
        // What the chain buys is the lane-steps outside the band; it pays while windows are narrow against a strip's slope.
        // Limits of the kernel: ring addressing (base | offset) and one CU's LDS.
        if (p.code_cols > 2048 || p.ring_depth > 64) return p;
        if (p.unit_delay) p.ring_depth = 0;
        if (BandLds<kBandK>::total(p.code_cols, p.ring_depth) > 40 * 1024) return p;
        p.usable = true;
        return p;
    }

    // score_alignments(SW, linear gaps, band_width > 0) on the block chain; false: not applicable here (strips run instead)
    bool score_band_device(long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream) {
        if (no_band_chain_ || sc_.affine || band_width_ <= 0) return false;
        if (band_plan_width_ != band_width_) {
            band_plan_ = make_band_plan();
            band_plan_width_ = band_width_;
            hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            if (d_band_blocks_) (void)hipFree(d_band_blocks_);
            if (d_band_fill_) (void)hipFree(d_band_fill_);
            d_band_blocks_ = nullptr;
            d_band_fill_ = nullptr;
            if (band_plan_.usable) {
                hip_check(hipMalloc((void **)&d_band_blocks_, band_plan_.blocks.size() * sizeof(BandBlock)), "hipMalloc(band blocks)");
                hip_check(hipMalloc((void **)&d_band_fill_, band_plan_.fill_to.size() * sizeof(int)), "hipMalloc(band fill)");
                hip_check(hipMemcpy(d_band_blocks_, band_plan_.blocks.data(), band_plan_.blocks.size() * sizeof(BandBlock), hipMemcpyHostToDevice), "hipMemcpy");
                hip_check(hipMemcpy(d_band_fill_, band_plan_.fill_to.data(), band_plan_.fill_to.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
            }
        }
        if (!band_plan_.usable) return false;
        const BandPlan &p = band_plan_;
        BandArgs a;
        a.reads = d_reads;
        a.refs = d_refs;
        a.scores = d_scores;
        a.blocks = d_band_blocks_;
        a.fill_to = d_band_fill_;
        a.n = n;
        a.R = R_;
        a.F = F_;
        a.nb = p.nb;
        a.pad_rows = p.pad_rows;
        a.d = p.d;
        a.ring_depth = p.ring_depth;
        a.code_cols = p.code_cols;
        a.match = (short)sc_.match;
        a.mismatch = (short)sc_.mismatch;
        a.gap_read = (short)sc_.gap_read;
        a.gap_ref = (short)sc_.gap_ref;
        const bool sym = sc_.gap_read == sc_.gap_ref && !no_sym_;
        const void *fn = p.unit_delay ? (sym ? (const void *)&score_band_kernel<kBandK, true, true> : (const void *)&score_band_kernel<kBandK, false, true>)
                                      : (sym ? (const void *)&score_band_kernel<kBandK, true, false> : (const void *)&score_band_kernel<kBandK, false, false>);
        int lds = BandLds<kBandK>::total(p.code_cols, p.ring_depth);
        if (const char *pad = getenv("VALIGN_HIP_BAND_LDS_PAD")) lds += atoi(pad);       // experiment: fewer waves per CU
        // as many one-wave blocks as run side by side; each takes quads of pairs in turn (band_kernels.hip.h)
        int per_cu = 0;
        hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kWave, (size_t)lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
        band_blocks_per_cu_ = per_cu;
        band_lds_ = lds;
        const long long resident = (long long)std::max(per_cu, 1) * std::max(cu_count_, 1);
        const long long blocks = no_band_persist_ ? (n + 3) / 4 : std::min<long long>((n + 3) / 4, resident);
        if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
        void *kargs[] = {&a};
        hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(kWave), kargs, (size_t)lds, stream), "hipLaunchKernel(score_band_kernel)");
        return true;
    }
    bool band_chain_in_use() const {
        return !no_band_chain_ && !sc_.affine && band_width_ > 0 && (band_plan_width_ == band_width_ ? band_plan_.usable : make_band_plan().usable);
    }

    // Long sequences: strips of kLongG*kLongK rows, boundary rows through an HBM scratch.
    void score_long_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores,
                           hipStream_t stream, bool wide) {
        if (band_width_ > 0 && alg != kAlgSW)
            throw std::runtime_error("band_width applies to Smith-Waterman scores only");
        // linear gaps, banded: the cyclic block chain (int32 cells whatever score_width says: same results in the int16 range)
        if (band_width_ > 0 && alg == kAlgSW && score_band_device(n, d_reads, d_refs, d_scores, stream)) return;
        const int rows = kLongG * kLongK;
        const int ppw = 2 * (kWave / kLongG);
        LongArgs a;
        a.R = R_;
        a.F = F_;
        a.strips = std::max(1, (R_ + rows - 1) / rows);
        a.row_dwords = ((F_ + kLongG + kPhase - 1) / kPhase) * kPhase + kPhase;
        a.band_half = (band_width_ > 0 && alg == kAlgSW) ? band_width_ / 2 : -1;
        a.match = (short)sc_.match;
        a.mismatch = (short)sc_.mismatch;
        a.gap_read = (short)sc_.gap_read;
        a.gap_ref = (short)sc_.gap_ref;
        a.open_read = (short)sc_.open_read;
        a.ext_read = (short)sc_.ext_read;
        a.open_ref = (short)sc_.open_ref;
        a.ext_ref = (short)sc_.ext_ref;
        const int row_sets = (wide ? 2 : 1) * (sc_.affine ? 2 : 1);        // boundary rows per pair-of-pairs: per half (int32), H and F (affine)
        const size_t bytes_per_wave = (size_t)2 * (ppw / 2) * a.row_dwords * 4 * row_sets;
        long long chunk = (long long)((8ull << 30) / bytes_per_wave) * ppw;
        chunk = std::max<long long>(ppw, std::min(chunk, (n + ppw - 1) / ppw * ppw));
        const long long waves = chunk / ppw;
        if ((size_t)waves * bytes_per_wave > brow_bytes_) {
            hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            if (d_brow_) (void)hipFree(d_brow_);
            d_brow_ = nullptr;
            brow_bytes_ = (size_t)waves * bytes_per_wave;
            hip_check(hipMalloc((void **)&d_brow_, brow_bytes_), "hipMalloc(boundary rows)");
        }
        const bool affine_sym = sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_;
        const void *fn = sc_.affine ? kLongAffineKernels[alg][affine_sym ? 1 : 0][wide ? 1 : 0]
                                    : kLongKernels[alg][(sc_.gap_read == sc_.gap_ref && !no_sym_) ? 1 : 0][wide ? 1 : 0];
        int long_lds = sc_.affine ? LongLds<kLongG, kLongK, true>::kTotal : LongLds<kLongG, kLongK, false>::kTotal;
        if (const char *pad = getenv("VALIGN_HIP_LONG_LDS")) long_lds = std::max(long_lds, atoi(pad));     // tuning switch: fewer waves per CU
        for (long long begin = 0; begin < n; begin += chunk) {
            const long long cnt = std::min(chunk, n - begin);
            a.reads = d_reads + (size_t)begin * R_;
            a.refs = d_refs + (size_t)begin * F_;
            a.scores = d_scores + begin;
            a.brow = d_brow_;
            a.n = cnt;
            a.pp_total = waves * (ppw / 2) * row_sets;
            void *kargs[] = {&a};
            hip_check(hipLaunchKernel(fn, dim3((unsigned)((cnt + ppw - 1) / ppw)), dim3(kWave), kargs,
                                      (size_t)long_lds, stream),
                      "hipLaunchKernel(score_long_kernel)");
        }
    }

    // align_fill_affine_tag_kernel keeps 8 * cell + tag in int16; SW needs open scores < 0 (the tag rides on the
    // open constant) and, for the lane key, value << 4 (5) in range
    bool affine_tagged_range_ok(int alg) const {
        long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
        const int worst = std::min({sc_.open_read, sc_.open_ref, sc_.ext_read, sc_.ext_ref, sc_.mismatch, 0});
        // NW: every cell is at least the path "one gap up, one gap left"; E / F sit one open below H
        long long lo = alg == kAlgSW ? worst
                                     : 2ll * (std::min(sc_.open_read, 0) + std::min(sc_.open_ref, 0)) +
                                           (long long)(R_ + F_ + 2) * std::min({sc_.ext_read, sc_.ext_ref, 0}) + worst;
        if (alg == kAlgNW) {        // the kernel's tilted frame: cell (p, j) carries - ext_ref * p - ext_read * j on top
            const long long rows = (long long)plan_.geo->G * plan_.geo->K + 1, cols = F_ + 1;
            hi += std::max(0, -sc_.ext_ref) * rows + std::max(0, -sc_.ext_read) * cols;
            lo += std::min(0, -sc_.ext_ref) * rows + std::min(0, -sc_.ext_read) * cols;
            if (std::abs((long long)sc_.ext_ref) * rows > 3500 || std::abs((long long)sc_.ext_read) * cols > 3500) return false;
        }
        if (alg == kAlgSW && (sc_.open_read >= 0 || sc_.open_ref >= 0)) return false;
        const int key_bits = plan_.geo->K <= 16 ? 4 : 5;
        if (alg == kAlgSW && ((hi + 1) << key_bits) > 32000) return false;
        return 8 * hi + 8 <= 32000 && 8 * lo - 8 >= -28000 && std::abs(sc_.match) < 1000 && std::abs(sc_.mismatch) < 1000;
    }

    // align_fill_tag_kernel keeps 4 * cell + tag in int16
    bool tagged_range_ok(int alg, int rows) const {     // rows: padded rows of the sweep that would run
        long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
        if (alg == kAlgNW && !sse_policy_)           // the kernel's tilted frame: every cell plus -gap_ref * p - gap_read * j
            hi += (long long)-sc_.gap_ref * (rows + 1) + (long long)-sc_.gap_read * (F_ + 1);
        const int worst = std::min({sc_.gap_read, sc_.gap_ref, sc_.mismatch, 0});
        const long long lo = alg == kAlgSW ? worst : (long long)(R_ + F_ + 2) * worst;      // H(i,j) >= i gf + j gr
        if (alg == kAlgSW && !sse_policy_ && sc_.gap_ref >= 0) return false;
        return 4 * hi + 4 <= 32000 && 4 * lo - 4 >= -32000 && std::abs(sc_.match) < 2000 && std::abs(sc_.mismatch) < 2000;
    }

    // what score_alignments computes in for this mode at the engine's full shape
    const char *score_cell_format(int alg) const {
        if (alg > 1) return "none";
        if (alg == kAlgSW && band_chain_in_use()) return "int32";
        if (score_width_ == 32 || (score_width_ == 0 && !int16_range_ok(alg))) return "int32";
        if (!plan_.long_mode && sc_.affine && !no_f16_ && half_float_exact(alg, R_, F_, plan_.geo->G * plan_.geo->K)) return "f16";
        if (!plan_.long_mode && !sc_.affine && ((sc_.gap_read == sc_.gap_ref && !no_sym_) || alg == kAlgNW) && !no_f16_ &&
            (alg == kAlgNW ? half_float_exact(alg, R_, F_, plan_.geo->G * plan_.geo->K) : half_float_unit_exact(R_, F_)))
            return "f16";
        return "int16";
    }

    // Every cell of an R x F sweep and everything added to it stays an integer of magnitude <= 2048:
    // exact in half floats (kGapAffineSymF16 / kGapAffineF16).  SW cells are >= 0; cells of the NW
    // variant are bounded below by the cheaper border path (as in check_int16_range).
    // NW: plus what the kernels' tilted frame adds to a cell of a sweep of `rows` padded rows (score_kernel).
    bool half_float_exact(int alg, int R, int F, int rows) const {
        const long long top = (long long)std::min(R, F) * std::max({sc_.match, sc_.mismatch, 0});
        long long slack = std::max({std::abs(sc_.match), std::abs(sc_.mismatch), std::abs(sc_.open_read),
                                    std::abs(sc_.ext_read), std::abs(sc_.open_ref), std::abs(sc_.ext_ref)});
        if (alg == kAlgSW) return top + 2 * slack <= 2048 && slack <= 1024;
        // NW frame: H' of cell (p, j) is at least what its row or its column adds (the border path along the other axis
        // is free there) less one opening, at most top + the far corner's tilt; E' / F' sit at most one opening below H'.
        // The kernel centres that range on zero (nw_frame_centre, same formula).
        if (!sc_.affine) slack = std::max<long long>(slack, std::max(std::abs(sc_.gap_read), std::abs(sc_.gap_ref)));
        const long long span = nw_tilt_span(rows, F);
        const long long centre = (top + span) / 2;
        return span < 30000 && (top + span - centre) + 3 * slack <= 2048 && centre + 3 * slack <= 2048 && slack <= 512;
    }

    // kGapSymF16 for Smith-Waterman scales every value by 2^-10 and floors with the [0, 1] clamp of the
    // packed add: cells must stay below 1024, scores be integers of magnitude < 1024
    bool half_float_unit_exact(int R, int F) const {
        const long long top = (long long)std::min(R, F) * std::max({sc_.match, sc_.mismatch, 0});
        const long long slack = std::max({std::abs(sc_.match), std::abs(sc_.mismatch), std::abs(sc_.gap_read), std::abs(sc_.gap_ref)});
        return top + 2 * slack < 1024 && slack < 512;
    }

    // int16 DP cells: the reference wraps silently.  Scores switch to int32 cells on the strip path
    // where they could; alignments (int16 only) are refused.
    // The NW score kernels keep cell (p, j) plus -g_ref * p - g_read * j (g: gap / extension scores, <= 0): the most that
    // adds over a sweep of `rows` padded rows and F columns.
    long long nw_tilt_span(int rows, int F) const {
        const long long per_row = -(long long)(sc_.affine ? sc_.ext_ref : sc_.gap_ref);
        const long long per_col = -(long long)(sc_.affine ? sc_.ext_read : sc_.gap_read);
        return per_row * (rows + 1) + per_col * (F + 1);
    }
    int widest_sweep_rows() const {
        int rows = 0;
        if (!plan_.long_mode && plan_.geo) rows = plan_.geo->G * plan_.geo->K;
        if (latency_plan_.geo && !latency_plan_.long_mode) rows = std::max(rows, latency_plan_.geo->G * latency_plan_.geo->K);
        return rows;
    }

    bool int16_range_ok(int alg) const {
        try {
            check_int16_range(alg, true);
            return true;
        } catch (const std::runtime_error &) {
            return false;
        }
    }

    // score_path: score_alignments' register sweep (the NW variant's tilted frame counts)
    void check_int16_range(int alg, bool score_path = false) const {
        long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
        if (score_path && alg == kAlgNW && !plan_.long_mode) hi += nw_tilt_span(widest_sweep_rows(), F_);
        const int worst_gap = std::min({sc_.gap_read, sc_.gap_ref, sc_.open_read, sc_.open_ref, sc_.ext_read, sc_.ext_ref, 0});
        // SW cells are >= 0; NW-variant score cells are bounded below by the cheaper border path
        const long long lo = alg == kAlgSW ? (long long)std::min(sc_.mismatch, 0) + worst_gap
                                           : (long long)(std::min(R_, F_) + 2) * std::min(worst_gap, std::min(sc_.mismatch, 0));
        if (hi > 32000 || lo < -32000 || (sc_.affine && alg == kAlgNW && lo < -15000))
            throw std::runtime_error("shape x scoring can leave the int16 range of the DP cells (read_length " +
                                     std::to_string(R_) + ", ref_length " + std::to_string(F_) + ")");
    }

    // Host pointers in, host scores out.  Chunked over kSlots pinned slots, each with its own
    // stream: while the kernel of chunk c runs, chunk c+1 crosses PCIe and the host threads gather
    // chunk c+2 (two slots would serialise copy and kernel of a chunk behind the gather).
    //
    // Smith-Waterman chunks are length-sorted on the way (the reference pads
    // every sequence to the longest, src/util/versalignUtil.cpp:17-33, and sweeps the padding):
    // trailing bytes that are not ACGT score 0 against everything (DefaultKernel.h:83-97), so with
    // gap scores <= 0 no cell of a trailing row or column can exceed the maximum already seen and
    // the SW score of the trimmed pair is the score of the padded one.  Pairs are binned by trimmed
    // (read, ref) length class, each bin is packed at its own strides and swept by the geometry
    // that suits it; scores return through the permutation.  Bit-exact by construction, checked in
    // tests/test_gpu_ragged.py.
    // `d_dest` (the plugin's hip_devices_allgather): the scores stay on the device, pair i at d_dest[i], and `scores` is not
    // touched -- the caller gathers the shards of all devices there (RCCL) before anything goes to the host.
    void score_host(int opt, int n, const char *const *reads, const char *const *refs, short *scores,
                    int threads, int16_t *d_dest = nullptr) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const size_t per_pair = (size_t)R_ + F_;
        long long chunk = per_pair ? (long long)(score_chunk_bytes_ / per_pair) : n;
        chunk = whole_rounds(chunk);
        chunk = std::max<long long>(chunk, 1024);
        chunk = std::min<long long>(chunk, n);
        reset_pipeline();
        ensure_staging(chunk);
        if (threads < 1) threads = 1;
        threads = std::min(threads, 64);
        if (direct_call(n, per_pair) && (!ragged_applies(alg) || d_dest)) {      // (length-sorted batching is a property of the pipeline)
            // Small call (the reference's timing loop is 100 of them back to back, src/impl/main.cpp:278-287): what
            // it costs is API calls, not bytes.  The kernel reads the gathered sequences straight out of the pinned
            // staging over PCIe and writes its scores into pinned host memory: one launch and one wait instead of
            // three copies, a launch, an event and four event waits.
            host_stats_ = HostStats{};
            auto t0 = std::chrono::steady_clock::now();
            gather(reads, refs, n, h_reads_[0], h_refs_[0], threads);
            auto t1 = std::chrono::steady_clock::now();
            score_device(opt, n, dev_view(h_reads_[0]), dev_view(h_refs_[0]), d_dest ? d_dest : (int16_t *)dev_view(h_scores_[0]), streams_[0]);
            hip_check(hipStreamSynchronize(streams_[0]), "hipStreamSynchronize");
            auto t2 = std::chrono::steady_clock::now();
            if (!d_dest) memcpy(scores, h_scores_[0], sizeof(short) * (size_t)n);
            host_stats_.gather_ms = ms_between(t0, t1);
            host_stats_.wait_ms = ms_between(t1, t2);
            host_stats_.drain_ms = ms_between(t2, std::chrono::steady_clock::now());
            host_stats_.direct = 1;
            return;
        }
        // length-sorted batching: the decision is the host's (a sample of the call's tails), the work the device's -- every chunk
        // is classified, packed by length class and swept class by class in HBM (ragged_kernels.hip.h)
        const bool ragged = !d_dest && ragged_applies(alg) && ragged_fits(chunk) &&
                            (ragged_ == 2 || sampled_cell_fraction(reads, refs, n) < 0.67);
        const bool shared_scratch = plan_.long_mode || score_width_ == 32 || !int16_range_ok(alg);
        host_stats_ = HostStats{};
        auto drain = [&](int s) {
            if (slot_pending_[s] <= 0) return;
            if (d_dest) {                          // (the kernels wrote the device destination themselves)
                slot_pending_[s] = 0;
                return;
            }
            memcpy(scores + slot_begin_[s], h_scores_[s], sizeof(short) * (size_t)slot_pending_[s]);
            slot_pending_[s] = 0;
        };
        // what follows a chunk's kernels: the scores' way home and the slot's event.  A length-sorted chunk gets there one
        // iteration late: its classification runs on the device while the host gathers the next chunk, and only then does
        // the host read the histogram, lay the groups out and launch the sweeps (ragged_finish) -- no wait in between.
        auto finish_chunk = [&](int s, long long pairs) {
            hipStream_t cs = streams_[shared_scratch ? 0 : s];
            if (ragged) (void)ragged_finish(s, alg, pairs, d_scores_[s], cs, true);
            if (!d_dest)
                hip_check(hipMemcpyAsync(h_scores_[s], d_scores_[s], sizeof(short) * (size_t)pairs, hipMemcpyDeviceToHost, cs), "D2H scores");
            hip_check(hipEventRecord(slot_done_[s], cs), "hipEventRecord");
        };
        // (two iterations late, in fact: one gather is about as long as a chunk's copy + classification, two leave room)
        struct OpenChunk {
            int slot;
            long long pairs;
        };
        std::vector<OpenChunk> open;               // length-sorted chunks whose sweeps are not launched yet, oldest first
        constexpr size_t kOpenChunks = 2;          // (< kSlots - 1: a slot comes round again only after its chunk is finished)
        int slot = 0;
        // Ramp: the device idles until the first chunk is gathered and copied, so the first chunks are short (a quarter,
        // then half a chunk); chunks of many calls deep in the pipeline stay large (fewer launches, full waves).
        long long chunk_no = 0, cnt = 0;
        for (long long begin = 0; begin < n; begin += cnt, slot = (slot + 1) % kSlots, ++chunk_no) {
            const long long ramp = (ramp_ && n > 2 * chunk) ? (chunk_no == 0 ? whole_rounds(chunk / 4) : (chunk_no == 1 ? whole_rounds(chunk / 2) : chunk)) : chunk;
            cnt = std::min<long long>(std::max<long long>(ramp, 1024), n - begin);
            auto t0 = std::chrono::steady_clock::now();
            hip_check(hipEventSynchronize(slot_done_[slot]), "hipEventSynchronize");
            auto t1 = std::chrono::steady_clock::now();
            drain(slot);                            // the result of the chunk that used this slot
            auto t2 = std::chrono::steady_clock::now();
            host_stats_.wait_ms += ms_between(t0, t1);
            host_stats_.drain_ms += ms_between(t1, t2);
            // kernels that share a scratch (strip boundary rows) stay on one stream
            hipStream_t st = streams_[shared_scratch ? 0 : slot];
            const bool sweep_now = !ragged;
            if (pack_) {
                // two base classes per byte across PCIe, expanded in HBM to the canonical byte of each class
                const size_t PR = packed_length(R_), PF = packed_length(F_);
                packer_.gather_packed(reads + begin, refs + begin, cnt, h_reads_[slot], h_refs_[slot], threads);
                host_stats_.gather_ms += ms_between(t2, std::chrono::steady_clock::now());
                hip_check(hipMemcpyAsync(d_pack_reads_[slot], h_reads_[slot], (size_t)cnt * PR, hipMemcpyHostToDevice, st), "H2D reads (classes)");
                hip_check(hipMemcpyAsync(d_pack_refs_[slot], h_refs_[slot], (size_t)cnt * PF, hipMemcpyHostToDevice, st), "H2D refs (classes)");
                launch_unpack(d_pack_reads_[slot], d_reads_[slot], cnt, R_, st);
                launch_unpack(d_pack_refs_[slot], d_refs_[slot], cnt, F_, st);
                host_stats_.packed = 1;
            } else {
                gather(reads + begin, refs + begin, cnt, h_reads_[slot], h_refs_[slot], threads);
                host_stats_.gather_ms += ms_between(t2, std::chrono::steady_clock::now());
                hip_check(hipMemcpyAsync(d_reads_[slot], h_reads_[slot], (size_t)cnt * R_, hipMemcpyHostToDevice, st), "H2D reads");
                hip_check(hipMemcpyAsync(d_refs_[slot], h_refs_[slot], (size_t)cnt * F_, hipMemcpyHostToDevice, st), "H2D refs");
            }
            if (sweep_now) {
                score_device(opt, cnt, d_reads_[slot], d_refs_[slot], d_dest ? d_dest + begin : d_scores_[slot], st, false);
                finish_chunk(slot, cnt);
            } else {
                ragged_begin(slot, cnt, d_reads_[slot], d_refs_[slot], st);
                open.push_back(OpenChunk{slot, cnt});
                if (open.size() > kOpenChunks) {
                    finish_chunk(open.front().slot, open.front().pairs);
                    open.erase(open.begin());
                }
            }
            slot_begin_[slot] = begin;
            slot_pending_[slot] = cnt;
        }
        for (const OpenChunk &c : open) finish_chunk(c.slot, c.pairs);
        for (int k = 0; k < kSlots; ++k) {          // oldest chunk first
            const int s = (slot + k) % kSlots;
            auto t0 = std::chrono::steady_clock::now();
            hip_check(hipEventSynchronize(slot_done_[s]), "hipEventSynchronize");
            auto t1 = std::chrono::steady_clock::now();
            drain(s);
            host_stats_.wait_ms += ms_between(t0, t1);
            host_stats_.drain_ms += ms_between(t1, std::chrono::steady_clock::now());
        }
    }


    // ---- compute_alignments ----

    // Device-resident batch -> rows (n * 2 * (R+F) bytes: read row then ref row, right-justified,
    // zero before the start, NUL at R+F-1) and idx (n * 4 shorts).  Asynchronous on `stream`;
    // the pointer scratch is reused chunk after chunk in stream order.
    // `chain` (the chunk pipeline of align_host): the batch is ONE chunk of a sequence of calls.  Its traceback then runs on
    // the engine's helper stream behind the fill, in region `chain->region` (0 / 1) of a pointer scratch sized for two chunks
    // of `chain->chunk_pairs` pairs, and `stream` does NOT wait for it -- the fill of the next chunk (other region) runs
    // beside this walk; the caller chains whatever needs the rows behind trace_done(region).  Returns false where the call
    // ran in stream order instead (row strips, a chunk larger than half the scratch cap): everything is then on `stream`.
    struct WalkChain {
        int region;
        long long chunk_pairs;
    };
    hipEvent_t trace_done(int region) const { return trace_done_[region]; }

    bool align_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                      short *d_idx, hipStream_t stream, const WalkChain *chain = nullptr) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return false;
        // NW-variant alignments whose cells leave int16 (the reference's shorts would wrap): int32 cells on the row-strip path,
        // one pair per register -- linear gaps, default tie-breaks; anything else that leaves the range is refused
        const bool border_bad = alg == kAlgNW && (long long)(R_ + 1) * std::min({sc_.gap_ref, sc_.open_ref, sc_.ext_ref, 0}) < (sc_.affine ? -15000 : -32000);
        const bool wide_ok = alg == kAlgNW && !sc_.affine && !sse_policy_;
        if (wide_ok && (border_bad || !int16_range_ok(alg) || wide_align_)) {
            hip_check(hipSetDevice(device_), "hipSetDevice");
            align_strips_device(alg, n, d_reads, d_refs, d_rows, d_idx, stream, true);
            return false;
        }
        check_int16_range(alg);
        if (border_bad) throw std::runtime_error("NW alignment border (read_length * gap score) leaves the int16 range");
        hip_check(hipSetDevice(device_), "hipSetDevice");
        if (plan_.long_mode) {
            align_strips_device(alg, n, d_reads, d_refs, d_rows, d_idx, stream);
            return false;
        }
        const int G = plan_.geo->G, K = plan_.geo->K, AL = R_ + F_;
        // affine gaps with the traceback information tagged into the cells (4-bit codes, 4-step blocks)
        const bool affine_tagged = sc_.affine && !sse_policy_ && !no_tag_ && affine_tagged_range_ok(alg);
        int blocks8 = affine_tagged ? (F_ + G - 1 + 3) / 4 : (F_ + G - 1 + 7) / 8;            // blocks of steps per lane
        const long long ppb = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
        const size_t bytes_per_pp = (size_t)G * blocks8 * K * 4 * ((sc_.affine && !affine_tagged) ? 2 : 1);
        // Pointer scratch: as much of the batch per launch as memory allows (a 1 M-pair launch keeps
        // the traceback kernel at full occupancy), capped at 64 GiB -- one launch for a million affine pairs of
        // 150 x 500 (43.6 GB) on a 288 GB device -- and half the free HBM.
        size_t free_b = 0, total_b = 0;
        hip_check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
        const size_t have = trace_bytes_;
        size_t cap = std::min<size_t>(64ull << 30, std::max<size_t>((free_b + have) / 2, 256ull << 20));
        if (scratch_cap_mb_ > 0) cap = std::min<size_t>(cap, (size_t)scratch_cap_mb_ << 20);
        long long chunk = (long long)(cap / bytes_per_pp) * 2;
        chunk = std::max(ppb, chunk / ppb * ppb);
        const long long chain_pairs = chain ? (std::max(chain->chunk_pairs, n) + ppb - 1) / ppb * ppb : 0;
        if (chain && (2 * chain_pairs > chunk || no_overlap_)) chain = nullptr;        // two regions do not fit: stream order
        if (!chain) chain_regions_busy_[0] = chain_regions_busy_[1] = false;
        chunk = chain ? 2 * chain_pairs : std::min(chunk, (n + ppb - 1) / ppb * ppb);
        ensure_trace_scratch(chunk, bytes_per_pp, stream);
        if (sse_policy_ && sc_.affine)
            throw std::runtime_error("traceback_policy = 1 (SSE/AVX tie-breaks) exists for the linear gap model only");
        // linear gaps, Default tie-breaks: the pointer rides in the low bits of the cell where 4x the cell
        // range still fits int16 (and, for SW, gap_ref < 0); otherwise the equality-test kernels
        const bool tagged = !sc_.affine && !no_tag_ && tagged_range_ok(alg, G * K);        // (both tie-break policies)
        // SW: one (value, row) key per lane instead of a first-arg-max per row where value << 4 (5 bits of
        // row for more than 16 rows per lane) still fits int16
        const long long key_top = ((long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1) << (plan_.geo->K <= 16 ? 4 : 5);
        const bool lane_key = tagged && alg == kAlgSW && key_top <= 32000;
        // ... and where 64x the cell range fits (K <= 16), the key rides in the query profile instead of being computed
        const bool prof_key = lane_key && !sse_policy_ && !no_prof_key_ && K <= 16 &&
                              (((long long)std::min(R_, F_) * std::max(sc_.match, 0) + 2) << 6) <= 32000 &&
                              64ll * std::max(std::abs(sc_.gap_read), std::abs(sc_.gap_ref)) < 32000 && 64ll * std::abs(sc_.mismatch) < 16000;
        const bool affine_sym = sc_.affine && sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_;
        const void *fn = plan_.geo->fill[alg][prof_key ? 11 : tagged ? (sse_policy_ ? (lane_key ? 10 : 9) : (lane_key ? 5 : 4)) : (sse_policy_ ? 3 : (sc_.affine ? (affine_tagged ? (affine_sym ? 8 : 7) : (affine_sym ? 6 : 2)) : ((sc_.gap_read == sc_.gap_ref && !no_sym_) ? 1 : 0)))];
        const int block_lds = plan_.lds.total * plan_.waves_per_block;
        if (block_lds > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, block_lds),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        // Parts of the batch: the traceback of one part runs on a helper stream beside the fill of the next (the walk
        // waits on memory at 17 % VALU issue, the fill owns the VALU).  A batch that fits the scratch in one piece is cut
        // 7/8 + 1/8 -- the short fill covers the long walk, what stays exposed is the walk of the last eighth (cuts between
        // 3/4 and 7/8 measure the same, finer ones lose to the second fill's own tail); a batch
        // that needs several chunks alternates between the two halves of the scratch.
        struct Part { long long begin, cnt, slot; int region; };
        std::vector<Part> parts;
        const bool overlap = !no_overlap_ && (double)n * R_ * F_ >= 1e10 && !chain;
        if (chain) {
            parts.push_back(Part{0, n, chain->region * chain_pairs, chain->region});
        } else if (overlap && chunk >= n && n >= 16 * ppb && split_parts_ > 2) {
            // Geometric parts (3/4 of what is left each time, regions alternating): every walk but the last runs beside the
            // next, shorter fill and what stays exposed is the walk of a sliver.  Part 0 sits in region 0, part 1 behind it in
            // region 1; later parts are smaller than the first occupant of their region.
            long long begin = 0;
            long long first = 0;
            for (int k = 0; k < split_parts_ && begin < n; ++k) {
                long long cnt = (k + 1 == split_parts_) ? n - begin : std::max(ppb, (n - begin) * 3 / 4 / ppb * ppb);
                if (n - begin - cnt < 8 * ppb) cnt = n - begin;                  // no slivers below a few blocks
                if (k == 0) first = cnt;
                parts.push_back(Part{begin, cnt, (k & 1) ? first : 0, k & 1});
                begin += cnt;
            }
        } else if (overlap && chunk >= n && n >= 16 * ppb) {
            const long long big = std::max(ppb, n * 7 / 8 / ppb * ppb);
            parts.push_back(Part{0, big, 0, 0});
            parts.push_back(Part{big, n - big, big, 1});
        } else if (overlap && chunk < n && chunk >= 4 * ppb) {
            const long long half = chunk / 2 / ppb * ppb;
            for (long long begin = 0, i = 0; begin < n; begin += half, ++i)
                parts.push_back(Part{begin, std::min(half, n - begin), (i & 1) * half, (int)(i & 1)});
        } else {
            for (long long begin = 0; begin < n; begin += chunk) parts.push_back(Part{begin, std::min(chunk, n - begin), 0, 0});
        }
        const bool helper = (parts.size() > 1 && overlap) || chain;
        if (chain) {
            // rows are zeroed on the helper stream right before the walk that writes them (the caller has made sure the
            // previous user of d_rows is done: its copy-out event was waited for on the host)
            ensure_trace_stream();
        } else if (helper) {
            // the result rows are zeroed on the helper stream too (1.4 GB per million pairs of 150 x 500: the fills do not
            // touch them), behind whatever the caller's stream was still doing with them
            ensure_trace_stream();
            hip_check(hipEventRecord(entry_ev_, stream), "hipEventRecord");
            hip_check(hipStreamWaitEvent(trace_stream_, entry_ev_, 0), "hipStreamWaitEvent");
            hip_check(hipMemsetAsync(d_rows, 0, (size_t)n * 2 * AL, trace_stream_), "hipMemsetAsync(rows)");
        } else if (!chain) {
            hip_check(hipMemsetAsync(d_rows, 0, (size_t)n * 2 * AL, stream), "hipMemsetAsync(rows)");
        }
        bool region_used[2] = {chain && chain_regions_busy_[0], chain && chain_regions_busy_[1]};
        for (const Part &part : parts) {
            const long long begin = part.begin, cnt = part.cnt;
            unsigned *part_ptr = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(d_ptr_) + (size_t)(part.slot / 2) * bytes_per_pp);
            EndCell *part_ends = d_ends_ + part.slot;
            if (helper && region_used[part.region])          // the region's previous walk must be over before it is overwritten
                hip_check(hipStreamWaitEvent(stream, trace_done_[part.region], 0), "hipStreamWaitEvent");
            FillArgs f;
            f.reads = d_reads + (size_t)begin * R_;
            f.refs = d_refs + (size_t)begin * F_;
            f.ptr = part_ptr;
            f.ends = part_ends;
            f.n = cnt;
            f.R = R_;
            f.F = F_;
            f.prof_area = plan_.lds.prof_area;
            f.refc_stride = plan_.lds.refc_stride;
            f.wave_lds = plan_.lds.total;
            f.blocks8 = blocks8;
            f.match = (short)sc_.match;
            f.mismatch = (short)sc_.mismatch;
            f.gap_read = (short)sc_.gap_read;
            f.gap_ref = (short)sc_.gap_ref;
            f.open_read = (short)sc_.open_read;
            f.ext_read = (short)sc_.ext_read;
            f.open_ref = (short)sc_.open_ref;
            f.ext_ref = (short)sc_.ext_ref;
            void *fargs[] = {&f};
            const long long blocks = (cnt + ppb - 1) / ppb;
            hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(plan_.waves_per_block * kWave), fargs,
                                      (size_t)block_lds, stream),
                      "hipLaunchKernel(align_fill_kernel)");
            TraceArgs t{};
            t.reads = f.reads;
            t.refs = f.refs;
            t.ptr = part_ptr;
            t.ends = part_ends;
            t.rows = d_rows + (size_t)begin * 2 * AL;
            t.idx = d_idx + (size_t)begin * 4;
            t.n = cnt;
            t.R = R_;
            t.F = F_;
            t.G = G;
            t.K = K;
            t.pad_rows = G * K - R_;
            t.blocks8 = blocks8;
            t.alg = alg;
            t.match = f.match;
            t.mismatch = f.mismatch;
            t.gap_read = f.gap_read;
            t.gap_ref = f.gap_ref;
            t.affine = sc_.affine ? 1 : 0;
            t.sse_policy = sse_policy_ ? 1 : 0;
            t.tagged = affine_tagged ? 2 : ((tagged && !sse_policy_) ? 1 : 0);     // SSE tags are the stored states
            t.open_read = f.open_read;
            t.ext_read = f.ext_read;
            t.open_ref = f.open_ref;
            t.ext_ref = f.ext_ref;
            void *targs[] = {&t};
            hipStream_t walk_stream = stream;
            if (helper) {
                hip_check(hipEventRecord(fill_done_[part.region], stream), "hipEventRecord");
                hip_check(hipStreamWaitEvent(trace_stream_, fill_done_[part.region], 0), "hipStreamWaitEvent");
                walk_stream = trace_stream_;
                if (chain) hip_check(hipMemsetAsync(d_rows, 0, (size_t)n * 2 * AL, trace_stream_), "hipMemsetAsync(rows)");
            }
            hip_check(hipLaunchKernel((const void *)&traceback_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256),
                                      targs, 0, walk_stream),
                      "hipLaunchKernel(traceback_kernel)");
            if (helper) {
                hip_check(hipEventRecord(trace_done_[part.region], trace_stream_), "hipEventRecord");
                region_used[part.region] = true;
            }
        }
        if (chain) {                                       // the walk is the caller's to wait for (trace_done(region))
            chain_regions_busy_[chain->region] = true;
            return true;
        }
        if (helper)                                        // the call stays asynchronous on `stream`: it ends when the walks have
            for (int r = 0; r < 2; ++r)
                if (region_used[r]) hip_check(hipStreamWaitEvent(stream, trace_done_[r], 0), "hipStreamWaitEvent");
        return false;
    }

    void ensure_trace_stream() {
        if (trace_stream_) return;
        hip_check(hipStreamCreateWithFlags(&trace_stream_, hipStreamNonBlocking), "hipStreamCreate(traceback)");
        hip_check(hipEventCreateWithFlags(&entry_ev_, hipEventDisableTiming), "hipEventCreate");
        for (int r = 0; r < 2; ++r) {
            hip_check(hipEventCreateWithFlags(&fill_done_[r], hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&trace_done_[r], hipEventDisableTiming), "hipEventCreate");
        }
    }

    // Fill + traceback of a small batch in one launch (linear gaps, default tie-breaks, tagged cells): false when the
    // shape / scoring has no fused kernel (the caller takes the three-kernel path).
    bool align_fused(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows, short *d_idx,
                     hipStream_t stream) {
        if (no_fused_ || sc_.affine || sse_policy_ || no_tag_ || plan_.long_mode || force_g_ || force_k_ || !tagged_range_ok(alg, 256)) return false;     // (256: the tallest fused geometry)
        if (wide_align_ && alg == kAlgNW) return false;
        try {
            check_int16_range(alg);
        } catch (const std::runtime_error &) {
            return false;                           // let the regular path raise its error
        }
        if (alg == kAlgNW && (long long)(R_ + 1) * std::min(sc_.gap_ref, 0) < -32000) return false;
        const FusedGeometry *best = nullptr;
        WaveLds best_lds{};
        int best_total = 0, best_blocks = 0;
        double best_cost = 0;
        for (const FusedGeometry &g : kFusedGeometries) {
            if (g.G * g.K < R_) continue;
            const WaveLds w = g.lds(R_, F_);
            const int blocks8 = (F_ + g.G - 1 + 7) / 8;
            const int total = g.total(w.total, R_, F_, blocks8);
            if (total > kMaxBlockLds - 8192) continue;
            const double cost = (double)(F_ + g.G - 1) * (g.K * 9.0 + 7.0);         // single-wave latency
            if (!best || cost < best_cost) {
                best = &g;
                best_lds = w;
                best_total = total;
                best_blocks = blocks8;
                best_cost = cost;
            }
        }
        if (!best) return false;
        FillArgs f{};
        f.reads = d_reads;
        f.refs = d_refs;
        f.n = n;
        f.R = R_;
        f.F = F_;
        f.prof_area = best_lds.prof_area;
        f.refc_stride = best_lds.refc_stride;
        f.wave_lds = best_lds.total;
        f.blocks8 = best_blocks;
        f.match = (short)sc_.match;
        f.mismatch = (short)sc_.mismatch;
        f.gap_read = (short)sc_.gap_read;
        f.gap_ref = (short)sc_.gap_ref;
        f.out_rows = d_rows;
        f.out_idx = d_idx;
        const void *fn = best->kernel[alg];
        if (best_total > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, best_total),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        const long long ppw = 2 * (kWave / best->G);
        void *fargs[] = {&f};
        hip_check(hipLaunchKernel(fn, dim3((unsigned)((n + ppw - 1) / ppw)), dim3(kWave), fargs, (size_t)best_total, stream),
                  "hipLaunchKernel(align_fill_tag_kernel, fused)");
        return true;
    }

    // Reads beyond one register sweep: row strips of 64 * K rows, one launch per strip in stream order, boundary
    // rows ping-pong through HBM, one pointer region per strip, then the same traceback kernel (strip_kernels.hip.h).
    // Linear or affine gaps, Default tie-breaks, int16 cells (the reference's; where they would wrap the call is refused
    // by check_int16_range above instead of wrapping silently).
    void align_strips_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                             short *d_idx, hipStream_t stream, bool wide = false) {
        const bool affine = sc_.affine;
        if (sse_policy_ && affine)
            throw std::runtime_error("traceback_policy = 1 (SSE/AVX tie-breaks) exists for the linear gap model only");
        const StripGeometry *geo = nullptr;
        WaveLds lds{};
        for (int budget : {kMaxBlockLds / 2, kMaxBlockLds}) {            // two waves per CU if possible
            for (const StripGeometry &g : kStripGeometries) {
                if (affine && !g.affine_kernel[alg]) continue;
                if (sse_policy_ && !g.sse_kernel[alg]) continue;
                if (wide && !g.wide_kernel) continue;
                const WaveLds w = g.lds(64 * g.K, F_);
                if (w.total <= budget && !geo) {
                    geo = &g;
                    lds = w;
                }
            }
            if (geo) break;
        }
        if (!geo) throw std::runtime_error("ref_length " + std::to_string(F_) + " does not fit the LDS of one CU");
        const int K = geo->K, rows = 64 * K, AL = R_ + F_;
        const int strips = std::max(1, (R_ + rows - 1) / rows), pad_total = strips * rows - R_;
        const int blocks8 = (F_ + 63 + 7) / 8;
        const int row_dwords = ((F_ + 71) / 64 + 2) * 64;
        const size_t strip_words = (size_t)blocks8 * 64 * K * (affine ? 2 : 1);    // per wave (= pair-of-pairs) and strip
        const int row_sets = (affine || wide) ? 2 : 1;                             // boundary rows: H, and F beside it (affine); one per pair (int32 cells)
        const size_t bytes_per_pp = strip_words * 4 * strips + (size_t)2 * row_sets * row_dwords * 4;
        size_t free_b = 0, total_b = 0;
        hip_check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
        size_t cap = std::min<size_t>(24ull << 30, std::max<size_t>((free_b + trace_bytes_) / 2, 256ull << 20));
        if (scratch_cap_mb_ > 0) cap = std::min<size_t>(cap, (size_t)scratch_cap_mb_ << 20);
        long long chunk = std::max<long long>(2, (long long)(cap / bytes_per_pp) * 2);
        chunk = std::min(chunk, (n + 1) / 2 * 2);
        const long long waves = chunk / 2;
        const size_t need = (size_t)waves * bytes_per_pp;
        if (need > trace_bytes_ || chunk > trace_pairs_ || (size_t)2 * n * sizeof(int) > first_bad_bytes_) {
            hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            if (need > trace_bytes_) {
                if (d_ptr_) (void)hipFree(d_ptr_);
                d_ptr_ = nullptr;
                trace_bytes_ = 0;
                hip_check(hipMalloc((void **)&d_ptr_, need), "hipMalloc(pointer scratch)");
                trace_bytes_ = need;
            }
            if (chunk > trace_pairs_) {
                if (d_ends_) (void)hipFree(d_ends_);
                d_ends_ = nullptr;
                trace_pairs_ = 0;
                hip_check(hipMalloc((void **)&d_ends_, sizeof(EndCell) * (size_t)chunk), "hipMalloc(end cells)");
                trace_pairs_ = chunk;
            }
            if ((size_t)2 * n * sizeof(int) > first_bad_bytes_) {
                if (d_first_bad_) (void)hipFree(d_first_bad_);
                d_first_bad_ = nullptr;
                first_bad_bytes_ = 0;
                hip_check(hipMalloc((void **)&d_first_bad_, (size_t)2 * n * sizeof(int)), "hipMalloc(first invalid positions)");
                first_bad_bytes_ = (size_t)2 * n * sizeof(int);
            }
        }
        unsigned *boundary = d_ptr_ + (size_t)waves * strip_words * strips;        // two rows per pair-of-pairs behind the pointers
        hip_check(hipMemsetAsync(d_rows, 0, (size_t)n * 2 * AL, stream), "hipMemsetAsync(rows)");
        hipLaunchKernelGGL(first_invalid_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, d_reads, d_refs, n, R_, F_, d_first_bad_,
                           sse_policy_ ? 1 : 0);
        hip_check(hipGetLastError(), "hipLaunchKernel(first_invalid_kernel)");
        const void *fn = wide ? geo->wide_kernel : (affine ? geo->affine_kernel[alg] : (sse_policy_ ? geo->sse_kernel[alg] : geo->kernel[alg]));
        if (lds.total > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds.total),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        for (long long begin = 0; begin < n; begin += chunk) {
            const long long cnt = std::min(chunk, n - begin), cnt_waves = (cnt + 1) / 2;
            for (int s = 0; s < strips; ++s) {
                StripArgs a;
                a.reads = d_reads + (size_t)begin * R_;
                a.refs = d_refs + (size_t)begin * F_;
                a.ptr = d_ptr_ + (size_t)s * cnt_waves * strip_words;
                a.ends = d_ends_;
                a.first_bad = d_first_bad_ + 2 * begin;
                a.top = boundary + (size_t)((s & 1) ^ 1) * row_sets * waves * row_dwords;
                a.bottom = boundary + (size_t)(s & 1) * row_sets * waves * row_dwords;
                a.top_f = a.top + (size_t)waves * row_dwords;                // (only read / written by the affine kernel)
                a.bottom_f = a.bottom + (size_t)waves * row_dwords;
                a.n = cnt;
                a.R = R_;
                a.F = F_;
                a.prof_area = lds.prof_area;
                a.refc_stride = lds.refc_stride;
                a.wave_lds = lds.total;
                a.blocks8 = blocks8;
                a.strip = s;
                a.strips = strips;
                a.row_dwords = row_dwords;
                a.match = (short)sc_.match;
                a.mismatch = (short)sc_.mismatch;
                a.gap_read = (short)sc_.gap_read;
                a.gap_ref = (short)sc_.gap_ref;
                a.open_read = (short)sc_.open_read;
                a.ext_read = (short)sc_.ext_read;
                a.open_ref = (short)sc_.open_ref;
                a.ext_ref = (short)sc_.ext_ref;
                void *kargs[] = {&a};
                hip_check(hipLaunchKernel(fn, dim3((unsigned)cnt_waves), dim3(kWave), kargs, (size_t)lds.total, stream),
                          "hipLaunchKernel(align_strip_kernel)");
            }
            TraceArgs t{};
            t.reads = d_reads + (size_t)begin * R_;
            t.refs = d_refs + (size_t)begin * F_;
            t.ptr = d_ptr_;
            t.ends = d_ends_;
            t.rows = d_rows + (size_t)begin * 2 * AL;
            t.idx = d_idx + (size_t)begin * 4;
            t.n = cnt;
            t.R = R_;
            t.F = F_;
            t.G = 64;
            t.K = K;
            t.pad_rows = pad_total;
            t.blocks8 = blocks8;
            t.alg = alg;
            t.match = (short)sc_.match;
            t.mismatch = (short)sc_.mismatch;
            t.gap_read = (short)sc_.gap_read;
            t.gap_ref = (short)sc_.gap_ref;
            t.affine = affine ? 1 : 0;
            t.sse_policy = sse_policy_ ? 1 : 0;
            t.open_read = (short)sc_.open_read;
            t.ext_read = (short)sc_.ext_read;
            t.open_ref = (short)sc_.open_ref;
            t.ext_ref = (short)sc_.ext_ref;
            t.strip_rows = rows;
            t.strip_words = (long long)(cnt_waves * strip_words);
            void *targs[] = {&t};
            hip_check(hipLaunchKernel((const void *)&traceback_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), targs, 0, stream),
                      "hipLaunchKernel(traceback_kernel)");
        }
    }

    // Host pointers in, Alignment[] out: the rows of every pair are fresh operator new[] blocks
    // (the host's ~Alignment delete[]s them, include/AlignmentKernel.h:20-23).
    // Host pointers in, Alignment[] out.  Three streams: copy-in (H2D), kernels (fill + traceback; they
    // share the pointer scratch, so one stream), copy-out (D2H), chained per chunk with events, over
    // kSlots staging slots -- the 1.3 KB/pair result copy of chunk c, the kernels of chunk c+1 and the
    // input copy of chunk c+2 overlap, and the host gathers / scatters (2n operator new[] blocks, which
    // the ABI demands) meanwhile.
    // `alignments`: the ABI's Alignment array (rows become operator new[] blocks), or a FlatSink (caller-provided
    // contiguous buffers, for FFI callers that do not want 2n heap blocks)
    using FlatSink = valign::FlatSink;      // host_pipeline.h

    template <typename Sink>
    void align_host(int opt, int n, const char *const *reads, const char *const *refs, Sink alignments,
                    int threads) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const int AL = R_ + F_;
        const size_t per_pair = (size_t)3 * AL + 8;
        long long chunk = per_pair ? (long long)(align_chunk_bytes_ / per_pair) : n;
        chunk = whole_rounds(chunk);
        chunk = std::max<long long>(chunk, 1024);
        chunk = std::min<long long>(chunk, n);
        reset_pipeline();
        ensure_staging(chunk);
        ensure_align_staging(chunk);
        if (threads < 1) threads = 1;
        threads = std::min(threads, 64);
        hipStream_t kernels = streams_[0], copy_in = streams_[1], copy_out = streams_[2];
        host_stats_ = HostStats{};
        if (direct_call(n, (size_t)AL)) {
            // Small call: one stream, no events.  The kernels read the sequences out of the pinned staging; rows
            // and coordinates land next to each other in one device buffer and come back in ONE copy (scattered
            // 4-byte stores over PCIe would cost a bus transaction each).
            auto t0 = std::chrono::steady_clock::now();
            gather(reads, refs, n, h_reads_[0], h_refs_[0], threads);
            auto t1 = std::chrono::steady_clock::now();
            const size_t rows_bytes = ((size_t)n * 2 * AL + 15) / 16 * 16, all_bytes = rows_bytes + sizeof(short) * 4 * (size_t)n;
            if (align_fused(alg, n, dev_view(h_reads_[0]), dev_view(h_refs_[0]), dev_view(h_rows_[0]),
                            (short *)(dev_view(h_rows_[0]) + rows_bytes), kernels)) {
                // ONE launch: the wave that fills a pair's pointers (kept in LDS) walks it back and writes the rows
                // straight into the pinned staging
                hip_check(hipStreamSynchronize(kernels), "hipStreamSynchronize");
                auto t2 = std::chrono::steady_clock::now();
                scatter(alignments, n, h_rows_[0], (const short *)(h_rows_[0] + rows_bytes), threads);
                host_stats_.gather_ms = ms_between(t0, t1);
                host_stats_.wait_ms = ms_between(t1, t2);
                host_stats_.drain_ms = ms_between(t2, std::chrono::steady_clock::now());
                host_stats_.direct = 2;
                return;
            }
            // rows and coordinates sit next to each other in the slot's row buffer (it has room for both) and come
            // back in ONE copy
            short *d_idx = (short *)(d_rows_[0] + rows_bytes);
            align_device(opt, n, dev_view(h_reads_[0]), dev_view(h_refs_[0]), d_rows_[0], d_idx, kernels);
            hip_check(hipMemcpyAsync(h_rows_[0], d_rows_[0], all_bytes, hipMemcpyDeviceToHost, kernels), "D2H rows + idx");
            hip_check(hipStreamSynchronize(kernels), "hipStreamSynchronize");
            auto t2 = std::chrono::steady_clock::now();
            scatter(alignments, n, h_rows_[0], (const short *)(h_rows_[0] + rows_bytes), threads);
            host_stats_.gather_ms = ms_between(t0, t1);
            host_stats_.wait_ms = ms_between(t1, t2);
            host_stats_.drain_ms = ms_between(t2, std::chrono::steady_clock::now());
            host_stats_.direct = 1;
            return;
        }
        // A flat destination in page-locked memory (valign_hip_host_register) IS the device layout: the copy engine
        // writes the caller's buffers directly and the host has nothing left to scatter.
        uint8_t *direct_rows = nullptr;
        short *direct_idx = nullptr;
        if (!no_direct_out_) flat_destination(alignments, n, direct_rows, direct_idx);
        host_stats_.direct_out = direct_rows ? 1 : 0;
        auto drain = [&](int s) {
            if (slot_pending_[s] <= 0) return;
            if (!direct_rows) {
                const auto t0 = std::chrono::steady_clock::now();
                scatter(alignments + slot_begin_[s], slot_pending_[s], h_rows_[s], h_idx_[s], threads);
                host_stats_.drain_ms += ms_between(t0, std::chrono::steady_clock::now());
            }
            slot_pending_[s] = 0;
        };
        int slot = 0;
        long long chunk_no = 0;
        chain_regions_busy_[0] = chain_regions_busy_[1] = false;       // (every earlier call ended with its walks waited for)
        prime_copy_engines(copy_in, copy_out, chunk);
        CopyIssuer *copy_issuer = nullptr;
        if (!d2h_on_stream_) {
            if (!copy_issuer_) copy_issuer_.reset(new CopyIssuer(device_));
            copy_issuer = copy_issuer_.get();
        }
        // An error in the middle of the pipeline must not leave copies in flight into the CALLER's buffers (registered
        // result buffers receive them directly): quiesce the issuer and the streams before the exception leaves.
        struct Quiesce {
            Engine *e;
            bool armed = true;
            ~Quiesce() {
                if (!armed) return;
                if (e->copy_issuer_) {
                    try {
                        e->copy_issuer_->wait_idle();
                    } catch (...) {
                    }
                }
                if (e->trace_stream_) (void)hipStreamSynchronize(e->trace_stream_);
                for (int s = 0; s < kSlots; ++s) (void)hipStreamSynchronize(e->streams_[s]);
                for (int s = 0; s < kSlots; ++s) e->slot_pending_[s] = 0;
            }
        } quiesce{this};
        for (long long begin = 0; begin < n; begin += chunk, slot = (slot + 1) % kSlots) {
            const long long cnt = std::min<long long>(chunk, n - begin);
            auto t0 = std::chrono::steady_clock::now();
            if (copy_issuer) copy_issuer->wait_issued(slot);             // (only then is the slot's event the one of its last chunk)
            hip_check(hipEventSynchronize(slot_done_[slot]), "hipEventSynchronize");   // its last chunk is back on the host
            host_stats_.wait_ms += ms_between(t0, std::chrono::steady_clock::now());
            drain(slot);
            t0 = std::chrono::steady_clock::now();
            gather(reads + begin, refs + begin, cnt, h_reads_[slot], h_refs_[slot], threads);
            host_stats_.gather_ms += ms_between(t0, std::chrono::steady_clock::now());
            hip_check(hipMemcpyAsync(d_reads_[slot], h_reads_[slot], (size_t)cnt * R_, hipMemcpyHostToDevice, copy_in), "H2D reads");
            hip_check(hipMemcpyAsync(d_refs_[slot], h_refs_[slot], (size_t)cnt * F_, hipMemcpyHostToDevice, copy_in), "H2D refs");
            hip_check(hipEventRecord(in_done_[slot], copy_in), "hipEventRecord");
            hip_check(hipStreamWaitEvent(kernels, in_done_[slot], 0), "hipStreamWaitEvent");
            // the walk of this chunk runs on the helper stream beside the fill of the next one (two scratch regions)
            const WalkChain chain{(int)(chunk_no & 1), chunk};
            const bool chained = align_device(opt, cnt, d_reads_[slot], d_refs_[slot], d_rows_[slot], d_idx_[slot], kernels, &chain);
            hip_check(hipEventRecord(kernels_done_[slot], chained ? trace_stream_ : kernels), "hipEventRecord");      // the chunk's last kernel
            ++chunk_no;
            uint8_t *rows_to = direct_rows ? direct_rows + (size_t)begin * 2 * AL : h_rows_[slot];
            short *idx_to = direct_idx ? direct_idx + 4 * begin : h_idx_[slot];
            if (copy_issuer) {
                // SDMA, not a blit kernel beside the next fill: the copies are issued once the host has seen the kernels end
                copy_issuer->submit(CopyIssuer::Job{kernels_done_[slot], {rows_to, idx_to}, {d_rows_[slot], d_idx_[slot]},
                                                    {(size_t)cnt * 2 * AL, sizeof(short) * 4 * (size_t)cnt}, copy_out, slot_done_[slot], slot});
            } else {
                hip_check(hipStreamWaitEvent(copy_out, kernels_done_[slot], 0), "hipStreamWaitEvent");
                hip_check(hipMemcpyAsync(rows_to, d_rows_[slot], (size_t)cnt * 2 * AL, hipMemcpyDeviceToHost, copy_out), "D2H rows");
                hip_check(hipMemcpyAsync(idx_to, d_idx_[slot], sizeof(short) * 4 * (size_t)cnt, hipMemcpyDeviceToHost, copy_out), "D2H idx");
                hip_check(hipEventRecord(slot_done_[slot], copy_out), "hipEventRecord");
            }
            slot_begin_[slot] = begin;
            slot_pending_[slot] = cnt;
        }
        for (int k = 0; k < kSlots; ++k) {              // oldest chunk first
            const int s = (slot + k) % kSlots;
            const auto t0 = std::chrono::steady_clock::now();
            if (copy_issuer) copy_issuer->wait_issued(s);
            hip_check(hipEventSynchronize(slot_done_[s]), "hipEventSynchronize");
            host_stats_.wait_ms += ms_between(t0, std::chrono::steady_clock::now());
            drain(s);
        }
        quiesce.armed = false;               // (everything has been waited for)
    }

    // host-side phases of the last score_host / align_host call
    std::string host_phases() const {
        char buf[200];
        snprintf(buf, sizeof buf, "{\"host_gather_ms\": %.3f, \"host_wait_ms\": %.3f, \"host_drain_ms\": %.3f, \"host_classify_ms\": %.3f, \"host_launch_ms\": %.3f}",
                 host_stats_.gather_ms, host_stats_.wait_ms, host_stats_.drain_ms, host_stats_.classify_ms, host_stats_.launch_ms);
        return buf;
    }

    std::string describe(int opt, long long n) const {
        const long long ppb = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
        char buf[1600];
        snprintf(buf, sizeof buf,
                 "{\"arch\": \"%s\", \"device\": %d, \"alg\": %d, \"affine\": %d, \"group_lanes\": %d, "
                 "\"rows_per_lane\": %d, \"padded_rows\": %d, \"pairs_per_wave\": %d, \"waves_per_block\": %d, "
                 "\"lds_per_wave\": %d, \"lds_per_block\": %d, \"steps\": %d, \"blocks\": %lld, \"long_mode\": %d, "
                 "\"band_width\": %d, \"ragged_batching\": %d, \"ragged_launches\": %d, \"ragged_cell_fraction\": %.4f, "
                 "\"score_cells\": \"%s\", \"direct_call\": %d, \"packed_classes\": %d, \"direct_out\": %d, \"band_block_rows\": %d, \"band_col_align\": %d, \"band_waves_per_cu\": %d, \"band_lds_per_wave\": %d, \"host_gather_ms\": %.3f, \"host_classify_ms\": %.3f, \"host_wait_ms\": %.3f, \"host_drain_ms\": %.3f}",
                 arch_.c_str(), device_, opt & 0xF, sc_.affine ? 1 : 0, plan_.geo->G, plan_.geo->K,
                 plan_.geo->G * plan_.geo->K, plan_.pairs_per_wave, plan_.waves_per_block, plan_.lds.total,
                 plan_.lds.total * plan_.waves_per_block, F_ + plan_.geo->G - 1, n > 0 ? (n + ppb - 1) / ppb : 0,
                 plan_.long_mode ? 1 : 0, band_width_, ragged_, host_stats_.launches,
                 host_stats_.cells_padded > 0 ? host_stats_.cells_swept / host_stats_.cells_padded : 1.0,
                 score_cell_format(opt & 0xF), host_stats_.direct, host_stats_.packed, host_stats_.direct_out,
                 ((opt & 0xF) == kAlgSW && band_chain_in_use()) ? kBandK : VALIGN_HIP_BAND_BLOCK_ROWS,
                 ((opt & 0xF) == kAlgSW && band_chain_in_use()) ? 1 : VALIGN_HIP_BAND_COL_ALIGN, band_blocks_per_cu_, band_lds_, host_stats_.gather_ms, host_stats_.classify_ms, host_stats_.wait_ms,
                 host_stats_.drain_ms);
        return buf;
    }

private:
    void validate_scoring() {
        auto fits = [](int v) { return v >= -32768 && v <= 32767; };
        if (!fits(sc_.match) || !fits(sc_.mismatch) || !fits(sc_.gap_read) || !fits(sc_.gap_ref) ||
            !fits(sc_.open_read) || !fits(sc_.ext_read) || !fits(sc_.open_ref) || !fits(sc_.ext_ref))
            throw std::runtime_error("scoring parameter outside int16");
        // The row padding and the unsigned floor-at-zero arithmetic need non-positive gap scores.
        const bool gaps_ok = sc_.affine ? (sc_.open_read <= 0 && sc_.ext_read <= 0 && sc_.open_ref <= 0 && sc_.ext_ref <= 0)
                                        : (sc_.gap_read <= 0 && sc_.gap_ref <= 0);
        if (!gaps_ok) throw std::runtime_error("positive gap scores are not supported by the HIP kernels");
        // Affine model: a maximal run of k gap bases costs open + (k - 1) * extend.  The Gotoh recurrence only
        // computes that while extending is not dearer than opening -- otherwise it re-opens instead (H of the
        // previous cell may itself end in a gap), which exhaustive enumeration exposes
        // (tests/golden/make_affine_golden.py).  Refused rather than silently computing another model.
        if (sc_.affine && (sc_.ext_read < sc_.open_read || sc_.ext_ref < sc_.open_ref))
            throw std::runtime_error("affine gap scores need extend >= open in each direction (an extension dearer "
                                     "than the opening is not an affine model)");
    }

    // latency: pick for the shortest single sweep (few pairs: every wave has a SIMD to itself and the call takes
    // as long as one wave does) instead of for the most cell updates per second
    LaunchPlan choose_plan(int R, int F, int force_g, int force_k, bool latency = false) const {
        LaunchPlan best;
        double best_cost = 0;
        for (int i = 0; i < kNumGeometries; ++i) {
            const Geometry &g = kGeometries[i];
            if (g.G * g.K < R) continue;
            if (force_g && (g.G != force_g || (force_k && g.K != force_k))) continue;
            if (!force_g && force_k && g.K != force_k) continue;
            LaunchPlan p;
            p.geo = &g;
            p.lds = g.lds(R, F);
            p.pairs_per_wave = 2 * (kWave / g.G);
            // Block size: 4-wave blocks put one wave on each SIMD and measured fastest whenever two
            // of them fit a CU's 160 KiB of LDS; otherwise take the size that keeps most waves resident.
            int best_waves = 0;
            if (p.lds.total * 8 <= kMaxBlockLds) {
                p.waves_per_block = 4;
                best_waves = std::min(32, (kMaxBlockLds / (p.lds.total * 4)) * 4);
            } else {
                for (int wpb = 4; wpb >= 1; wpb >>= 1) {
                    if (p.lds.total * wpb > kMaxBlockLds) continue;
                    const int resident = std::min(32, (kMaxBlockLds / (p.lds.total * wpb)) * wpb);
                    if (resident > best_waves) {
                        best_waves = resident;
                        p.waves_per_block = wpb;
                    }
                }
            }
            if (const char *force = getenv("VALIGN_HIP_WPB")) {                 // tuning switch
                const int wpb = atoi(force);
                if (wpb >= 1 && wpb <= 4 && p.lds.total * wpb <= kMaxBlockLds) {
                    p.waves_per_block = wpb;
                    best_waves = std::min(32, (kMaxBlockLds / (p.lds.total * wpb)) * wpb);
                }
            }
            if (best_waves == 0) continue;
            // lane-steps per pair, weighted by instructions per step: per-row work + fixed part, measured
            // on the score kernels (5.6 / 9.3 packed instructions per register, linear / affine), and a
            // penalty for long register tiles, which lose occupancy (16x12: +14 %, 16x16: +35 %,
            // 8x20: +60 % per step over the linear estimate; tools/shape_sweep.sh)
            double per_step = g.K * (sc_.affine ? 9.3 : 5.6) + 7.0;
            if (g.K > 10) per_step *= 1.0 + 0.06 * (g.K - 10);
            double cost = (double)(F + g.G - 1) * per_step * g.G / 2.0;
            if (best_waves < 8) cost *= 1.0 + 0.08 * (8 - best_waves);         // fewer than two waves per SIMD
            if (latency) cost = (double)(F + g.G - 1) * (g.K * (sc_.affine ? 9.3 : 5.6) + 7.0);
            if (!best.geo || cost < best_cost) {
                best = p;
                best_cost = cost;
            }
        }
        if (!best.geo || (getenv("VALIGN_HIP_FORCE_LONG") && !force_g)) {
            if (force_g || force_k)
                throw std::runtime_error("the forced kernel geometry does not fit read_length=" + std::to_string(R) +
                                         ", ref_length=" + std::to_string(F));
            return long_plan();
        }
        return best;
    }

    static LaunchPlan long_plan() {        // row strips + column phases: any length the ABI allows
        LaunchPlan p;
        p.long_mode = true;
        p.pairs_per_wave = 2 * (kWave / kLongG);
        p.waves_per_block = 1;
        p.lds.total = LongLds<kLongG, kLongK>::kTotal;
        for (int i = 0; i < kNumGeometries; ++i)
            if (kGeometries[i].G == kLongG && kGeometries[i].K == kLongK) p.geo = &kGeometries[i];
        return p;
    }


    // A chunk of the host-pointer pipeline is one kernel launch: sized in whole "rounds" of the waves the device runs side by
    // side (16,384 pairs at 16 x 10: 8 pairs per wave, 8 waves per CU, 256 CUs), it leaves no partly filled last round --
    // a 48 MB chunk of 150 x 500 was 4.57 rounds and paid for 5.
    long long whole_rounds(long long pairs) const {
        if (plan_.long_mode || !plan_.geo || cu_count_ <= 0) return pairs;
        const long long waves_per_cu = std::min<long long>(32, (kMaxBlockLds / std::max(1, plan_.lds.total * plan_.waves_per_block)) * plan_.waves_per_block);
        const long long round_pairs = std::max<long long>(1, waves_per_cu) * cu_count_ * plan_.pairs_per_wave;
        return pairs >= round_pairs ? pairs / round_pairs * round_pairs : pairs;
    }

    // Small calls skip the chunk pipeline (VALIGN_HIP_DIRECT_BYTES: sequence bytes up to which they do; 0 = never)
    bool direct_call(long long n, size_t per_pair) const {
        return direct_bytes_ > 0 && (size_t)n * per_pair <= direct_bytes_ && n <= staged_pairs_;
    }

    // Input copies (H2D) and result copies (D2H) of align_host must not share an SDMA engine: 0.65 GB in and 1.36 GB out
    // per million pairs of 150 x 500 would queue up behind each other (36 ms of copying beside 28 ms of kernels).
    // The runtime gives a stream the lowest-numbered engine that is FREE at the stream's first copy and keeps it there
    // (AMD_LOG_LEVEL=4: "Last copy mask 0x1" on both streams, profiles/r03_copy_engines.txt) -- and at the start of a
    // pipeline the first result copy finds the input engine idle.  So, once per engine: a small result copy is issued
    // while a long input copy keeps engine 0 busy, which lands the result stream on the next engine for good.  No
    // API promises this; where it does not work the only loss is the overlap.
    void prime_copy_engines(hipStream_t copy_in, hipStream_t copy_out, long long staged_pairs) {
        if (copy_engines_primed_ || no_engine_priming_) return;
        copy_engines_primed_ = true;
        const size_t in_bytes = std::min<size_t>((size_t)staged_pairs * F_, 128u << 20);
        const size_t out_bytes = std::min<size_t>(sizeof(short) * 4 * (size_t)staged_pairs, 4096);
        if (in_bytes < (16u << 20) || out_bytes == 0) return;         // (too short to still be running when the second copy is issued)
        hip_check(hipStreamSynchronize(copy_in), "hipStreamSynchronize");
        hip_check(hipStreamSynchronize(copy_out), "hipStreamSynchronize");
        hip_check(hipMemcpyAsync(d_refs_[0], h_refs_[0], in_bytes, hipMemcpyHostToDevice, copy_in), "H2D (engine priming)");
        // the input copy must have reached its engine before the result stream asks which engines are free: >= 16 MB
        // take >= 0.3 ms on the wire, a tenth of that is plenty for the submission
        for (const auto t0 = std::chrono::steady_clock::now(); ms_between(t0, std::chrono::steady_clock::now()) < 0.1;) {
        }
        hip_check(hipMemcpyAsync(h_idx_[0], d_idx_[0], out_bytes, hipMemcpyDeviceToHost, copy_out), "D2H (engine priming)");
        hip_check(hipStreamSynchronize(copy_out), "hipStreamSynchronize");
        hip_check(hipStreamSynchronize(copy_in), "hipStreamSynchronize");
    }

    // the caller's result buffers, when they can take the device's copies directly (page-locked, contiguous)
    void flat_destination(FlatSink sink, long long n, uint8_t *&rows, short *&idx) const {
        if (HostRegistry::instance().covers(sink.rows, (size_t)n * 2 * sink.AL) &&
            HostRegistry::instance().covers(sink.idx, sizeof(short) * 4 * (size_t)n)) {
            rows = sink.rows;
            idx = sink.idx;
        }
    }
    template <typename AlignmentT>
    void flat_destination(AlignmentT *, long long, uint8_t *&, short *&) const {}      // 2n heap rows: always scattered

    // device-side address of pinned host memory of this engine (hipHostMalloc: mapped, same address on ROCm)
    static uint8_t *dev_view(void *pinned) {
        void *d = nullptr;
        hip_check(hipHostGetDevicePointer(&d, pinned, 0), "hipHostGetDevicePointer");
        return (uint8_t *)d;
    }

    // A call that threw in the middle of the pipeline (a HIP error, `too many length groups`) leaves chunks
    // pending in the slots; draining them into the NEXT caller's arrays would write at the old offsets.  Every
    // host-pointer call starts from idle streams and empty slots.
    void reset_pipeline() {
        bool stale = false;
        for (int s = 0; s < kSlots; ++s) stale = stale || slot_pending_[s] != 0;
        if (!stale) return;
        if (copy_issuer_) {
            try {
                copy_issuer_->wait_idle();
            } catch (...) {
            }
        }
        if (trace_stream_) (void)hipStreamSynchronize(trace_stream_);
        for (int s = 0; s < kSlots; ++s) {
            (void)hipStreamSynchronize(streams_[s]);
            slot_pending_[s] = 0;
            slot_begin_[s] = 0;
        }
    }

    void release_trace_scratch() {
        if (d_ptr_) (void)hipFree(d_ptr_);
        if (d_ends_) (void)hipFree(d_ends_);
        d_ptr_ = nullptr;
        d_ends_ = nullptr;
        trace_pairs_ = 0;
        trace_bytes_ = 0;
        if (d_first_bad_) (void)hipFree(d_first_bad_);
        d_first_bad_ = nullptr;
        first_bad_bytes_ = 0;
        for (int s = 0; s < kSlots; ++s) {
            if (h_rows_[s]) (void)hipHostFree(h_rows_[s]);
            if (h_idx_[s]) (void)hipHostFree(h_idx_[s]);
            if (d_rows_[s]) (void)hipFree(d_rows_[s]);
            if (d_idx_[s]) (void)hipFree(d_idx_[s]);
            h_rows_[s] = nullptr;
            h_idx_[s] = nullptr;
            d_rows_[s] = nullptr;
            d_idx_[s] = nullptr;
        }
        align_staged_pairs_ = 0;
    }

    // The pointer stream's bytes per pair-of-pairs depend on the fill kernel the call selects (tagged /
    // untagged, 4- or 8-step blocks, one or two code words): capacity is tracked in BYTES, so a call with a
    // wider stream than the one that sized the scratch reallocates instead of writing past it.
    void ensure_trace_scratch(long long pairs, size_t bytes_per_pp, hipStream_t stream) {
        const long long ppw = plan_.pairs_per_wave;
        const long long waves = (pairs + ppw - 1) / ppw;
        const size_t need = (size_t)(waves * (ppw / 2)) * bytes_per_pp;
        if (need <= trace_bytes_ && pairs <= trace_pairs_) return;
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");   // nothing may still read the old scratch
        if (trace_stream_) hip_check(hipStreamSynchronize(trace_stream_), "hipStreamSynchronize");
        chain_regions_busy_[0] = chain_regions_busy_[1] = false;
        if (need > trace_bytes_) {
            if (d_ptr_) (void)hipFree(d_ptr_);
            d_ptr_ = nullptr;
            trace_bytes_ = 0;
            hip_check(hipMalloc((void **)&d_ptr_, need), "hipMalloc(pointer scratch)");
            trace_bytes_ = need;
        }
        if (pairs > trace_pairs_) {
            if (d_ends_) (void)hipFree(d_ends_);
            d_ends_ = nullptr;
            trace_pairs_ = 0;
            hip_check(hipMalloc((void **)&d_ends_, sizeof(EndCell) * (size_t)(waves * ppw)), "hipMalloc(end cells)");
            trace_pairs_ = pairs;
        }
    }

    void ensure_align_staging(long long pairs) {
        if (pairs <= align_staged_pairs_) return;
        const size_t AL = (size_t)R_ + F_;
        for (int s = 0; s < kSlots; ++s) {
            if (h_rows_[s]) (void)hipHostFree(h_rows_[s]);
            if (h_idx_[s]) (void)hipHostFree(h_idx_[s]);
            if (d_rows_[s]) (void)hipFree(d_rows_[s]);
            if (d_idx_[s]) (void)hipFree(d_idx_[s]);
            // (room for the coordinates behind the rows: small calls bring both back in one piece)
            const size_t rows_cap = (size_t)pairs * 2 * AL + sizeof(short) * 4 * (size_t)pairs + 32;
            hip_check(hipHostMalloc((void **)&h_rows_[s], rows_cap, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_idx_[s], sizeof(short) * 4 * (size_t)pairs, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc((void **)&d_rows_[s], rows_cap), "hipMalloc");
            hip_check(hipMalloc((void **)&d_idx_[s], sizeof(short) * 4 * (size_t)pairs), "hipMalloc");
        }
        align_staged_pairs_ = pairs;
    }

    // gather / scatter between the caller's scattered blocks and the staging: host_pipeline.h (host-only, sanitizer-tested)
    template <typename Sink>
    void scatter(Sink sink, long long cnt, const uint8_t *rows, const short *idx, int threads) {
        packer_.scatter(sink, cnt, rows, idx, threads);
    }

    void release_staging() {
        for (int s = 0; s < kSlots; ++s) {
            if (h_reads_[s]) (void)hipHostFree(h_reads_[s]);
            if (h_refs_[s]) (void)hipHostFree(h_refs_[s]);
            if (h_scores_[s]) (void)hipHostFree(h_scores_[s]);
            if (d_reads_[s]) (void)hipFree(d_reads_[s]);
            if (d_refs_[s]) (void)hipFree(d_refs_[s]);
            if (d_scores_[s]) (void)hipFree(d_scores_[s]);
            if (d_pack_reads_[s]) (void)hipFree(d_pack_reads_[s]);
            if (d_pack_refs_[s]) (void)hipFree(d_pack_refs_[s]);
            d_pack_reads_[s] = d_pack_refs_[s] = nullptr;
            h_reads_[s] = h_refs_[s] = nullptr;
            h_scores_[s] = nullptr;
            d_reads_[s] = d_refs_[s] = nullptr;
            d_scores_[s] = nullptr;
        }
        staged_pairs_ = 0;
    }

    void ensure_staging(long long pairs) {
        if (pairs <= staged_pairs_) return;
        release_staging();
        for (int s = 0; s < kSlots; ++s) {
            // (write-combined pinned memory for the input staging was tried: no gain -- the call is bound by the H2D
            // copies, 12-14 ms per 650 MB while the host threads gather, and by what else runs on the box)
            hip_check(hipHostMalloc((void **)&h_reads_[s], std::max<size_t>((size_t)pairs * R_, 16), hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_refs_[s], std::max<size_t>((size_t)pairs * F_, 16), hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_scores_[s], sizeof(short) * (size_t)pairs, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc((void **)&d_reads_[s], std::max<size_t>((size_t)pairs * R_, 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_refs_[s], std::max<size_t>((size_t)pairs * F_, 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_scores_[s], sizeof(short) * (size_t)pairs), "hipMalloc");
            // (the 4-bit class copies of the score path; the pinned staging above is large enough for them)
            hip_check(hipMalloc((void **)&d_pack_reads_[s], std::max<size_t>((size_t)pairs * packed_length(R_), 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_pack_refs_[s], std::max<size_t>((size_t)pairs * packed_length(F_), 16)), "hipMalloc");
        }
        staged_pairs_ = pairs;
    }

    // ---- length-sorted batching (score_host, Smith-Waterman) ----

    static double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    }

    // Both modes (round 3): rows and columns of trailing non-ACGT bytes score 0 against everything, so every value of the
    // real matrix's last row / last column runs down its diagonal unchanged to the padded matrix's last row / column (gap
    // moves only lose) -- the NW variant's max(0, last column, last row) of the trimmed pair IS that of the padded pair,
    // as the Smith-Waterman maximum is.  Checked on the oracle (tests/test_oracle_golden.py) and on the GPU.
    bool ragged_applies(int alg) const {
        return ragged_ && alg <= kAlgNW && !plan_.long_mode && !force_g_ && !force_k_ && score_width_ != 32 &&
               R_ > 0 && F_ > 0 && int16_range_ok(alg);
    }

    // Share of the padded cells a length-sorted sweep would still visit, from 256 pairs spread over
    // the call: sorting costs the host a pass over every tail, so it has to buy something.
    double sampled_cell_fraction(const char *const *reads, const char *const *refs, long long n) const {
        const long long samples = std::min<long long>(n, 256);
        double swept = 0;
        for (long long k = 0; k < samples; ++k) {
            const long long i = k * n / samples;
            const int r = read_caps_[read_class_[trimmed_length((const unsigned char *)reads[i], R_)]];
            const int f = ref_caps_[ref_class_[trimmed_length((const unsigned char *)refs[i], F_)]];
            swept += (double)r * f;
        }
        return swept / ((double)samples * R_ * F_);
    }

    // Read classes: the row capacities of the compiled geometries below read_length, then
    // read_length itself.  Reference classes: multiples of 64 columns, then ref_length.
    void build_length_classes() {
        std::vector<int> caps;
        for (int i = 0; i < kNumGeometries; ++i) {
            const int rows = kGeometries[i].G * kGeometries[i].K;
            if (rows < R_) caps.push_back(rows);
        }
        caps.push_back(R_);
        std::sort(caps.begin(), caps.end());
        caps.erase(std::unique(caps.begin(), caps.end()), caps.end());
        read_caps_ = caps;
        ref_caps_.clear();
        const int width = 64 * std::max(1, (F_ + 64 * kMaxScoreGroups - 1) / (64 * kMaxScoreGroups));
        for (int c = width; c < F_; c += width) ref_caps_.push_back(c);
        ref_caps_.push_back(F_);
        read_class_.assign((size_t)R_ + 1, 0);
        for (int len = 0, c = 0; len <= R_; ++len) {
            while (read_caps_[c] < len) ++c;
            read_class_[len] = (unsigned char)c;
        }
        ref_class_.assign((size_t)F_ + 1, 0);
        for (int len = 0, c = 0; len <= F_; ++len) {
            while (ref_caps_[c] < len) ++c;
            ref_class_[len] = (unsigned short)c;
        }
        if (const char *m = getenv("VALIGN_HIP_RAGGED_MIN")) ragged_min_ = std::max(1, atoi(m));   // tuning switch
        if (const char *m = getenv("VALIGN_HIP_CHUNK_BYTES")) score_chunk_bytes_ = (size_t)std::max(4096ll, atoll(m));
        if (const char *m = getenv("VALIGN_HIP_ALIGN_CHUNK_BYTES")) align_chunk_bytes_ = (size_t)std::max(4096ll, atoll(m));
        if (const char *m = getenv("VALIGN_HIP_DIRECT_BYTES")) direct_bytes_ = (size_t)std::max(0ll, atoll(m));
    }

    const LaunchPlan &class_plan(int R, int F) {
        const std::pair<int, int> key(R, F);
        auto it = class_plans_.find(key);
        if (it == class_plans_.end()) it = class_plans_.emplace(key, choose_plan(R, F, 0, 0)).first;
        return it->second;
    }

    static int trimmed_length(const unsigned char *s, int len) {
        static const struct Table {
            bool acgt[256] = {};
            Table() { for (const char *p = "ACGTacgt"; *p; ++p) acgt[(unsigned char)*p] = true; }
        } table;
        while (len >= 8) {                      // NUL padding, eight bytes at a time
            uint64_t tail;
            memcpy(&tail, s + len - 8, 8);
            if (tail != 0) break;
            len -= 8;
        }
        while (len > 0 && !table.acgt[s[len - 1]]) --len;
        return len;
    }

    template <typename Fn>
    void for_ranges(int threads, long long cnt, long long serial_below, Fn fn) {
        packer_.for_ranges(threads, cnt, serial_below, fn);
    }

    // ---- length-sorted batching on the device (ragged_kernels.hip.h) ----
    // One context per pipeline slot (chunks of different slots are in flight side by side) and one for device-resident
    // batches (score_device, on the caller's stream).
    struct RaggedCtx {
        long long cap = 0;                     // pairs the buffers hold
        uint8_t *reads = nullptr, *refs = nullptr;     // the packed groups
        int16_t *scores = nullptr;             // ... and their scores, packed order
        uint16_t *bin = nullptr;               // length bin of every pair
        int *pos = nullptr;                    // packed place of every pair
        RaggedPlace *place = nullptr;          // ... as byte offsets + strides, for the copy kernel
        unsigned *counters = nullptr;          // bins' pair counts, then the groups' fill cursors
        uint8_t *tables = nullptr;             // device: group_of_bin[bins] then RaggedGroupDev[groups]
        unsigned *h_counts = nullptr;          // pinned: the histogram's way to the host
        uint8_t *h_tables = nullptr;           // pinned: the tables' way to the device
        hipEvent_t counted = nullptr;
        const uint8_t *src_reads = nullptr, *src_refs = nullptr;      // of the chunk between begin and finish
    };
    static constexpr size_t kRaggedTableBytes = sizeof(uint16_t) * kRaggedMaxBins + sizeof(RaggedGroupDev) * kRaggedMaxGroups;

    int ragged_bins() const { return (int)(read_caps_.size() * ref_caps_.size()); }
    bool ragged_fits(long long n) const {
        return ragged_bins() <= kRaggedMaxBins && read_caps_.size() * (size_t)kMaxScoreGroups <= (size_t)kRaggedMaxGroups &&
               n < 0x7FFFFFFFll;
    }

    void ensure_ragged(int c, long long n) {
        RaggedCtx &x = rag_[c];
        if (!d_read_class_) {
            hip_check(hipMalloc((void **)&d_read_class_, read_class_.size()), "hipMalloc(read classes)");
            hip_check(hipMalloc((void **)&d_ref_class_, sizeof(uint16_t) * ref_class_.size()), "hipMalloc(ref classes)");
            hip_check(hipMemcpy(d_read_class_, read_class_.data(), read_class_.size(), hipMemcpyHostToDevice), "hipMemcpy");
            hip_check(hipMemcpy(d_ref_class_, ref_class_.data(), sizeof(uint16_t) * ref_class_.size(), hipMemcpyHostToDevice), "hipMemcpy");
        }
        if (!x.counted) {
            hip_check(hipEventCreateWithFlags(&x.counted, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipMalloc((void **)&x.counters, sizeof(unsigned) * (kRaggedMaxBins + kRaggedMaxGroups)), "hipMalloc(ragged counters)");
            hip_check(hipMalloc((void **)&x.tables, kRaggedTableBytes), "hipMalloc(ragged tables)");
            hip_check(hipHostMalloc((void **)&x.h_counts, sizeof(unsigned) * kRaggedMaxBins, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&x.h_tables, kRaggedTableBytes, hipHostMallocDefault), "hipHostMalloc");
        }
        if (x.cap >= n) return;
        for (void *p : {(void *)x.reads, (void *)x.refs, (void *)x.scores, (void *)x.bin, (void *)x.pos, (void *)x.place})
            if (p) (void)hipFree(p);                         // (hipFree waits for the device: nothing is still reading them)
        x.reads = x.refs = nullptr;
        x.scores = nullptr;
        x.bin = nullptr;
        x.pos = nullptr;
        x.place = nullptr;
        x.cap = 0;
        hip_check(hipMalloc((void **)&x.reads, std::max<size_t>((size_t)n * R_, 16)), "hipMalloc(ragged reads)");
        hip_check(hipMalloc((void **)&x.refs, std::max<size_t>((size_t)n * F_, 16)), "hipMalloc(ragged refs)");
        hip_check(hipMalloc((void **)&x.scores, sizeof(int16_t) * (size_t)n), "hipMalloc(ragged scores)");
        hip_check(hipMalloc((void **)&x.bin, sizeof(uint16_t) * (size_t)n), "hipMalloc(ragged bins)");
        hip_check(hipMalloc((void **)&x.pos, sizeof(int) * (size_t)n), "hipMalloc(ragged places)");
        hip_check(hipMalloc((void **)&x.place, sizeof(RaggedPlace) * (size_t)n), "hipMalloc(ragged place records)");
        x.cap = n;
    }
    void release_ragged() {
        for (RaggedCtx &x : rag_) {
            for (void *p : {(void *)x.reads, (void *)x.refs, (void *)x.scores, (void *)x.bin, (void *)x.pos, (void *)x.place, (void *)x.counters, (void *)x.tables})
                if (p) (void)hipFree(p);
            if (x.h_counts) (void)hipHostFree(x.h_counts);
            if (x.h_tables) (void)hipHostFree(x.h_tables);
            if (x.counted) (void)hipEventDestroy(x.counted);
            x = RaggedCtx{};
        }
        if (d_read_class_) (void)hipFree(d_read_class_);
        if (d_ref_class_) (void)hipFree(d_ref_class_);
        d_read_class_ = nullptr;
        d_ref_class_ = nullptr;
    }

    // first half: trimmed lengths -> bins, the histogram on its way to the host.  Asynchronous on `stream`.
    void ragged_begin(int c, long long n, const uint8_t *d_reads, const uint8_t *d_refs, hipStream_t stream) {
        ensure_ragged(c, n);
        RaggedCtx &x = rag_[c];
        x.src_reads = d_reads;
        x.src_refs = d_refs;
        const int NG = ragged_bins();
        hip_check(hipMemsetAsync(x.counters, 0, sizeof(unsigned) * (kRaggedMaxBins + kRaggedMaxGroups), stream), "hipMemsetAsync");
        RaggedClassifyArgs a{d_reads, d_refs, n, R_, F_, d_read_class_, d_ref_class_, (int)ref_caps_.size(), NG, x.bin, x.counters};
        void *kargs[] = {&a};
        const long long blocks = (n + kRaggedClassifyPairs - 1) / kRaggedClassifyPairs;
        hip_check(hipLaunchKernel((const void *)&ragged_classify_kernel, dim3((unsigned)blocks), dim3(256), kargs, 0, stream),
                  "hipLaunchKernel(ragged_classify_kernel)");
        hip_check(hipMemcpyAsync(x.h_counts, x.counters, sizeof(unsigned) * (size_t)NG, hipMemcpyDeviceToHost, stream), "D2H histogram");
        hip_check(hipEventRecord(x.counted, stream), "hipEventRecord");
    }

    // Fold bins too small to be worth a launch into the next larger one and lay the groups out: a read class with too few
    // pairs for a launch of its own joins the next read class (bin by bin); inside a class, a reference bin smaller than a
    // few blocks joins the next wider one.  Both dimensions only ever grow, so the padded sweep still covers the pair.
    std::vector<LengthGroup> fold_groups(std::vector<long long> &total, std::vector<int> &group_of_bin) const {
        const int NR = (int)read_caps_.size(), NF = (int)ref_caps_.size(), NG = NR * NF;
        std::vector<int> target((size_t)NG);
        for (int g = 0; g < NG; ++g) target[g] = g;
        const long long bin_min = std::min<long long>(ragged_min_, 256);
        for (int rc = 0; rc < NR; ++rc) {
            long long in_class = 0;
            for (int fc = 0; fc < NF; ++fc) in_class += total[rc * NF + fc];
            if (in_class == 0) continue;
            if (in_class < ragged_min_ && rc < NR - 1) {
                for (int fc = 0; fc < NF; ++fc) {
                    const int g = rc * NF + fc;
                    total[g + NF] += total[g];
                    total[g] = 0;
                    target[g] = g + NF;
                }
                continue;
            }
            for (int fc = 0; fc < NF - 1; ++fc) {
                const int g = rc * NF + fc;
                if (total[g] == 0 || total[g] >= bin_min) continue;
                total[g + 1] += total[g];
                total[g] = 0;
                target[g] = g + 1;
            }
        }
        for (int g = NG - 1; g >= 0; --g) target[g] = target[target[g]];      // targets only point forward
        std::vector<LengthGroup> groups;
        std::vector<int> group_at((size_t)NG, -1);
        long long pair_ofs = 0;
        size_t read_ofs = 0, ref_ofs = 0;
        for (int g = 0; g < NG; ++g) {
            if (total[g] == 0) continue;
            LengthGroup lg;
            lg.R = read_caps_[g / NF];
            lg.F = ref_caps_[g % NF];
            lg.pairs = total[g];
            lg.pair_ofs = pair_ofs;
            lg.read_ofs = read_ofs;
            lg.ref_ofs = ref_ofs;
            pair_ofs += lg.pairs;
            read_ofs += (size_t)lg.pairs * lg.R;
            ref_ofs += (size_t)lg.pairs * lg.F;
            group_at[g] = (int)groups.size();
            groups.push_back(lg);
        }
        group_of_bin.assign((size_t)NG, 0);
        for (int g = 0; g < NG; ++g) group_of_bin[g] = std::max(group_at[target[g]], 0);     // (an empty bin: any group, no pair asks)
        return groups;
    }

    // second half: waits for the histogram, lays the groups out, then -- asynchronously on `stream` -- packs the pairs by
    // group, sweeps class by class and puts the scores back in the caller's order.  `always` false: false is returned, and
    // nothing launched, where the classes would still visit two thirds of the padded cells or more.
    bool ragged_finish(int c, int alg, long long n, int16_t *d_scores, hipStream_t stream, bool always) {
        RaggedCtx &x = rag_[c];
        const int NG = ragged_bins();
        const auto t0 = std::chrono::steady_clock::now();
        hip_check(hipEventSynchronize(x.counted), "hipEventSynchronize");
        const auto t_counted = std::chrono::steady_clock::now();
        host_stats_.classify_ms += ms_between(t0, t_counted);
        std::vector<long long> total((size_t)NG);
        long long seen = 0;
        for (int g = 0; g < NG; ++g) seen += (total[g] = (long long)x.h_counts[g]);
        if (seen != n) throw std::runtime_error("length classification lost pairs");
        std::vector<int> group_of_bin;
        const std::vector<LengthGroup> groups = fold_groups(total, group_of_bin);
        double swept = 0;
        for (const LengthGroup &g : groups) swept += (double)g.pairs * g.R * g.F;
        const double padded = (double)n * R_ * F_;
        if (!always && swept >= 0.67 * padded) return false;
        const int NL = (int)groups.size();
        if (NL > kRaggedMaxGroups) throw std::runtime_error("too many length groups");
        uint16_t *h_map = reinterpret_cast<uint16_t *>(x.h_tables);
        RaggedGroupDev *h_groups = reinterpret_cast<RaggedGroupDev *>(x.h_tables + sizeof(uint16_t) * kRaggedMaxBins);
        for (int g = 0; g < NG; ++g) h_map[g] = (uint16_t)group_of_bin[g];
        for (int l = 0; l < NL; ++l)
            h_groups[l] = RaggedGroupDev{groups[l].R, groups[l].F, groups[l].pair_ofs, (long long)groups[l].read_ofs, (long long)groups[l].ref_ofs};
        hip_check(hipMemcpyAsync(x.tables, x.h_tables, kRaggedTableBytes, hipMemcpyHostToDevice, stream), "H2D length groups");
        RaggedPermuteArgs pa{x.src_reads, x.src_refs, n, R_, F_, x.bin, reinterpret_cast<const uint16_t *>(x.tables),
                             reinterpret_cast<const RaggedGroupDev *>(x.tables + sizeof(uint16_t) * kRaggedMaxBins), NL,
                             x.counters + kRaggedMaxBins, x.reads, x.refs, x.pos, x.place};
        void *pargs[] = {&pa};
        hip_check(hipLaunchKernel((const void *)&ragged_place_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), pargs, 0, stream),
                  "hipLaunchKernel(ragged_place_kernel)");
        hip_check(hipLaunchKernel((const void *)&ragged_copy_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), pargs, 0, stream),
                  "hipLaunchKernel(ragged_copy_kernel)");
        // one launch per read class: its reference-length groups ride in the kernel's group table
        for (size_t first = 0; first < groups.size();) {
            size_t end = first;
            int widest = 0;
            while (end < groups.size() && groups[end].R == groups[first].R) widest = std::max(widest, groups[end++].F);
            launch_score(class_plan(groups[first].R, widest), alg, groups[first].R, widest, 0, x.reads, x.refs, x.scores, stream,
                         groups.data() + first, (int)(end - first));
            host_stats_.launches += 1;
            first = end;
        }
        RaggedUnpermuteArgs ua{x.scores, x.pos, d_scores, n};
        void *uargs[] = {&ua};
        hip_check(hipLaunchKernel((const void *)&ragged_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), uargs, 0, stream),
                  "hipLaunchKernel(ragged_unpermute_kernel)");
        host_stats_.cells_swept += swept;
        host_stats_.cells_padded += padded;
        host_stats_.launch_ms += ms_between(t_counted, std::chrono::steady_clock::now());
        return true;
    }

    void gather(const char *const *reads, const char *const *refs, long long cnt, uint8_t *dst_reads,
                uint8_t *dst_refs, int threads) {
        packer_.gather(reads, refs, cnt, dst_reads, dst_refs, threads);
    }

    int device_, R_, F_;
    Scoring sc_;
    bool sse_policy_ = false;
    int band_width_ = 0;
    int score_width_ = 0;
    int ragged_ = 0, force_g_ = 0, force_k_ = 0;
    size_t score_chunk_bytes_ = 48u << 20;                   // staging chunk of score_host (VALIGN_HIP_CHUNK_BYTES)
    size_t align_chunk_bytes_ = 128u << 20;                  // staging chunk of align_host, inputs + results (VALIGN_HIP_ALIGN_CHUNK_BYTES)
    size_t direct_bytes_ = 768u << 10;                       // calls with at most this many sequence bytes run on the pinned staging directly
    long long ragged_min_ = 2048;                             // pairs a length bin needs for its own launch
    std::vector<int> read_caps_, ref_caps_;
    std::vector<unsigned char> read_class_;
    std::vector<unsigned short> ref_class_;
    std::map<std::pair<int, int>, LaunchPlan> class_plans_;
    RaggedCtx rag_[kSlots + 1];                                  // one per pipeline slot, the last for device-resident batches
    uint8_t *d_read_class_ = nullptr;
    uint16_t *d_ref_class_ = nullptr;
    HostPacker packer_{R_, F_};                               // (declared after R_ / F_)
    HostStats host_stats_;
    bool no_sym_ = getenv("VALIGN_HIP_NO_SYM") != nullptr;   // tuning switch: use the two-gap kernel always
    bool no_tag_ = getenv("VALIGN_HIP_NO_TAG") != nullptr;   // tuning switch: equality-test pointer kernels for linear alignments
    bool no_f16_ = getenv("VALIGN_HIP_NO_F16") != nullptr;   // tuning switch: int16 cells for symmetric affine SW too
    bool no_fused_ = getenv("VALIGN_HIP_NO_FUSED") != nullptr;   // tuning switch: small alignment calls as fill + traceback kernels
    bool no_prof_key_ = getenv("VALIGN_HIP_NO_PROF_KEY") != nullptr;   // tuning switch: compute the SW lane key instead of carrying it in the profile
    bool copy_engines_primed_ = false;
    bool no_engine_priming_ = getenv("VALIGN_HIP_NO_ENGINE_PRIMING") != nullptr;   // tuning switch
    bool d2h_on_stream_ = getenv("VALIGN_HIP_D2H_ON_STREAM") != nullptr;   // tuning switch: result copies behind a stream wait (a blit kernel on this stack)
    std::unique_ptr<CopyIssuer> copy_issuer_;
    bool ramp_ = getenv("VALIGN_HIP_NO_RAMP") == nullptr;                 // tuning switch: every chunk of a host-pointer call full-sized
    int split_parts_ = getenv("VALIGN_HIP_SPLIT_PARTS") ? atoi(getenv("VALIGN_HIP_SPLIT_PARTS")) : 2;   // tuning switch: > 2: geometric parts of align_device
    bool wide_align_ = getenv("VALIGN_HIP_WIDE_ALIGN") != nullptr;         // test switch: NW alignments (linear gaps, default tie-breaks) on int32 cells always
    bool no_direct_out_ = getenv("VALIGN_HIP_NO_DIRECT_OUT") != nullptr;   // tuning switch: stage + scatter even into registered result buffers
    bool no_overlap_ = getenv("VALIGN_HIP_NO_OVERLAP") != nullptr;   // tuning switch: tracebacks in stream order behind their fills
    long long scratch_cap_mb_ = getenv("VALIGN_HIP_SCRATCH_CAP_MB") ? atoll(getenv("VALIGN_HIP_SCRATCH_CAP_MB")) : 0;   // test switch: small pointer scratch
    hipStream_t trace_stream_ = nullptr;                          // helper stream of align_device (walks beside the next fill)
    bool chain_regions_busy_[2] = {false, false};                 // WalkChain: the region's last walk may still be running
    hipEvent_t fill_done_[2] = {nullptr, nullptr}, trace_done_[2] = {nullptr, nullptr}, entry_ev_ = nullptr;
    std::string arch_;
    LaunchPlan plan_, latency_plan_;
    hipStream_t streams_[kSlots] = {};
    hipEvent_t slot_done_[kSlots] = {};
    long long slot_begin_[kSlots] = {}, slot_pending_[kSlots] = {};
    long long staged_pairs_ = 0;
    uint8_t *h_reads_[kSlots] = {}, *h_refs_[kSlots] = {};
    short *h_scores_[kSlots] = {};
    uint8_t *d_reads_[kSlots] = {}, *d_refs_[kSlots] = {};
    uint8_t *d_pack_reads_[kSlots] = {}, *d_pack_refs_[kSlots] = {};     // 4-bit classes as they arrive (score path)
    bool pack_ = getenv("VALIGN_HIP_NO_PACK") == nullptr;                // host_packing (tuning switch: ASCII across PCIe)
    int16_t *d_scores_[kSlots] = {};
    // compute_alignments: pointer scratch + end cells (device), result staging (both sides)
    BandPlan band_plan_;               // banded linear SW: the block chain's plan for band_plan_width_, its tables on the device
    int band_plan_width_ = -1;
    int band_blocks_per_cu_ = 0, band_lds_ = 0;          // of the last block-chain launch (describe)
    BandBlock *d_band_blocks_ = nullptr;
    int *d_band_fill_ = nullptr;
    int cu_count_ = 0;
    bool no_band_persist_ = getenv("VALIGN_HIP_BAND_BLOCK_PER_QUAD") != nullptr;   // tuning switch: one block per four pairs
    bool no_band_chain_ = getenv("VALIGN_HIP_NO_BAND_CHAIN") != nullptr;      // tuning switch: banded scores on score_long_kernel's strips
    unsigned *d_brow_ = nullptr;       // long-read path: strip boundary rows
    size_t brow_bytes_ = 0;
    unsigned *d_ptr_ = nullptr;
    EndCell *d_ends_ = nullptr;
    long long trace_pairs_ = 0, align_staged_pairs_ = 0;
    size_t trace_bytes_ = 0;            // capacity of d_ptr_
    int *d_first_bad_ = nullptr;        // row strips: first invalid read / ref position per pair
    size_t first_bad_bytes_ = 0;
    uint8_t *h_rows_[kSlots] = {}, *d_rows_[kSlots] = {};
    short *h_idx_[kSlots] = {}, *d_idx_[kSlots] = {};
    hipEvent_t in_done_[kSlots] = {}, kernels_done_[kSlots] = {};   // align_host: H2D / kernels of the slot's chunk finished
};

}  // namespace valign
