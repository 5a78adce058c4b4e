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
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dp_kernels.hip.h"
#include "trace_kernels.hip.h"
#include "long_kernels.hip.h"

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

// One compiled (G, K) geometry with its four kernel variants.
struct Geometry {
    int G, K;
    WaveLds (*lds)(int R, int F);
    const void *kernel[2][4];      // score kernels [alg][linear, symmetric linear, affine, symmetric affine]
    const void *fill[2][4];        // alignment fill kernels [alg][linear, symmetric linear, affine, SSE policy]
};

template <int G, int K>
constexpr Geometry make_geometry() {
    return Geometry{G, K, &wave_lds<G, K>,
                    {{(const void *)&score_kernel<G, K, kAlgSW, kGapLinear>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapSym>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffine>,
                      (const void *)&score_kernel<G, K, kAlgSW, kGapAffineSym>},
                     {(const void *)&score_kernel<G, K, kAlgNW, kGapLinear>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapSym>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffine>,
                      (const void *)&score_kernel<G, K, kAlgNW, kGapAffineSym>}},
                    {{(const void *)&align_fill_kernel<G, K, kAlgSW, false>, (const void *)&align_fill_kernel<G, K, kAlgSW, true>,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgSW>, (const void *)&align_fill_sse_kernel<G, K, kAlgSW>},
                     {(const void *)&align_fill_kernel<G, K, kAlgNW, false>, (const void *)&align_fill_kernel<G, K, kAlgNW, true>,
                      (const void *)&align_fill_affine_kernel<G, K, kAlgNW>, (const void *)&align_fill_sse_kernel<G, K, kAlgNW>}}};
}

// Rows covered = G*K.  Ordered by capacity; selection is by estimated cost.
static const Geometry kGeometries[] = {
    make_geometry<8, 4>(),   make_geometry<8, 8>(),   make_geometry<16, 4>(),  make_geometry<8, 12>(),
    make_geometry<8, 16>(),  make_geometry<16, 8>(),  make_geometry<8, 20>(),  make_geometry<16, 10>(),
    make_geometry<16, 12>(), make_geometry<16, 16>(), make_geometry<32, 8>(),  make_geometry<32, 12>(),
    make_geometry<32, 16>(), make_geometry<64, 12>(), make_geometry<64, 16>(), make_geometry<64, 24>(),
    make_geometry<64, 32>(),
};
constexpr int kNumGeometries = sizeof(kGeometries) / sizeof(kGeometries[0]);

constexpr int kMaxBlockLds = 160 * 1024;       // gfx950: 160 KiB per CU, one block may take it all
constexpr int kDefaultBlockLds = 64 * 1024;    // above this the kernel attribute must be raised

// Long-read path (row strips + column phases, long_kernels.hip.h): one geometry, linear gaps.
constexpr int kLongG = 16, kLongK = 10;
static const void *const kLongKernels[2][2][2] = {     // [alg][gap_read == gap_ref][int32 cells]
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, true>}},
    {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, true>},
     {(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, false>, (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, true>}}};

struct LaunchPlan {
    bool long_mode = false;        // sequences too long for one register sweep / LDS-resident reference
    const Geometry *geo = nullptr;
    WaveLds lds{};
    int waves_per_block = 4;
    int pairs_per_wave = 0;
};

class Engine {
public:
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
        if (arch_.find("gfx950") == std::string::npos)
            throw std::runtime_error("device is " + arch_ + "; this library carries gfx950 code only");
        plan_ = choose_plan(force_g, force_k);
        for (int s = 0; s < 2; ++s) hip_check(hipStreamCreateWithFlags(&streams_[s], hipStreamNonBlocking), "hipStreamCreate");
        for (int s = 0; s < 2; ++s) hip_check(hipEventCreateWithFlags(&slot_done_[s], hipEventDisableTiming), "hipEventCreate");
    }

    ~Engine() {
        (void)hipSetDevice(device_);
        release_staging();
        release_trace_scratch();
        if (d_brow_) (void)hipFree(d_brow_);
        for (int s = 0; s < 2; ++s) {
            if (slot_done_[s]) (void)hipEventDestroy(slot_done_[s]);
            if (streams_[s]) (void)hipStreamDestroy(streams_[s]);
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
    int device() const { return device_; }
    const LaunchPlan &plan() const { return plan_; }
    hipStream_t own_stream() const { return streams_[0]; }

    // Device-resident batch, asynchronous on `stream`.
    void score_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs,
                      int16_t *d_scores, hipStream_t stream) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;          // reference: unsupported mode is a silent no-op
        hip_check(hipSetDevice(device_), "hipSetDevice");
        if (score_width_ == 16) check_int16_range(alg);
        const bool wide = score_width_ == 32 || (score_width_ == 0 && !int16_range_ok(alg));
        if (plan_.long_mode || wide) {      // int32 cells exist on the strip path only
            score_long_device(alg, n, d_reads, d_refs, d_scores, stream, wide);
            return;
        }
        ScoreArgs a;
        a.reads = d_reads;
        a.refs = d_refs;
        a.scores = d_scores;
        a.n = n;
        a.R = R_;
        a.F = F_;
        a.prof_area = plan_.lds.prof_area;
        a.refc_stride = plan_.lds.refc_stride;
        a.wave_lds = plan_.lds.total;
        a.match = (short)sc_.match;
        a.mismatch = (short)sc_.mismatch;
        a.gap_read = (short)sc_.gap_read;
        a.gap_ref = (short)sc_.gap_ref;
        a.open_read = (short)sc_.open_read;
        a.ext_read = (short)sc_.ext_read;
        a.open_ref = (short)sc_.open_ref;
        a.ext_ref = (short)sc_.ext_ref;
        int gaps;
        if (sc_.affine)
            gaps = (sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_) ? kGapAffineSym : kGapAffine;
        else
            gaps = (sc_.gap_read == sc_.gap_ref && !no_sym_) ? kGapSym : kGapLinear;
        const void *fn = plan_.geo->kernel[alg][gaps];
        const int block_lds = plan_.lds.total * plan_.waves_per_block;
        if (block_lds > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, block_lds),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        const long long pairs_per_block = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
        const long long blocks = (n + pairs_per_block - 1) / pairs_per_block;
        if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
        void *kargs[] = {&a};
        hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(plan_.waves_per_block * kWave), kargs,
                                  (size_t)block_lds, stream),
                  "hipLaunchKernel(score_kernel)");
    }


    // Long sequences: strips of kLongG*kLongK rows, boundary rows through an HBM scratch.
    void score_long_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores,
                           hipStream_t stream, bool wide) {
        if (band_width_ > 0 && alg != kAlgSW)
            throw std::runtime_error("band_width applies to Smith-Waterman scores only");
        if (sc_.affine)
            throw std::runtime_error("the long-read path implements the linear gap model only (read_length " +
                                     std::to_string(R_) + " needs row strips)");
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
        const size_t bytes_per_wave = (size_t)2 * (ppw / 2) * a.row_dwords * 4 * (wide ? 2 : 1);
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
        const void *fn = kLongKernels[alg][(sc_.gap_read == sc_.gap_ref && !no_sym_) ? 1 : 0][wide ? 1 : 0];
        const int long_lds = LongLds<kLongG, kLongK>::kTotal;
        for (long long begin = 0; begin < n; begin += chunk) {
            const long long cnt = std::min(chunk, n - begin);
            a.reads = d_reads + (size_t)begin * R_;
            a.refs = d_refs + (size_t)begin * F_;
            a.scores = d_scores + begin;
            a.brow = d_brow_;
            a.n = cnt;
            a.pp_total = waves * (ppw / 2) * (wide ? 2 : 1);
            void *kargs[] = {&a};
            hip_check(hipLaunchKernel(fn, dim3((unsigned)((cnt + ppw - 1) / ppw)), dim3(kWave), kargs,
                                      (size_t)long_lds, stream),
                      "hipLaunchKernel(score_long_kernel)");
        }
    }

    // int16 DP cells: the reference wraps silently.  Scores switch to int32 cells on the strip path
    // where they could; alignments (int16 only) are refused.
    bool int16_range_ok(int alg) const {
        try {
            check_int16_range(alg);
            return true;
        } catch (const std::runtime_error &) {
            return false;
        }
    }

    void check_int16_range(int alg) const {
        const long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
        const int worst_gap = std::min({sc_.gap_read, sc_.gap_ref, sc_.open_read, sc_.open_ref, sc_.ext_read, sc_.ext_ref, 0});
        // SW cells are >= 0; NW-variant score cells are bounded below by the cheaper border path
        const long long lo = alg == kAlgSW ? (long long)std::min(sc_.mismatch, 0) + worst_gap
                                           : (long long)(std::min(R_, F_) + 2) * std::min(worst_gap, std::min(sc_.mismatch, 0));
        if (hi > 32000 || lo < -32000 || (sc_.affine && alg == kAlgNW && lo < -15000))
            throw std::runtime_error("shape x scoring can leave the int16 range of the DP cells (read_length " +
                                     std::to_string(R_) + ", ref_length " + std::to_string(F_) + ")");
    }

    // Host pointers in, host scores out.  Chunked: while chunk c runs on the device the
    // host threads gather chunk c+1 into the other pinned slot.
    void score_host(int opt, int n, const char *const *reads, const char *const *refs, short *scores,
                    int threads) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const size_t per_pair = (size_t)R_ + F_;
        long long chunk = per_pair ? (long long)((48u << 20) / per_pair) : n;
        chunk = std::max<long long>(chunk, 1024);
        chunk = std::min<long long>(chunk, n);
        ensure_staging(chunk);
        if (threads < 1) threads = 1;
        threads = std::min(threads, 64);
        int slot = 0;
        for (long long begin = 0; begin < n; begin += chunk, slot ^= 1) {
            const long long cnt = std::min<long long>(chunk, n - begin);
            hip_check(hipEventSynchronize(slot_done_[slot]), "hipEventSynchronize");
            if (slot_pending_[slot] > 0) {         // drain the result of the chunk that used this slot
                memcpy(scores + slot_begin_[slot], h_scores_[slot], sizeof(short) * (size_t)slot_pending_[slot]);
                slot_pending_[slot] = 0;
            }
            gather(reads + begin, refs + begin, cnt, h_reads_[slot], h_refs_[slot], threads);
            hipStream_t st = streams_[slot];
            hip_check(hipMemcpyAsync(d_reads_[slot], h_reads_[slot], (size_t)cnt * R_, hipMemcpyHostToDevice, st), "H2D reads");
            hip_check(hipMemcpyAsync(d_refs_[slot], h_refs_[slot], (size_t)cnt * F_, hipMemcpyHostToDevice, st), "H2D refs");
            score_device(opt, cnt, d_reads_[slot], d_refs_[slot], d_scores_[slot], st);
            hip_check(hipMemcpyAsync(h_scores_[slot], d_scores_[slot], sizeof(short) * (size_t)cnt, hipMemcpyDeviceToHost, st), "D2H scores");
            hip_check(hipEventRecord(slot_done_[slot], st), "hipEventRecord");
            slot_begin_[slot] = begin;
            slot_pending_[slot] = cnt;
        }
        for (int s = 0; s < 2; ++s) {
            hip_check(hipEventSynchronize(slot_done_[s]), "hipEventSynchronize");
            if (slot_pending_[s] > 0) {
                memcpy(scores + slot_begin_[s], h_scores_[s], sizeof(short) * (size_t)slot_pending_[s]);
                slot_pending_[s] = 0;
            }
        }
    }


    // ---- compute_alignments ----

    // Device-resident batch -> rows (n * 2 * (R+F) bytes: read row then ref row, right-justified,
    // zero before the start, NUL at R+F-1) and idx (n * 4 shorts).  Asynchronous on `stream`;
    // the pointer scratch is reused chunk after chunk in stream order.
    void align_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                      short *d_idx, hipStream_t stream) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;
        if (plan_.long_mode)
            throw std::runtime_error("compute_alignments needs the pair to fit one register sweep (read_length <= 2048, "
                                     "reference resident in LDS); this shape only supports score_alignments");
        check_int16_range(alg);
        if (alg == kAlgNW && (long long)(R_ + 1) * std::min({sc_.gap_ref, sc_.open_ref, sc_.ext_ref, 0}) < (sc_.affine ? -15000 : -32000))
            throw std::runtime_error("NW alignment border (read_length * gap score) leaves the int16 range");
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const int G = plan_.geo->G, K = plan_.geo->K, AL = R_ + F_;
        const int blocks8 = (F_ + G - 1 + 7) / 8;
        const long long ppb = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
        const size_t bytes_per_pp = (size_t)G * blocks8 * K * 4 * (sc_.affine ? 2 : 1);
        // Pointer scratch: as much of the batch per launch as memory allows (a 1 M-pair launch keeps
        // the latency-bound traceback kernel at full occupancy), capped at 24 GiB and half the free HBM.
        size_t free_b = 0, total_b = 0;
        hip_check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
        const size_t have = trace_pairs_ > 0 ? (size_t)(trace_pairs_ / 2) * bytes_per_pp : 0;
        const size_t cap = std::min<size_t>(24ull << 30, std::max<size_t>((free_b + have) / 2, 256ull << 20));
        long long chunk = (long long)(cap / bytes_per_pp) * 2;
        chunk = std::max(ppb, chunk / ppb * ppb);
        chunk = std::min(chunk, (n + ppb - 1) / ppb * ppb);
        ensure_trace_scratch(chunk, bytes_per_pp, stream);
        hip_check(hipMemsetAsync(d_rows, 0, (size_t)n * 2 * AL, stream), "hipMemsetAsync(rows)");
        if (sse_policy_ && sc_.affine)
            throw std::runtime_error("traceback_policy = 1 (SSE/AVX tie-breaks) exists for the linear gap model only");
        const void *fn = plan_.geo->fill[alg][sse_policy_ ? 3 : (sc_.affine ? 2 : ((sc_.gap_read == sc_.gap_ref && !no_sym_) ? 1 : 0))];
        const int block_lds = plan_.lds.total * plan_.waves_per_block;
        if (block_lds > kDefaultBlockLds)
            hip_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, block_lds),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        for (long long begin = 0; begin < n; begin += chunk) {
            const long long cnt = std::min(chunk, n - begin);
            FillArgs f;
            f.reads = d_reads + (size_t)begin * R_;
            f.refs = d_refs + (size_t)begin * F_;
            f.ptr = d_ptr_;
            f.ends = d_ends_;
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
            TraceArgs t;
            t.reads = f.reads;
            t.refs = f.refs;
            t.ptr = d_ptr_;
            t.ends = d_ends_;
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
            t.open_read = f.open_read;
            t.ext_read = f.ext_read;
            t.open_ref = f.open_ref;
            t.ext_ref = f.ext_ref;
            void *targs[] = {&t};
            hip_check(hipLaunchKernel((const void *)&traceback_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256),
                                      targs, 0, stream),
                      "hipLaunchKernel(traceback_kernel)");
        }
    }

    // Host pointers in, Alignment[] out: the rows of every pair are fresh operator new[] blocks
    // (the host's ~Alignment delete[]s them, include/AlignmentKernel.h:20-23).
    template <typename AlignmentT>
    void align_host(int opt, int n, const char *const *reads, const char *const *refs, AlignmentT *alignments,
                    int threads) {
        const int alg = opt & 0xF;
        if (alg > 1 || n <= 0) return;
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const int AL = R_ + F_;
        const size_t per_pair = (size_t)3 * AL + 8;
        long long chunk = per_pair ? (long long)((256u << 20) / per_pair) : n;
        chunk = std::max<long long>(chunk, 1024);
        chunk = std::min<long long>(chunk, n);
        ensure_staging(chunk);
        ensure_align_staging(chunk);
        if (threads < 1) threads = 1;
        threads = std::min(threads, 64);
        auto drain = [&](int s) {
            if (slot_pending_[s] <= 0) return;
            scatter(alignments + slot_begin_[s], slot_pending_[s], h_rows_[s], h_idx_[s], threads);
            slot_pending_[s] = 0;
        };
        int slot = 0;
        for (long long begin = 0; begin < n; begin += chunk, slot ^= 1) {
            const long long cnt = std::min<long long>(chunk, n - begin);
            hip_check(hipEventSynchronize(slot_done_[slot]), "hipEventSynchronize");
            drain(slot);
            gather(reads + begin, refs + begin, cnt, h_reads_[slot], h_refs_[slot], threads);
            hipStream_t st = streams_[0];          // one stream: the pointer scratch is shared
            hip_check(hipMemcpyAsync(d_reads_[slot], h_reads_[slot], (size_t)cnt * R_, hipMemcpyHostToDevice, st), "H2D reads");
            hip_check(hipMemcpyAsync(d_refs_[slot], h_refs_[slot], (size_t)cnt * F_, hipMemcpyHostToDevice, st), "H2D refs");
            align_device(opt, cnt, d_reads_[slot], d_refs_[slot], d_rows_[slot], d_idx_[slot], st);
            hip_check(hipMemcpyAsync(h_rows_[slot], d_rows_[slot], (size_t)cnt * 2 * AL, hipMemcpyDeviceToHost, st), "D2H rows");
            hip_check(hipMemcpyAsync(h_idx_[slot], d_idx_[slot], sizeof(short) * 4 * (size_t)cnt, hipMemcpyDeviceToHost, st), "D2H idx");
            hip_check(hipEventRecord(slot_done_[slot], st), "hipEventRecord");
            slot_begin_[slot] = begin;
            slot_pending_[slot] = cnt;
        }
        for (int s = 0; s < 2; ++s) {
            hip_check(hipEventSynchronize(slot_done_[s]), "hipEventSynchronize");
            drain(s);
        }
    }

    std::string describe(int opt, long long n) const {
        const long long ppb = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
        char buf[640];
        snprintf(buf, sizeof buf,
                 "{\"arch\": \"%s\", \"device\": %d, \"alg\": %d, \"affine\": %d, \"group_lanes\": %d, "
                 "\"rows_per_lane\": %d, \"padded_rows\": %d, \"pairs_per_wave\": %d, \"waves_per_block\": %d, "
                 "\"lds_per_wave\": %d, \"lds_per_block\": %d, \"steps\": %d, \"blocks\": %lld, \"long_mode\": %d, "
                 "\"band_width\": %d}",
                 arch_.c_str(), device_, opt & 0xF, sc_.affine ? 1 : 0, plan_.geo->G, plan_.geo->K,
                 plan_.geo->G * plan_.geo->K, plan_.pairs_per_wave, plan_.waves_per_block, plan_.lds.total,
                 plan_.lds.total * plan_.waves_per_block, F_ + plan_.geo->G - 1, n > 0 ? (n + ppb - 1) / ppb : 0,
                 plan_.long_mode ? 1 : 0, band_width_);
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
    }

    LaunchPlan choose_plan(int force_g, int force_k) const {
        LaunchPlan best;
        double best_cost = 0;
        for (int i = 0; i < kNumGeometries; ++i) {
            const Geometry &g = kGeometries[i];
            if (g.G * g.K < R_) continue;
            if (force_g && (g.G != force_g || (force_k && g.K != force_k))) continue;
            if (!force_g && force_k && g.K != force_k) continue;
            LaunchPlan p;
            p.geo = &g;
            p.lds = g.lds(R_, F_);
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
                if ((wpb == 1 || wpb == 2 || wpb == 4) && p.lds.total * wpb <= kMaxBlockLds) {
                    p.waves_per_block = wpb;
                    best_waves = std::min(32, (kMaxBlockLds / (p.lds.total * wpb)) * wpb);
                }
            }
            if (best_waves == 0) continue;
            // lane-steps per pair, weighted by instructions per step (per-row work + fixed part)
            const double per_step = g.K * (sc_.affine ? 11.0 : 7.0) + 14.0;
            double cost = (double)(F_ + g.G - 1) * per_step * g.G / 2.0;
            if (best_waves < 8) cost *= 1.0 + 0.08 * (8 - best_waves);         // fewer than two waves per SIMD
            if (!best.geo || cost < best_cost) {
                best = p;
                best_cost = cost;
            }
        }
        if (!best.geo || (getenv("VALIGN_HIP_FORCE_LONG") && !force_g)) {
            if (force_g || force_k)
                throw std::runtime_error("the forced kernel geometry does not fit read_length=" + std::to_string(R_) +
                                         ", ref_length=" + std::to_string(F_));
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


    void release_trace_scratch() {
        if (d_ptr_) (void)hipFree(d_ptr_);
        if (d_ends_) (void)hipFree(d_ends_);
        d_ptr_ = nullptr;
        d_ends_ = nullptr;
        trace_pairs_ = 0;
        for (int s = 0; s < 2; ++s) {
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

    void ensure_trace_scratch(long long pairs, size_t bytes_per_pp, hipStream_t stream) {
        if (pairs <= trace_pairs_) return;
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");   // nothing may still read the old scratch
        if (d_ptr_) (void)hipFree(d_ptr_);
        if (d_ends_) (void)hipFree(d_ends_);
        d_ptr_ = nullptr;
        d_ends_ = nullptr;
        const long long ppw = plan_.pairs_per_wave;
        const long long waves = (pairs + ppw - 1) / ppw;
        hip_check(hipMalloc((void **)&d_ptr_, (size_t)(waves * (ppw / 2)) * bytes_per_pp), "hipMalloc(pointer scratch)");
        hip_check(hipMalloc((void **)&d_ends_, sizeof(EndCell) * (size_t)(waves * ppw)), "hipMalloc(end cells)");
        trace_pairs_ = pairs;
    }

    void ensure_align_staging(long long pairs) {
        if (pairs <= align_staged_pairs_) return;
        const size_t AL = (size_t)R_ + F_;
        for (int s = 0; s < 2; ++s) {
            if (h_rows_[s]) (void)hipHostFree(h_rows_[s]);
            if (h_idx_[s]) (void)hipHostFree(h_idx_[s]);
            if (d_rows_[s]) (void)hipFree(d_rows_[s]);
            if (d_idx_[s]) (void)hipFree(d_idx_[s]);
            hip_check(hipHostMalloc((void **)&h_rows_[s], std::max<size_t>((size_t)pairs * 2 * AL, 16), hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_idx_[s], sizeof(short) * 4 * (size_t)pairs, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc((void **)&d_rows_[s], std::max<size_t>((size_t)pairs * 2 * AL, 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_idx_[s], sizeof(short) * 4 * (size_t)pairs), "hipMalloc");
        }
        align_staged_pairs_ = pairs;
    }

    template <typename AlignmentT>
    void scatter(AlignmentT *alignments, long long cnt, const uint8_t *rows, const short *idx, int threads) const {
        const size_t AL = (size_t)R_ + F_;
        auto work = [=](long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                AlignmentT &a = alignments[i];
                a.read = new char[AL ? AL : 1];
                a.ref = new char[AL ? AL : 1];
                memcpy(a.read, rows + (size_t)i * 2 * AL, AL);
                memcpy(a.ref, rows + (size_t)i * 2 * AL + AL, AL);
                a.readStart = idx[4 * i + 0];
                a.readEnd = idx[4 * i + 1];
                a.refStart = idx[4 * i + 2];
                a.refEnd = idx[4 * i + 3];
            }
        };
        if (threads <= 1 || cnt < 2048) {
            work(0, cnt);
            return;
        }
        std::vector<std::thread> pool;
        const long long per = (cnt + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            const long long lo = t * per, hi = std::min(cnt, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto &th : pool) th.join();
    }

    void release_staging() {
        for (int s = 0; s < 2; ++s) {
            if (h_reads_[s]) (void)hipHostFree(h_reads_[s]);
            if (h_refs_[s]) (void)hipHostFree(h_refs_[s]);
            if (h_scores_[s]) (void)hipHostFree(h_scores_[s]);
            if (d_reads_[s]) (void)hipFree(d_reads_[s]);
            if (d_refs_[s]) (void)hipFree(d_refs_[s]);
            if (d_scores_[s]) (void)hipFree(d_scores_[s]);
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
        for (int s = 0; s < 2; ++s) {
            hip_check(hipHostMalloc((void **)&h_reads_[s], std::max<size_t>((size_t)pairs * R_, 16), hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_refs_[s], std::max<size_t>((size_t)pairs * F_, 16), hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc((void **)&h_scores_[s], sizeof(short) * (size_t)pairs, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc((void **)&d_reads_[s], std::max<size_t>((size_t)pairs * R_, 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_refs_[s], std::max<size_t>((size_t)pairs * F_, 16)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_scores_[s], sizeof(short) * (size_t)pairs), "hipMalloc");
        }
        staged_pairs_ = pairs;
    }

    void gather(const char *const *reads, const char *const *refs, long long cnt, uint8_t *dst_reads,
                uint8_t *dst_refs, int threads) const {
        const int R = R_, F = F_;
        auto work = [=](long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                memcpy(dst_reads + (size_t)i * R, reads[i], (size_t)R);
                memcpy(dst_refs + (size_t)i * F, refs[i], (size_t)F);
            }
        };
        if (threads <= 1 || cnt < 4096) {
            work(0, cnt);
            return;
        }
        std::vector<std::thread> pool;
        const long long per = (cnt + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            const long long lo = t * per, hi = std::min(cnt, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto &th : pool) th.join();
    }

    int device_, R_, F_;
    Scoring sc_;
    bool sse_policy_ = false;
    int band_width_ = 0;
    int score_width_ = 0;
    bool no_sym_ = getenv("VALIGN_HIP_NO_SYM") != nullptr;   // tuning switch: use the two-gap kernel always
    std::string arch_;
    LaunchPlan plan_;
    hipStream_t streams_[2] = {nullptr, nullptr};
    hipEvent_t slot_done_[2] = {nullptr, nullptr};
    long long slot_begin_[2] = {0, 0}, slot_pending_[2] = {0, 0};
    long long staged_pairs_ = 0;
    uint8_t *h_reads_[2] = {nullptr, nullptr}, *h_refs_[2] = {nullptr, nullptr};
    short *h_scores_[2] = {nullptr, nullptr};
    uint8_t *d_reads_[2] = {nullptr, nullptr}, *d_refs_[2] = {nullptr, nullptr};
    int16_t *d_scores_[2] = {nullptr, nullptr};
    // compute_alignments: pointer scratch + end cells (device), result staging (both sides)
    unsigned *d_brow_ = nullptr;       // long-read path: strip boundary rows
    size_t brow_bytes_ = 0;
    unsigned *d_ptr_ = nullptr;
    EndCell *d_ends_ = nullptr;
    long long trace_pairs_ = 0, align_staged_pairs_ = 0;
    uint8_t *h_rows_[2] = {nullptr, nullptr}, *d_rows_[2] = {nullptr, nullptr};
    short *h_idx_[2] = {nullptr, nullptr}, *d_idx_[2] = {nullptr, nullptr};
};

}  // namespace valign
