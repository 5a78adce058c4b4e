// engine.hip.h -- the host side of libHIPKernel.so: one Engine per (device, read_length, ref_length, scoring).
//
// The class is declared here; its parts are separate translation units, compiled in parallel and testable apart:
//   engine_core.hip ..... construction, launch-plan selection (cost model), cell-range checks, staging, describe()
//   engine_score.hip .... score_alignments: register-sweep launches, the host-pointer chunk pipeline, 4-bit class
//                         unpacking, length-sorted batches (reference: DefaultKernel.cpp:52-202)
//   engine_long.hip ..... long reads: row strips (score_long_kernel) and the banded block chain (score_band_kernel)
//   engine_align.hip .... compute_alignments: fill + traceback launches, row strips, the fused small-batch launch, the
//                         host-pointer pipeline with its copy-issuing thread (reference: DefaultKernel.cpp:21-50, 204-525)
//   hip_plugin.hip ...... the plugin ABI and the flat C API over it
// The closest reference precedent for the staging loops is the OpenCL backend's gather / copy / launch / collect loop
// (src/Kernels/OpenCL/OpenCLKernel.cpp:57-108); unlike it, chunks here are large, asynchronous and overlapped.
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
#define VALIGN_DECLARE_FULL(G, K) VALIGN_FAST_KERNELS(extern template, G, K) VALIGN_FALLBACK_KERNELS(extern template, G, K)
#define VALIGN_DECLARE_FAST(G, K) VALIGN_FAST_KERNELS(extern template, G, K)
VALIGN_ALL_PARTS(VALIGN_DECLARE_FULL, VALIGN_DECLARE_FAST)
#undef VALIGN_DECLARE_FULL
#undef VALIGN_DECLARE_FAST

// alignment fill kernels of a geometry, by what the engine selects (Engine::fill_kernel_index)
enum FillKernel {
    kFillLinear = 0,        // equality-test pointers, two gap scores            (full geometries only)
    kFillLinearSym,         // ... one shared gap score                           (full)
    kFillAffine,            // affine, equality tests                             (full)
    kFillSse,               // SSE2 / AVX2 tie-breaks, equality tests             (full)
    kFillTag,               // pointer tagged into the cell; SW: per-row arg-max  (NW: every geometry; SW: full)
    kFillTagKey,            // ... SW with one end-cell key per lane              (every geometry)
    kFillAffineSym,         // affine with symmetric scores, equality tests       (full)
    kFillAffineTag,         // affine, tagged cells, different scores per direction (full)
    kFillAffineTagSym,      // ... symmetric scores                               (every geometry)
    kFillSseTag,            // SSE tie-breaks, tagged; SW: per-row arg-max        (full)
    kFillSseTagKey,         // ... SW with the lane key                           (full)
    kFillTagProfKey,        // SW, the end-cell key rides in the query profile    (every geometry)
    kFillKernels
};

// One compiled (G, K) geometry with its kernel variants.
struct Geometry {
    int G, K;
    bool full;                     // carries the fallback kernels too (kernel_instances.hip.h)
    WaveLds (*lds)(int R, int F);
    const void *kernel[2][7];      // score kernels [alg][linear, symmetric linear, affine, symmetric affine,
                                   //                     symmetric affine / affine / symmetric linear on half floats]
    const void *fill[2][kFillKernels];      // alignment fill kernels [alg][FillKernel]; nullptr: not compiled for this geometry
};

template <int G, int K>
constexpr void set_fast_kernels(Geometry &g) {
    g.kernel[0][0] = (const void *)&score_kernel<G, K, kAlgSW, kGapLinear>;
    g.kernel[0][1] = (const void *)&score_kernel<G, K, kAlgSW, kGapSym>;
    g.kernel[0][2] = (const void *)&score_kernel<G, K, kAlgSW, kGapAffine>;
    g.kernel[0][3] = (const void *)&score_kernel<G, K, kAlgSW, kGapAffineSym>;
    g.kernel[0][4] = (const void *)&score_kernel<G, K, kAlgSW, kGapAffineSymF16>;
    g.kernel[0][5] = (const void *)&score_kernel<G, K, kAlgSW, kGapAffineF16>;
    g.kernel[0][6] = (const void *)&score_kernel<G, K, kAlgSW, kGapSymF16>;
    g.kernel[1][0] = (const void *)&score_kernel<G, K, kAlgNW, kGapLinear>;
    g.kernel[1][1] = (const void *)&score_kernel<G, K, kAlgNW, kGapSym>;
    g.kernel[1][2] = (const void *)&score_kernel<G, K, kAlgNW, kGapAffine>;
    g.kernel[1][3] = (const void *)&score_kernel<G, K, kAlgNW, kGapAffineSym>;
    g.kernel[1][4] = (const void *)&score_kernel<G, K, kAlgNW, kGapAffineSymF16>;
    g.kernel[1][5] = (const void *)&score_kernel<G, K, kAlgNW, kGapAffineF16>;
    g.kernel[1][6] = (const void *)&score_kernel<G, K, kAlgNW, kGapSymF16>;
    g.fill[0][kFillTagKey] = (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, false>;
    g.fill[0][kFillTagProfKey] = (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, false, false, true>;
    g.fill[0][kFillAffineTagSym] = (const void *)&align_fill_affine_tag_kernel<G, K, kAlgSW, true>;
    g.fill[1][kFillTag] = (const void *)&align_fill_tag_kernel<G, K, kAlgNW, false, false>;
    g.fill[1][kFillAffineTagSym] = (const void *)&align_fill_affine_tag_kernel<G, K, kAlgNW, true>;
}

template <int G, int K>
constexpr Geometry fast_geometry() {
    Geometry g{G, K, false, &wave_lds<G, K>, {}, {}};
    set_fast_kernels<G, K>(g);
    return g;
}

template <int G, int K>
constexpr Geometry full_geometry() {
    Geometry g{G, K, true, &wave_lds<G, K>, {}, {}};
    set_fast_kernels<G, K>(g);
    g.fill[0][kFillLinear] = (const void *)&align_fill_kernel<G, K, kAlgSW, false>;
    g.fill[0][kFillLinearSym] = (const void *)&align_fill_kernel<G, K, kAlgSW, true>;
    g.fill[0][kFillAffine] = (const void *)&align_fill_affine_kernel<G, K, kAlgSW, false>;
    g.fill[0][kFillAffineSym] = (const void *)&align_fill_affine_kernel<G, K, kAlgSW, true>;
    g.fill[0][kFillSse] = (const void *)&align_fill_sse_kernel<G, K, kAlgSW>;
    g.fill[0][kFillTag] = (const void *)&align_fill_tag_kernel<G, K, kAlgSW, false, false>;
    g.fill[0][kFillSseTag] = (const void *)&align_fill_tag_kernel<G, K, kAlgSW, false, true>;
    g.fill[1][kFillLinear] = (const void *)&align_fill_kernel<G, K, kAlgNW, false>;
    g.fill[1][kFillLinearSym] = (const void *)&align_fill_kernel<G, K, kAlgNW, true>;
    g.fill[1][kFillAffine] = (const void *)&align_fill_affine_kernel<G, K, kAlgNW, false>;
    g.fill[1][kFillAffineSym] = (const void *)&align_fill_affine_kernel<G, K, kAlgNW, true>;
    g.fill[1][kFillSse] = (const void *)&align_fill_sse_kernel<G, K, kAlgNW>;
    g.fill[0][kFillSseTagKey] = (const void *)&align_fill_tag_kernel<G, K, kAlgSW, true, true>;
    g.fill[0][kFillAffineTag] = (const void *)&align_fill_affine_tag_kernel<G, K, kAlgSW, false>;
    g.fill[1][kFillSseTag] = (const void *)&align_fill_tag_kernel<G, K, kAlgNW, false, true>;
    g.fill[1][kFillAffineTag] = (const void *)&align_fill_affine_tag_kernel<G, K, kAlgNW, false>;
    return g;
}

// Rows covered = G*K.  Ordered by capacity; selection is by estimated cost (Engine::choose_plan).  The full geometries --
// one per row capacity 64 / 160 / 320 / 512 / 1024 / 2048 -- are where calls that need a fallback kernel are re-planned to.
static const Geometry kGeometries[] = {
    fast_geometry<8, 4>(),   fast_geometry<8, 6>(),   full_geometry<8, 8>(),   fast_geometry<16, 4>(),
    fast_geometry<8, 10>(),  fast_geometry<8, 12>(),  fast_geometry<16, 8>(),  full_geometry<16, 10>(),
    fast_geometry<16, 12>(), fast_geometry<32, 8>(),  full_geometry<32, 10>(), fast_geometry<32, 12>(),
    full_geometry<64, 8>(),  fast_geometry<64, 12>(), full_geometry<64, 16>(), fast_geometry<64, 24>(),
    full_geometry<64, 32>(),
};
constexpr int kNumGeometries = sizeof(kGeometries) / sizeof(kGeometries[0]);
#define VALIGN_COUNT(G, K) +1
static_assert(kNumGeometries == 0 VALIGN_ALL_PARTS(VALIGN_COUNT, VALIGN_COUNT), "kernel_instances.hip.h lists other geometries than this table");
#undef VALIGN_COUNT

constexpr int kMaxBlockLds = 160 * 1024;       // gfx950: 160 KiB per CU, one block may take it all
constexpr int kSlots = 4;                      // staging slots of the host-pointer pipeline
constexpr int kDefaultBlockLds = 64 * 1024;    // above this the kernel attribute must be raised

// Long-read path (row strips + column phases, long_kernels.hip.h): one geometry.
constexpr int kLongG = 16, kLongK = 10;

struct LaunchPlan {
    bool long_mode = false;        // sequences too long for one register sweep / LDS-resident reference
    const Geometry *geo = nullptr;
    WaveLds lds{};
    int waves_per_block = 4;
    int pairs_per_wave = 0;
};

// ONE environment variable for tests and experiments -- VALIGN_HIP_DEBUG="name[=value],name[=value],..." -- read when an
// engine is created; production hosts never set it.  Unknown names are refused: a typo must not silently test nothing.
class DebugSwitches {
public:
    DebugSwitches() {
        const char *env = getenv("VALIGN_HIP_DEBUG");
        if (!env) return;
        static const char *const known[] = {"no_sym", "no_tag", "no_f16", "no_fused", "no_prof_key", "no_overlap", "no_band_chain",
                                            "force_long", "wide_align", "strip_k", "no_single_strip", "no_direct_out", "ragged_min", "chunk_bytes",
                                            "align_chunk_bytes", "direct_bytes", "scratch_cap_mb", "whole_rows", "short_strips"};
        std::string s(env);
        for (size_t at = 0; at <= s.size();) {
            const size_t end = std::min(s.find(',', at), s.size());
            const std::string item = s.substr(at, end - at);
            at = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            const std::string name = item.substr(0, eq);
            bool ok = false;
            for (const char *k : known) ok = ok || name == k;
            if (!ok) throw std::runtime_error("VALIGN_HIP_DEBUG: unknown switch '" + name + "'");
            kv_[name] = eq == std::string::npos ? 1 : atoll(item.c_str() + eq + 1);
        }
    }
    bool on(const char *name) const { return kv_.count(name) != 0; }
    long long value(const char *name, long long fallback) const {
        auto it = kv_.find(name);
        return it == kv_.end() ? fallback : it->second;
    }

private:
    std::map<std::string, long long> kv_;
};

}  // namespace valign

#include "host_runtime.hip.h"

namespace valign {

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
        double d2h_row_bytes = 0, full_row_bytes = 0;        // align_host: result-row bytes that crossed PCIe / that whole rows would have been
    };

    Engine(int device, int R, int F, const Scoring &sc, int force_g, int force_k);

    ~Engine();

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
    // Half-float cells for score_alignments where they are exact (identical scores, fewer instructions): 1 on (default),
    // 0 integer cells only (what BASELINE.json's "int16" headline is measured with)
    void set_half_float_cells(int mode) {
        if (mode != 0 && mode != 1) throw std::runtime_error("half_float_cells must be 0 or 1");
        no_f16_ = mode == 0 || dbg_.on("no_f16");
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
                      int16_t *d_scores, hipStream_t stream, bool length_sorted = true);

    // One launch of the register-sweep score kernel over n pairs of shape R x F (sequences laid
    // out pair-major at exactly those strides) with the geometry of `plan`.
    // `groups` (length-sorted batches): packed groups sharing the read stride R, each with its own
    // reference stride <= F, swept by one launch; offsets are relative to d_reads / d_refs / d_scores.
    void launch_score(const LaunchPlan &plan, int alg, int R, int F, long long n, const uint8_t *d_reads,
                      const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream,
                      const LengthGroup *groups = nullptr, int n_groups = 0);


    // n sequences of `len` 4-bit classes -> n * len canonical bytes (pack_kernels.hip.h)
    void launch_unpack(const uint8_t *d_packed, uint8_t *d_out, long long n, int len, hipStream_t stream);

    // ---- banded Smith-Waterman scores, linear gaps: the cyclic block chain of band_kernels.hip.h ----
    static constexpr int kBandK = VALIGN_HIP_BAND_CHAIN_BLOCK_ROWS;              // rows per block = the band definition's block (describe: band_block_rows)

    struct BandPlan {
        bool usable = false, unit_delay = false;
        int nb = 0, first_block = 0, pad_rows = 0, d = 0, ring_depth = 0, code_cols = 0, events = 0;
        long long cells = 0;                // DP cells one pair's band windows hold (what the chain actually sweeps)
        std::vector<BandBlock> blocks;
        std::vector<int> fill_to;
    };

    // Windows, start distance, delays and ring sizes of the block chain for (R, F, band): what tools/band_schedule_model.py
    // calls plan().  `usable` is false where the chain does not pay or does not fit (then score_long_kernel's strips run).
    BandPlan make_band_plan() const;

    // score_alignments(SW, linear gaps, band_width > 0) on the block chain; false: not applicable here (strips run instead)
    bool score_band_device(long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream);
    bool band_chain_in_use() const;
    bool long_single_strip(bool wide) const;

    // Long sequences: strips of kLongG*kLongK rows, boundary rows through an HBM scratch.
    void score_long_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores,
                           hipStream_t stream, bool wide);

    // align_fill_affine_tag_kernel keeps 8 * cell + tag in int16; SW needs open scores < 0 (the tag rides on the
    // open constant) and, for the lane key, value << 4 (5) in range
    bool affine_tagged_range_ok(int alg, int rows, int K) const;

    // align_fill_tag_kernel keeps 4 * cell + tag in int16
    bool tagged_range_ok(int alg, int rows) const;

    // what score_alignments computes in for this mode at the engine's full shape
    const char *score_cell_format(int alg) const;

    // Every cell of an R x F sweep and everything added to it stays an integer of magnitude <= 2048:
    // exact in half floats (kGapAffineSymF16 / kGapAffineF16).  SW cells are >= 0; cells of the NW
    // variant are bounded below by the cheaper border path (as in check_int16_range).
    // NW: plus what the kernels' tilted frame adds to a cell of a sweep of `rows` padded rows (score_kernel).
    bool half_float_exact(int alg, int R, int F, int rows) const;

    // kGapSymF16 for Smith-Waterman scales every value by 2^-10 and floors with the [0, 1] clamp of the
    // packed add: cells must stay below 1024, scores be integers of magnitude < 1024
    bool half_float_unit_exact(int R, int F) const;

    // int16 DP cells: the reference wraps silently.  Scores switch to int32 cells on the strip path
    // where they could; alignments (int16 only) are refused.
    // The NW score kernels keep cell (p, j) plus -g_ref * p - g_read * j (g: gap / extension scores, <= 0): the most that
    // adds over a sweep of `rows` padded rows and F columns.
    long long nw_tilt_span(int rows, int F) const;
    int widest_sweep_rows() const;

    bool int16_range_ok(int alg) const;

    // score_path: score_alignments' register sweep (the NW variant's tilted frame counts)
    void check_int16_range(int alg, bool score_path = false) const;

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
                    int threads, int16_t *d_dest = nullptr);


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
        int *min_start;             // device word: the traceback leaves the chunk's smallest readStart there (or nullptr)
        uint8_t *packed;            // ... and the rows, packed to their columns from there on, go here (compact_rows_kernel)
    };
    hipEvent_t trace_done(int region) const { return trace_done_[region]; }
    struct FillChoice {
        int kernel = 0;                 // FillKernel
        bool affine_tagged = false, tagged = false;
    };
    FillChoice fill_choice(int alg, const Geometry &geo) const;
    const LaunchPlan &align_plan_for(int alg, FillChoice &choice);
    // The plan compute_alignments starts from: the engine's own -- unless that is the long-read one only because the score
    // kernels prefer it (a reference that starves LDS) while the read fits a register sweep with two waves per CU or so:
    // row strips are 512 rows tall, 150-row reads fill a third of them (150 x 4 000: 1.06 TCUPS on strips).
    const LaunchPlan &align_base_plan() const {
        if (plan_.long_mode && align_resident_.geo && !align_resident_.long_mode && R_ <= 1024 && align_resident_.lds.total <= 140 * 1024) {
            const int wpb = align_resident_.waves_per_block;
            const int waves_per_cu = (kMaxBlockLds / std::max(1, align_resident_.lds.total * wpb)) * wpb;
            if (waves_per_cu >= 2 || !sc_.affine) return align_resident_;      // (150 x 8 000 affine, one wave per CU: 60 ms against 53 on strips)
        }
        return plan_;
    }

    bool align_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                      short *d_idx, hipStream_t stream, const WalkChain *chain = nullptr);

    void ensure_trace_stream();

    // Fill + traceback of a small batch in one launch (linear gaps, default tie-breaks, tagged cells): false when the
    // shape / scoring has no fused kernel (the caller takes the three-kernel path).
    bool align_fused(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows, short *d_idx,
                     hipStream_t stream);

    // Reads beyond one register sweep: row strips of 64 * K rows, one launch per strip in stream order, boundary
    // rows ping-pong through HBM, one pointer region per strip, then the same traceback kernel (strip_kernels.hip.h).
    // Linear or affine gaps, Default tie-breaks, int16 cells (the reference's; where they would wrap the call is refused
    // by check_int16_range above instead of wrapping silently).
    void align_strips_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                             short *d_idx, hipStream_t stream, bool wide = false);

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
                    int threads);

    // host-side phases of the last score_host / align_host call
    std::string host_phases() const;

    std::string describe(int opt, long long n) const;

private:
    void validate_scoring();

    // latency: pick for the shortest single sweep (few pairs: every wave has a SIMD to itself and the call takes
    // as long as one wave does) instead of for the most cell updates per second
    LaunchPlan choose_plan(int R, int F, int force_g, int force_k, bool latency = false, bool full_only = false) const;

    static LaunchPlan long_plan();


    // A chunk of the host-pointer pipeline is one kernel launch: sized in whole "rounds" of the waves the device runs side by
    // side (16,384 pairs at 16 x 10: 8 pairs per wave, 8 waves per CU, 256 CUs), it leaves no partly filled last round --
    // a 48 MB chunk of 150 x 500 was 4.57 rounds and paid for 5.
    long long whole_rounds(long long pairs) const;

    // Small calls skip the chunk pipeline (VALIGN_HIP_DIRECT_BYTES: sequence bytes up to which they do; 0 = never)
    bool direct_call(long long n, size_t per_pair) const;

    // Input copies (H2D) and result copies (D2H) of align_host must not share an SDMA engine: 0.65 GB in and 1.36 GB out
    // per million pairs of 150 x 500 would queue up behind each other (36 ms of copying beside 28 ms of kernels).
    // The runtime gives a stream the lowest-numbered engine that is FREE at the stream's first copy and keeps it there
    // (AMD_LOG_LEVEL=4: "Last copy mask 0x1" on both streams, profiles/r03_copy_engines.txt) -- and at the start of a
    // pipeline the first result copy finds the input engine idle.  So, once per engine: a small result copy is issued
    // while a long input copy keeps engine 0 busy, which lands the result stream on the next engine for good.  No
    // API promises this; where it does not work the only loss is the overlap.
    void prime_copy_engines(hipStream_t copy_in, hipStream_t copy_out, long long staged_pairs);

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
    void reset_pipeline();

    void release_trace_scratch();

    // The pointer stream's bytes per pair-of-pairs depend on the fill kernel the call selects (tagged /
    // untagged, 4- or 8-step blocks, one or two code words): capacity is tracked in BYTES, so a call with a
    // wider stream than the one that sized the scratch reallocates instead of writing past it.
    void ensure_trace_scratch(long long pairs, size_t bytes_per_pp, long long ppw, hipStream_t stream);

    void ensure_align_staging(long long pairs);

    // gather / scatter between the caller's scattered blocks and the staging: host_pipeline.h (host-only, sanitizer-tested)
    template <typename Sink>
    void scatter(Sink sink, long long cnt, const uint8_t *rows, const short *idx, int threads, size_t first_col = 0) {
        packer_.scatter(sink, cnt, rows, idx, threads, first_col);
    }

    void release_staging();

    void ensure_staging(long long pairs);

    // ---- length-sorted batching (score_host, Smith-Waterman) ----

    static double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    }

    // Both modes (round 3): rows and columns of trailing non-ACGT bytes score 0 against everything, so every value of the
    // real matrix's last row / last column runs down its diagonal unchanged to the padded matrix's last row / column (gap
    // moves only lose) -- the NW variant's max(0, last column, last row) of the trimmed pair IS that of the padded pair,
    // as the Smith-Waterman maximum is.  Checked on the oracle (tests/test_oracle_golden.py) and on the GPU.
    bool ragged_applies(int alg) const;

    // Share of the padded cells a length-sorted sweep would still visit, from 256 pairs spread over
    // the call: sorting costs the host a pass over every tail, so it has to buy something.
    double sampled_cell_fraction(const char *const *reads, const char *const *refs, long long n) const;

    // Read classes: the row capacities of the compiled geometries below read_length, then
    // read_length itself.  Reference classes: multiples of 64 columns, then ref_length.
    void build_length_classes();

    const LaunchPlan &class_plan(int R, int F);

    static int trimmed_length(const unsigned char *s, int len);

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

    void ensure_ragged(int c, long long n);
    void release_ragged();

    // first half: trimmed lengths -> bins, the histogram on its way to the host.  Asynchronous on `stream`.
    void ragged_begin(int c, long long n, const uint8_t *d_reads, const uint8_t *d_refs, hipStream_t stream);

    // Fold bins too small to be worth a launch into the next larger one and lay the groups out: a read class with too few
    // pairs for a launch of its own joins the next read class (bin by bin); inside a class, a reference bin smaller than a
    // few blocks joins the next wider one.  Both dimensions only ever grow, so the padded sweep still covers the pair.
    std::vector<LengthGroup> fold_groups(std::vector<long long> &total, std::vector<int> &group_of_bin) const;

    // second half: waits for the histogram, lays the groups out, then -- asynchronously on `stream` -- packs the pairs by
    // group, sweeps class by class and puts the scores back in the caller's order.  `always` false: false is returned, and
    // nothing launched, where the classes would still visit two thirds of the padded cells or more.
    bool ragged_finish(int c, int alg, long long n, int16_t *d_scores, hipStream_t stream, bool always);

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
    size_t score_chunk_bytes_ = 48u << 20;                   // staging chunk of score_host (debug switch chunk_bytes)
    size_t align_chunk_bytes_ = 128u << 20;                  // staging chunk of align_host, inputs + results (debug switch align_chunk_bytes)
    size_t direct_bytes_ = 768u << 10;                       // calls with at most this many sequence bytes run on the pinned staging directly
    long long ragged_min_ = 2048;                             // pairs a length bin needs for its own launch
    std::vector<int> read_caps_, ref_caps_;
    std::vector<unsigned char> read_class_;
    std::vector<unsigned short> ref_class_;
    std::map<std::pair<int, int>, LaunchPlan> class_plans_;
    RaggedCtx rag_[kSlots + 1];                                  // one per pipeline slot, the last for device-resident batches
    hipStream_t ragged_dev_stream_ = nullptr;                    // stream and end of the last device-resident length-sorted call
    hipEvent_t ragged_dev_done_ = nullptr;
    uint8_t *d_read_class_ = nullptr;
    uint16_t *d_ref_class_ = nullptr;
    HostPacker packer_{R_, F_};                               // (declared after R_ / F_)
    HostStats host_stats_;
    DebugSwitches dbg_;                                       // VALIGN_HIP_DEBUG (tests and experiments only)
    bool no_sym_ = dbg_.on("no_sym");                         // the two-gap kernels always
    bool no_tag_ = dbg_.on("no_tag");   // equality-test pointer kernels for linear alignments
    bool no_f16_ = dbg_.on("no_f16");   // int16 cells for symmetric affine SW too
    bool no_fused_ = dbg_.on("no_fused");   // small alignment calls as fill + traceback kernels
    bool no_prof_key_ = dbg_.on("no_prof_key");   // compute the SW lane key instead of carrying it in the profile
    bool copy_engines_primed_ = false;
    std::unique_ptr<CopyIssuer> copy_issuer_;
    bool wide_align_ = dbg_.on("wide_align");         // alignments on int32 cells always (align_strip_wide_kernel)
    bool whole_rows_ = dbg_.on("whole_rows");         // result rows cross PCIe whole (A/B of the device-side packing)
    bool no_direct_out_ = dbg_.on("no_direct_out");   // stage + scatter even into registered result buffers
    bool no_overlap_ = dbg_.on("no_overlap");   // tracebacks in stream order behind their fills
    int strip_k_ = (int)dbg_.value("strip_k", 0);                  // rows per lane of the strip alignment kernels (16 / 12 / 8): tests
    long long scratch_cap_mb_ = dbg_.value("scratch_cap_mb", 0);   // small pointer scratch: chunked alignment batches in tests (key: pointer_scratch_cap_mb)
    hipStream_t trace_stream_ = nullptr;                          // helper stream of align_device (walks beside the next fill)
    bool chain_regions_busy_[2] = {false, false};                 // WalkChain: the region's last walk may still be running
    hipEvent_t fill_done_[2] = {nullptr, nullptr}, trace_done_[2] = {nullptr, nullptr}, entry_ev_ = nullptr;
    std::string arch_;
    LaunchPlan plan_, latency_plan_;
    LaunchPlan align_resident_;         // what choose_plan picked before the score path's preferences for the long-read kernels
    LaunchPlan fallback_plan_;          // alignments that need a kernel only the full geometries carry (align_plan_for)
    hipStream_t streams_[kSlots] = {};
    hipEvent_t slot_done_[kSlots] = {};
    long long slot_begin_[kSlots] = {}, slot_pending_[kSlots] = {};
    long long staged_pairs_ = 0;
    uint8_t *h_reads_[kSlots] = {}, *h_refs_[kSlots] = {};
    short *h_scores_[kSlots] = {};
    uint8_t *d_reads_[kSlots] = {}, *d_refs_[kSlots] = {};
    uint8_t *d_pack_reads_[kSlots] = {}, *d_pack_refs_[kSlots] = {};     // 4-bit classes as they arrive (score path)
    bool pack_ = true;                                                   // host_packing: 4-bit base classes across PCIe
    int16_t *d_scores_[kSlots] = {};
    // compute_alignments: pointer scratch + end cells (device), result staging (both sides)
    BandPlan band_plan_;               // banded linear SW: the block chain's plan for band_plan_width_, its tables on the device
    int band_plan_width_ = -1;
    int band_blocks_per_cu_ = 0, band_lds_ = 0;          // of the last block-chain launch (describe)
    int long_strip_rows_ = 0;                            // rows per strip of the last score_long_kernel launch (describe)
    BandBlock *d_band_blocks_ = nullptr;
    int *d_band_fill_ = nullptr;
    int cu_count_ = 0;
    bool no_band_chain_ = dbg_.on("no_band_chain");      // banded scores on score_long_kernel's strips
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
    uint8_t *d_packed_rows_[kSlots] = {};                        // the chunk's rows without their all-zero leading columns
    int *d_min_start_ = nullptr, *h_min_start_ = nullptr;       // per slot: first column of the chunk's rows that holds a string (device / pinned)
    int start_col_[kSlots] = {};                                 // ... as the copy issuer used it (columns before it were not copied)
    hipEvent_t in_done_[kSlots] = {}, kernels_done_[kSlots] = {};   // align_host: H2D / kernels of the slot's chunk finished
};

}  // namespace valign
