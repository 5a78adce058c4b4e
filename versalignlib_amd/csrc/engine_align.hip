// engine_align.hip -- Engine: compute_alignments (reference: src/Kernels/default/DefaultKernel.cpp:21-50, 204-525) -- fill +
// traceback launches, row strips for long reads, the fused small-batch launch, and the host-pointer pipeline.  traceback_kernel
// and first_invalid_kernel (not templates) are defined in this translation unit.
#define VALIGN_TU_ALIGN 1
#include "engine.hip.h"
#include "versalign_plugin_abi.h"

namespace valign {

// compute_alignments for reads beyond one register sweep (strip_kernels.hip.h): a wave per pair-of-pairs, K rows per lane
struct StripGeometry {
    int K;
    WaveLds (*lds)(int R, int F);
    const void *kernel[2];
    const void *affine_kernel[2];      // nullptr: too many rows per lane for the affine kernel's registers
    const void *sse_kernel[2];         // traceback_policy = 1 (linear gaps)
    const void *wide_kernel[2][3];     // int32 cells, [alg][0 linear gaps, 1 affine, 2 SSE tie-breaks]; nullptr: no such instance
};
// int32 cells are the rare path: every mode at 8 rows per lane, the NW variant with linear gaps (the reference's model: long
// reads whose column-0 border leaves int16) at 16 / 12 as well
#define VALIGN_STRIP_WIDE_ALL(K)                                                                                       \
    {{(const void *)&align_strip_wide_kernel<K, kAlgSW>, (const void *)&align_strip_wide_kernel<K, kAlgSW, true>,      \
      (const void *)&align_strip_wide_kernel<K, kAlgSW, false, true>},                                                 \
     {(const void *)&align_strip_wide_kernel<K, kAlgNW>, (const void *)&align_strip_wide_kernel<K, kAlgNW, true>,      \
      (const void *)&align_strip_wide_kernel<K, kAlgNW, false, true>}}
#define VALIGN_STRIP_WIDE_NW(K) {{nullptr, nullptr, nullptr}, {(const void *)&align_strip_wide_kernel<K, kAlgNW>, nullptr, nullptr}}
#define VALIGN_STRIP_WIDE_NONE {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}
#define VALIGN_STRIP_SSE(K) {(const void *)&align_strip_kernel<K, kAlgSW, false, true>, (const void *)&align_strip_kernel<K, kAlgNW, false, true>}
template <int K>
static WaveLds strip_lds(int, int) {            // the profile of 64 K rows and the ring of slab numbers: no term in F
    return WaveLds{StripLds<K>::kRing, 0, StripLds<K>::kTotal};
}
static const StripGeometry kStripGeometries[] = {
    {16, &strip_lds<16>, {(const void *)&align_strip_kernel<16, kAlgSW>, (const void *)&align_strip_kernel<16, kAlgNW>},
     {(const void *)&align_strip_kernel<16, kAlgSW, true>, (const void *)&align_strip_kernel<16, kAlgNW, true>}, VALIGN_STRIP_SSE(16),
     VALIGN_STRIP_WIDE_NW(16)},
    {12, &strip_lds<12>, {(const void *)&align_strip_kernel<12, kAlgSW>, (const void *)&align_strip_kernel<12, kAlgNW>},
     {(const void *)&align_strip_kernel<12, kAlgSW, true>, (const void *)&align_strip_kernel<12, kAlgNW, true>}, VALIGN_STRIP_SSE(12),
     VALIGN_STRIP_WIDE_NW(12)},
    {8, &strip_lds<8>, {(const void *)&align_strip_kernel<8, kAlgSW>, (const void *)&align_strip_kernel<8, kAlgNW>},
     {(const void *)&align_strip_kernel<8, kAlgSW, true>, (const void *)&align_strip_kernel<8, kAlgNW, true>}, VALIGN_STRIP_SSE(8),
     VALIGN_STRIP_WIDE_ALL(8)},
};
#undef VALIGN_STRIP_SSE
#undef VALIGN_STRIP_WIDE_ALL
#undef VALIGN_STRIP_WIDE_NW
#undef VALIGN_STRIP_WIDE_NONE

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

// Which fill kernel a call of this mode takes on geometry `geo`, and what its pointer stream looks like.
Engine::FillChoice Engine::fill_choice(int alg, const Geometry &geo) const {
    FillChoice c;
    const int rows = geo.G * geo.K;
    // affine gaps with the traceback information tagged into the cells (4-bit codes, 4-step blocks)
    c.affine_tagged = sc_.affine && !sse_policy_ && !no_tag_ && affine_tagged_range_ok(alg, rows, geo.K);
    // linear gaps: the pointer rides in the low bits of the cell where 4x the cell range still fits int16 (and, for SW,
    // gap_ref < 0); otherwise the equality-test kernels (both tie-break policies)
    c.tagged = !sc_.affine && !no_tag_ && tagged_range_ok(alg, rows);
    // SW: one (value, row) key per lane instead of a first-arg-max per row where value << 4 (5 bits of row for more than
    // 16 rows per lane) still fits int16
    const long long key_top = ((long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1) << (geo.K <= 16 ? 4 : 5);
    const bool lane_key = c.tagged && alg == kAlgSW && key_top <= 32000;
    // ... and where 64x the cell range fits (K <= 16), the key rides in the query profile instead of being computed
    const bool prof_key = lane_key && !sse_policy_ && !no_prof_key_ && geo.K <= 16 &&
                          (((long long)std::min(R_, F_) * std::max(sc_.match, 0) + 2) << 6) <= 32000 &&
                          64ll * std::max(std::abs(sc_.gap_read), std::abs(sc_.gap_ref)) < 32000 && 64ll * std::abs(sc_.mismatch) < 16000;
    const bool affine_sym = sc_.affine && sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_;
    if (prof_key) c.kernel = kFillTagProfKey;
    else if (c.tagged) c.kernel = sse_policy_ ? (lane_key ? kFillSseTagKey : kFillSseTag) : (lane_key ? kFillTagKey : kFillTag);
    else if (sse_policy_) c.kernel = kFillSse;
    else if (sc_.affine) c.kernel = c.affine_tagged ? (affine_sym ? kFillAffineTagSym : kFillAffineTag) : (affine_sym ? kFillAffineSym : kFillAffine);
    else c.kernel = (sc_.gap_read == sc_.gap_ref && !no_sym_) ? kFillLinearSym : kFillLinear;
    return c;
}

// The plan an alignment call of this mode runs on: the engine's own where its geometry carries the kernel the call needs;
// otherwise (a fallback kernel on a geometry compiled with the fast set only, kernel_instances.hip.h) the cheapest FULL
// geometry that fits the read -- same results, the sweep a few per cent longer.
const LaunchPlan &Engine::align_plan_for(int alg, FillChoice &choice) {
    const LaunchPlan &base = align_base_plan();
    choice = fill_choice(alg, *base.geo);
    if (base.geo->fill[alg][choice.kernel]) return base;
    if (!fallback_plan_.geo) fallback_plan_ = choose_plan(R_, F_, 0, 0, false, true);
    choice = fill_choice(alg, *fallback_plan_.geo);
    if (!fallback_plan_.geo->fill[alg][choice.kernel]) throw std::runtime_error("no alignment kernel for this mode");
    return fallback_plan_;
}

bool Engine::align_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                  short *d_idx, hipStream_t stream, const WalkChain *chain) {
    const int alg = opt & 0xF;
    if (alg > 1 || n <= 0) return false;
    // Alignments whose cells leave int16 (the reference's shorts would wrap): int32 cells on the row-strip path, one pair per
    // register (align_strip_wide_kernel) -- every mode; only scores so large that (R + F) * |score| nears 2^28 are refused
    // (column 0 of the NW variant: a gap of the whole read -- linear (R + 1) gap_ref; affine open_ref + R ext_ref, which
    // check_int16_range covers)
    const bool border_bad = alg == kAlgNW && !sc_.affine && (long long)(R_ + 1) * std::min(sc_.gap_ref, 0) < -32000;
    bool in_range = true;
    try {
        check_int16_range(alg);
    } catch (const std::runtime_error &) {
        in_range = false;
    }
    if (border_bad || !in_range || wide_align_) {
        const long long worst = std::max({std::abs((long long)sc_.match), std::abs((long long)sc_.mismatch),
                                          std::abs((long long)(sc_.affine ? sc_.open_read : sc_.gap_read)), std::abs((long long)(sc_.affine ? sc_.open_ref : sc_.gap_ref)),
                                          sc_.affine ? std::abs((long long)sc_.ext_read) : 0ll, sc_.affine ? std::abs((long long)sc_.ext_ref) : 0ll});
        if ((long long)(R_ + F_ + 2) * worst >= (1ll << 28))
            throw std::runtime_error("shape x scoring can leave the int32 range of the DP cells (read_length " + std::to_string(R_) +
                                     ", ref_length " + std::to_string(F_) + ")");
        hip_check(hipSetDevice(device_), "hipSetDevice");
        align_strips_device(alg, n, d_reads, d_refs, d_rows, d_idx, stream, true);
        return false;
    }
    hip_check(hipSetDevice(device_), "hipSetDevice");
    // row strips: reads beyond one register sweep, and -- measured, profiles/r04_rate_sweep.txt -- reads of more than 1 024
    // rows, whose resident geometries (64 x 24 / 64 x 32: 34 to 53 KB of LDS) fill at 0.8-2.1 TCUPS where 12- or 16-row
    // strips at eight waves per CU do 1.6-2.3 (1 200 x 3 000: 41 / 74 ms -> 26 / 37 ms, linear / affine)
    if (align_base_plan().long_mode || (!force_g_ && !force_k_ && R_ > 1024)) {
        align_strips_device(alg, n, d_reads, d_refs, d_rows, d_idx, stream);
        return false;
    }
    if (sse_policy_ && sc_.affine)
        throw std::runtime_error("traceback_policy = 1 (SSE/AVX tie-breaks) exists for the linear gap model only");
    // the fill kernel of this mode -- and the geometry that has it: the plan's own, or the next full one (fallback kernels)
    FillChoice fc;
    const LaunchPlan &plan = align_plan_for(alg, fc);
    const bool affine_tagged = fc.affine_tagged, tagged = fc.tagged;
    const int G = plan.geo->G, K = plan.geo->K, AL = R_ + F_;
    int blocks8 = affine_tagged ? (F_ + G - 1 + 3) / 4 : (F_ + G - 1 + 7) / 8;            // blocks of steps per lane
    const long long ppb = (long long)plan.pairs_per_wave * plan.waves_per_block;
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
    ensure_trace_scratch(chunk, bytes_per_pp, plan.pairs_per_wave, stream);
    const void *fn = plan.geo->fill[alg][fc.kernel];
    const int block_lds = plan.lds.total * plan.waves_per_block;
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
        f.prof_area = plan.lds.prof_area;
        f.refc_stride = plan.lds.refc_stride;
        f.wave_lds = plan.lds.total;
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
        hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(plan.waves_per_block * kWave), fargs,
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
            if (chain && chain->min_start) {
                hip_check(hipMemsetD32Async((hipDeviceptr_t)chain->min_start, AL, 1, trace_stream_), "hipMemsetD32Async(first column)");
                t.min_start = chain->min_start;
            }
        }
        hip_check(hipLaunchKernel((const void *)&traceback_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256),
                                  targs, 0, walk_stream),
                  "hipLaunchKernel(traceback_kernel)");
        if (chain && chain->min_start && chain->packed) {
            CompactArgs c{t.rows, chain->packed, chain->min_start, 2 * cnt, AL};
            void *cargs[] = {&c};
            hip_check(hipLaunchKernel((const void *)&compact_rows_kernel, dim3((unsigned)(2 * cnt)), dim3(256), cargs, 0, walk_stream),
                      "hipLaunchKernel(compact_rows_kernel)");
        }
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

void Engine::ensure_trace_stream() {
    if (trace_stream_) return;
    hip_check(hipStreamCreateWithFlags(&trace_stream_, hipStreamNonBlocking), "hipStreamCreate(traceback)");
    hip_check(hipEventCreateWithFlags(&entry_ev_, hipEventDisableTiming), "hipEventCreate");
    for (int r = 0; r < 2; ++r) {
        hip_check(hipEventCreateWithFlags(&fill_done_[r], hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&trace_done_[r], hipEventDisableTiming), "hipEventCreate");
    }
}

bool Engine::align_fused(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows, short *d_idx,
                 hipStream_t stream) {
    if (no_fused_ || sc_.affine || sse_policy_ || no_tag_ || align_base_plan().long_mode || force_g_ || force_k_ || !tagged_range_ok(alg, 256)) return false;     // (256: the tallest fused geometry)
    if (wide_align_) return false;
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

void Engine::align_strips_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, uint8_t *d_rows,
                         short *d_idx, hipStream_t stream, bool wide) {
    const bool affine = sc_.affine;
    if (sse_policy_ && affine)
        throw std::runtime_error("traceback_policy = 1 (SSE/AVX tie-breaks) exists for the linear gap model only");
    const int wide_mode = affine ? 1 : (sse_policy_ ? 2 : 0);
    // 16 rows per lane unless fewer leave less padding: 1 024-row strips cost 136 ms where 768-row strips cost 152 and 512-row
    // strips 156 (10 kbp x 10 kbp, 4 096 pairs; 24 and 32 rows per lane -- 218 / 221 ms, two waves per SIMD by their
    // registers -- are gone).  LDS no longer depends on the shape: the profile of the strip's rows and a 256-byte ring.
    const StripGeometry *geo = nullptr;
    WaveLds lds{};
    double best_cost = 0.0;
    for (const StripGeometry &g : kStripGeometries) {
        if (affine && !g.affine_kernel[alg]) continue;
        if (sse_policy_ && !g.sse_kernel[alg]) continue;
        if (wide && !g.wide_kernel[alg][wide_mode]) continue;
        if (strip_k_ && g.K != strip_k_) continue;
        const int rows_g = 64 * g.K;
        const double cost = (double)((R_ + rows_g - 1) / rows_g) * rows_g * (g.K == 16 ? 1.0 : (g.K == 12 ? 1.115 : 1.147));
        if (!geo || cost < best_cost) {
            geo = &g;
            lds = g.lds(rows_g, F_);
            best_cost = cost;
        }
    }
    if (!geo) throw std::runtime_error("no strip alignment kernel for this mode");
    const int K = geo->K, rows = 64 * K, AL = R_ + F_;
    const int strips = std::max(1, (R_ + rows - 1) / rows), pad_total = strips * rows - R_;
    const int blocks8 = (F_ + 63 + 7) / 8;
    const int row_dwords = ((F_ + 71) / 64 + 2) * 64;
    const size_t strip_words = (size_t)blocks8 * 64 * K * (affine ? 2 : 1);    // per wave (= pair-of-pairs) and strip
    // boundary row sets: H, and F beside it (affine); int32 cells: those per pair
    const int row_sets = (affine ? 2 : 1) * (wide ? 2 : 1);
    const size_t bytes_per_pp = strip_words * 4 * strips + (size_t)2 * row_sets * row_dwords * 4;
    size_t free_b = 0, total_b = 0;
    hip_check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
    // (the pointer stream of a 10 kbp x 10 kbp pair-of-pairs is 50 MB: what fits the scratch is what runs side by side --
    // 24 GB, the bound until round 4, kept 480 waves on 1 024 SIMDs; half of the free HBM, at most 128 GB, now)
    size_t cap = std::min<size_t>(128ull << 30, std::max<size_t>((free_b + trace_bytes_) / 2, 256ull << 20));
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
    const void *fn = wide ? geo->wide_kernel[alg][wide_mode] : (affine ? geo->affine_kernel[alg] : (sse_policy_ ? geo->sse_kernel[alg] : geo->kernel[alg]));
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
        t.wide_score = wide ? 1 : 0;
        void *targs[] = {&t};
        hip_check(hipLaunchKernel((const void *)&traceback_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), targs, 0, stream),
                  "hipLaunchKernel(traceback_kernel)");
    }
}

template <typename Sink>
void Engine::align_host(int opt, int n, const char *const *reads, const char *const *refs, Sink alignments,
                int threads) {
    const int alg = opt & 0xF;
    if (alg > 1 || n <= 0) return;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const int AL = R_ + F_;
    const size_t per_pair = (size_t)3 * AL + 8;
    // (row strips run chunk after chunk on one pointer scratch: chunks that fill the device -- 2 000 pairs-of-pairs and more --
    // instead of 128 MB of staging, which is 1 100 of them at 10 kbp x 10 kbp: 253 -> ~190 ms per 4 096 pairs through the ABI)
    const bool by_strips = align_base_plan().long_mode || (!force_g_ && !force_k_ && R_ > 1024);
    const size_t chunk_bytes = by_strips && !dbg_.on("align_chunk_bytes") ? std::max<size_t>(align_chunk_bytes_, 384u << 20) : align_chunk_bytes_;
    long long chunk = per_pair ? (long long)(chunk_bytes / per_pair) : n;
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
    // Result rows are right-justified strings behind zeros: on the staged paths only the columns from the chunk's smallest
    // readStart on cross PCIe, packed on the device (compact_rows_kernel; 0.42 instead of 1.36 GB per million pairs of
    // 150 x 500) -- the scatter unpacks them and writes the zeros in front, as it always did.  A registered flat
    // destination still receives whole rows straight from the copy engine: there the zeros would be the host's to write.
    auto drain = [&](int s) {
        if (slot_pending_[s] <= 0) return;
        const auto t0 = std::chrono::steady_clock::now();
        if (!direct_rows) scatter(alignments + slot_begin_[s], slot_pending_[s], h_rows_[s], h_idx_[s], threads, (size_t)start_col_[s]);
        host_stats_.drain_ms += ms_between(t0, std::chrono::steady_clock::now());
        host_stats_.d2h_row_bytes += (double)slot_pending_[s] * 2 * (AL - start_col_[s]);
        host_stats_.full_row_bytes += (double)slot_pending_[s] * 2 * AL;
        slot_pending_[s] = 0;
    };
    int slot = 0;
    long long chunk_no = 0;
    chain_regions_busy_[0] = chain_regions_busy_[1] = false;       // (every earlier call ended with its walks waited for)
    prime_copy_engines(copy_in, copy_out, chunk);
    if (!copy_issuer_) copy_issuer_.reset(new CopyIssuer(device_));
    CopyIssuer *copy_issuer = copy_issuer_.get();
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
        copy_issuer->wait_issued(slot);             // (only then is the slot's event the one of its last chunk)
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
        const bool packed = !direct_rows && !whole_rows_;
        const WalkChain chain{(int)(chunk_no & 1), chunk, packed ? d_min_start_ + slot : nullptr, packed ? d_packed_rows_[slot] : nullptr};
        const bool chained = align_device(opt, cnt, d_reads_[slot], d_refs_[slot], d_rows_[slot], d_idx_[slot], kernels, &chain);
        hip_check(hipEventRecord(kernels_done_[slot], chained ? trace_stream_ : kernels), "hipEventRecord");      // the chunk's last kernel
        ++chunk_no;
        uint8_t *rows_to = direct_rows ? direct_rows + (size_t)begin * 2 * AL : h_rows_[slot];
        short *idx_to = direct_idx ? direct_idx + 4 * begin : h_idx_[slot];
        // SDMA, not a blit kernel beside the next fill: the copies are issued once the host has seen the kernels end
        CopyIssuer::Job job{kernels_done_[slot], {rows_to, idx_to}, {d_rows_[slot], d_idx_[slot]},
                            {(size_t)cnt * 2 * AL, sizeof(short) * 4 * (size_t)cnt}, copy_out, slot_done_[slot], slot};
        start_col_[slot] = 0;
        if (chained && packed) {        // (the stream-order fallback -- row strips, a scratch too small for two regions -- copies whole rows)
            job.src[0] = d_packed_rows_[slot];
            job.d_min = d_min_start_ + slot;
            job.h_min = h_min_start_ + slot;
            job.row_bytes = AL;
            job.rows = 2 * cnt;
            job.col = &start_col_[slot];
        }
        copy_issuer->submit(job);
        slot_begin_[slot] = begin;
        slot_pending_[slot] = cnt;
    }
    for (int k = 0; k < kSlots; ++k) {              // oldest chunk first
        const int s = (slot + k) % kSlots;
        const auto t0 = std::chrono::steady_clock::now();
        copy_issuer->wait_issued(s);
        hip_check(hipEventSynchronize(slot_done_[s]), "hipEventSynchronize");
        host_stats_.wait_ms += ms_between(t0, std::chrono::steady_clock::now());
        drain(s);
    }
    quiesce.armed = false;               // (everything has been waited for)
}

void Engine::prime_copy_engines(hipStream_t copy_in, hipStream_t copy_out, long long staged_pairs) {
    if (copy_engines_primed_) return;
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

void Engine::release_trace_scratch() {
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
        if (d_packed_rows_[s]) (void)hipFree(d_packed_rows_[s]);
        d_packed_rows_[s] = nullptr;
        h_rows_[s] = nullptr;
        h_idx_[s] = nullptr;
        d_rows_[s] = nullptr;
        d_idx_[s] = nullptr;
    }
    align_staged_pairs_ = 0;
    if (d_min_start_) (void)hipFree(d_min_start_);
    if (h_min_start_) (void)hipHostFree(h_min_start_);
    d_min_start_ = nullptr;
    h_min_start_ = nullptr;
}

void Engine::ensure_trace_scratch(long long pairs, size_t bytes_per_pp, long long ppw, hipStream_t stream) {
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

void Engine::ensure_align_staging(long long pairs) {
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
        if (d_packed_rows_[s]) (void)hipFree(d_packed_rows_[s]);
        d_packed_rows_[s] = nullptr;
        hip_check(hipMalloc((void **)&d_packed_rows_[s], (size_t)pairs * 2 * AL + 64), "hipMalloc(packed rows)");
    }
    if (!d_min_start_) {
        hip_check(hipMalloc((void **)&d_min_start_, sizeof(int) * kSlots), "hipMalloc(first columns)");
        hip_check(hipHostMalloc((void **)&h_min_start_, sizeof(int) * kSlots, hipHostMallocDefault), "hipHostMalloc(first columns)");
    }
    align_staged_pairs_ = pairs;
}

// the two sinks of align_host: the ABI's Alignment array, and caller-provided contiguous buffers (valign_hip_align_host)
template void Engine::align_host<Alignment *>(int, int, const char *const *, const char *const *, Alignment *, int);
template void Engine::align_host<FlatSink>(int, int, const char *const *, const char *const *, FlatSink, int);

}  // namespace valign
