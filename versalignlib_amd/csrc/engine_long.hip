// engine_long.hip -- Engine: long reads -- row strips with boundary rows through HBM (long_kernels.hip.h) and the banded
// cyclic block chain (band_kernels.hip.h); their kernel templates are instantiated in this translation unit.
#include "engine.hip.h"

namespace valign {

// The strip kernels of one geometry: [affine][alg][same scores both ways][int32 cells]
struct LongGeometry {
    int G, K;
    int lds[2];                        // per wave: linear / affine
    const void *kernel[2][2][2][2];
};
template <int G, int K>
constexpr LongGeometry long_geometry() {
    return LongGeometry{
        G, K, {LongLds<G, K, false>::kTotal, LongLds<G, K, true>::kTotal},
        {{{{(const void *)&score_long_kernel<G, K, kAlgSW, false, false>, (const void *)&score_long_kernel<G, K, kAlgSW, false, true>},
           {(const void *)&score_long_kernel<G, K, kAlgSW, true, false>, (const void *)&score_long_kernel<G, K, kAlgSW, true, true>}},
          {{(const void *)&score_long_kernel<G, K, kAlgNW, false, false>, (const void *)&score_long_kernel<G, K, kAlgNW, false, true>},
           {(const void *)&score_long_kernel<G, K, kAlgNW, true, false>, (const void *)&score_long_kernel<G, K, kAlgNW, true, true>}}},
         {{{(const void *)&score_long_kernel<G, K, kAlgSW, false, false, true>, (const void *)&score_long_kernel<G, K, kAlgSW, false, true, true>},
           {(const void *)&score_long_kernel<G, K, kAlgSW, true, false, true>, (const void *)&score_long_kernel<G, K, kAlgSW, true, true, true>}},
          {{(const void *)&score_long_kernel<G, K, kAlgNW, false, false, true>, (const void *)&score_long_kernel<G, K, kAlgNW, false, true, true>},
           {(const void *)&score_long_kernel<G, K, kAlgNW, true, false, true>, (const void *)&score_long_kernel<G, K, kAlgNW, true, true, true>}}}}};
}
// 16 x 10: strips of 160 rows -- the blocks the strip band is defined on (include/valign_hip.h), four lane groups = eight
// pairs per wave.  64 x 10 (round 4): strips of 640 rows for UNBANDED sweeps -- a quarter of the boundary-row traffic and
// of the per-strip work, and ONE lane group per wave, so the LDS rings shrink from four sets to one (affine: 20.7 -> 14 KB
// per wave, 7 -> 11 waves per CU).  These kernels are bound by how often a wave may issue, not by latency
// (profiles/r04_band_two_chains.txt): waves per SIMD are what they lacked.
static const LongGeometry kLongStrips = long_geometry<kLongG, kLongK>();
static const LongGeometry kLongTall = long_geometry<64, 8>();
static_assert(kLongG * kLongK == VALIGN_HIP_BAND_BLOCK_ROWS, "banded strips are the API's blocks");

Engine::BandPlan Engine::make_band_plan() const {
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
    p.first_block = std::max(first_real, 0);
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
    p.unit_delay = dmin == dmax && p.d == dmax + 1;
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
            const int r_lo = std::max(b * K - p.pad_rows, 0), r_hi = std::min((b + 1) * K - p.pad_rows - 1, R - 1);
            p.cells += (long long)(k.span + 1) * (r_hi - r_lo + 1);
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
    if (BandLds<kBandK>::total(p.code_cols, p.ring_depth, sc_.affine) > 40 * 1024) return p;
    p.usable = true;
    return p;
}

bool Engine::score_band_device(long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream) {
    if (no_band_chain_ || band_width_ <= 0) return false;
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
    a.first_block = p.first_block;
    a.pad_rows = p.pad_rows;
    a.d = p.d;
    a.ring_depth = p.ring_depth;
    a.code_cols = p.code_cols;
    a.match = (short)sc_.match;
    a.mismatch = (short)sc_.mismatch;
    a.gap_read = (short)sc_.gap_read;
    a.gap_ref = (short)sc_.gap_ref;
    a.open_read = (short)sc_.open_read;
    a.ext_read = (short)sc_.ext_read;
    a.open_ref = (short)sc_.open_ref;
    a.ext_ref = (short)sc_.ext_ref;
    const bool sym = (sc_.affine ? (sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref) : sc_.gap_read == sc_.gap_ref) && !no_sym_;
    static const void *const kernels[2][2][2] = {        // [affine][one score both ways][unit delay]
        {{(const void *)&score_band_kernel<kBandK, false, false>, (const void *)&score_band_kernel<kBandK, false, true>},
         {(const void *)&score_band_kernel<kBandK, true, false>, (const void *)&score_band_kernel<kBandK, true, true>}},
        {{(const void *)&score_band_kernel<kBandK, false, false, true>, (const void *)&score_band_kernel<kBandK, false, true, true>},
         {(const void *)&score_band_kernel<kBandK, true, false, true>, (const void *)&score_band_kernel<kBandK, true, true, true>}}};
    const void *fn = kernels[sc_.affine ? 1 : 0][sym ? 1 : 0][p.unit_delay ? 1 : 0];
    const int lds = BandLds<kBandK>::total(p.code_cols, p.ring_depth, sc_.affine);
    // as many one-wave blocks as run side by side; each takes quads of pairs in turn (band_kernels.hip.h)
    int per_cu = 0;
    hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kWave, (size_t)lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    band_blocks_per_cu_ = per_cu;
    band_lds_ = lds;
    const long long resident = (long long)std::max(per_cu, 1) * std::max(cu_count_, 1);
    const long long blocks = std::min<long long>((n + 3) / 4, resident);
    if (blocks > 0x7FFFFFFFll) throw std::runtime_error("batch too large for one launch");
    void *kargs[] = {&a};
    hip_check(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(kWave), kargs, (size_t)lds, stream), "hipLaunchKernel(score_band_kernel)");
    return true;
}

bool Engine::band_chain_in_use() const {
    return !no_band_chain_ && band_width_ > 0 && (band_plan_width_ == band_width_ ? band_plan_.usable : make_band_plan().usable);
}

// One 160-row strip, packed cells, no band: the long-read instances that keep nothing in HBM between launches.
bool Engine::long_single_strip(bool wide) const {
    return R_ <= kLongG * kLongK && !wide && band_width_ == 0 && !dbg_.on("no_single_strip");
}

void Engine::score_long_device(int alg, long long n, const uint8_t *d_reads, const uint8_t *d_refs, int16_t *d_scores,
                       hipStream_t stream, bool wide) {
    if (band_width_ > 0 && alg != kAlgSW)
        throw std::runtime_error("band_width applies to Smith-Waterman scores only");
    // linear gaps, banded: the cyclic block chain (int32 cells whatever score_width says: same results in the int16 range)
    if (band_width_ > 0 && alg == kAlgSW && score_band_device(n, d_reads, d_refs, d_scores, stream)) return;
    // unbanded sweeps of reads beyond a few strips take the tall strips; a band is defined on the 160-row blocks
    const LongGeometry &geo = (band_width_ == 0 && R_ > 2 * kLongTall.G * kLongTall.K && !dbg_.on("short_strips")) ? kLongTall : kLongStrips;
    const int rows = geo.G * geo.K;
    const int ppw = 2 * (kWave / geo.G);
    long_strip_rows_ = rows;
    LongArgs a;
    a.R = R_;
    a.F = F_;
    a.strips = std::max(1, (R_ + rows - 1) / rows);
    a.row_dwords = ((F_ + geo.G + kPhase - 1) / kPhase) * kPhase + kPhase;
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
    const bool single_strip = long_single_strip(wide);       // (no boundary rows: nothing to allocate, nothing shared between launches)
    if (!single_strip && (size_t)waves * bytes_per_wave > brow_bytes_) {
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        if (d_brow_) (void)hipFree(d_brow_);
        d_brow_ = nullptr;
        brow_bytes_ = (size_t)waves * bytes_per_wave;
        hip_check(hipMalloc((void **)&d_brow_, brow_bytes_), "hipMalloc(boundary rows)");
    }
    const bool affine_sym = sc_.open_read == sc_.open_ref && sc_.ext_read == sc_.ext_ref && !no_sym_;
    const bool sym = sc_.affine ? affine_sym : (sc_.gap_read == sc_.gap_ref && !no_sym_);
    const void *fn = geo.kernel[sc_.affine ? 1 : 0][alg][sym ? 1 : 0][wide ? 1 : 0];
    // half-float cells (score_long_kernel<..., F16>): Smith-Waterman with one gap score on the 160-row strips while every cell
    // stays below 1024 -- short reads against a reference the resident kernels' LDS cannot hold (150 x 8 000: 8.2 -> ~11 TCUPS)
    if (&geo == &kLongStrips && alg == kAlgSW && !sc_.affine && sym && !wide && band_width_ == 0 && !no_f16_ && half_float_unit_exact(R_, F_))
        fn = (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, false, true>;
    int long_lds = geo.lds[sc_.affine ? 1 : 0];
    // a read of ONE strip (short reads sent here for their reference's length): the instances without boundary rings
    if (single_strip) {
        static const void *const single[2][2][2] = {
            {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, false, false, false, true>,
              (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, false, false, true>},
             {(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, false, false, false, true>,
              (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, false, false, false, true>}},
            {{(const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, false, false, true, false, true>,
              (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, true, false, true>},
             {(const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, false, false, true, false, true>,
              (const void *)&score_long_kernel<kLongG, kLongK, kAlgNW, true, false, true, false, true>}}};
        const bool f16 = fn == (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, false, true>;
        fn = f16 ? (const void *)&score_long_kernel<kLongG, kLongK, kAlgSW, true, false, false, true, true>
                 : single[sc_.affine ? 1 : 0][alg][sym ? 1 : 0];
        long_lds = sc_.affine ? LongLds<kLongG, kLongK, true, true>::kTotal : LongLds<kLongG, kLongK, false, true>::kTotal;
    }
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

}  // namespace valign
