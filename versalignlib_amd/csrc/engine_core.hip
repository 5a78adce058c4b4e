// engine_core.hip -- Engine: construction, launch-plan selection, cell-range checks, staging, describe().
#include "engine.hip.h"

namespace valign {

Engine::Engine(int device, int R, int F, const Scoring &sc, int force_g, int force_k)
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
    align_resident_ = plan_;
    // Reads of more than 1 536 rows would take the 64 x 32 geometry -- 45 to 53 KB of LDS, three waves per CU --: the long-read
    // kernels (512-row strips, 12 to 15 waves per CU) sweep them 15 to 55 % faster (2 000 x 4 000: 6.2 -> 7.9 TCUPS linear,
    // 3.6 -> 5.6 affine; profiles/r04_rate_sweep.txt).  A forced geometry is a forced geometry.
    if (!force_g && !force_k && !plan_.long_mode && plan_.geo && plan_.geo->G * plan_.geo->K > 1536) plan_ = long_plan();
    // The resident kernels keep the slab numbers of the whole reference in LDS (2 ref_length bytes per lane group, and as
    // much again to stage it): at 150 x 8 000 one wave fills a CU's LDS -- 2.3 TCUPS, 0.7 at 150 x 32 000 -- where the
    // long-read kernels, whose slab numbers go through a ring, sweep 10.9 / 10.2 (affine: 1.5 / 0.4 against 5.9 / 5.6; their
    // single-strip instances, profiles/r04_rate_sweep.txt).  One wave per CU: always.  Two: with linear gaps, or when the
    // 160-row strips pad the read no worse than the resident geometry (150 x 4 000 affine: 2.9 -> 6.0).  Up to four (from about
    // 1 500 columns on): under the same padding condition (150 x 2 000: 8.7 -> 11.0 / 5.7 -> 6.0; 100 x 2 000 stays: 128 rows
    // against 160).
    if (!force_g && !force_k && !plan_.long_mode && plan_.geo) {
        const int waves_per_cu = std::min(32, (kMaxBlockLds / std::max(1, plan_.lds.total * plan_.waves_per_block)) * plan_.waves_per_block);
        const int strip_rows = kLongG * kLongK;
        // (reads of up to 1 024 rows take the 160-row strips; beyond, the 512-row ones, which the 64 x 24 geometry still beats)
        const bool pads_alike = R_ <= 1024 && ((R_ + strip_rows - 1) / strip_rows) * strip_rows * 100 <= plan_.geo->G * plan_.geo->K * 115;
        if (waves_per_cu <= 1 || (waves_per_cu <= 2 && (!sc_.affine || pads_alike)) || (waves_per_cu <= 4 && pads_alike)) plan_ = long_plan();
    }
    latency_plan_ = (force_g || force_k || plan_.long_mode) ? plan_ : choose_plan(R_, F_, 0, 0, true);
    build_length_classes();
    for (int s = 0; s < kSlots; ++s) hip_check(hipStreamCreateWithFlags(&streams_[s], hipStreamNonBlocking), "hipStreamCreate");
    for (int s = 0; s < kSlots; ++s) {
        hip_check(hipEventCreateWithFlags(&slot_done_[s], hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&in_done_[s], hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&kernels_done_[s], hipEventDisableTiming | hipEventBlockingSync), "hipEventCreate");     // (the copy issuer sleeps on it)
    }
}

Engine::~Engine() {
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

bool Engine::affine_tagged_range_ok(int alg, int geo_rows, int K) const {
    long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
    const int worst = std::min({sc_.open_read, sc_.open_ref, sc_.ext_read, sc_.ext_ref, sc_.mismatch, 0});
    // NW: every cell is at least the path "one gap up, one gap left"; E / F sit one open below H
    long long lo = alg == kAlgSW ? worst
                                 : 2ll * (std::min(sc_.open_read, 0) + std::min(sc_.open_ref, 0)) +
                                       (long long)(R_ + F_ + 2) * std::min({sc_.ext_read, sc_.ext_ref, 0}) + worst;
    if (alg == kAlgNW) {        // the kernel's tilted frame: cell (p, j) carries - ext_ref * p - ext_read * j on top
        const long long rows = (long long)geo_rows + 1, cols = F_ + 1;
        hi += std::max(0, -sc_.ext_ref) * rows + std::max(0, -sc_.ext_read) * cols;
        lo += std::min(0, -sc_.ext_ref) * rows + std::min(0, -sc_.ext_read) * cols;
        if (std::abs((long long)sc_.ext_ref) * rows > 3500 || std::abs((long long)sc_.ext_read) * cols > 3500) return false;
    }
    if (alg == kAlgSW && (sc_.open_read >= 0 || sc_.open_ref >= 0)) return false;
    const int key_bits = K <= 16 ? 4 : 5;
    if (alg == kAlgSW && ((hi + 1) << key_bits) > 32000) return false;
    return 8 * hi + 8 <= 32000 && 8 * lo - 8 >= -28000 && std::abs(sc_.match) < 1000 && std::abs(sc_.mismatch) < 1000;
}

bool Engine::tagged_range_ok(int alg, int rows) const {     // rows: padded rows of the sweep that would run
    long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
    if (alg == kAlgNW && !sse_policy_)           // the kernel's tilted frame: every cell plus -gap_ref * p - gap_read * j
        hi += (long long)-sc_.gap_ref * (rows + 1) + (long long)-sc_.gap_read * (F_ + 1);
    const int worst = std::min({sc_.gap_read, sc_.gap_ref, sc_.mismatch, 0});
    const long long lo = alg == kAlgSW ? worst : (long long)(R_ + F_ + 2) * worst;      // H(i,j) >= i gf + j gr
    if (alg == kAlgSW && !sse_policy_ && sc_.gap_ref >= 0) return false;
    return 4 * hi + 4 <= 32000 && 4 * lo - 4 >= -32000 && std::abs(sc_.match) < 2000 && std::abs(sc_.mismatch) < 2000;
}

const char *Engine::score_cell_format(int alg) const {
    if (alg > 1) return "none";
    if (alg == kAlgSW && band_chain_in_use()) return "int32";
    if (score_width_ == 32 || (score_width_ == 0 && !int16_range_ok(alg))) return "int32";
    if (!plan_.long_mode && sc_.affine && !no_f16_ && half_float_exact(alg, R_, F_, plan_.geo->G * plan_.geo->K)) return "f16";
    if (!plan_.long_mode && !sc_.affine && ((sc_.gap_read == sc_.gap_ref && !no_sym_) || alg == kAlgNW) && !no_f16_ &&
        (alg == kAlgNW ? half_float_exact(alg, R_, F_, plan_.geo->G * plan_.geo->K) : half_float_unit_exact(R_, F_)))
        return "f16";
    // (long-read kernels: Smith-Waterman with one gap score on the 160-row strips, engine_long.hip)
    if (plan_.long_mode && alg == kAlgSW && !sc_.affine && sc_.gap_read == sc_.gap_ref && !no_sym_ && !no_f16_ && band_width_ == 0 &&
        R_ <= 1024 && half_float_unit_exact(R_, F_))
        return "f16";
    return "int16";
}

bool Engine::half_float_exact(int alg, int R, int F, int rows) const {
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

bool Engine::half_float_unit_exact(int R, int F) const {
    const long long top = (long long)std::min(R, F) * std::max({sc_.match, sc_.mismatch, 0});
    const long long slack = std::max({std::abs(sc_.match), std::abs(sc_.mismatch), std::abs(sc_.gap_read), std::abs(sc_.gap_ref)});
    return top + 2 * slack < 1024 && slack < 512;
}

long long Engine::nw_tilt_span(int rows, int F) const {
    const long long per_row = -(long long)(sc_.affine ? sc_.ext_ref : sc_.gap_ref);
    const long long per_col = -(long long)(sc_.affine ? sc_.ext_read : sc_.gap_read);
    return per_row * (rows + 1) + per_col * (F + 1);
}

int Engine::widest_sweep_rows() const {
    int rows = 0;
    if (!plan_.long_mode && plan_.geo) rows = plan_.geo->G * plan_.geo->K;
    if (latency_plan_.geo && !latency_plan_.long_mode) rows = std::max(rows, latency_plan_.geo->G * latency_plan_.geo->K);
    return rows;
}

bool Engine::int16_range_ok(int alg) const {
    try {
        check_int16_range(alg, true);
        return true;
    } catch (const std::runtime_error &) {
        return false;
    }
}

void Engine::check_int16_range(int alg, bool score_path) const {
    long long hi = (long long)std::min(R_, F_) * std::max(sc_.match, 0) + 1;
    if (score_path && alg == kAlgNW && !plan_.long_mode) hi += nw_tilt_span(widest_sweep_rows(), F_);
    const int worst_gap = std::min({sc_.gap_read, sc_.gap_ref, sc_.open_read, sc_.open_ref, sc_.ext_read, sc_.ext_ref, 0});
    // SW cells are >= 0; NW-variant score cells are bounded below by the cheaper border path
    long long lo = alg == kAlgSW ? (long long)std::min(sc_.mismatch, 0) + worst_gap
                                 : (long long)(std::min(R_, F_) + 2) * std::min(worst_gap, std::min(sc_.mismatch, 0));
    // NW-variant alignments with affine gaps (plain frame, "minus infinity" = -16384): row 0 is free, so every H is at least a
    // gap straight down from it -- open_ref + (R - 1) ext_ref -- and E / F lie at most one opening below an H; a candidate
    // adds one mismatch.  (The product above charged every step an opening: -50 010 for 10 kbp reads at -5 / -1, whose cells
    // never go below -10 010.)
    if ((!score_path || plan_.long_mode) && sc_.affine && alg == kAlgNW)       // (the long-read score kernels: the same plain frame)
        lo = (long long)std::min(sc_.open_ref, 0) + (long long)R_ * std::min(sc_.ext_ref, 0) + std::min({sc_.open_read, sc_.open_ref, 0}) +
             std::min(sc_.mismatch, 0);
    if (hi > 32000 || lo < -32000 || (sc_.affine && alg == kAlgNW && lo < -15000))
        throw std::runtime_error("shape x scoring can leave the int16 range of the DP cells (read_length " +
                                 std::to_string(R_) + ", ref_length " + std::to_string(F_) + ")");
}

std::string Engine::host_phases() const {
    char buf[320];
    snprintf(buf, sizeof buf, "{\"host_gather_ms\": %.3f, \"host_wait_ms\": %.3f, \"host_drain_ms\": %.3f, \"host_classify_ms\": %.3f, \"host_launch_ms\": %.3f, \"d2h_row_mb\": %.1f, \"full_row_mb\": %.1f}",
             host_stats_.gather_ms, host_stats_.wait_ms, host_stats_.drain_ms, host_stats_.classify_ms, host_stats_.launch_ms,
             host_stats_.d2h_row_bytes / 1e6, host_stats_.full_row_bytes / 1e6);
    return buf;
}

std::string Engine::describe(int opt, long long n) const {
    const long long ppb = (long long)plan_.pairs_per_wave * plan_.waves_per_block;
    char buf[1600];
    snprintf(buf, sizeof buf,
             "{\"arch\": \"%s\", \"device\": %d, \"alg\": %d, \"affine\": %d, \"group_lanes\": %d, "
             "\"rows_per_lane\": %d, \"padded_rows\": %d, \"pairs_per_wave\": %d, \"waves_per_block\": %d, "
             "\"lds_per_wave\": %d, \"lds_per_block\": %d, \"steps\": %d, \"blocks\": %lld, \"long_mode\": %d, "
             "\"band_width\": %d, \"ragged_batching\": %d, \"ragged_launches\": %d, \"ragged_cell_fraction\": %.4f, "
             "\"score_cells\": \"%s\", \"direct_call\": %d, \"packed_classes\": %d, \"direct_out\": %d, \"band_block_rows\": %d, \"band_col_align\": %d, \"band_waves_per_cu\": %d, \"band_lds_per_wave\": %d, \"band_cells_per_pair\": %lld, \"long_strip_rows\": %d, \"d2h_row_mb\": %.1f, \"full_row_mb\": %.1f, \"host_gather_ms\": %.3f, \"host_classify_ms\": %.3f, \"host_wait_ms\": %.3f, \"host_drain_ms\": %.3f}",
             arch_.c_str(), device_, opt & 0xF, sc_.affine ? 1 : 0, plan_.geo->G, plan_.geo->K,
             plan_.geo->G * plan_.geo->K, plan_.pairs_per_wave, plan_.waves_per_block, plan_.lds.total,
             plan_.lds.total * plan_.waves_per_block, F_ + plan_.geo->G - 1, n > 0 ? (n + ppb - 1) / ppb : 0,
             plan_.long_mode ? 1 : 0, band_width_, ragged_, host_stats_.launches,
             host_stats_.cells_padded > 0 ? host_stats_.cells_swept / host_stats_.cells_padded : 1.0,
             score_cell_format(opt & 0xF), host_stats_.direct, host_stats_.packed, host_stats_.direct_out,
             ((opt & 0xF) == kAlgSW && band_chain_in_use()) ? kBandK : VALIGN_HIP_BAND_BLOCK_ROWS,
             ((opt & 0xF) == kAlgSW && band_chain_in_use()) ? 1 : VALIGN_HIP_BAND_COL_ALIGN, band_blocks_per_cu_, band_lds_, (band_plan_width_ == band_width_ && band_plan_.usable) ? band_plan_.cells : 0ll, long_strip_rows_, host_stats_.d2h_row_bytes / 1e6, host_stats_.full_row_bytes / 1e6, host_stats_.gather_ms, host_stats_.classify_ms, host_stats_.wait_ms,
             host_stats_.drain_ms);
    return buf;
}

void Engine::validate_scoring() {
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

LaunchPlan Engine::choose_plan(int R, int F, int force_g, int force_k, bool latency, bool full_only) const {
    LaunchPlan best;
    double best_cost = 0;
    for (int i = 0; i < kNumGeometries; ++i) {
        const Geometry &g = kGeometries[i];
        if (g.G * g.K < R) continue;
        if (full_only && !g.full) continue;
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
                // (a smaller block for half as many waves again or more: five one-wave blocks were slower than one block of
                // four at 150 x 2 000, 10.9 against 9.7 ms per 131 072 alignments)
                if (resident > best_waves && (best_waves == 0 || 2 * resident >= 3 * best_waves)) {
                    best_waves = resident;
                    p.waves_per_block = wpb;
                }
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
    if (!best.geo || (dbg_.on("force_long") && !force_g)) {
        if (force_g || force_k)
            throw std::runtime_error("the forced kernel geometry does not fit read_length=" + std::to_string(R) +
                                     ", ref_length=" + std::to_string(F));
        return long_plan();
    }
    return best;
}

LaunchPlan Engine::long_plan() {        // row strips + column phases: any length the ABI allows
    LaunchPlan p;
    p.long_mode = true;
    p.pairs_per_wave = 2 * (kWave / kLongG);
    p.waves_per_block = 1;
    p.lds.total = LongLds<kLongG, kLongK>::kTotal;
    for (int i = 0; i < kNumGeometries; ++i)
        if (kGeometries[i].G == kLongG && kGeometries[i].K == kLongK) p.geo = &kGeometries[i];
    return p;
}

long long Engine::whole_rounds(long long pairs) const {
    if (plan_.long_mode || !plan_.geo || cu_count_ <= 0) return pairs;
    const long long waves_per_cu = std::min<long long>(32, (kMaxBlockLds / std::max(1, plan_.lds.total * plan_.waves_per_block)) * plan_.waves_per_block);
    const long long round_pairs = std::max<long long>(1, waves_per_cu) * cu_count_ * plan_.pairs_per_wave;
    return pairs >= round_pairs ? pairs / round_pairs * round_pairs : pairs;
}

bool Engine::direct_call(long long n, size_t per_pair) const {
    return direct_bytes_ > 0 && (size_t)n * per_pair <= direct_bytes_ && n <= staged_pairs_;
}

void Engine::reset_pipeline() {
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

void Engine::release_staging() {
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

void Engine::ensure_staging(long long pairs) {
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

void Engine::build_length_classes() {
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
    // debug switches (VALIGN_HIP_DEBUG): small bins / chunks so that tests reach the pipeline's corners at test sizes
    ragged_min_ = std::max(1ll, dbg_.value("ragged_min", ragged_min_));
    score_chunk_bytes_ = (size_t)std::max(4096ll, dbg_.value("chunk_bytes", (long long)score_chunk_bytes_));
    align_chunk_bytes_ = (size_t)std::max(4096ll, dbg_.value("align_chunk_bytes", (long long)align_chunk_bytes_));
    direct_bytes_ = (size_t)std::max(0ll, dbg_.value("direct_bytes", (long long)direct_bytes_));
}

const LaunchPlan &Engine::class_plan(int R, int F) {
    const std::pair<int, int> key(R, F);
    auto it = class_plans_.find(key);
    if (it == class_plans_.end()) it = class_plans_.emplace(key, choose_plan(R, F, 0, 0)).first;
    return it->second;
}

int Engine::trimmed_length(const unsigned char *s, int len) {
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

}  // namespace valign
