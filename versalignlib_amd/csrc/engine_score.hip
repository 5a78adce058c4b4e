// engine_score.hip -- Engine: score_alignments (reference: src/Kernels/default/DefaultKernel.cpp:52-202) -- the register-sweep
// launches, the host-pointer chunk pipeline, 4-bit class unpacking and length-sorted batches.  The non-template kernels of
// pack_kernels.hip.h / ragged_kernels.hip.h are defined in this translation unit.
#define VALIGN_TU_SCORE 1
#include "engine.hip.h"

namespace valign {

void Engine::score_device(int opt, long long n, const uint8_t *d_reads, const uint8_t *d_refs,
                  int16_t *d_scores, hipStream_t stream, bool length_sorted) {
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
        // One context (pinned histogram / tables, counters, packed buffers) serves the device-resident calls of this
        // engine: a call on ANOTHER stream than the last one first waits, on the host, until that one's kernels and
        // table copies are through -- otherwise it would rewrite the pinned tables under them.  Calls on one stream are
        // ordered by the stream.
        if (ragged_dev_done_ && ragged_dev_stream_ != stream)
            hip_check(hipEventSynchronize(ragged_dev_done_), "hipEventSynchronize(previous length-sorted call)");
        if (!ragged_dev_done_) hip_check(hipEventCreateWithFlags(&ragged_dev_done_, hipEventDisableTiming), "hipEventCreate");
        ragged_dev_stream_ = stream;
        ragged_begin(kSlots, n, d_reads, d_refs, stream);           // (a context of its own: the pipeline's slots may be busy on the engine's streams)
        const bool swept = ragged_finish(kSlots, alg, n, d_scores, stream, ragged_ == 2);
        hip_check(hipEventRecord(ragged_dev_done_, stream), "hipEventRecord");
        if (swept) return;
    }
    // a batch that leaves most SIMDs with at most one wave is over when its slowest wave is: shortest sweep
    const bool few = n <= (long long)latency_plan_.pairs_per_wave * 1024 && band_width_ == 0;
    launch_score(few ? latency_plan_ : plan_, alg, R_, F_, n, d_reads, d_refs, d_scores, stream);
}

void Engine::launch_score(const LaunchPlan &plan, int alg, int R, int F, long long n, const uint8_t *d_reads,
                  const uint8_t *d_refs, int16_t *d_scores, hipStream_t stream,
                  const LengthGroup *groups, int n_groups) {
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

void Engine::launch_unpack(const uint8_t *d_packed, uint8_t *d_out, long long n, int len, hipStream_t stream) {
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

void Engine::score_host(int opt, int n, const char *const *reads, const char *const *refs, short *scores,
                int threads, int16_t *d_dest) {
    const int alg = opt & 0xF;
    if (alg > 1 || n <= 0) return;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const size_t per_pair = (size_t)R_ + F_;
    // Long reads on row strips: their launches follow one another on one stream (the strips' boundary rows are one scratch) and
    // a 48 MB chunk of 10 kbp pairs is 1 200 waves for 3 500 resident ones -- each launch runs at a third of the device.
    // Chunks of up to 192 MB there (the banded block chain needs no scratch: its small chunks run side by side instead).
    const bool wide_cells = score_width_ == 32 || (score_width_ == 0 && !int16_range_ok(alg));
    const bool scratch_free = (band_width_ > 0 && alg == kAlgSW && band_chain_in_use()) || (plan_.long_mode && long_single_strip(wide_cells));
    const bool strips_in_turn = plan_.long_mode && !scratch_free;
    const size_t chunk_bytes = strips_in_turn && !dbg_.on("chunk_bytes") ? std::max<size_t>(score_chunk_bytes_, 192u << 20) : score_chunk_bytes_;
    long long chunk = per_pair ? (long long)(chunk_bytes / per_pair) : n;
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
    // (the banded block chain keeps nothing in HBM between its steps: its chunks may run side by side on the slots' streams --
    // a chunk of 2 400 pairs of 10 kbp fills 600 of the 4 096 resident waves and takes a launch's latency whatever its size)
    // ... and so do the single-strip instances of the long-read kernel (short reads against a long reference)
    const bool shared_scratch = !scratch_free && (plan_.long_mode || score_width_ == 32 || !int16_range_ok(alg));
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
        const long long ramp = (n > 2 * chunk) ? (chunk_no == 0 ? whole_rounds(chunk / 4) : (chunk_no == 1 ? whole_rounds(chunk / 2) : chunk)) : chunk;
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

bool Engine::ragged_applies(int alg) const {
    return ragged_ && alg <= kAlgNW && !plan_.long_mode && !force_g_ && !force_k_ && score_width_ != 32 &&
           R_ > 0 && F_ > 0 && int16_range_ok(alg);
}

double Engine::sampled_cell_fraction(const char *const *reads, const char *const *refs, long long n) const {
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

void Engine::ensure_ragged(int c, long long n) {
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

void Engine::release_ragged() {
    if (ragged_dev_done_) (void)hipEventDestroy(ragged_dev_done_);
    ragged_dev_done_ = nullptr;
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

void Engine::ragged_begin(int c, long long n, const uint8_t *d_reads, const uint8_t *d_refs, hipStream_t stream) {
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

std::vector<Engine::LengthGroup> Engine::fold_groups(std::vector<long long> &total, std::vector<int> &group_of_bin) const {
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

bool Engine::ragged_finish(int c, int alg, long long n, int16_t *d_scores, hipStream_t stream, bool always) {
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

}  // namespace valign
