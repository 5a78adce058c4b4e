// hip_plugin.hip -- libHIPKernel.so: the versalignLib plugin boundary over the HIP engine.
//
// Exports the four C symbols every versalignLib backend exports
// (src/Kernels/default/DefaultKernel_dllexport.cpp:18-42) plus the flat C API of
// include/valign_hip.h.  There is deliberately no CPU path: without a gfx950 device
// spawn_alignment_kernel() fails loudly.
#include "valign_hip.h"
#include "versalign_plugin_abi.h"

#include <atomic>
#include <dlfcn.h>
#include <exception>
#include <malloc.h>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "engine.hip.h"

#define VALIGN_EXPORT __attribute__((visibility("default")))

// Plugin-private globals the ABI headers declare (AlignmentParameters.h:21, AlignmentLogger.h:21).
VALIGN_EXPORT AlignmentParameters *_parameters = 0;
VALIGN_EXPORT AlignmentLogger *_logger = 0;

namespace {

const char *const kModule = "HIP";
thread_local std::string g_last_error;

int opt_param(const char *key, int fallback) {
    return Parameters.has_key(key) ? Parameters.param_int(key) : fallback;
}

void log_line(int level, const std::string &msg) {
    if (_logger) Logger.log(level, kModule, msg.c_str());
}

// hip_devices_allgather = 1: the exchange step BASELINE.json's north star names, inside ONE process -- every device
// keeps its shard's scores in HBM, an RCCL all-gather over xGMI (ncclCommInitAll: one communicator per device, all
// driven by this process inside ncclGroupStart / ncclGroupEnd) leaves the whole score vector on every device, and
// the host copy comes from the first one.  The reference has no counterpart (its kernels are single-device,
// DefaultKernel.cpp:45-48).  RCCL is loaded with dlopen when the key is set: the plugin does not link it, hosts that never
// ask for the collective never pay for it.  RCCL has no int16 type: scores travel as bytes.
class ShardGather {
public:
    ShardGather(const std::vector<int> &devices) : devices_(devices) {
        for (size_t i = 0; i < devices.size(); ++i)
            for (size_t k = 0; k < i; ++k)
                if (devices[i] == devices[k])
                    throw std::runtime_error("hip_devices_allgather needs one distinct device per shard (device " +
                                             std::to_string(devices[i]) + " holds two): RCCL refuses duplicate ranks");
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib_) break;
        }
        if (!lib_) throw std::runtime_error(std::string("hip_devices_allgather: cannot load librccl.so (") + dlerror() + ")");
        init_all_ = (int (*)(void **, int, const int *))dlsym(lib_, "ncclCommInitAll");
        destroy_ = (int (*)(void *))dlsym(lib_, "ncclCommDestroy");
        group_start_ = (int (*)())dlsym(lib_, "ncclGroupStart");
        group_end_ = (int (*)())dlsym(lib_, "ncclGroupEnd");
        all_gather_ = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(lib_, "ncclAllGather");
        error_string_ = (const char *(*)(int))dlsym(lib_, "ncclGetErrorString");
        if (!init_all_ || !destroy_ || !group_start_ || !group_end_ || !all_gather_ || !error_string_)
            throw std::runtime_error("hip_devices_allgather: librccl.so lacks an expected symbol");
        comms_.assign(devices.size(), nullptr);
        check(init_all_(comms_.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
        streams_.assign(devices.size(), nullptr);
        for (size_t i = 0; i < devices.size(); ++i) {
            valign::hip_check(hipSetDevice(devices[i]), "hipSetDevice");
            valign::hip_check(hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking), "hipStreamCreate");
        }
        send_.assign(devices.size(), nullptr);
        recv_.assign(devices.size(), nullptr);
    }
    ~ShardGather() {
        for (size_t i = 0; i < devices_.size(); ++i) {
            (void)hipSetDevice(devices_[i]);
            if (i < send_.size() && send_[i]) (void)hipFree(send_[i]);
            if (i < recv_.size() && recv_[i]) (void)hipFree(recv_[i]);
            if (i < streams_.size() && streams_[i]) (void)hipStreamDestroy(streams_[i]);
            if (i < comms_.size() && comms_[i] && destroy_) (void)destroy_(comms_[i]);
        }
        // (librccl.so stays loaded: a collective library with service threads is not something to unmap under them)
    }
    // shard buffer of device i: `per` scores (the all-gather needs equal counts: the last shard's tail is padding)
    void reserve(int per) { ensure(per); }                                   // (before the shard threads start)
    int16_t *shard(int i) const { return send_[(size_t)i]; }
    // after every shard's kernels have finished: gather on all devices, copy the first n scores out of device 0's vector
    void gather_to_host(int per, int n, short *scores) {
        check(group_start_(), "ncclGroupStart");
        for (size_t i = 0; i < devices_.size(); ++i)
            check(all_gather_(send_[i], recv_[i], (size_t)per * 2, 0 /* ncclInt8 */, comms_[i], streams_[i]), "ncclAllGather");
        check(group_end_(), "ncclGroupEnd");
        valign::hip_check(hipSetDevice(devices_[0]), "hipSetDevice");
        valign::hip_check(hipMemcpyAsync(scores, recv_[0], sizeof(short) * (size_t)n, hipMemcpyDeviceToHost, streams_[0]), "D2H gathered scores");
        for (size_t i = 0; i < devices_.size(); ++i) {
            valign::hip_check(hipSetDevice(devices_[i]), "hipSetDevice");
            valign::hip_check(hipStreamSynchronize(streams_[i]), "hipStreamSynchronize");
        }
    }
    // (tests) the gathered vector of device i, first n scores
    void copy_gathered(int i, int n, short *out) {
        valign::hip_check(hipSetDevice(devices_[(size_t)i]), "hipSetDevice");
        valign::hip_check(hipMemcpy(out, recv_[(size_t)i], sizeof(short) * (size_t)n, hipMemcpyDeviceToHost), "D2H");
    }

private:
    void check(int rc, const char *what) {
        if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (error_string_ ? error_string_(rc) : "RCCL error"));
    }
    void ensure(int per) {
        if (per <= cap_) return;
        for (size_t i = 0; i < devices_.size(); ++i) {
            valign::hip_check(hipSetDevice(devices_[i]), "hipSetDevice");
            if (send_[i]) (void)hipFree(send_[i]);
            if (recv_[i]) (void)hipFree(recv_[i]);
            send_[i] = recv_[i] = nullptr;
            valign::hip_check(hipMalloc((void **)&send_[i], sizeof(short) * (size_t)per), "hipMalloc(shard scores)");
            valign::hip_check(hipMalloc((void **)&recv_[i], sizeof(short) * (size_t)per * devices_.size()), "hipMalloc(gathered scores)");
            valign::hip_check(hipMemset(send_[i], 0, sizeof(short) * (size_t)per), "hipMemset");
        }
        cap_ = per;
    }
    std::vector<int> devices_;
    void *lib_ = nullptr;
    int (*init_all_)(void **, int, const int *) = nullptr;
    int (*destroy_)(void *) = nullptr;
    int (*group_start_)() = nullptr;
    int (*group_end_)() = nullptr;
    int (*all_gather_)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*error_string_)(int) = nullptr;
    std::vector<void *> comms_;
    std::vector<hipStream_t> streams_;
    std::vector<int16_t *> send_, recv_;
    int cap_ = 0;
};

// hip_devices = N: the one rule by which a call's pairs are cut into contiguous shards -- `per` pairs each (the last one
// short, trailing shards empty when n < N * per) -- used by the shard threads AND by the gather path's buffer offsets.
struct ShardSplit {
    int per;
    ShardSplit(int n, int shards) : per(shards > 0 ? (n + shards - 1) / shards : n) {}
    int begin(int d) const { return d * per; }
    int count(int d, int n) const {
        const int b = begin(d);
        return b < n ? (n - b < per ? n - b : per) : 0;
    }
    int shard_of(int first_pair) const { return per > 0 ? first_pair / per : 0; }
};

// The kernel object handed to the host.  Reads the same six required keys as every
// reference backend at construction (DefaultKernel.h:70-81) and num_threads per call
// (DefaultKernel.cpp:45); the latter sizes the host gather pool here.
class HIPKernel : public AlignmentKernel {
public:
    HIPKernel() {
        if (!_parameters) throw "Cannot instantiate Kernel. Lacking parameters";
        static const char *const required[] = {"score_match", "score_mismatch", "score_gap_read",
                                               "score_gap_ref", "read_length", "ref_length"};
        for (const char *key : required)
            if (!Parameters.has_key(key)) throw "Cannot instantiate Kernel. Lacking parameters";
        valign::Scoring sc;
        sc.match = Parameters.param_int("score_match");
        sc.mismatch = Parameters.param_int("score_mismatch");
        sc.gap_read = Parameters.param_int("score_gap_read");
        sc.gap_ref = Parameters.param_int("score_gap_ref");
        const int R = Parameters.param_int("read_length");
        const int F = Parameters.param_int("ref_length");
        const bool any_affine = Parameters.has_key("score_gap_open_read") || Parameters.has_key("score_gap_extend_read") ||
                                Parameters.has_key("score_gap_open_ref") || Parameters.has_key("score_gap_extend_ref");
        if (any_affine) {
            sc.affine = true;
            sc.open_read = opt_param("score_gap_open_read", sc.gap_read);
            sc.ext_read = opt_param("score_gap_extend_read", sc.gap_read);
            sc.open_ref = opt_param("score_gap_open_ref", sc.gap_ref);
            sc.ext_ref = opt_param("score_gap_extend_ref", sc.gap_ref);
        } else {
            sc.open_read = sc.ext_read = sc.gap_read;
            sc.open_ref = sc.ext_ref = sc.gap_ref;
        }
        try {
            // hip_devices = N: the pairs of every call are split into N contiguous shards, one device each
            // (hip_device, hip_device + 1, ...), each shard on its own host thread with its own streams and
            // staging.  Results land in the caller's host arrays, so no collective is needed inside one process.
            const int shards = opt_param("hip_devices", 1);
            if (shards < 1 || shards > 64) throw std::runtime_error("hip_devices must be 1..64");
            const int first = opt_param("hip_device", 0), visible = valign_hip_device_count();
            if (shards > 1 && first + shards > visible) {
                // More shards than devices: several shards share a device (each with its own engine).  Results are
                // the same, the speed-up is not there -- a host that asked for N devices should hear about it.
                if (opt_param("hip_devices_strict", 0) != 0)
                    throw std::runtime_error("hip_devices = " + std::to_string(shards) + " from device " + std::to_string(first) +
                                             " but only " + std::to_string(visible) + " visible (hip_devices_strict)");
                log_line(1, "hip_devices = " + std::to_string(shards) + " from device " + std::to_string(first) + " but only " +
                                std::to_string(visible) + " visible: shards are folded onto the visible devices (same results, no "
                                "speed-up; hip_devices_strict = 1 refuses instead)");
            }
            for (int d = 0; d < shards; ++d) {
                std::unique_ptr<valign::Engine> e(new valign::Engine((first + d) % (visible > 0 ? visible : 1), R, F, sc,
                                                                     opt_param("hip_group_lanes", 0), opt_param("hip_rows_per_lane", 0)));
                e->set_traceback_policy(opt_param("traceback_policy", 0));
                e->set_band_width(opt_param("band_width", 0));
                e->set_score_width(opt_param("score_width", 0));
                e->set_ragged_batching(opt_param("ragged_batching", 0));
                // pointer scratch of compute_alignments (device memory, internal): capped at 64 GiB / half the free HBM
                // unless the host says otherwise -- larger batches simply run in more chunks
                e->set_pointer_scratch_cap_mb(opt_param("pointer_scratch_cap_mb", 0));
                if (Parameters.has_key("host_packing")) e->set_host_packing(Parameters.param_int("host_packing"));
                if (Parameters.has_key("half_float_cells")) e->set_half_float_cells(Parameters.param_int("half_float_cells"));
                if (d == 0) engine_ = std::move(e);
                else more_.push_back(std::move(e));
            }
            if (opt_param("hip_devices_allgather", 0) != 0) {
                std::vector<int> devs{engine_->device()};
                for (auto &e : more_) devs.push_back(e->device());
                gather_.reset(new ShardGather(devs));
                log_line(0, "hip_devices_allgather = 1: RCCL all-gather of the per-shard scores over " + std::to_string(devs.size()) +
                                " device(s), host copy from device " + std::to_string(devs[0]));
            }
            if (shards > 1) {
                std::string where = std::to_string(engine_->device());
                for (auto &e : more_) where += ", " + std::to_string(e->device());
                log_line(0, "hip_devices = " + std::to_string(shards) + ": shards on devices [" + where + "] of " +
                                std::to_string(visible) + " visible");
            }
            // compute_alignments must hand out 2n operator new[] blocks (the caller delete[]s them,
            // include/AlignmentKernel.h:20-23) -- 1.4 GB per million pairs of 150 x 500, allocated by the scatter threads.
            // glibc grows a thread arena in steps of M_TOP_PAD (128 KB by default), each one an mprotect() under the
            // process's mm lock: with 16 threads that is what the call costs (650-850 ms per million pairs, the bare
            // allocation loop without any plugin included -- tools/microbench/host_alloc.cpp; 45-60 ms with 256 MB steps).
            // That is the HOST's allocator and the host's decision: by default (0) the plugin touches nothing global --
            // INTEGRATION.md section 0 tells hosts to start with MALLOC_TOP_PAD_=268435456 in their environment.  Only a
            // host that sets the key itself gets the mallopt() call from here: 2 = M_TOP_PAD 256 MB (address space is
            // reserved in larger steps, pages are still committed on first touch and returned on free), 1 = that + never
            // trim (freed rows stay with the process for the next call).
            const int tuning = opt_param("host_malloc_tuning", 0);
            if (tuning < 0 || tuning > 2) throw std::runtime_error("host_malloc_tuning must be 0, 1 or 2");
            if (tuning == 1) mallopt(M_TRIM_THRESHOLD, 0x7FFFFFFF);
            if (tuning >= 1) {
                mallopt(M_TOP_PAD, 256 << 20);
                // a process-wide setting of the HOST's allocator: said out loud (WARNING level) the first time
                static std::atomic<bool> told{false};
                log_line(told.exchange(true) ? 0 : 1, std::string("host_malloc_tuning = ") + std::to_string(tuning) +
                                ": mallopt(M_TOP_PAD, 256 MB)" + (tuning == 1 ? " + M_TRIM_THRESHOLD off" : "") +
                                " for the result rows of compute_alignments (the host asked for it; 0, the default, leaves the host's allocator alone)");
            }
        } catch (const std::exception &e) {
            what_ = std::string("Cannot instantiate Kernel. ") + e.what();
            log_line(3, what_);
            static thread_local std::string thrown;
            thrown = what_;
            throw thrown.c_str();          // hosts catch `const char *` (reference convention)
        }
#ifndef NDEBUG
        log_line(0, "Successfully instantiated HIP Kernel.");
#endif
    }

    ~HIPKernel() override {}

    void score_alignments(int const &opt, int const &aln_number, char const *const *const reads,
                          char const *const *const refs, short *const scores) override {
        if ((opt & 0xF) > 1) return;       // unsupported mode: silent no-op, like the reference
        const int threads = Parameters.has_key("num_threads") ? Parameters.param_int("num_threads") : 1;
        log_line(0, "Running HIPKernel score with " + std::to_string(threads) + " host threads on " +
                        engine_->describe(opt, aln_number));
        try {
            const int shards = 1 + (int)more_.size();
            if (gather_ && aln_number >= shards) {
                // every device keeps its shard in HBM; the RCCL all-gather puts the whole vector on each of them
                const ShardSplit split(aln_number, shards);
                const int per = split.per;
                gather_->reserve(per);
                sharded(aln_number, threads, [&](valign::Engine &e, int begin, int count, int th) {
                    e.score_host(opt, count, reads + begin, refs + begin, nullptr, th, gather_->shard(split.shard_of(begin)));
                });
                gather_->gather_to_host(per, aln_number, scores);
                log_line(0, "HIPKernel score done (RCCL all-gather of " + std::to_string(shards) + " shard(s) of " + std::to_string(per) +
                                " scores), host phases " + engine_->host_phases());
                return;
            }
            sharded(aln_number, threads, [&](valign::Engine &e, int begin, int count, int th) {
                e.score_host(opt, count, reads + begin, refs + begin, scores + begin, th);
            });
            log_line(0, "HIPKernel score done, host phases " + engine_->host_phases());
        } catch (const std::exception &e) {
            rethrow_for_host(e.what());
        }
    }

    void compute_alignments(int const &opt, int const &aln_number, char const *const *const reads,
                            char const *const *const refs, Alignment *const alignments) override {
        if ((opt & 0xF) > 1) return;
        const int threads = Parameters.has_key("num_threads") ? Parameters.param_int("num_threads") : 1;
        log_line(0, "Running HIPKernel align with " + std::to_string(threads) + " host threads on " +
                        engine_->describe(opt, aln_number));
        try {
            sharded(aln_number, threads, [&](valign::Engine &e, int begin, int count, int th) {
                e.align_host(opt, count, reads + begin, refs + begin, alignments + begin, th);
            });
            log_line(0, "HIPKernel align done, host phases " + engine_->host_phases());
        } catch (const std::exception &e) {
            rethrow_for_host(e.what());
        }
    }

private:
    // Errors leave the virtuals the way the reference's kernels raise theirs: logged at level 3, then thrown
    // as `const char *` (DefaultKernel.h:79-81) -- reference-style hosts catch nothing else, a std::exception
    // crossing the plugin boundary would end in std::terminate.
    [[noreturn]] void rethrow_for_host(const char *what) {
        log_line(3, what);
        static thread_local std::string thrown;
        thrown = what;
        throw thrown.c_str();
    }

    // fn(engine, first pair, pairs, host threads) once per device shard, concurrently; the first error wins
    template <typename Fn>
    void sharded(int n, int threads, Fn fn) {
        const int shards = 1 + (int)more_.size();
        if (shards == 1 || n < shards) {
            fn(*engine_, 0, n, threads);
            return;
        }
        const ShardSplit split(n, shards);
        const int th = threads / shards > 0 ? threads / shards : 1;
        std::vector<std::exception_ptr> errors((size_t)shards);
        std::vector<std::thread> workers;
        for (int d = 0; d < shards; ++d) {
            const int begin = split.begin(d), count = split.count(d, n);
            if (count <= 0) continue;
            valign::Engine *e = d == 0 ? engine_.get() : more_[(size_t)d - 1].get();
            workers.emplace_back([&, e, begin, count, d] {
                try {
                    fn(*e, begin, count, th);
                } catch (...) {
                    errors[(size_t)d] = std::current_exception();
                }
            });
        }
        for (auto &w : workers) w.join();
        for (auto &err : errors)
            if (err) std::rethrow_exception(err);
    }

    std::unique_ptr<valign::Engine> engine_;
    std::vector<std::unique_ptr<valign::Engine>> more_;      // hip_devices > 1: one engine per further device
    std::unique_ptr<ShardGather> gather_;                    // hip_devices_allgather = 1
    std::string what_;
};

template <typename Fn>
int flat_guard(Fn &&fn) {
    try {
        fn();
        return 0;
    } catch (const std::exception &e) {
        g_last_error = e.what();
    } catch (const char *msg) {
        g_last_error = msg ? msg : "error";
    } catch (...) {
        g_last_error = "unknown error";
    }
    return 1;
}

}  // namespace

struct valign_hip_engine {
    std::unique_ptr<valign::Engine> impl;
};

extern "C" {

VALIGN_EXPORT AlignmentKernel *spawn_alignment_kernel() {
    AlignmentKernel *kernel = new HIPKernel();
    return kernel;
}

VALIGN_EXPORT void set_parameters(AlignmentParameters *parameters) { _parameters = parameters; }

VALIGN_EXPORT void set_logger(AlignmentLogger *logger) { _logger = logger; }

VALIGN_EXPORT void delete_alignment_kernel(AlignmentKernel *instance) {
    if (instance != 0) delete instance;
}

VALIGN_EXPORT const char *valign_hip_last_error(void) { return g_last_error.c_str(); }

VALIGN_EXPORT int valign_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

// (for tests, no device needed) shard d of a call of n pairs under hip_devices = shards: first pair and pair count
VALIGN_EXPORT int valign_hip_shard_range(int n, int shards, int d, int *begin, int *count) {
    if (n < 0 || shards < 1 || d < 0 || d >= shards || !begin || !count) return 1;
    const ShardSplit split(n, shards);
    *begin = split.begin(d);
    *count = split.count(d, n);
    return 0;
}

VALIGN_EXPORT int valign_hip_engine_create(int device, int read_length, int ref_length,
                                           const valign_hip_scoring *s, int force_group_lanes,
                                           int force_rows_per_lane, valign_hip_engine **out) {
    if (!s || !out) {
        g_last_error = "null argument";
        return 1;
    }
    return flat_guard([&] {
        valign::Scoring sc;
        sc.match = s->match;
        sc.mismatch = s->mismatch;
        sc.gap_read = s->gap_read;
        sc.gap_ref = s->gap_ref;
        sc.affine = s->affine != 0;
        sc.open_read = sc.affine ? s->open_read : s->gap_read;
        sc.ext_read = sc.affine ? s->ext_read : s->gap_read;
        sc.open_ref = sc.affine ? s->open_ref : s->gap_ref;
        sc.ext_ref = sc.affine ? s->ext_ref : s->gap_ref;
        std::unique_ptr<valign_hip_engine> e(new valign_hip_engine());
        e->impl.reset(new valign::Engine(device, read_length, ref_length, sc, force_group_lanes, force_rows_per_lane));
        *out = e.release();
    });
}

VALIGN_EXPORT void valign_hip_engine_destroy(valign_hip_engine *e) { delete e; }

VALIGN_EXPORT int valign_hip_set_band_width(valign_hip_engine *e, int diagonals) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_band_width(diagonals); });
}

VALIGN_EXPORT int valign_hip_set_score_width(valign_hip_engine *e, int bits) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_score_width(bits); });
}

VALIGN_EXPORT int valign_hip_set_ragged_batching(valign_hip_engine *e, int mode) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_ragged_batching(mode); });
}

VALIGN_EXPORT int valign_hip_host_register(void *ptr, unsigned long long bytes) {
    return flat_guard([&] { valign::HostRegistry::instance().add(ptr, (size_t)bytes); });
}

VALIGN_EXPORT int valign_hip_host_unregister(void *ptr) {
    return flat_guard([&] { valign::HostRegistry::instance().remove(ptr); });
}

VALIGN_EXPORT int valign_hip_set_host_packing(valign_hip_engine *e, int mode) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_host_packing(mode); });
}

VALIGN_EXPORT int valign_hip_set_half_float_cells(valign_hip_engine *e, int mode) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_half_float_cells(mode); });
}

VALIGN_EXPORT int valign_hip_set_pointer_scratch_cap_mb(valign_hip_engine *e, long long mb) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_pointer_scratch_cap_mb(mb); });
}

VALIGN_EXPORT int valign_hip_set_traceback_policy(valign_hip_engine *e, int policy) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->set_traceback_policy(policy); });
}

VALIGN_EXPORT int valign_hip_score_device(valign_hip_engine *e, int opt, long long n, const void *d_reads,
                                          const void *d_refs, void *d_scores, void *hip_stream) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] {
        e->impl->score_device(opt, n, (const uint8_t *)d_reads, (const uint8_t *)d_refs, (int16_t *)d_scores,
                              (hipStream_t)hip_stream);
    });
}

VALIGN_EXPORT int valign_hip_align_device(valign_hip_engine *e, int opt, long long n, const void *d_reads,
                                          const void *d_refs, void *d_rows, void *d_idx, void *hip_stream) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] {
        e->impl->align_device(opt, n, (const uint8_t *)d_reads, (const uint8_t *)d_refs, (uint8_t *)d_rows,
                              (short *)d_idx, (hipStream_t)hip_stream);
    });
}

VALIGN_EXPORT int valign_hip_score_host(valign_hip_engine *e, int opt, int n, const char *const *reads,
                                        const char *const *refs, short *scores, int threads) {
    if (!e) {
        g_last_error = "null engine";
        return 1;
    }
    return flat_guard([&] { e->impl->score_host(opt, n, reads, refs, scores, threads); });
}

VALIGN_EXPORT int valign_hip_align_host(valign_hip_engine *e, int opt, int n, const char *const *reads,
                                        const char *const *refs, void *rows, short *idx, int threads) {
    if (!e || !rows || !idx) {
        g_last_error = "null argument";
        return 1;
    }
    return flat_guard([&] {
        valign::Engine::FlatSink sink{(uint8_t *)rows, idx, (size_t)e->impl->read_length() + (size_t)e->impl->ref_length()};
        e->impl->align_host(opt, n, reads, refs, sink, threads);
    });
}

VALIGN_EXPORT int valign_hip_describe(valign_hip_engine *e, int opt, long long n, char *buf, int cap) {
    if (!e || !buf || cap <= 0) return 1;
    const std::string s = e->impl->describe(opt, n);
    snprintf(buf, (size_t)cap, "%s", s.c_str());
    return 0;
}

}  // extern "C"
