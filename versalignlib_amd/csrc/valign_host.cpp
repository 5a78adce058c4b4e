// valign_host.cpp -- host side of the versalignLib plugin protocol behind a flat C API.
// See include/valign_host.h for the contract and the reference call sites mirrored.
#include "valign_host.h"
#include "versalign_plugin_abi.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <string.h>

#include <chrono>
#include <stdio.h>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(const std::string &msg) {
    g_error = msg;
    return -1;
}

// Key -> int store injected into the plugin.  Counterpart of the reference host's
// CustomParameters (src/impl/CustomParameters.h): same defaults, same "unknown key"
// behaviour (has_key false; param_int throws a C string).
class HostParameters : public AlignmentParameters {
public:
    HostParameters() {
        values["score_match"] = 2;
        values["score_mismatch"] = -1;
        values["score_gap_read"] = -3;
        values["score_gap_ref"] = -3;
        values["num_threads"] = 1;
    }
    int param_int(char const *const key) override {
        auto it = values.find(key);
        if (it == values.end()) {
            scratch = std::string("Unknown int parameter: ") + key;
            throw scratch.c_str();
        }
        return it->second;
    }
    bool has_key(char const *const key) override { return values.count(key) != 0; }

    std::map<std::string, int> values;

private:
    std::string scratch;
};

// Thread-safe sink (the reference's CustomLogger shares one stringstream and is not).
class HostLogger : public AlignmentLogger {
public:
    void log(int const level, char const *const main, char const *const msg,
             size_t const &arg_num = 0, ...) override {
        const char *sev = "ERROR";
        if (level == 0) sev = "INFO";
        else if (level == 1) sev = "WARNING";
        else if (level == 3) sev = "DRASTIC";
        std::string out = std::string(sev) + "\t[" + (main ? main : "") + "]\t" + (msg ? msg : "") + "\n";
        if (arg_num > 0) {
            va_list args;
            va_start(args, arg_num);
            for (size_t i = 0; i < arg_num; ++i) {
                const char *extra = va_arg(args, const char *);
                out += std::string(sev) + "\t[" + (main ? main : "") + "]\t" + (extra ? extra : "") + "\n";
            }
            va_end(args);
        }
        std::lock_guard<std::mutex> lock(mu);
        if (echo) fputs(out.c_str(), stderr);
        if (lines.size() < (1u << 20)) lines += out;
    }
    std::mutex mu;
    std::string lines;
    bool echo = false;
};

}  // namespace

struct vh_plugin {
    void *dl = nullptr;
    fp_load_alignment_kernel spawn = nullptr;
    fp_delete_alignment_kernel destroy = nullptr;
    void (*set_params)(AlignmentParameters *) = nullptr;
    void (*set_log)(AlignmentLogger *) = nullptr;
    AlignmentKernel *kernel = nullptr;
    HostParameters params;
    HostLogger logger;
    double last_call_seconds = 0.0;      // wall time of the last compute_alignments virtual call
};

namespace {

bool lengths(vh_plugin *p, int *R, int *F) {
    auto r = p->params.values.find("read_length");
    auto f = p->params.values.find("ref_length");
    if (r == p->params.values.end() || f == p->params.values.end()) return false;
    *R = r->second;
    *F = f->second;
    return *R >= 0 && *F >= 0;
}

template <typename Fn>
int guarded(const char *what, Fn &&fn) {
    try {
        fn();
        return 0;
    } catch (const char *msg) {
        return fail(std::string(what) + ": " + (msg ? msg : "(null)"));
    } catch (const std::exception &e) {
        return fail(std::string(what) + ": " + e.what());
    } catch (...) {
        return fail(std::string(what) + ": unknown exception");
    }
}

}  // namespace

extern "C" {

const char *vh_last_error(void) { return g_error.c_str(); }

vh_plugin *vh_open(const char *so_path) {
    void *dl = dlopen(so_path, RTLD_LAZY);
    if (!dl) {
        const char *why = dlerror();
        fail(std::string("dlopen failed: ") + (why ? why : so_path));
        return nullptr;
    }
    vh_plugin *p = new (std::nothrow) vh_plugin();
    if (!p) {
        dlclose(dl);
        fail("out of memory");
        return nullptr;
    }
    p->dl = dl;
    p->spawn = (fp_load_alignment_kernel)dlsym(dl, VERSALIGN_SYM_SPAWN);
    p->destroy = (fp_delete_alignment_kernel)dlsym(dl, VERSALIGN_SYM_DELETE);
    p->set_params = (void (*)(AlignmentParameters *))dlsym(dl, VERSALIGN_SYM_SET_PARAMS);
    p->set_log = (void (*)(AlignmentLogger *))dlsym(dl, VERSALIGN_SYM_SET_LOGGER);
    if (!p->spawn || !p->destroy || !p->set_params || !p->set_log) {
        fail(std::string("plugin lacks one of the four versalignLib symbols: ") + so_path);
        dlclose(dl);
        delete p;
        return nullptr;
    }
    return p;
}

int vh_set_param(vh_plugin *p, const char *key, int value) {
    if (!p || !key) return fail("null argument");
    p->params.values[key] = value;
    return 0;
}

int vh_unset_param(vh_plugin *p, const char *key) {
    if (!p || !key) return fail("null argument");
    p->params.values.erase(key);
    return 0;
}

int vh_reapply_params(vh_plugin *p) {
    if (!p) return fail("null plugin");
    p->set_params(&p->params);
    return 0;
}

int vh_spawn(vh_plugin *p) {
    if (!p) return fail("null plugin");
    if (p->kernel) return fail("kernel already spawned");
    p->set_params(&p->params);
    p->set_log(&p->logger);
    int rc = guarded("spawn_alignment_kernel", [&] { p->kernel = p->spawn(); });
    if (rc == 0 && !p->kernel) return fail("spawn_alignment_kernel returned null");
    return rc;
}

int vh_score(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs,
             int16_t *scores) {
    if (!p || !p->kernel) return fail("kernel not spawned");
    int R, F;
    if (!lengths(p, &R, &F)) return fail("read_length / ref_length not set");
    if (n < 0) return fail("negative pair count");
    std::vector<const char *> rp((size_t)n), fp((size_t)n);
    for (int i = 0; i < n; ++i) {
        rp[i] = (const char *)reads + (size_t)i * R;
        fp[i] = (const char *)refs + (size_t)i * F;
    }
    return guarded("score_alignments",
                   [&] { p->kernel->score_alignments(opt, n, rp.data(), fp.data(), scores); });
}

int vh_score_scattered(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs,
                       int16_t *scores, double *seconds_out) {
    if (!p || !p->kernel) return fail("kernel not spawned");
    int R, F;
    if (!lengths(p, &R, &F)) return fail("read_length / ref_length not set");
    if (n < 0) return fail("negative pair count");
    std::vector<char *> rp((size_t)n), fp((size_t)n);
    for (int i = 0; i < n; ++i) {          // one heap block per sequence, as pad() leaves them
        rp[i] = new char[R > 0 ? R : 1];
        fp[i] = new char[F > 0 ? F : 1];
        memcpy(rp[i], reads + (size_t)i * R, (size_t)R);
        memcpy(fp[i], refs + (size_t)i * F, (size_t)F);
    }
    auto t0 = std::chrono::steady_clock::now();
    int rc = guarded("score_alignments", [&] {
        p->kernel->score_alignments(opt, n, rp.data(), fp.data(), scores);
    });
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    for (int i = 0; i < n; ++i) {
        delete[] rp[i];
        delete[] fp[i];
    }
    return rc;
}

int vh_align(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs,
             uint8_t *rows, int16_t *idx, int normalise) {
    if (!p || !p->kernel) return fail("kernel not spawned");
    int R, F;
    if (!lengths(p, &R, &F)) return fail("read_length / ref_length not set");
    if (n < 0) return fail("negative pair count");
    const int AL = R + F;
    std::vector<const char *> rp((size_t)n), fp((size_t)n);
    for (int i = 0; i < n; ++i) {
        rp[i] = (const char *)reads + (size_t)i * R;
        fp[i] = (const char *)refs + (size_t)i * F;
    }
    Alignment *alns = new Alignment[(size_t)n]();     // value-initialised, as main.cpp:123
    auto t0 = std::chrono::steady_clock::now();
    int rc = guarded("compute_alignments",
                     [&] { p->kernel->compute_alignments(opt, n, rp.data(), fp.data(), alns); });
    p->last_call_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc == 0) {
        for (int i = 0; i < n; ++i) {
            uint8_t *rr = rows + (size_t)i * 2 * AL, *fr = rr + AL;
            const Alignment &a = alns[i];
            if (!a.read || !a.ref) {          // opt with no algorithm: plugin left it untouched
                memset(rr, 0, (size_t)2 * AL);
                idx[4 * i] = idx[4 * i + 1] = idx[4 * i + 2] = idx[4 * i + 3] = 0;
                continue;
            }
            memcpy(rr, a.read, (size_t)AL);
            memcpy(fr, a.ref, (size_t)AL);
            idx[4 * i + 0] = a.readStart;
            idx[4 * i + 1] = a.readEnd;
            idx[4 * i + 2] = a.refStart;
            idx[4 * i + 3] = a.refEnd;
            if (normalise) {
                int s = a.readStart;
                if (s < 0) s = 0;
                if (s > AL) s = AL;
                memset(rr, 0, (size_t)s);
                memset(fr, 0, (size_t)s);
                if (AL > 0) rr[AL - 1] = fr[AL - 1] = 0;
            }
        }
    }
    delete[] alns;
    return rc;
}

double vh_last_call_seconds(vh_plugin *p) { return p ? p->last_call_seconds : 0.0; }

int vh_alloc_probe(int n, int row_bytes, int threads, double *seconds_out) {
    if (n < 0 || row_bytes < 1 || threads < 1 || !seconds_out) return fail("bad argument");
    std::vector<char *> rows((size_t)2 * n);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const size_t lo = (size_t)2 * n * t / threads, hi = (size_t)2 * n * (t + 1) / threads;
            for (size_t i = lo; i < hi; ++i) {
                rows[i] = new char[row_bytes];
                memset(rows[i], (int)(i & 0x7F), (size_t)row_bytes);
            }
        });
    for (auto &th : pool) th.join();
    const auto t1 = std::chrono::steady_clock::now();
    for (char *r : rows) delete[] r;
    const auto t2 = std::chrono::steady_clock::now();
    seconds_out[0] = std::chrono::duration<double>(t1 - t0).count();
    seconds_out[1] = std::chrono::duration<double>(t2 - t1).count();
    return 0;
}

int vh_time_calls(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs, int reps,
                  int align, int free_between, double *seconds_out) {
    if (!p || !p->kernel) return fail("kernel not spawned");
    int R, F;
    if (!lengths(p, &R, &F)) return fail("read_length / ref_length not set");
    if (n < 0 || reps < 1 || !seconds_out) return fail("bad argument");
    std::vector<char *> rp((size_t)n), fp((size_t)n);
    for (int i = 0; i < n; ++i) {          // one heap block per sequence, as pad() leaves them
        rp[i] = new char[R > 0 ? R : 1];
        fp[i] = new char[F > 0 ? F : 1];
        memcpy(rp[i], reads + (size_t)i * R, (size_t)R);
        memcpy(fp[i], refs + (size_t)i * F, (size_t)F);
    }
    std::vector<short> scores((size_t)n);
    std::vector<Alignment *> results;
    double total = 0.0;
    int rc = 0;
    for (int r = 0; r < reps && rc == 0; ++r) {
        Alignment *alns = align ? new Alignment[(size_t)n]() : nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        rc = guarded(align ? "compute_alignments" : "score_alignments", [&] {
            if (align) p->kernel->compute_alignments(opt, n, rp.data(), fp.data(), alns);
            else p->kernel->score_alignments(opt, n, rp.data(), fp.data(), scores.data());
        });
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        seconds_out[1 + r] = sec;
        total += sec;
        if (alns) {
            if (free_between) delete[] alns;          // ~Alignment delete[]s the rows
            else results.push_back(alns);
        }
    }
    seconds_out[0] = total;
    for (Alignment *a : results) delete[] a;
    for (int i = 0; i < n; ++i) {
        delete[] rp[i];
        delete[] fp[i];
    }
    return rc;
}

void vh_close(vh_plugin *p) {
    if (!p) return;
    if (p->kernel) {
        guarded("delete_alignment_kernel", [&] { p->destroy(p->kernel); });
        p->kernel = nullptr;
    }
    // The library stays loaded, as under the reference's host (src/util/versalignUtil.cpp:45-76 never closes a kernel
    // library): a plugin that owns device state, worker threads and registered code objects is not something to map and
    // unmap once per kernel object.
    delete p;
}

int vh_drain_log(vh_plugin *p, char *buf, int cap) {
    if (!p || !buf || cap <= 0) return 0;
    std::lock_guard<std::mutex> lock(p->logger.mu);
    size_t n = p->logger.lines.size();
    if (n > (size_t)cap - 1) n = (size_t)cap - 1;
    memcpy(buf, p->logger.lines.data(), n);
    buf[n] = 0;
    p->logger.lines.clear();
    return (int)n;
}

void vh_log_to_stderr(vh_plugin *p, int on) {
    if (p) p->logger.echo = on != 0;
}

// ---- host data formats ----

int vh_parse_fasta(const char *path, char **blob, int *count) {
    if (!path || !blob || !count) return fail("null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(std::string("cannot open ") + path);
    std::string text;
    char chunk[1 << 16];
    for (size_t got; (got = fread(chunk, 1, sizeof chunk, fp)) > 0;) text.append(chunk, got);
    fclose(fp);
    // One pass over the file image with a three-state record machine.  What is accepted is what the
    // reference host accepts (src/util/versalignUtil.h:52-92, exercised by tests/test_host_and_abi.py):
    // a header is '>' plus at least one byte; the record's sequence lines are joined until the next
    // '>' line or an empty line; a sequence line with a blank in it voids the whole record; a final
    // line without '\n' is not part of the file; a sequence ends at its first NUL byte.
    enum { kOutside, kInRecord, kVoided } state = kOutside;
    std::string packed;                  // the accepted sequences, each followed by its NUL
    size_t record_at = 0;                // where the open record's sequence starts in `packed`
    int records = 0;
    auto close_record = [&] {
        if (state == kInRecord) {
            const size_t nul = packed.find('\0', record_at);
            if (nul != std::string::npos) packed.resize(nul);
            packed.push_back('\0');
            ++records;
        } else {
            packed.resize(record_at);
        }
        state = kOutside;
    };
    const char *cursor = text.data(), *const stop = cursor + text.size();
    while (cursor < stop) {
        const char *nl = (const char *)memchr(cursor, '\n', (size_t)(stop - cursor));
        if (!nl) break;
        const size_t len = (size_t)(nl - cursor);
        if (len == 0 || cursor[0] == '>') {
            if (state != kOutside) close_record();
            record_at = packed.size();
            if (len > 1) state = kInRecord;           // (a bare ">" names nothing: its lines are skipped)
        } else if (state == kInRecord) {
            if (memchr(cursor, ' ', len)) {
                packed.resize(record_at);
                state = kVoided;
            } else {
                packed.append(cursor, len);
            }
        }
        cursor = nl + 1;
    }
    if (state != kOutside) close_record();
    char *out = (char *)malloc(packed.size() ? packed.size() : 1);
    if (!out) return fail("out of memory");
    memcpy(out, packed.data(), packed.size());
    *blob = out;
    *count = records;
    return 0;
}

int vh_pad(const char *blob, int count, char fill, uint8_t **out, int *length) {
    if (!blob || !out || !length || count < 0) return fail("bad argument");
    size_t longest = 0;
    const char *s = blob;
    for (int i = 0; i < count; ++i) {
        size_t len = strlen(s);
        if (len > longest) longest = len;
        s += len + 1;
    }
    uint8_t *buf = (uint8_t *)malloc(longest > 0 && count > 0 ? longest * (size_t)count : 1);
    if (!buf) return fail("out of memory");
    s = blob;
    for (int i = 0; i < count; ++i) {
        size_t len = strlen(s);
        uint8_t *dst = buf + (size_t)i * longest;
        memcpy(dst, s, len);
        memset(dst + len, (unsigned char)fill, longest - len);
        s += len + 1;
    }
    *out = buf;
    *length = (int)longest;
    return 0;
}

int vh_cigar(const uint8_t *read_row, const uint8_t *ref_row, int start, int end, int extended, char *buf,
             int cap) {
    if (!read_row || !ref_row || !buf || cap <= 0 || start < 0 || end < start) {
        fail("bad argument");
        return -1;
    }
    int len = 0, run = 0;
    char op = 0;
    auto flush = [&]() -> bool {
        if (run == 0) return true;
        char tmp[16];
        const int w = snprintf(tmp, sizeof tmp, "%d%c", run, op);
        if (len + w >= cap) return false;
        memcpy(buf + len, tmp, (size_t)w);
        len += w;
        run = 0;
        return true;
    };
    for (int c = start; c < end; ++c) {
        const uint8_t a = read_row[c], b = ref_row[c];
        if (a == 0 && b == 0) break;                      // the terminating NUL column
        char now;
        if (a == '-' && b != '-') now = 'D';
        else if (b == '-' && a != '-') now = 'I';
        else if (!extended) now = 'M';
        else now = ((a | 0x20) == (b | 0x20)) ? '=' : 'X';
        if (now != op && !flush()) {
            fail("CIGAR buffer too small");
            return -1;
        }
        op = now;
        ++run;
    }
    if (!flush()) {
        fail("CIGAR buffer too small");
        return -1;
    }
    buf[len] = 0;
    return len;
}

void vh_free(void *ptr) { free(ptr); }

}  // extern "C"
