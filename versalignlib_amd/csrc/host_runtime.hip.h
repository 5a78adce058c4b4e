// host_runtime.hip.h -- two host-side helpers of the alignment pipeline that are not the Engine's own state: the thread
// that issues result copies, and the process-wide registry of caller-registered (page-locked) result buffers.
#pragma once

namespace valign {

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
        // Packed rows (align_host): src[0] holds `rows` rows of `row_bytes` bytes packed to their columns from `col` on --
        // col = *d_min (the chunk's smallest readStart, left by the traceback) rounded down to 64, the packing done by
        // compact_rows_kernel before `ready` fired.  rows * (row_bytes - col) bytes cross PCIe as one linear copy; col is
        // left in *col for the host, which unpacks and owns the zeros in front.  d_min null: whole rows (bytes[0]).
        const int *d_min = nullptr;
        int *h_min = nullptr;       // pinned word the column travels through
        int row_bytes = 0;
        long long rows = 0;
        int *col = nullptr;
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
            for (int s = 0; s < kSlots; ++s) issued_[s] = submitted_[s];
            std::rethrow_exception(e);
        }
    }
    void wait_idle() {
        for (int s = 0; s < kSlots; ++s) wait_issued(s);
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
                int first = 0;
                if (job.d_min && job.rows > 0) {
                    hip_check(hipMemcpyAsync(job.h_min, job.d_min, sizeof(int), hipMemcpyDeviceToHost, job.stream), "D2H first column");
                    hip_check(hipStreamSynchronize(job.stream), "hipStreamSynchronize");
                    int c = *job.h_min;
                    c = c < 0 ? 0 : (c > job.row_bytes ? job.row_bytes : c);
                    c &= ~63;                                   // (first_copied_column of trace_kernels.hip.h)
                    *job.col = c;
                    if (c < job.row_bytes)
                        hip_check(hipMemcpyAsync(job.dst[0], job.src[0], (size_t)job.rows * (size_t)(job.row_bytes - c), hipMemcpyDeviceToHost, job.stream),
                                  "D2H result rows (packed columns)");
                    first = 1;
                }
                for (int k = first; k < 2; ++k)
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
    long long submitted_[kSlots] = {}, issued_[kSlots] = {};       // per staging slot of the engine's pipeline (Job::slot)
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

}  // namespace valign
