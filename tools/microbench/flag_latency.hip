// How long does the host wait for a tiny kernel: hipStreamSynchronize / hipEventSynchronize vs spinning on a flag the
// kernel writes into pinned host memory (results land there anyway in the plugin's small-call path).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/flag_latency.hip -o tools/microbench/bin/flag_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <vector>

__global__ void tiny(volatile unsigned *flag, unsigned *counter, unsigned epoch, int spin) {
    // a little dependent work, like a short sweep
    unsigned x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0xFFFFFFFFu) counter[1] = x;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(counter, 1u);
        if (old == gridDim.x - 1) {
            counter[0] = 0;
            __threadfence_system();
            if (flag) *flag = epoch;
        }
    }
}

static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
}

int main() {
    unsigned *flag, *counter;
    hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
    hipMalloc((void **)&counter, 64);
    hipMemset(counter, 0, 64);
    *flag = 0;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int blocks : {1, 63, 250}) {
        for (int spin : {0, 2000}) {
            std::vector<double> a, b, c;
            unsigned epoch = 0;
            for (int it = 0; it < 300; ++it) {
                auto t0 = std::chrono::steady_clock::now();
                tiny<<<blocks, 64, 0, s>>>(nullptr, counter, 0, spin);
                hipStreamSynchronize(s);
                auto t1 = std::chrono::steady_clock::now();
                tiny<<<blocks, 64, 0, s>>>(nullptr, counter, 0, spin);
                hipEventRecord(ev, s);
                hipEventSynchronize(ev);
                auto t2 = std::chrono::steady_clock::now();
                ++epoch;
                tiny<<<blocks, 64, 0, s>>>((volatile unsigned *)flag, counter, epoch, spin);
                while (*(volatile unsigned *)flag != epoch) {}
                auto t3 = std::chrono::steady_clock::now();
                if (it >= 50) { a.push_back(us(t0, t1)); b.push_back(us(t1, t2)); c.push_back(us(t2, t3)); }
            }
            hipStreamSynchronize(s);
            auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("blocks %3d spin %4d: launch + hipStreamSynchronize %.1f us | + event record/sync %.1f us | + flag in pinned memory, host spins %.1f us\n",
                   blocks, spin, med(a), med(b), med(c));
        }
    }
    return 0;
}
