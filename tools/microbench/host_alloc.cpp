#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <malloc.h>
#include <stdint.h>
#include <sys/mman.h>
struct A { char *read, *ref; short a,b,c,d; };
int main(int argc, char **argv) {
    const long n = atol(argv[1]); const int T = atoi(argv[2]); const int mode = atoi(argv[3]); const int reps = argc > 4 ? atoi(argv[4]) : 3;
    const size_t AL = 650;
    std::vector<char> src(2 * AL * 4096, 'A');
    for (int rep = 0; rep < reps; ++rep) {
        A *al = new A[n]();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            long lo = n * t / T, hi = n * (t + 1) / T;
            if (mode == 1) {   // prime: grow this thread's arena in big steps
                for (long i = lo; i < hi; ++i) {
                    if ((i - lo) % 160 == 0) { void *p = malloc(112 << 10); free(p); }
                    al[i].read = new char[AL]; al[i].ref = new char[AL];
                    memcpy(al[i].read, &src[(i % 4096) * 2 * AL], AL); memcpy(al[i].ref, &src[(i % 4096) * 2 * AL + AL], AL);
                }
            } else if (mode == 2) {   // allocate all first, then copy
                for (long i = lo; i < hi; ++i) { al[i].read = new char[AL]; al[i].ref = new char[AL]; }
                for (long i = lo; i < hi; ++i) { memcpy(al[i].read, &src[(i % 4096) * 2 * AL], AL); memcpy(al[i].ref, &src[(i % 4096) * 2 * AL + AL], AL); }
            } else if (mode == 4) {   // transparent huge pages for the arena the rows come from (glibc: 64 MB heaps, aligned)
                uintptr_t heap = 0;
                for (long i = lo; i < hi; ++i) {
                    al[i].read = new char[AL]; al[i].ref = new char[AL];
                    const uintptr_t h = (uintptr_t)al[i].ref >> 26;
                    if (h != heap) { heap = h; madvise((void *)(h << 26), 64u << 20, MADV_HUGEPAGE); }
                    memcpy(al[i].read, &src[(i % 4096) * 2 * AL], AL); memcpy(al[i].ref, &src[(i % 4096) * 2 * AL + AL], AL);
                }
            } else {
                for (long i = lo; i < hi; ++i) {
                    al[i].read = new char[AL]; al[i].ref = new char[AL];
                    memcpy(al[i].read, &src[(i % 4096) * 2 * AL], AL); memcpy(al[i].ref, &src[(i % 4096) * 2 * AL + AL], AL);
                }
            }
        });
        for (auto &x : th) x.join();
        auto t1 = std::chrono::steady_clock::now();
        // caller frees (single thread), like ~Alignment
        for (long i = 0; i < n; ++i) { delete[] al[i].read; delete[] al[i].ref; }
        delete[] al;
        auto t2 = std::chrono::steady_clock::now();
        printf("mode %d rep %d: alloc+copy %.1f ms, free %.1f ms\n", mode, rep, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
}
