// Host gather rate into different kinds of staging memory (developer microbenchmark):
// n pairs of (150 + 500) bytes, each sequence its own heap block, copied by T threads into
// pageable memory / hipHostMalloc default / write-combined / non-coherent memory.
// hipcc -O2 -o gather_rate gather_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

static double run(uint8_t *dr, uint8_t *df, char **reads, char **refs, long n, int R, int F, int threads, bool prefetch) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    long per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        long lo = t * per, hi = std::min(n, lo + per);
        pool.emplace_back([=] {
            for (long i = lo; i < hi; ++i) {
                if (prefetch && i + 6 < hi) {
                    const char *a = reads[i + 6], *b = refs[i + 6];
                    __builtin_prefetch(a); __builtin_prefetch(a + 64); __builtin_prefetch(a + 128);
                    for (int k = 0; k < F; k += 64) __builtin_prefetch(b + k);
                }
                memcpy(dr + (size_t)i * R, reads[i], R);
                memcpy(df + (size_t)i * F, refs[i], F);
            }
        });
    }
    for (auto &th : pool) th.join();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 1 << 20;
    const int R = 150, F = 500;
    std::vector<char *> reads(n), refs(n);
    for (long i = 0; i < n; ++i) {
        reads[i] = (char *)malloc(R);
        refs[i] = (char *)malloc(F);
        memset(reads[i], 'A' + (i & 3), R);
        memset(refs[i], 'C', F);
    }
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    struct Kind { const char *name; unsigned flags; bool pageable; } kinds[] = {
        {"pageable malloc", 0, true},
        {"hipHostMallocDefault", hipHostMallocDefault, false},
        {"hipHostMallocWriteCombined", hipHostMallocWriteCombined, false},
        {"hipHostMallocNonCoherent", hipHostMallocNonCoherent, false},
        {"hipHostMallocNumaUser", hipHostMallocNumaUser, false},
    };
    for (auto &k : kinds) {
        uint8_t *dr = nullptr, *df = nullptr;
        if (k.pageable) {
            dr = (uint8_t *)malloc((size_t)n * R);
            df = (uint8_t *)malloc((size_t)n * F);
            memset(dr, 1, (size_t)n * R);
            memset(df, 1, (size_t)n * F);
        } else {
            if (hipHostMalloc((void **)&dr, (size_t)n * R, k.flags) != hipSuccess ||
                hipHostMalloc((void **)&df, (size_t)n * F, k.flags) != hipSuccess) {
                printf("%s: allocation failed\n", k.name);
                continue;
            }
        }
        for (int threads : {1, 4, 8, 16, 32}) {
            for (int pf = 0; pf < 2; ++pf) {
                double best = 1e9;
                for (int rep = 0; rep < 3; ++rep) best = std::min(best, run(dr, df, reads.data(), refs.data(), n, R, F, threads, pf));
                printf("%-28s threads %2d prefetch %d: %7.2f ms  %6.1f GB/s\n", k.name, threads, pf, best,
                       (double)n * (R + F) / best / 1e6);
            }
        }
        if (k.pageable) { free(dr); free(df); } else { (void)hipHostFree(dr); (void)hipHostFree(df); }
    }
    return 0;
}
