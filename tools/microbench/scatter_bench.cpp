// scatter_bench.cpp -- host-only timing of HostPacker::scatter into the ABI's Alignment rows (2n operator new[] blocks):
// the rows' strings only (zeros written, not copied) against whole rows, huge-page hint on / off, interleaved.
//   g++ -O2 -std=c++17 -pthread -I versalignlib_amd/csrc tools/microbench/scatter_bench.cpp -o /tmp/scatter_bench
//   MALLOC_TOP_PAD_=268435456 /tmp/scatter_bench [pairs] [threads]
#include "host_pipeline.h"

#include <stdio.h>
#include <stdlib.h>

#include <chrono>

struct Alignment {
    char *read = nullptr, *ref = nullptr;
    short readStart = 0, readEnd = 0, refStart = 0, refEnd = 0;
};

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 1 << 20;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const int R = 150, F = 500;
    const size_t AL = R + F;
    std::vector<uint8_t> rows((size_t)n * 2 * AL, 0);
    std::vector<short> idx((size_t)n * 4);
    unsigned long long x = 88172645463325252ull;
    for (long long i = 0; i < n; ++i) {                        // strings of 150-230 characters, right-justified
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const size_t len = 150 + x % 80, start = AL - 1 - len;
        memset(rows.data() + (size_t)i * 2 * AL + start, 'A', len);
        memset(rows.data() + (size_t)i * 2 * AL + AL + start, 'C', len);
        idx[(size_t)i * 4] = idx[(size_t)i * 4 + 2] = (short)start;
        idx[(size_t)i * 4 + 1] = idx[(size_t)i * 4 + 3] = (short)(AL - 1);
    }
    valign::HostPacker packer(R, F);
    for (int rep = 0; rep < 3; ++rep)
        for (int variant = 0; variant < 4; ++variant) {
            packer.set_whole_rows(variant & 1);
            packer.set_huge_rows(variant & 2);
            std::vector<Alignment> out((size_t)n);
            const auto t0 = std::chrono::steady_clock::now();
            packer.scatter(out.data(), n, rows.data(), idx.data(), threads);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("rep %d  %-12s %-14s %7.2f ms\n", rep, (variant & 1) ? "whole rows" : "strings only", (variant & 2) ? "huge-page hint" : "4 KB pages", ms);
            fflush(stdout);
            for (auto &a : out) { delete[] a.read; delete[] a.ref; }
        }
    return 0;
}
