// host_pipeline_check.cpp -- CPU unit test of versalignlib_amd/csrc/host_pipeline.h (the worker pool and the
// gather / scatter that 16 host threads run into caller-owned arrays), meant to be built with
// -fsanitize=thread and with -fsanitize=address,undefined (tests/test_sanitizers.py, `make sanitize`).
// No GPU, no HIP: a fake "device" is a memcpy between the staging buffers.
//
// What it drives: scattered heap blocks of odd lengths in, pair-major staging out (gather); staging into a flat
// sink and into an Alignment-shaped array of operator new[] rows (scatter); chunk sizes around the serial / parallel
// thresholds; thread counts that make the pool resize; thousands of back-to-back tiny jobs (a worker that wakes up
// late must never touch a finished job); an exception thrown by one part.
#include "host_pipeline.h"

#include <stdio.h>
#include <stdlib.h>

#include <stdexcept>
#include <string>

namespace {

struct FakeAlignment {          // include/AlignmentKernel.h:12-24 in shape: the caller delete[]s the rows
    char *read = nullptr, *ref = nullptr;
    short readStart = 0, readEnd = 0, refStart = 0, refEnd = 0;
    ~FakeAlignment() {
        delete[] read;
        delete[] ref;
    }
};

uint64_t mix(uint64_t x) {      // splitmix64
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int failures = 0;
void expect(bool ok, const std::string &what) {
    if (!ok) {
        fprintf(stderr, "FAIL: %s\n", what.c_str());
        ++failures;
    }
}

void round_trip(int R, int F, long long n, int threads) {
    const size_t AL = (size_t)R + F;
    std::vector<char *> reads((size_t)n), refs((size_t)n);
    for (long long i = 0; i < n; ++i) {                      // one heap block per sequence, exactly R / F bytes
        reads[(size_t)i] = new char[R ? R : 1];
        refs[(size_t)i] = new char[F ? F : 1];
        for (int k = 0; k < R; ++k) reads[(size_t)i][k] = (char)mix((uint64_t)i * 131 + k);
        for (int k = 0; k < F; ++k) refs[(size_t)i][k] = (char)mix((uint64_t)i * 257 + k + 7);
    }
    std::vector<uint8_t> st_reads((size_t)n * R + 1), st_refs((size_t)n * F + 1);
    valign::HostPacker packer(R, F);
    packer.gather(reads.data(), refs.data(), n, st_reads.data(), st_refs.data(), threads);
    bool same = true;
    for (long long i = 0; i < n && same; ++i)
        same = memcmp(st_reads.data() + (size_t)i * R, reads[(size_t)i], (size_t)R) == 0 &&
               memcmp(st_refs.data() + (size_t)i * F, refs[(size_t)i], (size_t)F) == 0;
    expect(same, "gather " + std::to_string(R) + "x" + std::to_string(F) + " n=" + std::to_string(n) + " threads=" + std::to_string(threads));

    // a fake device: rows = read bytes then ref bytes, idx = four values of the pair -- with what the kernels guarantee:
    // both rows of a pair are zero in front of offset idx[0] (the strings are right-justified and start there)
    std::vector<uint8_t> st_rows((size_t)n * 2 * AL + 1);
    std::vector<short> st_idx((size_t)n * 4 + 1);
    for (long long i = 0; i < n; ++i) {
        uint8_t *row = st_rows.data() + (size_t)i * 2 * AL;
        memcpy(row, st_reads.data() + (size_t)i * R, (size_t)R);
        memcpy(row + R, st_refs.data() + (size_t)i * F, (size_t)F);
        memset(row + AL, (int)(i & 0xFF) | 1, AL);
        const size_t start = AL ? (size_t)(mix((uint64_t)i + 99) % (AL + 1)) : 0;      // 0 .. AL (AL: an empty string)
        memset(row, 0, start);
        memset(row + AL, 0, start);
        st_idx[(size_t)i * 4] = (short)start;
        for (int k = 1; k < 4; ++k) st_idx[(size_t)i * 4 + k] = (short)(i * 4 + k);
    }
    std::vector<uint8_t> flat_rows((size_t)n * 2 * AL + 1, 0xEE);
    std::vector<short> flat_idx((size_t)n * 4 + 1, -1);
    packer.scatter(valign::FlatSink{flat_rows.data(), flat_idx.data(), AL}, n, st_rows.data(), st_idx.data(), threads);
    expect(memcmp(flat_rows.data(), st_rows.data(), (size_t)n * 2 * AL) == 0 && flat_rows[(size_t)n * 2 * AL] == 0xEE,
           "flat scatter rows");
    expect(memcmp(flat_idx.data(), st_idx.data(), sizeof(short) * 4 * (size_t)n) == 0 && flat_idx[(size_t)n * 4] == -1,
           "flat scatter idx");
    // in two halves at an odd offset (FlatSink + k), as the chunk pipeline does
    if (n > 3) {
        std::fill(flat_rows.begin(), flat_rows.end(), (uint8_t)0xEE);
        const long long k = n / 3;
        valign::FlatSink sink{flat_rows.data(), flat_idx.data(), AL};
        packer.scatter(sink, k, st_rows.data(), st_idx.data(), threads);
        packer.scatter(sink + k, n - k, st_rows.data() + (size_t)k * 2 * AL, st_idx.data() + 4 * k, threads);
        expect(memcmp(flat_rows.data(), st_rows.data(), (size_t)n * 2 * AL) == 0, "flat scatter in two chunks");
    }
    {
        std::vector<FakeAlignment> out((size_t)n);
        packer.scatter(out.data(), n, st_rows.data(), st_idx.data(), threads);
        bool ok = true;
        for (long long i = 0; i < n && ok; ++i) {
            const FakeAlignment &a = out[(size_t)i];
            ok = a.read && a.ref && memcmp(a.read, st_rows.data() + (size_t)i * 2 * AL, AL) == 0 &&
                 memcmp(a.ref, st_rows.data() + (size_t)i * 2 * AL + AL, AL) == 0 && a.readStart == st_idx[(size_t)i * 4] &&
                 a.readEnd == (short)(i * 4 + 1) && a.refStart == (short)(i * 4 + 2) && a.refEnd == (short)(i * 4 + 3);
        }
        expect(ok, "Alignment scatter n=" + std::to_string(n));
    }
    // packed staging (what the device sends since round 4): every row from column `col` on only -- col: the smallest start of
    // the chunk rounded down to 64; the scatter unpacks and writes the zeros in front
    {
        size_t col = AL;
        for (long long i = 0; i < n; ++i) col = std::min(col, (size_t)st_idx[(size_t)i * 4]);
        col &= ~(size_t)63;
        const size_t W = AL - col;
        std::vector<uint8_t> packed((size_t)n * 2 * W + 1, 0x77);
        for (long long r = 0; r < 2 * n; ++r) memcpy(packed.data() + (size_t)r * W, st_rows.data() + (size_t)r * AL + col, W);
        std::fill(flat_rows.begin(), flat_rows.end(), (uint8_t)0xEE);
        packer.scatter(valign::FlatSink{flat_rows.data(), flat_idx.data(), AL}, n, packed.data(), st_idx.data(), threads, col);
        expect(memcmp(flat_rows.data(), st_rows.data(), (size_t)n * 2 * AL) == 0 && flat_rows[(size_t)n * 2 * AL] == 0xEE,
               "flat scatter from packed rows");
        std::vector<FakeAlignment> out((size_t)n);
        packer.scatter(out.data(), n, packed.data(), st_idx.data(), threads, col);
        bool ok = true;
        for (long long i = 0; i < n && ok; ++i)
            ok = memcmp(out[(size_t)i].read, st_rows.data() + (size_t)i * 2 * AL, AL) == 0 &&
                 memcmp(out[(size_t)i].ref, st_rows.data() + (size_t)i * 2 * AL + AL, AL) == 0;
        expect(ok, "Alignment scatter from packed rows");
    }
    for (long long i = 0; i < n; ++i) {
        delete[] reads[(size_t)i];
        delete[] refs[(size_t)i];
    }
}

// 4-bit class packing: the vector form against the definition, every byte value at every position of the step
void packing(int R, int F, long long n, int threads) {
    std::vector<char *> reads((size_t)n), refs((size_t)n);
    static const char alphabet[] = "ACGTNacgtn";
    for (long long i = 0; i < n; ++i) {
        reads[(size_t)i] = new char[R ? R : 1];
        refs[(size_t)i] = new char[F ? F : 1];
        for (int k = 0; k < R; ++k) {
            const uint64_t r = mix((uint64_t)i * 977 + k);
            reads[(size_t)i][k] = (r & 7) ? alphabet[(r >> 8) % 10] : (char)(r >> 16);      // one in eight: any byte
        }
        for (int k = 0; k < F; ++k) {
            const uint64_t r = mix((uint64_t)i * 1031 + k + 3);
            refs[(size_t)i][k] = (r & 7) ? alphabet[(r >> 8) % 10] : (char)(r >> 16);
        }
    }
    const size_t PR = valign::packed_length(R), PF = valign::packed_length(F);
    std::vector<uint8_t> pr((size_t)n * PR + 1, 0xEE), pf((size_t)n * PF + 1, 0xEE);
    valign::HostPacker packer(R, F);
    packer.gather_packed(reads.data(), refs.data(), n, pr.data(), pf.data(), threads);
    bool ok = pr[(size_t)n * PR] == 0xEE && pf[(size_t)n * PF] == 0xEE;
    for (long long i = 0; i < n && ok; ++i) {
        for (int k = 0; k < R && ok; ++k)
            ok = ((pr[(size_t)i * PR + k / 2] >> (4 * (k & 1))) & 15) == valign::base_class_of((uint8_t)reads[(size_t)i][k]);
        for (int k = 0; k < F && ok; ++k)
            ok = ((pf[(size_t)i * PF + k / 2] >> (4 * (k & 1))) & 15) == valign::base_class_of((uint8_t)refs[(size_t)i][k]);
        if (R & 1) ok = ok && (pr[(size_t)i * PR + PR - 1] >> 4) == 0;            // the spare nibble of an odd length is 0
        if (F & 1) ok = ok && (pf[(size_t)i * PF + PF - 1] >> 4) == 0;
    }
    expect(ok, "gather_packed " + std::to_string(R) + "x" + std::to_string(F) + " n=" + std::to_string(n));
    for (long long i = 0; i < n; ++i) {
        delete[] reads[(size_t)i];
        delete[] refs[(size_t)i];
    }
}

}  // namespace

int main() {
    {   // every byte value, scalar == vector == definition
        uint8_t all[256 + 64], a[160], b[160];
        for (int c = 0; c < 256 + 64; ++c) all[c] = (uint8_t)c;
        valign::pack_classes(all, 256 + 64, a);
        valign::pack_classes_scalar(all, 256 + 64, b);
        bool ok = memcmp(a, b, 160) == 0;
        for (int c = 0; c < 256 && ok; ++c) ok = ((a[c / 2] >> (4 * (c & 1))) & 15) == valign::base_class_of((uint8_t)c);
        expect(ok, "pack_classes over every byte value");
    }
    packing(150, 500, 5003, 16);
    packing(37, 101, 4097, 3);
    packing(1, 33, 300, 1);
    packing(64, 31, 4500, 16);
    packing(0, 5, 10, 2);

    // chunk sizes around the serial thresholds (2048 / 4096), odd shapes, thread counts that resize the pool
    const long long counts[] = {0, 1, 2, 2047, 2048, 2049, 4095, 4096, 4097, 10007};
    const int threads[] = {1, 2, 3, 16, 5, 16};
    for (int th : threads)
        for (long long n : counts) round_trip(37, 101, n, th);
    round_trip(150, 500, 20011, 16);
    round_trip(1, 1, 5000, 16);
    round_trip(0, 3, 4100, 7);

    // thousands of tiny jobs back to back on one pool: a late waker must find either nothing or the next job
    {
        valign::WorkerPool pool(15);
        std::vector<int> hits(64);
        long long total = 0;
        for (int rep = 0; rep < 4000; ++rep) {
            const int parts = 1 + rep % 33;
            std::fill(hits.begin(), hits.end(), 0);
            pool.run(parts, [&](int p) { hits[(size_t)p] += 1; });
            for (int p = 0; p < parts; ++p) {
                expect(hits[(size_t)p] == 1, "every part runs exactly once");
                total += hits[(size_t)p];
            }
        }
        expect(total > 0, "jobs ran");
        // an exception from one part reaches the caller, the pool stays usable
        bool caught = false;
        try {
            pool.run(16, [&](int p) {
                if (p == 11) throw std::runtime_error("part 11");
            });
        } catch (const std::runtime_error &e) {
            caught = std::string(e.what()) == "part 11";
        }
        expect(caught, "exception propagates");
        int after = 0;
        std::mutex m;
        pool.run(16, [&](int) {
            std::lock_guard<std::mutex> lock(m);
            ++after;
        });
        expect(after == 16, "pool usable after an exception");
    }
    if (failures) {
        fprintf(stderr, "%d failure(s)\n", failures);
        return 1;
    }
    printf("host pipeline ok\n");
    return 0;
}
