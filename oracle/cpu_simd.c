/*
 * cpu_simd.c -- AVX2 inter-sequence Smith-Waterman / NW-variant SCORES on the CPU, 16 pairs per vector.
 *
 * TEST INFRASTRUCTURE ONLY (like cpu_ref.c): bench.py's cpu_baseline leg times it beside the GPU path and
 * tests/ check it against cpu_ref.c; nothing under versalignlib_amd/ may call, link or import it.
 *
 * What it is for: the reference's fastest CPU path is its AVX2 kernel -- 16 pairs per __m256i of int16 cells, one
 * DP cell of all 16 per instruction (src/Kernels/AVX-SSE/AVXKernel.cpp:804-920 is its Smith-Waterman score sweep) --
 * but that kernel has the linear gap model only, gathers its 16 bases lane by lane for every cell
 * (AVXKernel.cpp, the _mm256_set_epi16 of 16 pointers per cell) and runs on one thread (its OpenMP is compiled out
 * on Linux, AVXKernel.cpp:74-80).  BASELINE.json's headline is AFFINE gaps: this file is the CPU figure for that --
 * the same inter-sequence idea (16 int16 lanes = 16 pairs), written from scratch: bases are transposed once per batch
 * of 16 pairs (class codes, one vector per row / column), the sweep is row-major with the H and F rows in L1, the
 * recurrences and saturating adds are exactly those of cpu_ref.c's vref_score / vref_score_affine (which pin it), and
 * batches are spread over OpenMP threads.
 *
 *   E(i,j) = max(E(i,j-1) + ext_read, H(i,j-1) + open_read)       gap in the read (LEFT)
 *   F(i,j) = max(F(i-1,j) + ext_ref,  H(i-1,j) + open_ref)        gap in the reference (UP)
 *   H(i,j) = max([0,] H(i-1,j-1) + S, E, F)                       linear model: open == extend == gap
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t match, mismatch, gap_read, gap_ref;
    int32_t open_read, ext_read, open_ref, ext_ref;
} vref_scoring;

#define LANES 16
#define NEG_INF ((int16_t)-16384)

static int class_of(uint8_t ch) {          /* src/Kernels/default/DefaultKernel.h:43-60; bytes >= 0x80: 0 */
    switch (ch) {
        case 'A': case 'a': return 1;
        case 'T': case 't': return 2;
        case 'C': case 'c': return 3;
        case 'G': case 'g': return 4;
        case 'N': case 'n': return 5;
        default: return 0;
    }
}

/* affine != 0: the four open / extend scores; else the linear model (gap_read / gap_ref both ways). */
__attribute__((target("avx2")))
int vsimd_score(int opt, int affine, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
                const vref_scoring *sc, int16_t *scores, int threads) {
    const int alg = opt & 0xF;
    if (alg > 1 || n <= 0) return 0;
    if (threads < 1) threads = 1;
    int8_t cls[256];
    for (int c = 0; c < 256; ++c) cls[c] = (int8_t)class_of((uint8_t)c);
    const __m256i oR = _mm256_set1_epi16((int16_t)(affine ? sc->open_read : sc->gap_read));
    const __m256i eR = _mm256_set1_epi16((int16_t)(affine ? sc->ext_read : sc->gap_read));
    const __m256i oF = _mm256_set1_epi16((int16_t)(affine ? sc->open_ref : sc->gap_ref));
    const __m256i eF = _mm256_set1_epi16((int16_t)(affine ? sc->ext_ref : sc->gap_ref));
    const __m256i mat = _mm256_set1_epi16((int16_t)sc->match), mis = _mm256_set1_epi16((int16_t)sc->mismatch);
    const __m256i zero = _mm256_setzero_si256(), neg = _mm256_set1_epi16(NEG_INF);
    const int batches = (n + LANES - 1) / LANES;
#pragma omp parallel num_threads(threads)
    {
        /* per thread: class codes of the batch, one vector per read row / reference column; the H and F rows */
        __m256i *rrow = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(R + 1));
        __m256i *rval = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(R + 1));
        __m256i *fcol = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(F + 1));
        __m256i *fmis = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(F + 1));
        __m256i *H = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(F + 1));
        __m256i *Fv = (__m256i *)aligned_alloc(32, sizeof(__m256i) * (size_t)(F + 1));
        int16_t tmp[LANES] __attribute__((aligned(32)));
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < batches; ++b) {
            const int first = b * LANES;
            /* transpose: lane k = pair first + k (a short last batch repeats its last pair; the copies are dropped) */
            for (int i = 0; i < R; ++i) {
                int16_t v[LANES] __attribute__((aligned(32))), ok[LANES] __attribute__((aligned(32)));
                for (int k = 0; k < LANES; ++k) {
                    const int p = first + k < n ? first + k : n - 1;
                    const int c = cls[reads[(size_t)p * R + i]];
                    ok[k] = (c >= 1 && c <= 4) ? -1 : 0;
                    v[k] = ok[k] ? (int16_t)c : (int16_t)-2;           /* never equal to a column's code */
                }
                rrow[i] = _mm256_load_si256((const __m256i *)v);
                rval[i] = _mm256_load_si256((const __m256i *)ok);
            }
            for (int j = 0; j < F; ++j) {
                int16_t v[LANES] __attribute__((aligned(32))), ok[LANES] __attribute__((aligned(32)));
                for (int k = 0; k < LANES; ++k) {
                    const int p = first + k < n ? first + k : n - 1;
                    const int c = cls[refs[(size_t)p * F + j]];
                    ok[k] = (c >= 1 && c <= 4) ? -1 : 0;
                    v[k] = ok[k] ? (int16_t)c : (int16_t)-1;
                }
                fcol[j] = _mm256_load_si256((const __m256i *)v);
                fmis[j] = _mm256_and_si256(_mm256_load_si256((const __m256i *)ok), mis);     /* mismatch where the column is ACGT */
            }
            for (int j = 0; j <= F; ++j) {
                H[j] = zero;
                Fv[j] = neg;
            }
            __m256i best = zero;
            for (int i = 0; i < R; ++i) {
                const __m256i vr = rrow[i], vok = rval[i];
                __m256i hdiag = H[0], hleft = zero, e = neg;
                for (int j = 0; j < F; ++j) {
                    const __m256i hup = H[j + 1];
                    e = _mm256_max_epi16(_mm256_adds_epi16(e, eR), _mm256_adds_epi16(hleft, oR));
                    const __m256i f = _mm256_max_epi16(_mm256_adds_epi16(Fv[j + 1], eF), _mm256_adds_epi16(hup, oF));
                    /* S: match where the codes are equal (both ACGT then), mismatch where both are ACGT, else 0 */
                    const __m256i eq = _mm256_cmpeq_epi16(vr, fcol[j]);
                    const __m256i s = _mm256_blendv_epi8(_mm256_and_si256(fmis[j], vok), mat, eq);
                    __m256i h = _mm256_max_epi16(_mm256_max_epi16(_mm256_add_epi16(hdiag, s), e), f);
                    if (alg == 0) {
                        h = _mm256_max_epi16(h, zero);
                        best = _mm256_max_epi16(best, h);
                    }
                    Fv[j + 1] = f;
                    hdiag = hup;
                    H[j + 1] = h;
                    hleft = h;
                }
                H[0] = zero;
                if (alg == 1) best = _mm256_max_epi16(best, H[F]);
            }
            if (alg == 1)
                for (int j = 0; j <= F; ++j) best = _mm256_max_epi16(best, H[j]);
            _mm256_store_si256((__m256i *)tmp, best);
            for (int k = 0; k < LANES && first + k < n; ++k) scores[first + k] = tmp[k];
        }
        free(Fv);
        free(H);
        free(fmis);
        free(fcol);
        free(rval);
        free(rrow);
    }
    return n;
}

int vsimd_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }
