/*
 * cpu_ref.c -- CPU restatement of versalignLib's Default (scalar) AlignmentKernel.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP backend.
 * Nothing in the product path (versalignlib_amd/, libHIPKernel.so) may call, link
 * or import it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do.  It is a from-scratch restatement of the reference algorithm, pinned
 * against the compiled reference kernels (oracle/_ref, built by oracle/Makefile
 * from /root/reference where that tree exists) through tests/golden/ fixtures.
 *
 * Reference semantics followed (paths relative to the reference checkout):
 *   base classes ........ src/Kernels/default/DefaultKernel.h:43-60  (char_to_score)
 *   substitution table .. src/Kernels/default/DefaultKernel.h:83-97  (base_score)
 *   SW score ............ src/Kernels/default/DefaultKernel.cpp:83-138
 *   NW-variant score .... src/Kernels/default/DefaultKernel.cpp:140-202
 *   SW fill + end cell .. src/Kernels/default/DefaultKernel.cpp:204-280
 *   NW fill + end cell .. src/Kernels/default/DefaultKernel.cpp:282-389
 *   SW / NW traceback ... src/Kernels/default/DefaultKernel.cpp:391-456, 458-525
 *
 * Deliberate, documented differences from the Default kernel's *outputs*:
 *   - scores are returned as full int16 (Default stores only the low byte,
 *     DefaultKernel.cpp:137,199; SSE/AVX/OpenCL store the full short);
 *   - alignment rows are zero-filled before readStart and carry '\0' at index
 *     R+F-1 for SW too (Default leaves those bytes uninitialised for SW);
 *   - bytes >= 0x80 map to class 0 (the reference indexes a table with a signed
 *     char there, which is undefined).
 *
 * The affine-gap (Gotoh) functions at the bottom are an EXTENSION with no
 * reference counterpart ("parity unpinned by the reference").  They are pinned
 * independently of this file: (i) affine(open == extend == g) == the reference-pinned
 * linear(g) bit for bit; (ii) exhaustive enumeration of every alignment of tiny pairs
 * (tests/enumerate_alignments.py -> the .npz files of tests/golden/affine, no dynamic program);
 * (iii) the general-gap-function recurrence without E/F state at sizes enumeration
 * cannot reach (tests/test_affine_enumeration.py).  The banded and int32 entry points
 * are pinned by band >= matrix == unbanded == reference, the per-cell / block sandwich,
 * and int32 == int16 wherever int16 does not overflow.
 *
 * Data layout for every entry point: reads = n*R bytes, pair-major contiguous;
 * refs = n*F bytes likewise; each sequence exactly R / F bytes, short ones
 * right-padded with '\0' (the reference host's pad(), src/util/versalignUtil.cpp:17-33).
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PTR_START 0
#define PTR_DIAG  1
#define PTR_UP    2
#define PTR_LEFT  3

typedef struct {
    int32_t match, mismatch, gap_read, gap_ref;                 /* linear model (reference) */
    int32_t open_read, ext_read, open_ref, ext_ref;             /* affine extension         */
} vref_scoring;

/* ---- base classes: A/a 1, T/t 2, C/c 3, G/g 4, N/n 5, everything else 0 ---- */
static uint8_t g_class[256];
static int g_class_ready = 0;

static void class_init(void) {
    if (g_class_ready) return;
    memset(g_class, 0, sizeof g_class);
    g_class['A'] = g_class['a'] = 1;
    g_class['T'] = g_class['t'] = 2;
    g_class['C'] = g_class['c'] = 3;
    g_class['G'] = g_class['g'] = 4;
    g_class['N'] = g_class['n'] = 5;
    g_class_ready = 1;
}

/* 6x6 substitution table: 0 whenever either class is 0 or 5 */
static void subst_init(const vref_scoring *sc, int16_t tab[6][6]) {
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            int16_t v = 0;
            if (a >= 1 && a <= 4 && b >= 1 && b <= 4)
                v = (int16_t)(a == b ? sc->match : sc->mismatch);
            tab[a][b] = v;
        }
}

static inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ scores */

static int16_t sw_score_one(const uint8_t *read, const uint8_t *ref, int R, int F,
                            int16_t tab[6][6], int16_t gr, int16_t gf, int16_t *rows) {
    int16_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int16_t) * 2 * (size_t)(F + 1));
    int16_t best = 0;
    for (int i = 0; i < R; ++i) {
        const int16_t *srow = tab[g_class[read[i]]];
        for (int j = 0; j < F; ++j) {
            int16_t up = prev[j + 1], left = cur[j];
            int16_t diag = (int16_t)(prev[j] + srow[g_class[ref[j]]]);
            int16_t h = max16((int16_t)(up + gf), max16((int16_t)(left + gr), max16(diag, 0)));
            cur[j + 1] = h;
            best = max16(best, h);
        }
        int16_t *t = prev; prev = cur; cur = t;
    }
    return best;
}

static int16_t nw_score_one(const uint8_t *read, const uint8_t *ref, int R, int F,
                            int16_t tab[6][6], int16_t gr, int16_t gf, int16_t *rows) {
    int16_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int16_t) * 2 * (size_t)(F + 1));   /* row 0 and column 0 stay 0 */
    int16_t best = 0;
    for (int i = 0; i < R; ++i) {
        const int16_t *srow = tab[g_class[read[i]]];
        for (int j = 0; j < F; ++j) {
            int16_t up = prev[j + 1], left = cur[j];
            int16_t diag = (int16_t)(prev[j] + srow[g_class[ref[j]]]);
            cur[j + 1] = max16((int16_t)(up + gf), max16((int16_t)(left + gr), diag));
        }
        best = max16(best, cur[F]);                           /* last column, every row */
        int16_t *t = prev; prev = cur; cur = t;
    }
    for (int j = 0; j <= F; ++j) best = max16(best, prev[j]); /* last computed row      */
    return best;
}

/* opt: 0 = Smith-Waterman, 1 = reference's Needleman-Wunsch variant */
int vref_score(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
               const vref_scoring *sc, int16_t *scores, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;                            /* reference: silent no-op */
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int16_t gr = (int16_t)sc->gap_read, gf = (int16_t)sc->gap_ref;
    const int alg = opt & 0xF;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int16_t *rows = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            scores[p] = alg == 0 ? sw_score_one(rd, rf, R, F, tab, gr, gf, rows)
                                 : nw_score_one(rd, rf, R, F, tab, gr, gf, rows);
        }
        free(rows);
    }
    return n;
}

/* -------------------------------------------------------------- alignments */

/* SW fill: pointer priority START(cell==0) > DIAG > UP > LEFT; end cell = first
 * strictly greater cell in row-major order (DefaultKernel.cpp:238-256).          */
static void sw_fill(const uint8_t *read, const uint8_t *ref, int R, int F, int16_t tab[6][6],
                    int16_t gr, int16_t gf, int16_t *rows, uint8_t *ptr, int *end_i, int *end_j) {
    int16_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int16_t) * 2 * (size_t)(F + 1));
    int16_t best = 0; int bi = 0, bj = 0;
    for (int i = 0; i < R; ++i) {
        const int16_t *srow = tab[g_class[read[i]]];
        uint8_t *prow = ptr + (size_t)(i + 1) * (F + 1);
        for (int j = 0; j < F; ++j) {
            int16_t up = prev[j + 1], left = cur[j];
            int16_t diag = (int16_t)(prev[j] + srow[g_class[ref[j]]]);
            int16_t h = max16((int16_t)(up + gf), max16((int16_t)(left + gr), max16(diag, 0)));
            cur[j + 1] = h;
            uint8_t p = PTR_START;
            if (h == 0) p = PTR_START;
            else if (h == diag) p = PTR_DIAG;
            else if (h == up + gf) p = PTR_UP;
            else if (h == left + gr) p = PTR_LEFT;
            prow[j + 1] = p;
            if (h > best) { best = h; bi = i; bj = j; }
        }
        int16_t *t = prev; prev = cur; cur = t;
    }
    *end_i = bi; *end_j = bj;
}

/* NW-variant fill: column 0 = (i+1)*gap_ref with UP, row 0 = 0 with START, priority
 * DIAG > UP > LEFT, end cell from the first-invalid-character bookkeeping
 * (DefaultKernel.cpp:285-318, 340-355, 381-387).                                  */
static void nw_fill(const uint8_t *read, const uint8_t *ref, int R, int F, int16_t tab[6][6],
                    int16_t gr, int16_t gf, int16_t *rows, uint8_t *ptr, int *end_i, int *end_j) {
    int16_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int16_t) * 2 * (size_t)(F + 1));
    int16_t last_read = (int16_t)(R - 1), last_ref = (int16_t)(F - 1);
    int16_t row_best = INT16_MIN, row_arg = 0, snap_arg = -1;
    for (int i = 0; i < R; ++i) {
        uint8_t *prow = ptr + (size_t)(i + 1) * (F + 1);
        prow[0] = PTR_UP;
        cur[0] = (int16_t)((i + 1) * gf);
        if (last_read == R - 1 && g_class[read[i]] == 0) last_read = (int16_t)(i - 1);
        if (last_read + 1 == i) snap_arg = row_arg;           /* argmax of the previous row */
        row_best = cur[0]; row_arg = 0;
        const int16_t *srow = tab[g_class[read[i]]];
        for (int j = 0; j < F; ++j) {
            int16_t up = prev[j + 1], left = cur[j];
            int16_t diag = (int16_t)(prev[j] + srow[g_class[ref[j]]]);
            int16_t h = max16((int16_t)(up + gf), max16((int16_t)(left + gr), diag));
            cur[j + 1] = h;
            uint8_t p = PTR_START;
            if (h == diag) p = PTR_DIAG;
            else if (h == up + gf) p = PTR_UP;
            else if (h == left + gr) p = PTR_LEFT;
            if (last_ref == F - 1 && g_class[ref[j]] == 0) last_ref = (int16_t)(j - 1);
            if (h > row_best) { row_best = h; row_arg = (int16_t)j; }
            prow[j + 1] = p;
        }
        int16_t *t = prev; prev = cur; cur = t;
    }
    if (snap_arg < 0) snap_arg = row_arg;
    *end_i = last_read;
    *end_j = last_ref < snap_arg ? last_ref : snap_arg;
}

/* A traceback that would leave the matrix: the caller asked the int16 restatement for a shape x scoring outside its range
 * (tests must use the int32 one there).  Said once per process on stderr, counted for vref_walks_left_matrix().        */
static int g_walks_left = 0;
static void vref_note_walk_left_matrix(int R, int F) {
    int first;
#pragma omp atomic capture
    first = g_walks_left++;
    if (first == 0) fprintf(stderr, "oracle/cpu_ref: a traceback left the %d x %d matrix (cells outside the int16 range?)\n", R, F);
}
int vref_walks_left_matrix(void) { return g_walks_left; }

static void traceback(const uint8_t *read, const uint8_t *ref, int R, int F, const uint8_t *ptr,
                      int rp, int fp, uint8_t *row_read, uint8_t *row_ref, int16_t idx[4]) {
    const int AL = R + F;
    memset(row_read, 0, (size_t)AL);
    memset(row_ref, 0, (size_t)AL);
    int k = AL - 2;
    uint8_t p = ptr[(size_t)(rp + 1) * (F + 1) + fp + 1];
    while (p != PTR_START) {
        /* (a walk that would leave the matrix -- only possible where the int16 cells wrapped, i.e. outside the range this
         * restatement is meant for -- stops instead of reading and writing out of bounds) */
        if (k < 0 || ((p != PTR_LEFT) && rp < 0) || ((p != PTR_UP) && fp < 0)) {
            vref_note_walk_left_matrix(R, F);
            break;
        }
        if (p == PTR_UP)        { row_ref[k] = '-';      row_read[k] = read[rp--]; }
        else if (p == PTR_LEFT) { row_read[k] = '-';     row_ref[k] = ref[fp--];   }
        else                    { row_read[k] = read[rp--]; row_ref[k] = ref[fp--]; }
        p = ptr[(size_t)(rp + 1) * (F + 1) + fp + 1];
        --k;
    }
    idx[0] = (int16_t)(k + 1);      /* readStart */
    idx[1] = (int16_t)(AL - 1);     /* readEnd   */
    idx[2] = (int16_t)(k + 1);      /* refStart  */
    idx[3] = (int16_t)(AL - 1);     /* refEnd    */
}

/* rows: n * 2 * (R+F) bytes (read row then ref row per pair); idx: n * 4 int16
 * (readStart, readEnd, refStart, refEnd).                                        */
int vref_align(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
               const vref_scoring *sc, uint8_t *rows_out, int16_t *idx_out, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int16_t gr = (int16_t)sc->gap_read, gf = (int16_t)sc->gap_ref;
    const int alg = opt & 0xF, AL = R + F;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int16_t *rows = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(F + 1));
        uint8_t *ptr = (uint8_t *)malloc((size_t)(R + 1) * (F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            memset(ptr, PTR_START, (size_t)(R + 1) * (F + 1));
            int ei, ej;
            if (alg == 0) sw_fill(rd, rf, R, F, tab, gr, gf, rows, ptr, &ei, &ej);
            else          nw_fill(rd, rf, R, F, tab, gr, gf, rows, ptr, &ei, &ej);
            traceback(rd, rf, R, F, ptr, ei, ej, rows_out + (size_t)p * 2 * AL,
                      rows_out + (size_t)p * 2 * AL + AL, idx_out + (size_t)p * 4);
        }
        free(ptr);
        free(rows);
    }
    return n;
}

/* ---- the same fills on int32 cells: what the mathematics says where the reference's shorts would wrap
 * (NW-variant borders beyond -32768: read_length * gap_ref).  Identical to the int16 fills wherever those do not
 * overflow; libHIPKernel.so takes int32 cells for such alignments instead of refusing them (round 3).           */
static void nw_fill_wide(const uint8_t *read, const uint8_t *ref, int R, int F, int16_t tab[6][6],
                         int gr, int gf, int32_t *rows, uint8_t *ptr, int *end_i, int *end_j) {
    int32_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int32_t) * 2 * (size_t)(F + 1));
    int last_read = R - 1, last_ref = F - 1;
    int32_t row_best = INT32_MIN;
    int row_arg = 0, snap_arg = -1;
    for (int i = 0; i < R; ++i) {
        uint8_t *prow = ptr + (size_t)(i + 1) * (F + 1);
        prow[0] = PTR_UP;
        cur[0] = (i + 1) * gf;
        if (last_read == R - 1 && g_class[read[i]] == 0) last_read = i - 1;
        if (last_read + 1 == i) snap_arg = row_arg;
        row_best = cur[0]; row_arg = 0;
        const int16_t *srow = tab[g_class[read[i]]];
        for (int j = 0; j < F; ++j) {
            const int32_t up = prev[j + 1] + gf, left = cur[j] + gr, diag = prev[j] + srow[g_class[ref[j]]];
            int32_t h = up > left ? up : left;
            if (diag > h) h = diag;
            cur[j + 1] = h;
            prow[j + 1] = h == diag ? PTR_DIAG : (h == up ? PTR_UP : PTR_LEFT);
            if (last_ref == F - 1 && g_class[ref[j]] == 0) last_ref = j - 1;
            if (h > row_best) { row_best = h; row_arg = j; }
        }
        int32_t *t = prev; prev = cur; cur = t;
    }
    if (snap_arg < 0) snap_arg = row_arg;
    *end_i = last_read;
    *end_j = last_ref < snap_arg ? last_ref : snap_arg;
}

static void sw_fill_wide(const uint8_t *read, const uint8_t *ref, int R, int F, int16_t tab[6][6],
                         int gr, int gf, int32_t *rows, uint8_t *ptr, int *end_i, int *end_j) {
    int32_t *prev = rows, *cur = rows + (F + 1);
    memset(rows, 0, sizeof(int32_t) * 2 * (size_t)(F + 1));
    int32_t best = 0; int bi = 0, bj = 0;
    for (int i = 0; i < R; ++i) {
        const int16_t *srow = tab[g_class[read[i]]];
        uint8_t *prow = ptr + (size_t)(i + 1) * (F + 1);
        for (int j = 0; j < F; ++j) {
            const int32_t up = prev[j + 1] + gf, left = cur[j] + gr, diag = prev[j] + srow[g_class[ref[j]]];
            int32_t h = up > left ? up : left;
            if (diag > h) h = diag;
            if (h < 0) h = 0;
            cur[j + 1] = h;
            prow[j + 1] = h == 0 ? PTR_START : (h == diag ? PTR_DIAG : (h == up ? PTR_UP : PTR_LEFT));
            if (h > best) { best = h; bi = i; bj = j; }
        }
        int32_t *t = prev; prev = cur; cur = t;
    }
    *end_i = bi; *end_j = bj;
}

int vref_align_wide(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
                    const vref_scoring *sc, uint8_t *rows_out, int16_t *idx_out, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int gr = sc->gap_read, gf = sc->gap_ref;
    const int alg = opt & 0xF, AL = R + F;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(F + 1));
        uint8_t *ptr = (uint8_t *)malloc((size_t)(R + 1) * (F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            memset(ptr, PTR_START, (size_t)(R + 1) * (F + 1));
            int ei, ej;
            if (alg == 0) sw_fill_wide(rd, rf, R, F, tab, gr, gf, rows, ptr, &ei, &ej);
            else          nw_fill_wide(rd, rf, R, F, tab, gr, gf, rows, ptr, &ei, &ej);
            traceback(rd, rf, R, F, ptr, ei, ej, rows_out + (size_t)p * 2 * AL,
                      rows_out + (size_t)p * 2 * AL + AL, idx_out + (size_t)p * 4);
        }
        free(ptr);
        free(rows);
    }
    return n;
}

/* ------------------------------------------- affine-gap extension (Gotoh) */
/* Not in the reference.  open_* = cost of the FIRST gap base, ext_* = cost of each
 * further one, so open == ext == g reproduces the linear model exactly.
 *   E(i,j) = max(E(i,j-1) + ext_read, H(i,j-1) + open_read)      gap in the read (LEFT)
 *   F(i,j) = max(F(i-1,j) + ext_ref,  H(i-1,j) + open_ref)       gap in the ref  (UP)
 *   H(i,j) = max([0,] H(i-1,j-1) + S, E(i,j), F(i,j))
 * Borders follow the reference score kernels: H(0,*) = H(*,0) = 0; E(*,0) and
 * F(0,*) are "minus infinity" (NEG_INF, never selected).  NW result as in the
 * linear variant: max(0, last column, last row).                                  */
#define NEG_INF ((int16_t)-16384)

static inline int16_t sat_add(int16_t a, int16_t b) {
    int v = (int)a + (int)b;
    return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

int vref_score_affine(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
                      const vref_scoring *sc, int16_t *scores, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int16_t oR = (int16_t)sc->open_read, eR = (int16_t)sc->ext_read;
    const int16_t oF = (int16_t)sc->open_ref,  eF = (int16_t)sc->ext_ref;
    const int alg = opt & 0xF;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int16_t *H = (int16_t *)malloc(sizeof(int16_t) * (size_t)(F + 1));
        int16_t *Fv = (int16_t *)malloc(sizeof(int16_t) * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            for (int j = 0; j <= F; ++j) { H[j] = 0; Fv[j] = NEG_INF; }
            int16_t best = 0;
            for (int i = 0; i < R; ++i) {
                const int16_t *srow = tab[g_class[rd[i]]];
                int16_t hdiag = H[0];           /* H(i-1, 0) */
                int16_t hleft = 0;              /* H(i, 0)   */
                int16_t e = NEG_INF;
                for (int j = 0; j < F; ++j) {
                    int16_t hup = H[j + 1];
                    e = max16(sat_add(e, eR), sat_add(hleft, oR));
                    int16_t f = max16(sat_add(Fv[j + 1], eF), sat_add(hup, oF));
                    int16_t h = max16(max16((int16_t)(hdiag + srow[g_class[rf[j]]]), e), f);
                    if (alg == 0) { h = max16(h, 0); best = max16(best, h); }
                    Fv[j + 1] = f;
                    hdiag = hup;
                    H[j + 1] = h;
                    hleft = h;
                }
                H[0] = 0;
                if (alg == 1) best = max16(best, H[F]);
            }
            if (alg == 1) for (int j = 0; j <= F; ++j) best = max16(best, H[j]);
            scores[p] = best;
        }
        free(Fv);
        free(H);
    }
    return n;
}


/* The same recurrence on int32 cells (extension of the extension): for (shape, scoring) whose cells would leave
 * int16 -- long reads.  "Minus infinity" is far below any reachable cell; the result saturates at 32767, the
 * largest score the ABI's short can carry.  Inside the int16 range it equals vref_score_affine.               */
int vref_score_affine_wide(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
                           const vref_scoring *sc, int16_t *scores, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int oR = sc->open_read, eR = sc->ext_read, oF = sc->open_ref, eF = sc->ext_ref;
    const int alg = opt & 0xF;
    const int32_t ninf = -(1 << 29);
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *Fv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            for (int j = 0; j <= F; ++j) { H[j] = 0; Fv[j] = ninf; }
            int32_t best = 0;
            for (int i = 0; i < R; ++i) {
                const int16_t *srow = tab[g_class[rd[i]]];
                int32_t hdiag = H[0], hleft = 0, e = ninf;
                for (int j = 0; j < F; ++j) {
                    const int32_t hup = H[j + 1];
                    e = (e + eR > hleft + oR) ? e + eR : hleft + oR;
                    const int32_t f = (Fv[j + 1] + eF > hup + oF) ? Fv[j + 1] + eF : hup + oF;
                    int32_t h = hdiag + srow[g_class[rf[j]]];
                    if (e > h) h = e;
                    if (f > h) h = f;
                    if (alg == 0) { if (h < 0) h = 0; if (h > best) best = h; }
                    Fv[j + 1] = f;
                    hdiag = hup;
                    H[j + 1] = h;
                    hleft = h;
                }
                H[0] = 0;
                if (alg == 1 && H[F] > best) best = H[F];
            }
            if (alg == 1) for (int j = 0; j <= F; ++j) if (H[j] > best) best = H[j];
            scores[p] = (int16_t)(best > 32767 ? 32767 : best);
        }
        free(Fv);
        free(H);
    }
    return n;
}


/* ---- affine-gap alignments (extension, no reference counterpart) ----
 * Three-state Gotoh traceback whose tie-breaks are chosen so that open == extend == g walks
 * exactly the path of the linear model above (the only reference-pinned case):
 *   at H(i,j):  START if H == 0 (SW)  >  DIAG if H == H(i-1,j-1)+S  >  F (gap in the ref, UP)  >  E (LEFT)
 *   at F(i,j):  emit (read[i], '-'); it was OPENED from H(i-1,j) if F == H(i-1,j)+open_ref (preferred
 *               on ties), else extended from F(i-1,j)
 *   at E(i,j):  emit ('-', ref[j]); opened from H(i,j-1) if E == H(i,j-1)+open_read (preferred), else
 *               extended from E(i,j-1)
 * Borders: SW all 0; NW variant H(0,j) = 0 (START), H(i,0) = open_ref + (i-1)*ext_ref (UP all the
 * way), E(i,0) = F(0,j) = NEG_INF.  End cells as in the linear model.                              */
typedef struct {
    uint8_t h;      /* 0 START, 1 DIAG, 2 UP (came from F), 3 LEFT (came from E) */
    uint8_t f_ext;  /* F(i,j) extended F(i-1,j) (1) or opened from H(i-1,j) (0)   */
    uint8_t e_ext;  /* E(i,j) extended E(i,j-1) (1) or opened from H(i,j-1) (0)   */
} aff_ptr;

/* ---- second tie-break policy: the reference's SSE2 / AVX2 kernels ----
 * Same cell values, different pointers (src/Kernels/AVX-SSE/SSEKernel.cpp:366-379, 646-659):
 *   pointer = DIAG if the cell equals diag+S AND both bases are in ACGT, else LEFT if it equals
 *   left+gap_read, else UP if it equals up+gap_ref, else START -- there is no "cell == 0 -> START"
 *   rule, so a Smith-Waterman traceback walks through zero cells, and a cell whose only source
 *   is a diagonal over a non-ACGT base stops the traceback.  For the NW variant's end cell a
 *   base is "invalid" when it is not in ACGT (N counts as invalid; SSEKernel.cpp:532-536,
 *   673-677), otherwise the bookkeeping is the Default kernel's.  Row 0 is START, column 0 is
 *   UP (NW) / START (SW).                                                                    */
static int valid_acgt(uint8_t ch) { const int c = g_class[ch]; return c >= 1 && c <= 4; }


#define VMAX(a, b) ((a) > (b) ? (a) : (b))
#define VCELL int16_t
#define VNAME(x) x
#define VADD(a, b) sat_add(a, b)
#define VNINF NEG_INF
#define VMIN INT16_MIN
#include "cpu_ref_fills.inc"
#undef VCELL
#undef VNAME
#undef VADD
#undef VNINF
#undef VMIN
/* int32 cells: vref_align_affine_wide, vref_align_sse_wide */
#define VCELL int32_t
#define VNAME(x) x##_wide
#define VADD(a, b) ((a) + (b))
#define VNINF (-(1 << 29))
#define VMIN INT32_MIN
#include "cpu_ref_fills.inc"
#undef VCELL
#undef VNAME
#undef VADD
#undef VNINF
#undef VMIN

/* ---- banded Smith-Waterman score (extension: the reference has no banding) ----
 * DEFINITION (include/valign_hip.h, "band_width"): the read's rows are taken in blocks of `block_rows`
 * consecutive rows, the LAST block ending with the last row of the read (so a short first block when
 * block_rows does not divide R).  A block computes the columns
 *      [ floor(r_first * F / R) - w ,  floor(r_last * F / R) + w ]        (w = band_width / 2)
 * of its rows r_first .. r_last, clipped to the matrix, the lower end rounded down to a multiple of
 * `col_align`; every other cell counts as 0 and cannot hold the maximum.
 *   block_rows = 1, col_align = 1 ..... the per-cell band |j - floor(i * F / R)| <= w
 *   block_rows = 160, col_align = 4 ... what libHIPKernel.so computes (VALIGN_HIP_BAND_BLOCK_ROWS /
 *                                       VALIGN_HIP_BAND_COL_ALIGN): a superset of the per-cell band, so its
 *                                       score lies between the per-cell-band score and the unbanded one.
 * band_half < 0 computes every cell.  Cells are int32 here and the result saturates at 32767 (the ABI's
 * short), like vref_score_wide; inside the int16 range that is the reference's arithmetic.               */
static void band_columns(int b, int R, int F, int block_rows, int col_align, int pad, int w, int *lo, int *hi) {
    if (w < 0 || R <= 0) { *lo = 0; *hi = F - 1; return; }
    int r_lo = b * block_rows - pad, r_hi = (b + 1) * block_rows - pad - 1;
    if (r_lo < 0) r_lo = 0;
    if (r_hi > R - 1) r_hi = R - 1;
    long long l = (long long)r_lo * F / R - w, h = (long long)r_hi * F / R + w;
    if (l < 0) l = 0;
    *lo = (int)(l - l % col_align);
    *hi = (int)(h > F - 1 ? F - 1 : h);
}

int vref_score_banded_sw(int n, int R, int F, const uint8_t *reads, const uint8_t *refs, const vref_scoring *sc,
                         int block_rows, int col_align, int band_half, int16_t *scores, int threads) {
    class_init();
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int gr = sc->gap_read, gf = sc->gap_ref;
    if (block_rows < 1) block_rows = 1;
    if (col_align < 1) col_align = 1;
    const int blocks = R > 0 ? (R + block_rows - 1) / block_rows : 1;
    const int pad = blocks * block_rows - R;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            memset(prev, 0, sizeof(int32_t) * (size_t)(F + 1));
            int32_t best = 0;
            for (int i = 0; i < R; ++i) {
                int lo, hi;
                band_columns((i + pad) / block_rows, R, F, block_rows, col_align, pad, band_half, &lo, &hi);
                memset(cur, 0, sizeof(int32_t) * (size_t)(F + 1));
                const int16_t *srow = tab[g_class[rd[i]]];
                for (int j = lo; j <= hi; ++j) {
                    int32_t h = prev[j] + srow[g_class[rf[j]]];
                    if (prev[j + 1] + gf > h) h = prev[j + 1] + gf;
                    if (cur[j] + gr > h) h = cur[j] + gr;
                    if (h < 0) h = 0;
                    cur[j + 1] = h;
                    if (h > best) best = h;
                }
                int32_t *t = prev; prev = cur; cur = t;
            }
            scores[p] = (int16_t)(best > 32767 ? 32767 : best);
        }
        free(cur);
        free(prev);
    }
    return n;
}

/* The same band with affine gaps (Gotoh recurrence as vref_score_affine_wide; cells outside the band count as
 * H = 0, E = F = 0 -- Smith-Waterman's own floor -- and cannot hold the maximum).                              */
int vref_score_banded_sw_affine(int n, int R, int F, const uint8_t *reads, const uint8_t *refs, const vref_scoring *sc,
                                int block_rows, int col_align, int band_half, int16_t *scores, int threads) {
    class_init();
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int oR = sc->open_read, eR = sc->ext_read, oF = sc->open_ref, eF = sc->ext_ref;
    if (block_rows < 1) block_rows = 1;
    if (col_align < 1) col_align = 1;
    const int blocks = R > 0 ? (R + block_rows - 1) / block_rows : 1;
    const int pad = blocks * block_rows - R;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *fprev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *fcur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            memset(prev, 0, sizeof(int32_t) * (size_t)(F + 1));
            memset(fprev, 0, sizeof(int32_t) * (size_t)(F + 1));
            int32_t best = 0;
            for (int i = 0; i < R; ++i) {
                int lo, hi;
                band_columns((i + pad) / block_rows, R, F, block_rows, col_align, pad, band_half, &lo, &hi);
                memset(cur, 0, sizeof(int32_t) * (size_t)(F + 1));
                memset(fcur, 0, sizeof(int32_t) * (size_t)(F + 1));
                const int16_t *srow = tab[g_class[rd[i]]];
                int32_t e = 0;
                for (int j = lo; j <= hi; ++j) {
                    e = (e + eR > cur[j] + oR) ? e + eR : cur[j] + oR;
                    if (e < 0) e = 0;
                    int32_t f = (fprev[j + 1] + eF > prev[j + 1] + oF) ? fprev[j + 1] + eF : prev[j + 1] + oF;
                    if (f < 0) f = 0;
                    int32_t h = prev[j] + srow[g_class[rf[j]]];
                    if (e > h) h = e;
                    if (f > h) h = f;
                    if (h < 0) h = 0;
                    cur[j + 1] = h;
                    fcur[j + 1] = f;
                    if (h > best) best = h;
                }
                int32_t *t = prev; prev = cur; cur = t;
                t = fprev; fprev = fcur; fcur = t;
            }
            scores[p] = (int16_t)(best > 32767 ? 32767 : best);
        }
        free(fcur); free(fprev); free(cur); free(prev);
    }
    return n;
}

/* ---- int32 cells (extension): the same two score recurrences without the reference's int16
 * wrap-around, for (shape, scoring) whose cells leave int16; results saturate at 32767, the
 * largest score the ABI's short can carry.                                                   */
int vref_score_wide(int opt, int n, int R, int F, const uint8_t *reads, const uint8_t *refs,
                    const vref_scoring *sc, int16_t *scores, int threads) {
    class_init();
    if ((opt & 0xF) > 1) return 0;
    int16_t tab[6][6];
    subst_init(sc, tab);
    const int gr = sc->gap_read, gf = sc->gap_ref, alg = opt & 0xF;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
        int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F + 1));
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            const uint8_t *rd = reads + (size_t)p * R, *rf = refs + (size_t)p * F;
            memset(prev, 0, sizeof(int32_t) * (size_t)(F + 1));
            cur[0] = 0;
            int32_t best = 0;
            for (int i = 0; i < R; ++i) {
                const int16_t *srow = tab[g_class[rd[i]]];
                for (int j = 0; j < F; ++j) {
                    int32_t h = prev[j] + srow[g_class[rf[j]]];
                    if (prev[j + 1] + gf > h) h = prev[j + 1] + gf;
                    if (cur[j] + gr > h) h = cur[j] + gr;
                    if (alg == 0) { if (h < 0) h = 0; if (h > best) best = h; }
                    cur[j + 1] = h;
                }
                if (alg == 1 && cur[F] > best) best = cur[F];
                int32_t *t = prev; prev = cur; cur = t;
                cur[0] = 0;
            }
            if (alg == 1) for (int j = 0; j <= F; ++j) if (prev[j] > best) best = prev[j];
            scores[p] = (int16_t)(best > 32767 ? 32767 : best);
        }
        free(cur);
        free(prev);
    }
    return n;
}

int vref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
