#!/usr/bin/env python3
"""Generates tests/golden/affine/*.npz: tiny pairs whose affine-gap optimum was found by EXHAUSTIVE
ENUMERATION of every alignment (tests/enumerate_alignments.py -- no dynamic program involved).

The reference has no affine model, so neither its kernels nor its fixtures can pin
`open != extend`.  These fixtures pin it independently of the hand that wrote oracle/cpu_ref.c and the HIP
kernels: for every pair and scoring set they hold
  sw_score / nw_score ... the enumerated optimum (Smith-Waterman; the reference's NW-variant score rule)
  rows_* / idx_* ........ the alignments the oracle emits, stored only after this script has checked that
                          they re-score to the enumerated value of their end cell, spell the right
                          substrings, and end in the cell the reference's end-cell rules select from the
                          ENUMERATED cell values (first row-major maximum; last valid row + row arg-max).
Scorings keep |extend| <= |open| per direction: with a dearer extension the Gotoh recurrence re-opens a gap
instead of extending it (H may itself come from E), so it no longer scores maximal runs -- found by this very
enumeration on (open -1, extend -3); the engine refuses such scorings (hip_engine.hip.h, validate_scoring).
The script fails instead of writing a fixture the enumeration does not confirm.  tests/test_affine_enumeration.py
re-checks the oracle against them on CPU (and re-enumerates a sample), tests/test_gpu_affine_golden.py runs
the HIP path against them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import enumerate_alignments as en  # noqa: E402
from oracle import cpu_ref  # noqa: E402

OUT = os.path.join(HERE, "affine")

# match, mismatch, open_read, ext_read, open_ref, ext_ref
SCORINGS = [
    (2, -1, -5, -1, -5, -1),       # BASELINE configs 2-4
    (2, -1, -3, -3, -3, -3),       # open == extend: the reference's linear model
    (2, -1, -6, -2, -4, -1),       # asymmetric directions
    (3, -2, -2, -2, -7, -1),       # linear in one direction, affine in the other
    (1, -1, -3, -1, -2, -2),       # unit scores, linear in one direction
    (5, -4, -4, -1, -3, -2),       # gaps cheap against a match: optima with long gaps at these sizes
    (4, -6, -3, -1, -5, -1),
    (6, -5, -7, -1, -2, -2),
]
SHAPES = [(1, 1), (2, 3), (3, 3), (4, 4), (5, 4), (5, 5), (6, 6), (4, 7), (7, 4), (5, 8)]
PAIRS_PER_SHAPE = 40


def make_inputs(R, F, n, rng):
    """Related pairs over ACGT with N, lower case, NUL padding and a junk byte mixed in."""
    reads = np.zeros((n, R), np.uint8)
    refs = np.zeros((n, F), np.uint8)
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    for p in range(n):
        ref = alphabet[rng.integers(0, 2 + (p % 3), F)]          # small alphabets give many ties and repeats
        off = rng.integers(0, max(1, F - R + 1))
        read = np.resize(ref[off:off + R], R).copy()
        for k in range(R):
            u = rng.random()
            if u < 0.12:
                read[k] = alphabet[rng.integers(0, 4)]
            elif u < 0.32:
                read[k] = ord("N")
            elif u < 0.36:
                read[k] |= 0x20
        if p % 3 == 0 and F >= R + 2 and R >= 3:                     # the read skips a block of the reference
            cut = rng.integers(1, R)
            read = np.concatenate([ref[:cut], ref[cut + F - R:]])[:R].copy()
        elif p % 3 == 1 and R >= F + 2 and F >= 3:                   # the read carries a block the reference lacks
            cut = rng.integers(1, F)
            read = np.concatenate([ref[:cut], alphabet[rng.integers(0, 4, R - F)], ref[cut:]])[:R].copy()
        elif p % 3 == 2 and R >= 5 and F >= 5:                       # a two-base block on either side
            cut = rng.integers(1, R - 2)
            read = np.concatenate([read[:cut], read[cut + 2:], alphabet[rng.integers(0, 4, 2)]])[:R].copy()
        if rng.random() < 0.3 and R > 1:                             # an indel
            k = rng.integers(0, R)
            read = np.delete(read, k)
            read = np.append(read, alphabet[rng.integers(0, 4)])
        if rng.random() < 0.2:
            ref[rng.integers(0, F)] = ord("N")
        if rng.random() < 0.15:
            read[rng.integers(0, R):] = 0                            # short read, NUL padded
        if rng.random() < 0.15:
            ref[rng.integers(0, F):] = 0
        if rng.random() < 0.05:
            read[rng.integers(0, R)] = ord("#")
        reads[p], refs[p] = read, ref
    return reads, refs


def check_pair(read, ref, sc, rows_sw, idx_sw, rows_nw, idx_nw):
    """Enumerate; compare the oracle's alignments with what the enumeration says.  -> (sw, nw) optimum."""
    R, F = len(read), len(ref)
    AL = R + F
    sw = en.sw_score(read, ref, sc)
    nw = en.nw_variant_score(read, ref, sc)
    # Smith-Waterman alignment: ends in the first row-major cell holding the maximum, re-scores to it
    cells = en.sw_cells(read, ref, sc)
    s = int(idx_sw[0])
    a, b = bytes(rows_sw[0, s:AL - 1]), bytes(rows_sw[1, s:AL - 1])
    assert en.rescore_rows(a, b, sc) == sw, ("SW rows do not re-score to the optimum", read, ref, sc, a, b)
    if sw > 0:
        end = next((i, j) for i in range(1, R + 1) for j in range(1, F + 1) if cells[(i, j)] == sw)
        ra, rb = en.ungapped(a), en.ungapped(b)
        assert bytes(read[end[0] - len(ra):end[0]]) == ra and bytes(ref[end[1] - len(rb):end[1]]) == rb, \
            ("SW alignment does not end in the first maximal cell", read, ref, sc, end, a, b)
    else:
        assert a == b"" and b == b""
    # NW-variant alignment: end cell by the reference's rule over the ENUMERATED cell values
    cells = en.nw_variant_align_cells(read, ref, sc)
    ei, ej = en.nw_variant_end_cell(read, ref, cells)
    s = int(idx_nw[0])
    a, b = bytes(rows_nw[0, s:AL - 1]), bytes(rows_nw[1, s:AL - 1])
    ra, rb = en.ungapped(a), en.ungapped(b)
    assert ra == bytes(read[:ei + 1]), ("NW alignment must spell the read up to its last valid base", read, ref, sc, a)
    assert rb == bytes(ref[ej + 1 - len(rb):ej + 1]), ("NW alignment must end at the rule's column", read, ref, sc, ej, b)
    assert en.rescore_rows(a, b, sc) == cells[(ei + 1, ej + 1)], ("NW rows do not re-score to their end cell", read, ref, sc, a, b)
    return sw, nw


def discrimination(reads, refs, sc, sw, nw):
    """How many pairs tell the stated model from its likeliest mis-statements: directions swapped, every gap
    base at the open score, every gap base at the extension score (SW optimum, NW-variant optimum)."""
    alts = {"swapped": (sc[0], sc[1], sc[4], sc[5], sc[2], sc[3]), "all_open": (sc[0], sc[1], sc[2], sc[2], sc[4], sc[4]),
            "all_extend": (sc[0], sc[1], sc[3], sc[3], sc[5], sc[5])}
    out = []
    for name in ("swapped", "all_open", "all_extend"):
        a = alts[name]
        out.append(sum(int(en.sw_score(reads[p], refs[p], a) != sw[p]) + int(en.nw_variant_score(reads[p], refs[p], a) != nw[p])
                       for p in range(reads.shape[0])))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    cpu_ref.build()
    told_apart = np.zeros((len(SCORINGS), 3), np.int64)
    for R, F in SHAPES:
        rng = np.random.default_rng(1000 * R + F)
        reads, refs = make_inputs(R, F, PAIRS_PER_SHAPE, rng)
        n = reads.shape[0]
        out = {"reads": reads, "refs": refs, "scorings": np.array(SCORINGS, np.int32)}
        for s, sc in enumerate(SCORINGS):
            osc = cpu_ref.Scoring.make(sc[0], sc[1], sc[2], sc[4], sc[2], sc[3], sc[4], sc[5])
            rows_sw, idx_sw = cpu_ref.align(0, reads, refs, osc, affine=True)
            rows_nw, idx_nw = cpu_ref.align(1, reads, refs, osc, affine=True)
            sw = np.zeros(n, np.int16)
            nw = np.zeros(n, np.int16)
            for p in range(n):
                sw[p], nw[p] = check_pair(reads[p], refs[p], sc, rows_sw[p], idx_sw[p], rows_nw[p], idx_nw[p])
            assert np.array_equal(cpu_ref.score(0, reads, refs, osc, affine=True), sw), ("oracle SW affine score != enumeration", R, F, sc)
            assert np.array_equal(cpu_ref.score(1, reads, refs, osc, affine=True), nw), ("oracle NW affine score != enumeration", R, F, sc)
            told_apart[s] += discrimination(reads, refs, sc, sw, nw)
            out.update({"sw_score_%d" % s: sw, "nw_score_%d" % s: nw, "rows_sw_%d" % s: rows_sw, "idx_sw_%d" % s: idx_sw,
                        "rows_nw_%d" % s: rows_nw, "idx_nw_%d" % s: idx_nw})
        np.savez_compressed(os.path.join(OUT, "affine_enum_%dx%d.npz" % (R, F)), **out)
        print("affine_enum_%dx%d.npz: %d pairs x %d scoring sets confirmed by enumeration" % (R, F, n, len(SCORINGS)), flush=True)
    # the fixtures must be able to tell the model from its mis-statements, or they pin nothing.  Gaps of two
    # and more bases only enter an optimum of pairs this small when gaps are cheap against a match, so the
    # "every base costs open" mis-statement is told apart by the cheap-gap scorings (and, at sizes enumeration
    # cannot reach, by the general-gap-function cross-check of tests/test_affine_enumeration.py).
    for s, sc in enumerate(SCORINGS):
        print("scoring %s: optima that differ under swapped directions / all-open / all-extend: %s" % (sc, told_apart[s].tolist()))
        if sc[2] != sc[3] or sc[4] != sc[5]:
            assert told_apart[s][2] > 0, ("no pair shows that opening a gap costs more than extending it", sc)
    assert sum(1 for s in range(len(SCORINGS)) if told_apart[s][1] > 0) >= 3, "too few scorings tell open from extend"
    assert sum(1 for s in range(len(SCORINGS)) if told_apart[s][0] > 0) >= 3, "too few scorings tell the gap directions apart"
    np.save(os.path.join(OUT, "told_apart.npy"), told_apart)


if __name__ == "__main__":
    main()
