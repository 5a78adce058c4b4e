"""Exhaustive enumeration of alignments of tiny pairs -- an implementation-independent judge for the
affine-gap extension (the reference has no affine model, SURVEY.md F1, so nothing of the reference pins
`open != extend`; the repo's Gotoh restatement in oracle/cpu_ref.c was written by the same hand as the
kernels).  Nothing here is a dynamic program: every path of DIAG / UP / LEFT moves is walked and scored
column by column with the model as a user would state it:

  * a column with two bases scores `match` if both are the same one of A, C, G, T (any case),
    `mismatch` if both are one of those and differ, else 0 (substitution rule of the reference,
    src/Kernels/default/DefaultKernel.h:83-97);
  * a maximal run of k columns with '-' in the READ row (LEFT moves, the reference's score_gap_read
    direction) costs open_read + (k - 1) * ext_read; a run with '-' in the REF row (UP moves) costs
    open_ref + (k - 1) * ext_ref; runs of different kinds that touch are separate runs.

`best_ending_at(...)` returns, for every cell, the best score of any path from the given start cells, which
is all the score / end-cell rules of the reference need (they are restated below over those tables).
"""

MOVE_DIAG, MOVE_UP, MOVE_LEFT = 0, 1, 2


def base_class(ch):
    ch = ch & 0xDF if ch < 0x80 else 0
    return {ord("A"): 1, ord("T"): 2, ord("C"): 3, ord("G"): 4, ord("N"): 5}.get(ch, 0)


def substitution(a, b, match, mismatch):
    ca, cb = base_class(a), base_class(b)
    if 1 <= ca <= 4 and 1 <= cb <= 4:
        return match if ca == cb else mismatch
    return 0


def best_ending_at(read, ref, sc, starts):
    """sc = (match, mismatch, open_read, ext_read, open_ref, ext_ref).  starts: iterable of (i, j) matrix
    cells (0..R, 0..F) where a path may begin with score 0.  -> dict (i, j) -> best score of a path that
    ends there (the empty path counts for the start cells)."""
    R, F = len(read), len(ref)
    match, mismatch, open_read, ext_read, open_ref, ext_ref = sc
    best = {}

    def walk(i, j, last, score):
        if score > best.get((i, j), -10 ** 9):
            best[(i, j)] = score
        if i < R and j < F:
            walk(i + 1, j + 1, MOVE_DIAG, score + substitution(read[i], ref[j], match, mismatch))
        if i < R:
            walk(i + 1, j, MOVE_UP, score + (ext_ref if last == MOVE_UP else open_ref))
        if j < F:
            walk(i, j + 1, MOVE_LEFT, score + (ext_read if last == MOVE_LEFT else open_read))

    for (i, j) in starts:
        walk(i, j, MOVE_DIAG, 0)
    return best


def sw_score(read, ref, sc):
    """Smith-Waterman: best local alignment, the empty one scores 0."""
    R, F = len(read), len(ref)
    best = best_ending_at(read, ref, sc, [(i, j) for i in range(R + 1) for j in range(F + 1)])
    return max(0, max(best.values()))


def sw_cells(read, ref, sc):
    """H of every cell in Smith-Waterman semantics (floored at 0)."""
    R, F = len(read), len(ref)
    best = best_ending_at(read, ref, sc, [(i, j) for i in range(R + 1) for j in range(F + 1)])
    return {c: max(0, v) for c, v in best.items()}


def nw_variant_score(read, ref, sc):
    """The reference's "Needleman-Wunsch" score (DefaultKernel.cpp:140-202): row 0 and column 0 are 0 (a
    path may start anywhere on them), no floor, result = max(0, last column over all rows, last row over
    all columns)."""
    R, F = len(read), len(ref)
    starts = [(0, j) for j in range(F + 1)] + [(i, 0) for i in range(1, R + 1)]
    best = best_ending_at(read, ref, sc, starts)
    ends = [best[(i, F)] for i in range(1, R + 1) if (i, F) in best] + [best[(R, j)] for j in range(F + 1) if (R, j) in best]
    return max([0] + ends)


def nw_variant_align_cells(read, ref, sc):
    """H of every cell in the semantics of the reference's NW ALIGNMENT fill (DefaultKernel.cpp:282-389):
    row 0 is 0 (free start on any column), column 0 is a gap of i read bases -- i.e. paths start on row 0
    only, and (0, 0) followed by UP moves gives column 0."""
    F = len(ref)
    return best_ending_at(read, ref, sc, [(0, j) for j in range(F + 1)])


def rescore_rows(read_row, ref_row, sc):
    """Score of an emitted alignment (two gapped rows of equal length, bytes) under the model above."""
    match, mismatch, open_read, ext_read, open_ref, ext_ref = sc
    total, last = 0, MOVE_DIAG
    for a, b in zip(read_row, ref_row):
        if a == ord("-"):
            total += ext_read if last == MOVE_LEFT else open_read
            last = MOVE_LEFT
        elif b == ord("-"):
            total += ext_ref if last == MOVE_UP else open_ref
            last = MOVE_UP
        else:
            total += substitution(a, b, match, mismatch)
            last = MOVE_DIAG
    return total


def ungapped(row):
    return bytes(c for c in row if c != ord("-"))


def first_invalid(seq):
    """First position whose class is 0 (DefaultKernel.cpp:308-310, 348-350), else len(seq)."""
    for k, ch in enumerate(seq):
        if base_class(ch) == 0:
            return k
    return len(seq)


def nw_variant_end_cell(read, ref, cells):
    """End cell of the reference's NW alignment from a table of cell values:
    (last valid read position, min(last valid ref position, row arg-max of that row)), 0-based; the row
    arg-max starts from the column-0 value with index 0 and takes the first strictly greater cell."""
    R, F = len(read), len(ref)
    ir, jr = first_invalid(read), first_invalid(ref)
    row = ir if ir < R else R               # matrix row of the last valid read base (ir == R: last row)
    if ir == 0:
        arg = 0
    else:
        bestv, arg = cells[(row, 0)], 0
        for j0 in range(F):
            if cells[(row, j0 + 1)] > bestv:
                bestv, arg = cells[(row, j0 + 1)], j0
    return ir - 1, min(jr - 1, arg)


# ---- a second, structurally different statement of the model for sizes enumeration cannot reach ----

def general_gap_cells(read, ref, sc, mode):
    """Waterman-Smith-Beyer recurrence with an explicit gap-length cost w(k) = open + (k - 1) * extend: no E / F
    matrices, every gap run is one term `H(start of the run) + w(k)`.  O(R F (R + F)); numpy only.
    mode: "sw" (floor 0, borders 0), "nw_score" (borders 0, no floor), "nw_align" (row 0 = 0, column 0 = a gap of
    i read bases).  -> H as an (R + 1, F + 1) int64 array."""
    import numpy as np
    R, F = len(read), len(ref)
    match, mismatch, open_read, ext_read, open_ref, ext_ref = sc
    H = np.zeros((R + 1, F + 1), np.int64)
    if mode == "nw_align":
        for i in range(1, R + 1):
            H[i, 0] = open_ref + (i - 1) * ext_ref
    w_read = open_read + ext_read * np.arange(0, F + 1)          # w(k), k = index + 1
    w_ref = open_ref + ext_ref * np.arange(0, R + 1)
    for i in range(1, R + 1):
        for j in range(1, F + 1):
            best = H[i - 1, j - 1] + substitution(read[i - 1], ref[j - 1], match, mismatch)
            # a run of k LEFT moves ending here started in H(i, j - k); of k UP moves in H(i - k, j)
            best = max(best, int((H[i, j - 1::-1][:j] + w_read[:j]).max()), int((H[i - 1::-1, j][:i] + w_ref[:i]).max()))
            H[i, j] = max(best, 0) if mode == "sw" else best
    return H
