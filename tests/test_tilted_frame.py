"""The NW kernels' tilted frame, restated on the CPU (no GPU, no HIP code involved): cell (p, j) kept as
V - g_ref * p - g_read * j (affine: the extension scores) turns every gap step / extension into "take the neighbour as
it is" and charges the diagonal step for both -- DESIGN.md section 3.  This file checks the algebra the kernels rely on,
with plain Python loops on small matrices: the frame's recurrence (no gap constant left in it) un-tilts to exactly the
plain recurrence's values, picks the same candidate in every cell under the same tie-breaks (so pointers are the
frame's pointers), and the NW-variant score read out of the frame (last row, last column, zero floor) is the oracle's.
"""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import synth

NEG = -10 ** 6


_CLASS = {ord(c): k for k, cs in ((1, "Aa"), (2, "Tt"), (3, "Cc"), (4, "Gg"), (5, "Nn")) for c in cs}      # DefaultKernel.h:43-60


def _sub(a, b, match, mismatch):
    ca, cb = _CLASS.get(int(a), 0), _CLASS.get(int(b), 0)
    if 1 <= ca <= 4 and 1 <= cb <= 4:
        return match if ca == cb else mismatch
    return 0


def _linear_plain(read, ref, match, mismatch, g_read, g_ref, borders_zero=True):
    R, F = len(read), len(ref)
    H = np.zeros((R + 1, F + 1), dtype=np.int64)
    P = np.zeros((R + 1, F + 1), dtype=np.int64)
    if not borders_zero:
        for i in range(1, R + 1):
            H[i, 0] = i * g_ref
    for i in range(1, R + 1):
        for j in range(1, F + 1):
            cands = (H[i - 1, j - 1] + _sub(read[i - 1], ref[j - 1], match, mismatch), H[i - 1, j] + g_ref, H[i, j - 1] + g_read)
            H[i, j] = max(cands)
            P[i, j] = cands.index(H[i, j])                     # first wins: DIAG > UP > LEFT
    return H, P


def _linear_tilted(read, ref, match, mismatch, g_read, g_ref, borders_zero=True):
    R, F = len(read), len(ref)
    tr, tc = -g_ref, -g_read                                   # what a row / a column adds
    T = np.zeros((R + 1, F + 1), dtype=np.int64)
    P = np.zeros((R + 1, F + 1), dtype=np.int64)
    for j in range(F + 1):
        T[0, j] = tc * j                                       # the zero row, tilted
    for i in range(1, R + 1):
        T[i, 0] = (0 if borders_zero else i * g_ref) + tr * i
    for i in range(1, R + 1):
        for j in range(1, F + 1):
            cands = (T[i - 1, j - 1] + _sub(read[i - 1], ref[j - 1], match, mismatch) + tr + tc, T[i - 1, j], T[i, j - 1])
            T[i, j] = max(cands)
            P[i, j] = cands.index(T[i, j])
    rows, cols = np.arange(R + 1)[:, None], np.arange(F + 1)[None, :]
    return T - tr * rows - tc * cols, P


def _affine(read, ref, match, mismatch, o_read, e_read, o_ref, e_ref, tilted):
    """Gotoh, NW-variant score borders (H zero on row 0 and column 0, gap matrices minus infinity there).  Returns H
    and a per-cell code (source of H: 0 DIAG, 1 F, 2 E; E opened; F opened) with the kernels' tie-breaks."""
    R, F = len(read), len(ref)
    tr, tc = (-e_ref, -e_read) if tilted else (0, 0)
    H = np.zeros((R + 1, F + 1), dtype=np.int64)
    E = np.full((R + 1, F + 1), NEG, dtype=np.int64)
    Fm = np.full((R + 1, F + 1), NEG, dtype=np.int64)
    code = np.zeros((R + 1, F + 1, 3), dtype=np.int64)
    rows, cols = np.arange(R + 1)[:, None], np.arange(F + 1)[None, :]
    H += tr * rows + tc * cols
    for i in range(1, R + 1):
        for j in range(1, F + 1):
            if tilted:                                          # extensions are free, an opening costs open - extend
                e_ext, e_opn = E[i, j - 1], H[i, j - 1] + o_read - e_read
                f_ext, f_opn = Fm[i - 1, j], H[i - 1, j] + o_ref - e_ref
                d = H[i - 1, j - 1] + _sub(read[i - 1], ref[j - 1], match, mismatch) + tr + tc
            else:
                e_ext, e_opn = E[i, j - 1] + e_read, H[i, j - 1] + o_read
                f_ext, f_opn = Fm[i - 1, j] + e_ref, H[i - 1, j] + o_ref
                d = H[i - 1, j - 1] + _sub(read[i - 1], ref[j - 1], match, mismatch)
            E[i, j] = max(e_ext, e_opn)
            Fm[i, j] = max(f_ext, f_opn)
            cands = (d, Fm[i, j], E[i, j])                      # DIAG > F > E
            H[i, j] = max(cands)
            code[i, j] = (cands.index(H[i, j]), int(e_opn >= e_ext), int(f_opn >= f_ext))      # ties: opened
    return H - tr * rows - tc * cols, code


def _nw_variant_score(H):
    return max(0, int(H[1:, -1].max()), int(H[-1, :].max()))


@pytest.mark.parametrize("seed", range(6))
def test_linear_frame_is_the_plain_recurrence(seed):
    rng = np.random.default_rng(seed)
    for case in range(25):
        R, F = int(rng.integers(1, 14)), int(rng.integers(1, 19))
        reads, refs = synth.make_pairs(1, R, F, seed=1000 * seed + case, indel_rate=0.05, n_run_frac=0.3, short_frac=0.3)
        match, mismatch = int(rng.integers(0, 6)), -int(rng.integers(0, 5))
        g_read, g_ref = -int(rng.integers(0, 6)), -int(rng.integers(0, 6))
        for borders_zero in (True, False):                      # score variant / alignment variant (column 0 = i * gap_ref)
            H, P = _linear_plain(reads[0], refs[0], match, mismatch, g_read, g_ref, borders_zero)
            Ht, Pt = _linear_tilted(reads[0], refs[0], match, mismatch, g_read, g_ref, borders_zero)
            assert np.array_equal(H, Ht) and np.array_equal(P, Pt)
        sc = cpu_ref.Scoring.make(match, mismatch, g_read, g_ref)
        assert _nw_variant_score(_linear_tilted(reads[0], refs[0], match, mismatch, g_read, g_ref)[0]) == int(cpu_ref.score(1, reads, refs, sc)[0])


@pytest.mark.parametrize("seed", range(6))
def test_affine_frame_is_the_plain_recurrence(seed):
    rng = np.random.default_rng(100 + seed)
    for case in range(25):
        R, F = int(rng.integers(1, 13)), int(rng.integers(1, 17))
        reads, refs = synth.make_pairs(1, R, F, seed=5000 + 1000 * seed + case, indel_rate=0.05, n_run_frac=0.3, short_frac=0.3)
        match, mismatch = int(rng.integers(0, 6)), -int(rng.integers(0, 5))
        o_read, o_ref = -int(rng.integers(0, 9)), -int(rng.integers(0, 9))
        e_read, e_ref = max(-int(rng.integers(0, 4)), o_read), max(-int(rng.integers(0, 4)), o_ref)      # extend >= open
        args = (reads[0], refs[0], match, mismatch, o_read, e_read, o_ref, e_ref)
        H, code = _affine(*args, tilted=False)
        Ht, codet = _affine(*args, tilted=True)
        assert np.array_equal(H, Ht) and np.array_equal(code, codet)
        sc = cpu_ref.Scoring.make(match, mismatch, -3, -3, o_read, e_read, o_ref, e_ref)
        assert _nw_variant_score(Ht) == int(cpu_ref.score(1, reads, refs, sc, affine=True)[0])
