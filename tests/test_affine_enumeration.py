"""The affine-gap extension pinned independently of its authors: tests/golden/affine/*.npz
hold tiny pairs whose optimum was found by EXHAUSTIVE ENUMERATION of every alignment (no dynamic program:
tests/enumerate_alignments.py, generator tests/golden/make_affine_golden.py).  The reference has no affine
model, so this is the only judge of `open != extend` that does not share code or authorship
of idea with oracle/cpu_ref.c's Gotoh restatement."""
import glob
import os

import numpy as np
import pytest

import enumerate_alignments as en
from oracle import cpu_ref

from conftest import ROOT

FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "affine", "affine_enum_*.npz")))


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _oracle_scoring(sc):
    return cpu_ref.Scoring.make(int(sc[0]), int(sc[1]), int(sc[2]), int(sc[4]), int(sc[2]), int(sc[3]), int(sc[4]), int(sc[5]))


def test_fixtures_exist_and_discriminate():
    assert len(FILES) >= 8
    told = np.load(os.path.join(ROOT, "tests", "golden", "affine", "told_apart.npy"))
    g = _load(FILES[0])
    # per scoring set: optima that change when the gap directions are swapped / every gap base costs the open
    # score / every gap base costs the extension score.  Gaps of several bases only enter optima of pairs this
    # small when gaps are cheap against a match (the larger shapes of the general-gap-function test cover the rest)
    for s, sc in enumerate(g["scorings"]):
        if sc[2] != sc[3] or sc[4] != sc[5]:
            assert told[s][2] > 0
    assert (told[:, 1] > 0).sum() >= 3 and (told[:, 0] > 0).sum() >= 3


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_equals_the_enumerated_optimum(path):
    g = _load(path)
    for s, sc in enumerate(g["scorings"]):
        osc = _oracle_scoring(sc)
        assert np.array_equal(cpu_ref.score(0, g["reads"], g["refs"], osc, affine=True), g["sw_score_%d" % s]), (path, sc)
        assert np.array_equal(cpu_ref.score(1, g["reads"], g["refs"], osc, affine=True), g["nw_score_%d" % s]), (path, sc)
        for opt, tag in ((0, "sw"), (1, "nw")):
            rows, idx = cpu_ref.align(opt, g["reads"], g["refs"], osc, affine=True)
            assert np.array_equal(rows, g["rows_%s_%d" % (tag, s)]) and np.array_equal(idx, g["idx_%s_%d" % (tag, s)])


@pytest.mark.parametrize("path", [p for p in FILES if any(t in p for t in ("3x3", "4x4", "5x5", "4x7"))],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_fixture_values_are_what_enumeration_finds(path):
    """Re-enumerate a sample live, so the stored numbers are not taken on trust; emitted rows must re-score to
    the optimum under the model as a user states it (maximal run of k gap bases: open + (k - 1) extend)."""
    g = _load(path)
    AL = g["reads"].shape[1] + g["refs"].shape[1]
    for s, sc in enumerate(g["scorings"]):
        sc = tuple(int(v) for v in sc)
        for p in range(0, g["reads"].shape[0], 5):
            read, ref = g["reads"][p], g["refs"][p]
            assert en.sw_score(read, ref, sc) == g["sw_score_%d" % s][p]
            assert en.nw_variant_score(read, ref, sc) == g["nw_score_%d" % s][p]
            start = int(g["idx_sw_%d" % s][p, 0])
            a, b = bytes(g["rows_sw_%d" % s][p, 0, start:AL - 1]), bytes(g["rows_sw_%d" % s][p, 1, start:AL - 1])
            assert en.rescore_rows(a, b, sc) == g["sw_score_%d" % s][p]


def test_gotoh_needs_extend_not_dearer_than_open():
    """Found by the enumeration: with extension dearer than opening the recurrence re-opens instead of
    extending, so it no longer scores maximal runs.  The engine refuses such scorings
    (tests/test_gpu_affine_golden.py); this is the pair that shows why."""
    read = np.frombuffer(b"CCACC", np.uint8)
    ref = np.frombuffer(b"CCAAACCC", np.uint8)
    sc = (1, -1, -1, -3, -2, -2)
    cells = en.nw_variant_align_cells(read, ref, sc)
    rows, idx = cpu_ref.align(1, read[None, :], ref[None, :], _oracle_scoring(sc), affine=True)
    start = int(idx[0, 0])
    a, b = bytes(rows[0, 0, start:12]), bytes(rows[0, 1, start:12])
    ei, ej = en.nw_variant_end_cell(read, ref, cells)
    assert en.rescore_rows(a, b, sc) != cells[(ei + 1, ej + 1)]


@pytest.mark.parametrize("R,F,n,seed", [(12, 20, 60, 1), (20, 33, 40, 2), (33, 70, 24, 3), (40, 25, 24, 4)])
@pytest.mark.parametrize("sc", [(2, -1, -5, -1, -5, -1), (2, -1, -6, -2, -4, -1), (3, -2, -2, -2, -7, -1)])
def test_oracle_equals_the_general_gap_function_recurrence(R, F, n, seed, sc):
    """Sizes where optima carry gaps of many bases under the BASELINE scoring (open -5, extend -1), beyond the
    reach of enumeration: the oracle's Gotoh restatement against the Waterman-Smith-Beyer recurrence, which has no
    E / F state at all (every gap run is one term H(start) + open + (k - 1) extend) -- scores, end cells and
    the re-scored rows."""
    from versalignlib_amd import synth
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.08, n_run_frac=0.1, short_frac=0.1, lowercase_frac=0.05)
    rng = np.random.default_rng(seed)
    for p in range(0, n, 2):              # every other pair: the read skips / repeats a block of 2..6 reference bases
        k = int(rng.integers(2, 7))
        cut = int(rng.integers(2, max(3, min(R, F) - k - 2)))
        src = refs[p, :max(R, F) + k] if F >= R + k else np.resize(refs[p], R + k)
        reads[p] = np.concatenate([src[:cut], src[cut + k:]])[:R]
    osc = _oracle_scoring(sc)
    sw = cpu_ref.score(0, reads, refs, osc, affine=True)
    nw = cpu_ref.score(1, reads, refs, osc, affine=True)
    rows_sw, idx_sw = cpu_ref.align(0, reads, refs, osc, affine=True)
    rows_nw, idx_nw = cpu_ref.align(1, reads, refs, osc, affine=True)
    AL = R + F
    long_gaps = 0
    for p in range(n):
        read, ref = reads[p], refs[p]
        H = en.general_gap_cells(read, ref, sc, "sw")
        assert sw[p] == H.max(), (p, "SW")
        s = int(idx_sw[p, 0])
        a, b = bytes(rows_sw[p, 0, s:AL - 1]), bytes(rows_sw[p, 1, s:AL - 1])
        assert en.rescore_rows(a, b, sc) == sw[p]
        long_gaps += int(b"--" in a or b"--" in b)
        if sw[p] > 0:                      # ends in the first row-major cell holding the maximum
            ei, ej = np.unravel_index(np.argmax(H), H.shape)
            ra, rb = en.ungapped(a), en.ungapped(b)
            assert bytes(read[ei - len(ra):ei]) == ra and bytes(ref[ej - len(rb):ej]) == rb
        H = en.general_gap_cells(read, ref, sc, "nw_score")
        assert nw[p] == max(0, H[1:, F].max(), H[R, :].max()), (p, "NW")
        H = en.general_gap_cells(read, ref, sc, "nw_align")
        cells = {(i, j): int(H[i, j]) for i in range(R + 1) for j in range(F + 1)}
        ei, ej = en.nw_variant_end_cell(read, ref, cells)
        s = int(idx_nw[p, 0])
        a, b = bytes(rows_nw[p, 0, s:AL - 1]), bytes(rows_nw[p, 1, s:AL - 1])
        assert en.ungapped(a) == bytes(read[:ei + 1])
        rb = en.ungapped(b)
        assert rb == bytes(ref[ej + 1 - len(rb):ej + 1])
        assert en.rescore_rows(a, b, sc) == H[ei + 1, ej + 1], (p, "NW rows")
        long_gaps += int(b"--" in a or b"--" in b)
    assert long_gaps >= 5                  # the batch does put gaps of several bases into optimal alignments
