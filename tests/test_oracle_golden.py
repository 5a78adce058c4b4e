"""The oracle pinned against outputs of the reference's own compiled kernels.

tests/golden/*.npz were produced by tests/golden/make_golden.py from
libSSEKernel.so (full scores) and libDefaultKernel.so (score low byte, alignments) built
from the reference sources; the known-answer cases are SURVEY.md Appendix C."""
import numpy as np
import pytest

from oracle import cpu_ref

from golden_util import golden_files, load, rows_text


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_reproduces_reference_outputs(path):
    g = load(path)
    m, x, gr, gf = (int(v) for v in g["scoring"])
    sc = cpu_ref.Scoring.make(m, x, gr, gf)
    for opt, tag in ((0, "sw"), (1, "nw")):
        s = cpu_ref.score(opt, g["reads"], g["refs"], sc)
        assert np.array_equal(s, g["score_" + tag])                       # SSE: full int16
        assert np.array_equal((s & 0xFF).astype(np.uint8), g["lowbyte_" + tag])   # Default: low byte
        rows, idx = cpu_ref.align(opt, g["reads"], g["refs"], sc)
        assert np.array_equal(idx, g["idx_" + tag])
        assert np.array_equal(rows, g["rows_" + tag])


KAT_EXPECT = [  # SW score, SW rows, SW start, NW score, NW rows, NW start
    (8, (b"ACGT", b"ACGT"), 7, 8, (b"ACGT", b"ACGT"), 7),
    (8, (b"ACGT", b"ACGT"), 11, 8, (b"ACGT", b"ACGT"), 11),
    (6, (b"ACNT", b"ACGT"), 7, 6, (b"ACNT", b"ACGT"), 7),
    (0, (b"", b""), 11, 0, (b"AAAA", b"CCCC"), 7),
    (10, (b"TGACC", b"TGACC"), 12, 10, (b"ACGTTTGACC", b"ACG--TGACC"), 7),
    (4, (b"AT", b"AT"), 11, 3, (b"GATTACA", b"GCATGCT"), 6),
]


@pytest.mark.parametrize("k", range(6))
def test_known_answers(k):
    g = load([p for p in golden_files() if p.endswith("kat%d.npz" % (k + 1))][0])
    sw_score, sw_rows, sw_start, nw_score, nw_rows, nw_start = KAT_EXPECT[k]
    assert int(cpu_ref.score(0, g["reads"], g["refs"])[0]) == sw_score
    assert int(cpu_ref.score(1, g["reads"], g["refs"])[0]) == nw_score
    rows, idx = cpu_ref.align(0, g["reads"], g["refs"])
    assert rows_text(rows[0], idx[0]) == sw_rows and idx[0, 0] == sw_start == idx[0, 2]
    rows, idx = cpu_ref.align(1, g["reads"], g["refs"])
    assert rows_text(rows[0], idx[0]) == nw_rows and idx[0, 0] == nw_start
    R, F = g["reads"].shape[1], g["refs"].shape[1]
    assert idx[0, 1] == R + F - 1 == idx[0, 3]


def test_affine_degenerates_to_linear():
    """Affine extension is unpinned by the reference; open == extend must equal linear."""
    from versalignlib_amd import synth
    for R, F, seed in ((20, 30, 1), (64, 128, 2), (150, 500, 3)):
        reads, refs = synth.make_pairs(64, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.1, short_frac=0.1)
        for gr, gf in ((-3, -3), (-2, -5)):
            lin = cpu_ref.Scoring.make(2, -1, gr, gf)
            aff = cpu_ref.Scoring.make(2, -1, gr, gf, gr, gr, gf, gf)
            for opt in (0, 1):
                assert np.array_equal(cpu_ref.score(opt, reads, refs, lin),
                                      cpu_ref.score(opt, reads, refs, aff, affine=True))


def test_affine_prefers_one_long_gap():
    # read = ref with a 4-base deletion: affine (open -5, extend -1) keeps one gap, score 2*16-5-3
    ref = b"ACGTACGTTTGGCCAAGTCA"
    read = ref[:8] + ref[12:]
    reads = np.frombuffer(read, np.uint8)[None, :].copy()
    refs = np.frombuffer(ref, np.uint8)[None, :].copy()
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
    assert int(cpu_ref.score(0, reads, refs, sc, affine=True)[0]) == 2 * 16 - 5 - 3
    assert int(cpu_ref.score(0, reads, refs, cpu_ref.Scoring.make())[0]) < 2 * 16 - 5 - 3


def test_oracle_threads_and_noop():
    from versalignlib_amd import synth
    reads, refs = synth.make_pairs(50, 30, 40, seed=9)
    a = cpu_ref.score(0, reads, refs, threads=1)
    b = cpu_ref.score(0, reads, refs, threads=4)
    assert np.array_equal(a, b)
    assert not cpu_ref.score(2, reads, refs).any()       # opt & 0xF == 2: silent no-op


def test_affine_alignment_oracle_degenerates_to_the_reference_path():
    """The Gotoh traceback's tie-breaks (DIAG > gap-in-ref > gap-in-read; open preferred over
    extend) are chosen so that open == extend == g reproduces the linear alignments, which are
    pinned to the reference Default kernel by the golden fixtures."""
    from versalignlib_amd import synth
    for R, F, seed in ((12, 20, 1), (33, 70, 2), (64, 128, 3), (40, 9, 5)):
        reads, refs = synth.make_pairs(80, R, F, seed=seed, indel_rate=0.04, n_run_frac=0.1, short_frac=0.15)
        for gr, gf in ((-3, -3), (-2, -4)):
            lin = cpu_ref.Scoring.make(2, -1, gr, gf)
            aff = cpu_ref.Scoring.make(2, -1, gr, gf, gr, gr, gf, gf)
            for opt in (0, 1):
                a = cpu_ref.align(opt, reads, refs, lin)
                b = cpu_ref.align(opt, reads, refs, aff, affine=True)
                assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_affine_alignment_oracle_known_answer():
    ref = b"ACGTACGTTTGGCCAAGTCA"
    read = ref[:8] + ref[12:]
    reads = np.frombuffer(read, np.uint8)[None, :].copy()
    refs = np.frombuffer(ref, np.uint8)[None, :].copy()
    rows, idx = cpu_ref.align(0, reads, refs, cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1), affine=True)
    s = idx[0, 0]
    assert bytes(rows[0, 0, s:-1]) == b"ACGTACGT----CCAAGTCA" and bytes(rows[0, 1, s:-1]) == ref


@pytest.mark.parametrize("path", [p for p in golden_files() if "kat" not in p], ids=lambda p: p.split("/")[-1][:-4])
def test_sse_policy_oracle_reproduces_the_sse_kernel(path):
    """Second tie-break policy: alignments of the reference's SSE2 kernel."""
    g = load(path)
    m, x, gr, gf = (int(v) for v in g["scoring"])
    sc = cpu_ref.Scoring.make(m, x, gr, gf)
    differs = 0
    for opt, tag in ((0, "sw"), (1, "nw")):
        n8 = g["sse_idx_" + tag].shape[0]
        rows, idx = cpu_ref.align(opt, g["reads"][:n8], g["refs"][:n8], sc, policy="sse")
        assert np.array_equal(idx, g["sse_idx_" + tag]) and np.array_equal(rows, g["sse_rows_" + tag])
        differs += int((idx != g["idx_" + tag][:n8]).any(axis=1).sum())
    assert differs > 0 or "c2" in path        # the fixtures do exercise the difference to the Default policy


def test_nw_variant_score_is_invariant_under_trimming_of_trailing_padding():
    """What length-sorted NW batches rest on (hip_engine.hip.h: ragged_applies): sweeping a pair at any shape between its
    trimmed lengths (without trailing non-ACGT bytes) and the padded shape gives the padded shape's NW-variant score --
    linear and affine, N runs and junk bytes included."""
    from versalignlib_amd import synth
    R, F, n = 150, 500, 1500
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=7, n_run_frac=0.05, short_frac=0.05, junk_frac=0.03)

    def trimmed(a):
        ok = np.isin(a & 0xDF, np.frombuffer(b"ACGT", np.uint8)) & (a < 0x80)
        return np.where(ok.any(axis=1), a.shape[1] - np.argmax(ok[:, ::-1], axis=1), 0)

    tr, tf = trimmed(reads), trimmed(refs)
    rng = np.random.default_rng(1)
    for sc, aff in ((cpu_ref.Scoring.make(), False), (cpu_ref.Scoring.make(2, -1, -2, -4), False),
                    (cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -4, -2), True)):
        for opt in (0, 1):
            full = cpu_ref.score(opt, reads, refs, sc, threads=4, affine=aff)
            for lo in range(0, n, 300):
                sl = slice(lo, lo + 300)
                r2 = max(1, int(min(R, tr[sl].max() + rng.integers(0, 5))))
                f2 = max(1, int(min(F, tf[sl].max() + rng.integers(0, 9))))
                part = cpu_ref.score(opt, np.ascontiguousarray(reads[sl, :r2]), np.ascontiguousarray(refs[sl, :f2]), sc, threads=4, affine=aff)
                assert np.array_equal(part, full[sl]), (opt, aff, lo)


def test_oracle_tracebacks_stay_inside_the_matrix_outside_their_range():
    """The int16 restatement asked for a shape x scoring outside its range (5 509 x 6, affine NW: the column-0 border lies
    below NEG_INF) follows flags that no longer mean anything: its traceback must stop at the matrix border -- it used to walk
    on and write in front of the result rows (found by tools/fuzz_parity.py; heap corruption in the test process) -- and say
    so (vref_walks_left_matrix).  The int32 restatement of the same call is the valid answer and leaves the matrix nowhere."""
    from versalignlib_amd import synth
    R, F, n = 5509, 6, 24
    reads, refs = synth.make_pairs(n, R, F, seed=741250890, sub_rate=0.3, short_frac=0.2)
    sc = cpu_ref.Scoring.make(1, -4, -3, -3, -5, -3, -5, -3)
    L = cpu_ref.lib()
    before = L.vref_walks_left_matrix()
    rows, idx = cpu_ref.align(1, reads, refs, sc, threads=2, affine=True, wide=True)
    assert L.vref_walks_left_matrix() == before and (idx[:, 1] == R + F - 1).all()
    cpu_ref.align(1, reads, refs, sc, threads=2, affine=True)            # out of range: garbage, but inside its buffers
    assert L.vref_walks_left_matrix() > before
