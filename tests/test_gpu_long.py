"""Long-read score path (row strips + column phases, BASELINE config 5 shape): same results
as the oracle, SW and NW variant, including shapes that would also fit the resident kernel."""
import os

import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth

from conftest import band_constants, debug_switches

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced_long(monkeypatch):
    debug_switches(monkeypatch, force_long=1)


@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 203, 1), (160, 64, 50, 2), (161, 700, 33, 3),
                                         (400, 333, 40, 4), (12, 20, 100, 5), (1000, 1300, 17, 6)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_forced_long_path_matches_oracle(forced_long, R, F, n, seed, gaps):
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08,
                                   lowercase_frac=0.05, junk_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1]) as hip:
        assert '"long_mode": 1' in (hip.score_alignments(0, reads[:1], refs[:1]) is not None and hip.drain_log())
        for opt in (0, 1):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8)
            assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8], got[:8], exp[:8])


@pytest.mark.parametrize("R,F,n", [(3000, 3500, 12), (2500, 700, 9)])
def test_long_shapes_select_the_long_path(R, F, n):
    reads, refs = synth.make_pairs(n, R, F, seed=R, indel_rate=0.01, n_run_frac=0.1, short_frac=0.1)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        for opt in (0, 1):
            got = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got, cpu_ref.score(opt, reads, refs, threads=8))
        assert '"long_mode": 1' in hip.drain_log()


def test_config5_shape_10k_by_10k():
    """BASELINE config 5 at a size the oracle finishes in seconds: unbanded, int16 (SW cells of a
    10 kbp x 10 kbp pair stay below 20000)."""
    R = F = 10000
    n = 5
    reads, refs = synth.make_pairs(n, R, F, seed=55, sub_rate=0.1, indel_rate=0.0, n_run_frac=0.2, short_frac=0.2)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        got = hip.score_alignments(0, reads, refs)
    exp = cpu_ref.score(0, reads, refs, threads=8)
    assert np.array_equal(got, exp), (got, exp)
    assert exp.max() > 10000          # the batch does exercise large cell values


@pytest.mark.parametrize("R,F,n,seed", [(400, 450, 120, 1), (1000, 1300, 33, 2), (150, 500, 203, 3), (3000, 2800, 10, 4)])
@pytest.mark.parametrize("band", [16, 64, 512, 100000])
def test_banded_smith_waterman(R, F, n, seed, band):
    """band_width (extension).  The definition is geometry-free (include/valign_hip.h): AT LEAST the per-cell
    band |j - floor(i F / R)| <= w, exactly the block band with the two documented constants.  Checked:
    equality with the oracle's restatement of the block definition; the sandwich
    per-cell band <= result <= unbanded; and a band wider than the matrix is the reference's result."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08)
    block_rows, col_align = band_constants(R, F, band)
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=band) as hip:
        got = hip.score_alignments(0, reads, refs)
        assert '"band_width": %d' % band in hip.drain_log()
        with pytest.raises(host.PluginError, match="Smith-Waterman scores only"):
            hip.score_alignments(1, reads, refs)
    exp = cpu_ref.score_banded_sw(reads, refs, band, threads=8, block_rows=block_rows, col_align=col_align)
    assert np.array_equal(got, exp), (np.nonzero(got != exp)[0][:8], got[:8], exp[:8])
    per_cell = cpu_ref.score_banded_sw(reads, refs, band, threads=8)
    full = cpu_ref.score(0, reads, refs, threads=8)
    assert (per_cell <= got).all() and (got <= full).all()
    if band >= 2 * max(R, F):
        assert np.array_equal(got, full) and np.array_equal(per_cell, full)


def test_band_definitions_differ_where_they_should():
    """The sandwich is not vacuous: on reads with a long insertion the per-cell band, the block band and the
    full matrix give three different scores for some pairs (CPU restatements only), and the HIP result is the
    block one."""
    R, F, n, band = 1200, 1200, 48, 64
    rng = np.random.default_rng(9)
    refs = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, F))
    reads = refs.copy()
    for p in range(n):                     # shift the second half of the read by 20..120 bases: the optimum leaves a narrow band
        shift = int(rng.integers(20, 120))
        cut = int(rng.integers(300, 700))
        reads[p, cut + shift:] = refs[p, cut:F - shift]
        reads[p, cut:cut + shift] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=shift)
    block_rows, col_align = band_constants(R, F, band)
    per_cell = cpu_ref.score_banded_sw(reads, refs, band, threads=8)
    block = cpu_ref.score_banded_sw(reads, refs, band, threads=8, block_rows=block_rows, col_align=col_align)
    strips = cpu_ref.score_banded_sw(reads, refs, band, threads=8, block_rows=160, col_align=4)      # the strip kernel's blocks
    full = cpu_ref.score(0, reads, refs, threads=8)
    assert (per_cell <= block).all() and (block <= strips).all() and (strips <= full).all()
    assert (per_cell < strips).any() and (block < full).any()
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=band) as hip:
        assert np.array_equal(hip.score_alignments(0, reads, refs), block)


def test_config5_banded_10k():
    """BASELINE config 5 as stated: 10 kbp x 10 kbp, band of 512 diagonals, int32 cells (score_width = 32) --
    and the int16 cells the engine would pick by itself for these scores give the same result."""
    R = F = 10000
    n = 24
    reads, refs = synth.make_pairs(n, R, F, seed=56, sub_rate=0.1, indel_rate=0.0, n_run_frac=0.2, short_frac=0.1)
    block_rows, col_align = band_constants(R, F, 512)
    exp = cpu_ref.score_banded_sw(reads, refs, 512, threads=8, block_rows=block_rows, col_align=col_align)
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=512, score_width=32) as hip:
        got = hip.score_alignments(0, reads, refs)
        assert '"score_cells": "int32"' in hip.drain_log()
    assert np.array_equal(got, exp)
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=512) as hip:
        assert np.array_equal(hip.score_alignments(0, reads, refs), exp)
        # (linear gaps under a band run on the block chain of band_kernels.hip.h: int32 cells whatever the range needs)
        assert '"score_cells": "int32"' in hip.drain_log()
    full = cpu_ref.score(0, reads[:4], refs[:4], threads=8)
    assert (got[:4] <= full).all()
    assert (cpu_ref.score_banded_sw(reads[:4], refs[:4], 512, threads=8) <= got[:4]).all()


def test_config5_cells_that_need_int32():
    """10 kbp x 10 kbp with match = 5: cells climb past 32767, int16 would wrap (the reference does, silently).
    Auto mode must take the int32 cells -- unbanded and banded -- and saturate the ABI's short at 32767."""
    R = F = 10000
    n = 6
    reads, refs = synth.make_pairs(n, R, F, seed=58, sub_rate=0.12, indel_rate=0.0, n_run_frac=0.0, short_frac=0.5)
    sc = cpu_ref.Scoring.make(5, -4, -6, -6)
    keys = dict(score_match=5, score_mismatch=-4, score_gap_read=-6, score_gap_ref=-6)
    block_rows, col_align = band_constants(R, F, 512, hipkernel.Scoring.make(5, -4, -6, -6))
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        got = hip.score_alignments(0, reads, refs)
        assert '"score_cells": "int32"' in hip.drain_log()
    exp = cpu_ref.score(0, reads, refs, sc, threads=8, wide=True)
    assert np.array_equal(got, exp)
    assert (exp == 32767).any() and ((exp > 0) & (exp < 32767)).any()
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=512, **keys) as hip:
        got = hip.score_alignments(0, reads, refs)
        assert '"score_cells": "int32"' in hip.drain_log()
    assert np.array_equal(got, cpu_ref.score_banded_sw(reads, refs, 512, sc, threads=8, block_rows=block_rows, col_align=col_align))


def test_config5_unbanded_int32_cells_forced():
    """The 10 kbp x 10 kbp shape on int32 cells (BASELINE config 5 names int32) against the oracle."""
    R = F = 10000
    n = 5
    reads, refs = synth.make_pairs(n, R, F, seed=57, sub_rate=0.1, indel_rate=0.0, n_run_frac=0.2, short_frac=0.2)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_width=32) as hip:
        for opt in (0, 1):
            got = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got, cpu_ref.score(opt, reads, refs, threads=8, wide=True)), opt
        assert '"score_cells": "int32"' in hip.drain_log()


@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 203, 7), (400, 333, 40, 8), (2500, 700, 9, 9)])
def test_int32_cells_forced(R, F, n, seed):
    """score_width = 32: one pair per register, int32 cells (strip path); same results in range."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08)
    for gaps in ((-3, -3), (-2, -4)):
        sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
        with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1], score_width=32) as hip:
            for opt in (0, 1):
                assert np.array_equal(hip.score_alignments(opt, reads, refs), cpu_ref.score(opt, reads, refs, sc, threads=8))


def test_int32_cells_take_over_where_int16_would_wrap():
    """match = 20 on 2000 x 2000: cells reach 40000.  Auto mode computes them in int32 (saturating
    the ABI's short at 32767); score_width = 16 refuses instead."""
    R, F, n = 2000, 2000, 12
    reads, refs = synth.make_pairs(n, R, F, seed=77, sub_rate=0.02, n_run_frac=0.0, short_frac=0.2)
    sc = cpu_ref.Scoring.make(20, -1, -3, -3)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=20) as hip:
        got = hip.score_alignments(0, reads, refs)
    exp = cpu_ref.score(0, reads, refs, sc, threads=8, wide=True)
    assert np.array_equal(got, exp)
    assert (exp == 32767).any() and (exp < 32767).any()
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=20, score_width=16) as hip:
        with pytest.raises(host.PluginError, match="int16 range"):
            hip.score_alignments(0, reads, refs)
    # NW variant on a long pair: intermediate cells dip below -32768 with gap -8
    R, F, n = 5000, 5000, 6
    reads, refs = synth.make_pairs(n, R, F, seed=78, sub_rate=0.3, short_frac=0.0)
    sc = cpu_ref.Scoring.make(2, -8, -8, -8)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_mismatch=-8, score_gap_read=-8, score_gap_ref=-8) as hip:
        got = hip.score_alignments(1, reads, refs)
    assert np.array_equal(got, cpu_ref.score(1, reads, refs, sc, threads=8, wide=True))


def test_largest_shape_the_abi_allows():
    """read_length + ref_length = 32767 (the Alignment struct's coordinates are shorts)."""
    R, F, n = 16383, 16384, 3
    reads, refs = synth.make_pairs(n, R, F, seed=99, sub_rate=0.2, n_run_frac=0.0, short_frac=0.34)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        got = hip.score_alignments(0, reads, refs)
    assert np.array_equal(got, cpu_ref.score(0, reads, refs, threads=8, wide=True))
    with pytest.raises(host.PluginError, match="16-bit coordinates"):
        host.Plugin(build.HIP_PLUGIN, R + 1, F)


def test_long_path_over_many_staging_chunks(forced_long, monkeypatch):
    """Host-pointer calls cycle through the pinned slots; strip kernels share one boundary-row scratch,
    so their chunks must run back to back on one stream -- eight chunks against the oracle."""
    debug_switches(monkeypatch, chunk_bytes=256 << 10)       # 1024-pair chunks
    R, F, blk = 330, 420, 2048
    r0, f0 = synth.make_pairs(blk, R, F, seed=31, indel_rate=0.01, n_run_frac=0.02, short_frac=0.05)
    reads, refs = np.tile(r0, (4, 1)), np.tile(f0, (4, 1))
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4) as hip:
        for opt in (0, 1):
            exp = cpu_ref.score(opt, r0, f0, threads=8)
            got = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got, np.tile(exp, 4)), opt


AFFINE_SETS = [(-5, -1, -5, -1), (-6, -2, -4, -1), (-3, -3, -3, -3)]


@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 203, 11), (161, 700, 33, 13), (400, 333, 40, 14), (1000, 1300, 17, 16)])
@pytest.mark.parametrize("aff", AFFINE_SETS)
def test_forced_long_path_affine_matches_oracle(forced_long, R, F, n, seed, aff):
    """Affine gaps on the strip kernel (E in registers along the row, F handed from strip to strip next to H):
    shapes that also fit the resident kernel, forced onto the long path, against the Gotoh oracle (which the
    enumeration fixtures pin) -- and open == extend against the linear oracle."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.05, short_frac=0.08,
                                   lowercase_frac=0.05, junk_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1], score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        for opt in (0, 1):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8, affine=True)
            assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8], got[:8], exp[:8])
            if aff[0] == aff[1] and aff[2] == aff[3]:
                assert np.array_equal(got, cpu_ref.score(opt, reads, refs, cpu_ref.Scoring.make(2, -1, aff[0], aff[2]), threads=8))
        assert '"long_mode": 1' in hip.drain_log()


@pytest.mark.parametrize("R,F,n", [(3000, 3500, 10), (2500, 700, 9)])
def test_long_shapes_affine(R, F, n):
    """Shapes that need row strips, affine gaps, both cell widths (int16 where the range allows, int32 forced)."""
    reads, refs = synth.make_pairs(n, R, F, seed=R + 1, indel_rate=0.02, n_run_frac=0.1, short_frac=0.1)
    aff = (-5, -1, -5, -1)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1], score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    for width in (0, 32):
        with host.Plugin(build.HIP_PLUGIN, R, F, score_width=width, **keys) as hip:
            for opt in (0, 1):
                got = hip.score_alignments(opt, reads, refs)
                assert np.array_equal(got, cpu_ref.score(opt, reads, refs, sc, threads=8, affine=True, wide=True)), (width, opt)


def test_config5_shape_affine_and_banded_affine():
    """10 kbp x 10 kbp with the affine scoring of BASELINE configs 2-4: unbanded SW (int16 cells suffice), the NW
    variant (int16 as well since round 4: no cell lies below a gap straight down from the free row 0, -10 010 here; the
    bound used before charged every step an opening and sent the call to int32 cells -- same scores, two thirds of the
    rate), and the 512-diagonal band."""
    R = F = 10000
    n = 5
    reads, refs = synth.make_pairs(n, R, F, seed=59, sub_rate=0.1, indel_rate=0.01, n_run_frac=0.2, short_frac=0.2)
    aff = (-5, -1, -5, -1)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1], score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    block_rows, col_align = band_constants(R, F, 512, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    assert (block_rows, col_align) == (16, 1)            # the affine block chain (round 4)
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        assert np.array_equal(hip.score_alignments(0, reads, refs), cpu_ref.score(0, reads, refs, sc, threads=8, affine=True, wide=True))
        assert '"score_cells": "int16"' in hip.drain_log()
        assert np.array_equal(hip.score_alignments(1, reads, refs), cpu_ref.score(1, reads, refs, sc, threads=8, affine=True, wide=True))
        assert '"score_cells": "int16"' in hip.drain_log()
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=512, **keys) as hip:
        got = hip.score_alignments(0, reads, refs)
    exp = cpu_ref.score_banded_sw(reads, refs, 512, sc, threads=8, block_rows=block_rows, col_align=col_align, affine=True)
    assert np.array_equal(got, exp)
    assert (cpu_ref.score_banded_sw(reads, refs, 512, sc, threads=8, affine=True) <= got).all()
    assert (got <= cpu_ref.score(0, reads, refs, sc, threads=8, affine=True, wide=True)).all()


@pytest.mark.parametrize("band", [16, 64, 512])
def test_banded_affine(band):
    R, F, n = 1000, 1300, 33
    reads, refs = synth.make_pairs(n, R, F, seed=21, indel_rate=0.03, n_run_frac=0.05, short_frac=0.08)
    aff = (-6, -2, -4, -1)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1], score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    # (the blocks of the band the library computes for THIS shape and scoring, as it reports them: since round 4 affine
    # bands run on the block chain too -- 16 rows, column alignment 1 -- wherever the chain's plan fits)
    block_rows, col_align = band_constants(R, F, band, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=band, **keys) as hip:
        got = hip.score_alignments(0, reads, refs)
    assert np.array_equal(got, cpu_ref.score_banded_sw(reads, refs, band, sc, threads=8, block_rows=block_rows, col_align=col_align, affine=True))


def _same_alignments(got, exp, what):
    rows, idx = got
    erows, eidx = exp
    bad = np.nonzero((idx != eidx).any(axis=1))[0]
    assert bad.size == 0, (what, "idx", bad[:5], idx[bad[:3]], eidx[bad[:3]])
    bad = np.nonzero((rows != erows).any(axis=(1, 2)))[0]
    assert bad.size == 0, (what, "rows", bad[:5])


@pytest.mark.parametrize("R,F,n,seed", [(3000, 3500, 9, 1), (2500, 700, 11, 2), (2049, 300, 7, 3), (5000, 4000, 4, 4), (4100, 9000, 3, 5)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_alignments_of_long_reads(R, F, n, seed, gaps):
    """compute_alignments beyond one register sweep (read_length > 2048): row strips with boundary rows through
    HBM, one pointer region per strip, the traceback crossing strips -- rows and coordinates of both modes against
    the oracle (the reference computes alignments for any shape its short coordinates allow,
    DefaultKernel.cpp:391-456).  Odd pair counts leave a wave half empty."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.15, short_frac=0.25,
                                   lowercase_frac=0.05, junk_frac=0.03)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1], num_threads=4) as hip:
        for opt in (host.SW, host.NW):
            if opt == host.NW and (R + 1) * min(gaps) < -32000:
                continue
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _same_alignments(got, cpu_ref.align(opt, reads, refs, sc, threads=8), (R, F, opt))


def test_alignments_config5_shape():
    """10 kbp x 10 kbp alignments of both modes (five strips of 2048 rows): with the default scores the cells stay
    inside int16 (column 0 of the NW variant reaches -30 003); gap scores of -4 leave it -- where the reference's
    shorts wrap, the NW variant now runs on int32 cells (round 3; rounds 1-2 refused the call)."""
    R = F = 10000
    n = 3
    reads, refs = synth.make_pairs(n, R, F, seed=61, sub_rate=0.1, indel_rate=0.01, n_run_frac=0.3, short_frac=0.34)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4) as hip:
        for opt in (host.SW, host.NW):
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _same_alignments(got, cpu_ref.align(opt, reads, refs, threads=8), ("10k", opt))
    sc = cpu_ref.Scoring.make(2, -1, -4, -4)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, score_gap_read=-4, score_gap_ref=-4) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)
        _same_alignments(got, cpu_ref.align(host.NW, reads, refs, sc, threads=8, wide=True), "10k, gap -4: int32 cells")


@pytest.mark.parametrize("R,F,n,seed", [(3000, 3500, 7, 11), (2500, 700, 9, 12), (2049, 300, 5, 13), (4100, 6000, 3, 14)])
@pytest.mark.parametrize("aff", [(-5, -1, -5, -1), (-6, -2, -4, -1), (-3, -3, -3, -3)])
def test_affine_alignments_of_long_reads(R, F, n, seed, aff):
    """Affine alignments by row strips (E in registers, F handed from strip to strip, two code streams) against the
    Gotoh oracle the enumeration fixtures pin; open == extend also against the linear oracle."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.15, short_frac=0.25,
                                   lowercase_frac=0.05, junk_frac=0.03)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1], score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, **keys) as hip:
        for opt in (host.SW, host.NW):
            # (cells of the NW variant that would leave the affine int16 range: int32 cells since round 4, refused before)
            wide = opt == host.NW and (min(R, F) + 2) * min(aff) < -15000
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _same_alignments(got, cpu_ref.align(opt, reads, refs, sc, threads=8, affine=True, wide=wide), (R, F, aff, opt))
            if aff[0] == aff[1] and aff[2] == aff[3]:
                _same_alignments(got, cpu_ref.align(opt, reads, refs, cpu_ref.Scoring.make(2, -1, aff[0], aff[2]), threads=8),
                                 ("degenerate", opt))


def test_long_alignments_refuse_what_they_do_not_implement():
    """The SSE / AVX kernels have linear gaps only: their tie-breaks with affine scoring are refused (loudly, as a
    `const char *` like every plugin error), at long reads as at short ones."""
    R, F = 3000, 500
    reads, refs = synth.make_pairs(2, R, F, seed=5)
    with host.Plugin(build.HIP_PLUGIN, R, F, traceback_policy=1, score_gap_open_read=-5, score_gap_extend_read=-1) as hip:
        with pytest.raises(host.PluginError, match="linear gap model only"):
            hip.compute_alignments(host.SW, reads, refs)


@pytest.mark.parametrize("R,F,n,seed", [(3000, 3500, 9, 21), (2500, 700, 11, 22), (2049, 300, 7, 23), (5000, 4000, 4, 24), (4100, 9000, 3, 25)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_sse_policy_alignments_of_long_reads(R, F, n, seed, gaps):
    """traceback_policy = 1 on the row-strip path (reads beyond 2048 rows): DIAG only between ACGT bases > LEFT > UP,
    no stop at zero cells, N invalid for the NW end cell -- against the oracle's restatement of the SSE2 kernel's rules."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, sub_rate=0.12, indel_rate=0.01, n_run_frac=0.3, short_frac=0.2, lowercase_frac=0.1)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1], traceback_policy=1, num_threads=4) as hip:
        for opt in (host.SW, host.NW):
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _same_alignments(got, cpu_ref.align(opt, reads, refs, sc, threads=8, policy="sse"), ("sse strips", R, F, opt, gaps))


def test_sse_policy_long_reads_live_against_the_sse_kernel():
    """... and live against the reference's own libSSEKernel.so / libAVXKernel.so at 3000 x 3500 (the verdict's shape)."""
    from conftest import ref_kernel
    sse, avx = ref_kernel("SSE"), ref_kernel("AVX")
    if not sse or not avx:
        pytest.skip("oracle/_ref not built")
    R, F, n = 3000, 3500, 16
    reads, refs = synth.make_pairs(n, R, F, seed=26, sub_rate=0.1, indel_rate=0.01, n_run_frac=0.3, short_frac=0.2)
    with host.Plugin(build.HIP_PLUGIN, R, F, traceback_policy=1) as hip, host.Plugin(sse, R, F) as s, host.Plugin(avx, R, F) as a:
        for opt in (host.SW, host.NW):
            got = hip.compute_alignments(opt, reads, refs)
            _same_alignments(got, s.compute_alignments(opt, reads, refs), ("sse live", opt))
            _same_alignments(got, a.compute_alignments(opt, reads, refs), ("avx live", opt))


def test_nw_alignments_on_int32_cells_where_int16_would_wrap():
    """NW-variant alignments whose column-0 border (read_length * gap_ref) leaves int16: the reference's shorts wrap
    there (DefaultKernel.cpp:282-389); rounds 1-2 refused the call, round 3 computes it on int32 cells (row strips, one
    pair per register).  Against the oracle's int32 restatement; the Smith-Waterman call of the same object is untouched."""
    R, F, n = 11000, 10500, 3                       # 11001 * -3 = -33003
    reads, refs = synth.make_pairs(n, R, F, seed=61, sub_rate=0.1, indel_rate=0.002, n_run_frac=0.4, short_frac=0.4)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)
        _same_alignments(got, cpu_ref.align(host.NW, reads, refs, threads=4, wide=True), ("int32 NW", R, F))
        got = hip.compute_alignments(host.SW, reads, refs, normalise=False)
        _same_alignments(got, cpu_ref.align(host.SW, reads, refs, threads=4), ("SW beside it", R, F))
    R, F, n = 9000, 400, 5                          # gap -4: 9001 * -4 = -36004, a short reference
    reads, refs = synth.make_pairs(n, R, F, seed=62, sub_rate=0.1, n_run_frac=0.2, short_frac=0.2)
    sc = cpu_ref.Scoring.make(2, -1, -2, -4)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=-2, score_gap_ref=-4, num_threads=4) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)
        _same_alignments(got, cpu_ref.align(host.NW, reads, refs, sc, threads=4, wide=True), ("int32 NW", R, F))
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=-2, score_gap_ref=-4, traceback_policy=1) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)     # (refused up to round 3: the SSE rules on int32 cells)
        _same_alignments(got, cpu_ref.align(host.NW, reads, refs, sc, threads=4, policy="sse", wide=True), ("int32 NW, SSE rules", R, F))


_WIDE_MODES = {
    "linear": (dict(), dict(), dict()),
    "affine": (dict(score_gap_open_read=-7, score_gap_extend_read=-2, score_gap_open_ref=-6, score_gap_extend_ref=-1), dict(affine=True), dict()),
    "sse": (dict(traceback_policy=1), dict(policy="sse"), dict()),
}


def _wide_case(mode, scale, gaps=(-3, -4)):
    """Plugin keys and oracle scoring of one int32 case: scores of the mode multiplied by `scale`."""
    keys, okw, _ = _WIDE_MODES[mode]
    keys = {k: (v * scale if k.startswith("score_") else v) for k, v in keys.items()}
    keys.update(score_match=2 * scale, score_mismatch=-1 * scale, score_gap_read=gaps[0] * scale, score_gap_ref=gaps[1] * scale)
    aff = [keys.get(k, None) for k in ("score_gap_open_read", "score_gap_extend_read", "score_gap_open_ref", "score_gap_extend_ref")]
    sc = cpu_ref.Scoring.make(2 * scale, -1 * scale, gaps[0] * scale, gaps[1] * scale, *aff)
    return keys, sc, okw


@pytest.mark.parametrize("mode", ["linear", "affine", "sse"])
@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 203, 171), (33, 70, 101, 173), (1, 1, 5, 174), (700, 90, 40, 175), (1030, 300, 9, 176),
                                         (2049, 1500, 5, 177)])
def test_int32_alignment_cells_every_mode_forced(monkeypatch, mode, alg, R, F, n, seed):
    """Every mode of compute_alignments on the int32 strip kernel (VALIGN_HIP_DEBUG wide_align) on shapes where int16
    suffices: identical to the oracle's int16 restatement (which the golden vectors pin) -- one strip and several (512 rows
    each), odd pair counts, padding, invalid bases moving the end cell."""
    debug_switches(monkeypatch, wide_align=1)
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.1, short_frac=0.15, lowercase_frac=0.05, junk_frac=0.05)
    keys, sc, okw = _wide_case(mode, 1)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=3, **keys) as hip:
        got = hip.compute_alignments(alg, reads, refs, normalise=False)
    exp = cpu_ref.align(alg, reads, refs, sc, threads=8, **okw)
    _same_alignments(got, exp, ("forced int32", mode, alg, R, F))
    _same_alignments(exp, cpu_ref.align(alg, reads, refs, sc, threads=8, wide=True, **okw), "oracle int16 == int32")


@pytest.mark.parametrize("mode", ["linear", "affine", "sse"])
@pytest.mark.parametrize("alg", [0, 1])
def test_alignments_whose_cells_leave_int16_every_mode(mode, alg):
    """Scores large enough that Smith-Waterman cells pass 32767 (and 65535: the end value's high half travels in
    EndCell.pad) and NW-variant cells fall below -32768 -- the reference's shorts wrap there, rounds 1-3 refused everything
    but linear-gap NW.  Against the oracle's int32 restatement of the same rules; the same scores divided by the scale
    give the same alignments on int16 cells."""
    scale = 150
    for R, F, n, seed in ((600, 900, 21, 181), (1300, 700, 7, 182)):
        reads, refs = synth.make_pairs(n, R, F, seed=seed, sub_rate=0.08, indel_rate=0.02, n_run_frac=0.1, short_frac=0.2)
        keys, sc, okw = _wide_case(mode, scale)
        with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, **keys) as hip:
            got = hip.compute_alignments(alg, reads, refs, normalise=False)
        _same_alignments(got, cpu_ref.align(alg, reads, refs, sc, threads=8, wide=True, **okw), ("int32", mode, alg, R, F))
        keys1, sc1, _ = _wide_case(mode, 1)
        _same_alignments(got, cpu_ref.align(alg, reads, refs, sc1, threads=8, **okw), ("scores / scale on int16", mode, alg, R, F))


@pytest.mark.parametrize("k", [8, 12, 16])
@pytest.mark.parametrize("mode", ["linear", "affine", "sse"])
def test_strip_alignment_kernels_at_every_strip_height(monkeypatch, k, mode):
    """The row-strip alignment kernels exist at 16, 12 and 8 rows per lane (the engine takes the height that pads the read
    least; VALIGN_HIP_DEBUG strip_k forces one): each against the oracle on a shape of several strips whose reference is
    longer than the ring of slab numbers (128 columns, refilled 64 at a time) many times over, and on one whose reference is
    shorter than the ring."""
    debug_switches(monkeypatch, strip_k=k)
    keys, sc, okw = _wide_case(mode, 1)
    for R, F, n, seed in ((2300, 1700, 7, 201), (2100, 90, 9, 202)):
        reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.15, short_frac=0.25, lowercase_frac=0.05, junk_frac=0.05)
        with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=3, **keys) as hip:
            for alg in (host.SW, host.NW):
                got = hip.compute_alignments(alg, reads, refs, normalise=False)
                _same_alignments(got, cpu_ref.align(alg, reads, refs, sc, threads=8, **okw), ("strip_k", k, mode, alg, R, F))


@pytest.mark.parametrize("k", [12, 16])
def test_int32_strip_kernels_at_every_strip_height(monkeypatch, k):
    """... and the int32 kernel of the reference's own model (NW variant, linear gaps) at 12 and 16 rows per lane (every other
    int32 mode has 8 only, which test_int32_alignment_cells_every_mode_forced runs)."""
    debug_switches(monkeypatch, strip_k=k, wide_align=1)
    R, F, n = 2300, 1700, 7
    reads, refs = synth.make_pairs(n, R, F, seed=203, indel_rate=0.02, n_run_frac=0.15, short_frac=0.25)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=3) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)
    _same_alignments(got, cpu_ref.align(host.NW, reads, refs, threads=8), ("int32 strip_k", k))


def test_alignment_scores_beyond_int32_bound_refused():
    """(R + F) * |score| near 2^28: refused with a message, as every range check."""
    R, F = 16000, 16000
    reads, refs = synth.make_pairs(2, R, F, seed=191)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=20000, score_mismatch=-20000, score_gap_read=-20000, score_gap_ref=-20000) as hip:
        with pytest.raises(host.PluginError, match="int32"):
            hip.compute_alignments(host.SW, reads, refs)


@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 203, 71), (64, 128, 300, 72), (33, 70, 101, 73), (1, 1, 5, 74), (700, 90, 40, 75),
                                         (2049, 300, 7, 76), (3000, 3500, 6, 77)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_int32_alignment_cells_forced(monkeypatch, R, F, n, seed, gaps):
    """The int32 strip kernel on shapes where int16 suffices (VALIGN_HIP_DEBUG wide_align): identical to the int16 kernels'
    alignments and to the oracle -- one strip and many, odd pair counts, padding, invalid bases moving the end cell."""
    debug_switches(monkeypatch, wide_align=1)
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.1, short_frac=0.15, lowercase_frac=0.05, junk_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1], num_threads=3) as hip:
        got = hip.compute_alignments(host.NW, reads, refs, normalise=False)
    exp = cpu_ref.align(host.NW, reads, refs, sc, threads=8)
    _same_alignments(got, exp, ("forced int32", R, F, gaps))
    _same_alignments(exp, cpu_ref.align(host.NW, reads, refs, sc, threads=8, wide=True), "oracle int16 == int32")
