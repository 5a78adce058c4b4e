"""Banded Smith-Waterman scores with linear gaps: the cyclic block chain of band_kernels.hip.h (round 3).

Every lane sweeps only the band window of its own 16-row block and hands its bottom row on through an LDS delay
ring; the result must equal the oracle's block band with the constants the library reports for the engine
(valign_hip_describe: band_block_rows = 16, band_col_align = 1), sit between the per-cell band and the full matrix,
and equal score_long_kernel's strips (VALIGN_HIP_DEBUG no_band_chain) on THEIR block definition.  Shapes are chosen for the
schedule's corners: slopes F / R above and below one (window starts that advance by varying amounts), a single strip
that is mostly top padding, reads that end inside a block, windows clipped at both matrix edges, bands from 2
diagonals to almost the matrix, pair counts that leave the last wave short."""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth
from conftest import debug_switches

pytestmark = pytest.mark.gpu

SHAPES = [
    # (R, F, n, band, seed)
    (1000, 1000, 37, 64, 1),
    (1000, 1300, 33, 16, 2),
    (1300, 1000, 30, 32, 3),          # slope < 1
    (700, 2100, 18, 128, 4),          # slope 3
    (520, 530, 41, 2, 5),             # the narrowest band
    (100, 120, 50, 8, 6),             # one strip, mostly padding
    (513, 400, 21, 24, 7),            # one row into the second strip
    (2049, 2000, 9, 512, 8),
    (3000, 2800, 7, 1000, 9),
    (4000, 9000, 4, 300, 10),
    (31, 33, 64, 6, 11),
    (10000, 10000, 6, 512, 12),       # BASELINE config 5's shape
]


@pytest.mark.parametrize("R,F,n,band,seed", SHAPES)
def test_block_chain_matches_the_block_band(R, F, n, band, seed):
    reads, refs = synth.make_pairs(n, R, F, seed=seed, sub_rate=0.1, indel_rate=0.01 if R <= 4000 else 0.0, n_run_frac=0.1,
                                   short_frac=0.15, lowercase_frac=0.05, junk_frac=0.05)
    for gaps in ((-3, -3), (-2, -4)):
        sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gaps[0], gaps[1]))
        eng.set_band_width(band)
        d = eng.describe(0, n)
        assert (d["band_block_rows"], d["band_col_align"], d["score_cells"]) == (16, 1, "int32"), d
        got = eng.score_host(0, reads, refs, threads=4)
        eng.close()
        exp = cpu_ref.score_banded_sw(reads, refs, band, sc, threads=8, block_rows=16, col_align=1)
        assert np.array_equal(got, exp), (gaps, np.nonzero(got != exp)[0][:8], got[:8], exp[:8])
        assert (cpu_ref.score_banded_sw(reads, refs, band, sc, threads=8) <= got).all()
        if R <= 4000:
            assert (got <= cpu_ref.score(0, reads, refs, sc, threads=8)).all()


def test_block_chain_classifies_every_byte_value():
    """All 256 byte values in reads and references (the chain's event code takes base classes out of a 64-bit table constant:
    letters 'A' + t for t < 20, everything else -- control bytes, punctuation, bytes >= 0x80, the letters' neighbours -- is no
    base).  Each pair carries 32 consecutive byte values at the same read / reference offsets, upper and lower case bases
    around them; against the oracle on the chain's blocks, linear and affine gaps."""
    R, F, n, band = 640, 700, 8, 64
    reads, refs = synth.make_pairs(n, R, F, seed=71, sub_rate=0.05, n_run_frac=0.0, short_frac=0.0)
    refs[:, :R] = reads                                   # the diagonal is inside the band
    rng = np.random.default_rng(72)
    for p in range(n):
        low = rng.choice(R - 1, size=40, replace=False)
        reads[p, low] |= 0x20                              # some lower-case bases
        vals = np.arange(32 * p, 32 * p + 32, dtype=np.uint8)
        at = rng.choice(R - 1, size=32, replace=False)
        reads[p, at] = vals
        refs[p, at[::2]] = vals[::2]                       # half of them face the same byte, half a base
    seen = set(np.unique(reads).tolist())
    assert seen.issuperset(range(256))
    for aff in (None, (-5, -1, -5, -1)):
        sc = cpu_ref.Scoring.make(2, -1, -3, -3, *(aff or ()))
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **(dict(zip(("open_read", "ext_read", "open_ref", "ext_ref"), aff)) if aff else {})))
        eng.set_band_width(band)
        assert eng.describe(0, n)["band_block_rows"] == 16
        got = eng.score_host(0, reads, refs, threads=2)
        eng.close()
        exp = cpu_ref.score_banded_sw(reads, refs, band, sc, threads=4, block_rows=16, col_align=1, affine=aff is not None)
        assert np.array_equal(got, exp), (aff, got, exp)


def test_block_chain_against_the_strip_kernel(monkeypatch):
    """The two kernels compute two documented supersets of the per-cell band: blocks of 16 rows (the chain) inside
    blocks of 160 rows with columns aligned to 4 (the strips).  Each equals the oracle's statement of its own blocks."""
    R, F, n, band = 1500, 1700, 40, 48
    reads, refs = synth.make_pairs(n, R, F, seed=31, indel_rate=0.03, n_run_frac=0.05, short_frac=0.08)
    eng = hipkernel.Engine(R, F)
    eng.set_band_width(band)
    chain = eng.score_host(0, reads, refs, threads=4)
    eng.close()
    debug_switches(monkeypatch, no_band_chain=1)
    eng = hipkernel.Engine(R, F)
    eng.set_band_width(band)
    d = eng.describe(0, n)
    assert (d["band_block_rows"], d["band_col_align"]) == (160, 4)
    strips = eng.score_host(0, reads, refs, threads=4)
    eng.close()
    assert np.array_equal(chain, cpu_ref.score_banded_sw(reads, refs, band, threads=8, block_rows=16, col_align=1))
    assert np.array_equal(strips, cpu_ref.score_banded_sw(reads, refs, band, threads=8, block_rows=160, col_align=4))
    assert (chain <= strips).all()


def test_block_chain_cells_beyond_int16_and_the_plugin_keys():
    """int32 cells: scores past 32767 saturate at the ABI's short; the plugin keys band_width (+ score_width) reach
    the same kernel; a band wider than the matrix is the reference's unbanded result."""
    R = F = 9000
    n = 5
    reads, refs = synth.make_pairs(n, R, F, seed=41, sub_rate=0.08, indel_rate=0.0, n_run_frac=0.0, short_frac=0.4)
    sc = cpu_ref.Scoring.make(5, -4, -6, -6)
    keys = dict(score_match=5, score_mismatch=-4, score_gap_read=-6, score_gap_ref=-6)
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=400, **keys) as hip:
        got = hip.score_alignments(0, reads, refs)
        hip.score_alignments(0, reads[:1], refs[:1])
        assert '"band_block_rows": 16' in hip.drain_log()
    exp = cpu_ref.score_banded_sw(reads, refs, 400, sc, threads=8, block_rows=16, col_align=1)
    assert np.array_equal(got, exp) and (exp == 32767).any() and ((exp > 0) & (exp < 32767)).any()
    R, F, n = 600, 650, 50
    reads, refs = synth.make_pairs(n, R, F, seed=42, indel_rate=0.02)
    with host.Plugin(build.HIP_PLUGIN, R, F, band_width=100000) as hip:
        assert np.array_equal(hip.score_alignments(0, reads, refs), cpu_ref.score(0, reads, refs, threads=8))


def test_block_chain_on_a_large_batch():
    """4,099 pairs (a last wave with three of its four pairs) of 2,000 x 2,000: tiling of a 64-pair block."""
    import torch
    R = F = 2000
    blk = 64
    reads, refs = synth.make_pairs(blk, R, F, seed=51, indel_rate=0.01, n_run_frac=0.1, short_frac=0.1)
    exp = cpu_ref.score_banded_sw(reads, refs, 256, threads=8, block_rows=16, col_align=1)
    eng = hipkernel.Engine(R, F)
    eng.set_band_width(256)
    d_reads = torch.from_numpy(reads).cuda().repeat(65, 1)[:4099].contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(65, 1)[:4099].contiguous()
    got = eng.score_device(0, d_reads, d_refs).cpu().numpy()
    eng.close()
    assert np.array_equal(got, np.tile(exp, 65)[:4099])


@pytest.mark.parametrize("R,F,n,band,seed", [(1000, 1000, 37, 64, 61), (1000, 1300, 33, 16, 62), (1300, 1000, 30, 32, 63),
                                             (700, 2100, 18, 128, 64), (100, 120, 50, 8, 65), (2049, 2000, 9, 512, 66),
                                             (10000, 10000, 4, 512, 67)])
def test_affine_block_chain_matches_the_block_band(R, F, n, band, seed):
    """Round 4: affine gaps on the chain too (E in registers, F handed from lane to lane beside H): symmetric scores (H - open
    shared by E and F) and four different ones, unit-delay (DPP) and ring shapes -- against the oracle's banded Gotoh
    recurrence on the chain's own blocks (16 rows, column alignment 1), between the per-cell band and the full matrix."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, sub_rate=0.1, indel_rate=0.02 if R <= 4000 else 0.001, n_run_frac=0.1,
                                   short_frac=0.15, lowercase_frac=0.05, junk_frac=0.05)
    for aff in ((-5, -1, -5, -1), (-4, -2, -6, -1)):
        sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
        eng.set_band_width(band)
        d = eng.describe(0, n)
        assert (d["band_block_rows"], d["band_col_align"], d["score_cells"]) == (16, 1, "int32"), d
        got = eng.score_host(0, reads, refs, threads=4)
        eng.close()
        exp = cpu_ref.score_banded_sw(reads, refs, band, sc, threads=8, block_rows=16, col_align=1, affine=True)
        assert np.array_equal(got, exp), (aff, np.nonzero(got != exp)[0][:8], got[:8], exp[:8])
        assert (cpu_ref.score_banded_sw(reads, refs, band, sc, threads=8, affine=True) <= got).all()
        if R <= 2100:
            assert (got <= cpu_ref.score(0, reads, refs, sc, threads=8, affine=True, wide=True)).all()
