"""GPU parity of the score path: libHIPKernel.so vs the oracle, bit-exact (integer DP).

Every case goes through the C ABI: either the versalignLib plugin protocol
(host.Plugin -> spawn_alignment_kernel -> AlignmentKernel::score_alignments) or the flat
device-resident entry point (valign_hip_score_device)."""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth

from conftest import ref_kernel, debug_switches

pytestmark = pytest.mark.gpu

SHAPES = [
    # (R, F, n, seed)
    (64, 128, 1000, 11),      # BASELINE config 1
    (150, 500, 777, 12),      # headline shape, odd pair count (tail group / tail wave)
    (12, 20, 300, 13),
    (33, 70, 301, 14),
    (16, 16, 64, 15),
    (1, 1, 5, 16),
    (100, 37, 129, 17),       # read longer than ref
    (250, 300, 65, 18),
    (600, 700, 17, 19),
]


def _data(R, F, n, seed):
    return synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08,
                            lowercase_frac=0.05, junk_frac=0.05)


@pytest.mark.parametrize("R,F,n,seed", SHAPES)
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_plugin_scores_match_oracle(R, F, n, seed, gaps):
    reads, refs = _data(R, F, n, seed)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1],
                     num_threads=4) as hip:
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8)
            assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8])


@pytest.mark.parametrize("R,F,n,seed", SHAPES[:5])
def test_plugin_scores_match_reference_kernels(R, F, n, seed):
    """Against the reference's own compiled kernels where they travelled with the repo:
    SSE gives full 16-bit scores, Default only the low byte (DefaultKernel.cpp:137,199)."""
    sse, default = ref_kernel("SSE"), ref_kernel("Default")
    if not sse or not default:
        pytest.skip("oracle/_ref not built")
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip, host.Plugin(sse, R, F) as s, \
            host.Plugin(default, R, F) as d:
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got, s.score_alignments(opt, reads, refs))
            assert np.array_equal(got & 0xFF, d.score_alignments(opt, reads, refs) & 0xFF)


@pytest.mark.parametrize("R,F,n,seed", [SHAPES[0], SHAPES[1], SHAPES[3]])
@pytest.mark.parametrize("aff", [(-5, -1, -4, -2), (-3, -3, -3, -3), (-6, -2, -6, -2)])
def test_affine_scores(R, F, n, seed, aff):
    """Affine extension (not in the reference): own oracle, plus open == extend == linear."""
    reads, refs = _data(R, F, n, seed)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1],
                score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8, affine=True)
            assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8])
            if aff == (-3, -3, -3, -3):
                lin = cpu_ref.score(opt, reads, refs, cpu_ref.Scoring.make(2, -1, -3, -3), threads=8)
                assert np.array_equal(got, lin)


@pytest.mark.parametrize("geom", [(16, 10), (16, 12), (32, 8), (32, 10), (64, 8), (64, 12)])
def test_every_geometry_agrees(geom):
    """The same batch through forced kernel geometries (lanes per pair group, rows per lane)."""
    R, F, n = 150, 500, 203
    reads, refs = _data(R, F, n, 21)
    sc = cpu_ref.Scoring.make()
    with host.Plugin(build.HIP_PLUGIN, R, F, hip_group_lanes=geom[0], hip_rows_per_lane=geom[1]) as hip:
        for opt in (host.SW, host.NW):
            assert np.array_equal(hip.score_alignments(opt, reads, refs), cpu_ref.score(opt, reads, refs, sc, threads=8))


def test_device_resident_entry_point():
    import torch
    R, F, n = 150, 500, 4099
    reads, refs = _data(R, F, n, 31)
    eng = hipkernel.Engine(R, F)
    d_reads = torch.from_numpy(reads).cuda()
    d_refs = torch.from_numpy(refs).cuda()
    for opt in (0, 1):
        got = eng.score_device(opt, d_reads, d_refs).cpu().numpy()
        assert np.array_equal(got, cpu_ref.score(opt, reads, refs, threads=8))
    # unaligned views (odd byte offsets of the batch) must still be exact
    got = eng.score_device(0, d_reads[1:], d_refs[1:]).cpu().numpy()
    assert np.array_equal(got, cpu_ref.score(0, reads[1:], refs[1:], threads=8))
    eng.close()


def test_unsupported_opt_is_a_silent_noop_and_empty_batch():
    R, F = 20, 30
    reads, refs = _data(R, F, 10, 41)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        out = hip.score_alignments(2, reads, refs)          # opt & 0xF == 2: nothing happens
        assert not out.any()
        assert hip.score_alignments(0, reads[:0], refs[:0]).shape == (0,)


def test_missing_required_key_throws_like_the_reference():
    with pytest.raises(host.PluginError, match="Lacking parameters"):
        host.Plugin(build.HIP_PLUGIN, 10, 10, score_match=None)


def test_full_size_properties_linear_gap():
    """The BASELINE config 2 shape in the reference's own linear-gap model at a size the oracle cannot
    cover in seconds: properties only.  The batch is 64 copies of a 4096-pair block -> scores repeat with
    period 4096, and the first block equals the oracle."""
    import torch
    R, F, blk, reps = 150, 500, 4096, 64
    reads, refs = synth.make_pairs(blk, R, F, seed=51)
    eng = hipkernel.Engine(R, F)
    d_reads = torch.from_numpy(reads).cuda().repeat(reps, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(reps, 1).contiguous()
    for opt in (0, 1):
        got = eng.score_device(opt, d_reads, d_refs).cpu().numpy().reshape(reps, blk)
        assert (got == got[0]).all()
        assert np.array_equal(got[0], cpu_ref.score(opt, reads, refs, threads=8))
    eng.close()


@pytest.mark.parametrize("cells", ["f16", "int16"])
def test_full_size_properties_config2(monkeypatch, cells):
    """BASELINE config 2 as stated: 1,048,576 pairs of 150 x 500, Smith-Waterman AFFINE (open -5, extend -1),
    on the kernel the engine picks by itself (half-float cells) and on the int16-cell kernel bench.py headlines.
    The batch is 256 copies of a 4096-pair block: scores must repeat with that period (checked on the device)
    and the first block must equal the oracle; NW variant as well."""
    import torch
    if cells == "int16":
        debug_switches(monkeypatch, no_f16=1)
    R, F, blk, reps = 150, 500, 4096, 256
    reads, refs = synth.make_pairs(blk, R, F, seed=53, indel_rate=0.01)
    aff = (-5, -1, -5, -1)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    assert eng.describe(0, blk * reps)["score_cells"] == cells
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    d_reads = torch.from_numpy(reads).cuda().repeat(reps, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(reps, 1).contiguous()
    assert d_reads.shape[0] == 1 << 20
    for opt in (0, 1):
        got = eng.score_device(opt, d_reads, d_refs).view(reps, blk)
        assert bool((got == got[0:1]).all())
        assert np.array_equal(got[0].cpu().numpy(), cpu_ref.score(opt, reads, refs, osc, threads=8, affine=True))
    eng.close()


@pytest.mark.parametrize("aff", [(-5, -1, -5, -1), (-7, -2, -7, -2), (-4, -4, -4, -4), (-5, -1, -5, -2)])
def test_symmetric_and_asymmetric_affine_kernels(aff):
    """open_read == open_ref and ext_read == ext_ref selects the shared-subtract kernel variant;
    both variants must equal the affine oracle."""
    R, F, n = 150, 500, 515
    reads, refs = _data(R, F, n, 71)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1],
                score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8, affine=True)
            assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8])


@pytest.mark.parametrize("match,cells", [(13, "f16"), (14, "int16")])
def test_half_float_cells_are_exact_up_to_their_limit(match, cells):
    """Symmetric affine Smith-Waterman runs on packed half floats while every value stays an integer of
    magnitude <= 2048 (150 x 13 = 1950), on int16 beyond (150 x 14 = 2100).  Perfect matches drive the
    scores to the top of the range; both must equal the oracle.  The NW variant's tilted frame (gap extensions
    free, every cell plus 3 per row and column, centred on zero) spans 1950 + 1986 here: +-1968 plus three
    openings still fits, 2100 + 1986 does not."""
    R, F, n = 150, 500, 600
    reads, refs = synth.make_pairs(n, R, F, seed=81, sub_rate=0.0, n_run_frac=0.0, short_frac=0.0)
    noisy, _ = synth.make_pairs(n, R, F, seed=81, sub_rate=0.1, n_run_frac=0.0, short_frac=0.0)
    reads[n // 2:] = noisy[n // 2:]
    aff = dict(open_read=-20, ext_read=-3, open_ref=-20, ext_ref=-3)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(match, -11, -20, -20, **aff))
    assert eng.describe(host.SW)["score_cells"] == cells
    assert eng.describe(host.NW)["score_cells"] == cells
    import torch
    for opt in (host.SW, host.NW):
        got = eng.score_device(opt, torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()).cpu().numpy()
        exp = cpu_ref.score(opt, reads, refs, cpu_ref.Scoring.make(match, -11, -20, -20, **aff), threads=8, affine=True)
        assert exp.max() == 150 * match
        assert np.array_equal(got, exp), (opt, np.nonzero(got != exp)[0][:8])
    eng.close()


@pytest.mark.parametrize("R,F,n,long_scores", [(150, 8000, 33, 1), (150, 4000, 41, 1), (150, 2000, 50, 1), (150, 1000, 60, 0), (100, 2000, 50, 0),
                                                 (500, 20000, 9, 1)])
def test_short_reads_against_long_references(R, F, n, long_scores):
    """A reference that starves LDS (the resident kernels keep its slab numbers whole: four waves per CU at 2 000 columns,
    one at 8 000) sends score_alignments to the long-read kernels, whose slab numbers go through a ring (single-strip
    instances, half-float cells for SW with one gap score) -- unless their 160-row strips pad the read worse than the
    resident geometry (100 rows) -- and leaves compute_alignments on a resident plan of its own while that one still fits:
    both against the oracle, linear and affine."""
    reads, refs = synth.make_pairs(n, R, F, seed=R + F, indel_rate=0.01, n_run_frac=0.1, short_frac=0.2, lowercase_frac=0.05, junk_frac=0.05)
    eng = hipkernel.Engine(R, F)
    assert eng.describe()["long_mode"] == long_scores
    assert eng.describe(host.SW)["score_cells"] == "f16"        # (round 4: the long-read kernel has half-float cells too, SW / one gap score)
    eng.close()
    for keys, osc, okw in ((dict(), cpu_ref.Scoring.make(), dict()),
                           (dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1),
                            cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1), dict(affine=True))):
        with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, **keys) as hip:
            for opt in (host.SW, host.NW):
                assert np.array_equal(hip.score_alignments(opt, reads, refs), cpu_ref.score(opt, reads, refs, osc, threads=8, **okw)), (opt, keys)
                rows, idx = hip.compute_alignments(opt, reads, refs, normalise=False)
                exp_rows, exp_idx = cpu_ref.align(opt, reads, refs, osc, threads=8, **okw)
                assert np.array_equal(idx, exp_idx) and np.array_equal(rows, exp_rows), (opt, keys)


def test_int16_range_is_checked_per_call():
    """A shape whose cells could leave int16 never wraps silently (as the reference would): scores and (since round 4, every
    mode) alignments move to int32 cells."""
    R, F = 2000, 2000
    reads, refs = _data(R, F, 4, 81)
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=20) as hip:     # 2000 * 20 > 32000
        rows, idx = hip.compute_alignments(0, reads, refs, normalise=False)
        exp_rows, exp_idx = cpu_ref.align(0, reads, refs, cpu_ref.Scoring.make(20, -1, -3, -3), threads=8, wide=True)
        assert np.array_equal(idx, exp_idx) and np.array_equal(rows, exp_rows)
        got = hip.score_alignments(0, reads, refs)
        assert np.array_equal(got, cpu_ref.score(0, reads, refs, cpu_ref.Scoring.make(20, -1, -3, -3), threads=8, wide=True))
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:                      # default scores fit
        assert np.array_equal(hip.score_alignments(0, reads, refs), cpu_ref.score(0, reads, refs, threads=8))


def test_flat_host_entry_point():
    """valign_hip_score_host: the virtual without the C++ object (char** in, shorts out)."""
    import ctypes
    R, F, n = 64, 128, 777
    reads, refs = _data(R, F, n, 82)
    eng = hipkernel.Engine(R, F)
    rp = (ctypes.c_void_p * n)(*[reads[i].ctypes.data for i in range(n)])
    fp = (ctypes.c_void_p * n)(*[refs[i].ctypes.data for i in range(n)])
    out = np.zeros(n, dtype=np.int16)
    for opt in (0, 1):
        rc = hipkernel.lib().valign_hip_score_host(eng._h, opt, n, rp, fp, out.ctypes.data, 4)
        assert rc == 0
        assert np.array_equal(out, cpu_ref.score(opt, reads, refs, threads=8))
    eng.close()


def test_ragged_batches_skip_trailing_padding_exactly():
    """Smith-Waterman kernels do not sweep the trailing columns in which no reference of a wave has an
    ACGT base (NUL padding of ragged batches, N tails): scores and alignments must not change."""
    R, F, n = 150, 500, 1024
    reads, refs = synth.make_pairs(n, R, F, seed=88, indel_rate=0.01, n_run_frac=0.02, short_frac=0.0)
    keep = 150 + (np.arange(n) * 37) % 300                      # true reference lengths 150..449
    for i in range(n):
        refs[i, keep[i]:] = 0 if i % 3 else ord("N")              # NUL padding or an N tail
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        assert np.array_equal(hip.score_alignments(0, reads, refs), cpu_ref.score(0, reads, refs, threads=8))
        assert np.array_equal(hip.score_alignments(1, reads, refs), cpu_ref.score(1, reads, refs, threads=8))
        rows, idx = hip.compute_alignments(0, reads, refs, normalise=False)
        erows, eidx = cpu_ref.align(0, reads, refs, threads=8)
        assert np.array_equal(idx, eidx) and np.array_equal(rows, erows)


@pytest.mark.parametrize("aff", [None, (-5, -1, -5, -1), (-5, -1, -4, -2)])
def test_half_float_and_int16_forms_agree(monkeypatch, aff):
    """Every recurrence has an int16 form (the fallback beyond the half-float range) and, where cells stay
    small integers, a half-float form: same scores from both, on the same batch, and both equal the oracle."""
    import torch
    R, F, n = 150, 500, 3000
    reads, refs = _data(R, F, n, 91)
    kw = {} if aff is None else dict(open_read=aff[0], ext_read=aff[1], open_ref=aff[2], ext_ref=aff[3])
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, **kw)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    fast = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **kw))
    debug_switches(monkeypatch, no_f16=1)
    plain = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **kw))
    debug_switches(monkeypatch, no_f16=None)
    for opt in (host.SW, host.NW):
        assert fast.describe(opt)["score_cells"] == "f16" and plain.describe(opt)["score_cells"] == "int16"
        a = fast.score_device(opt, d_reads, d_refs).cpu().numpy()
        b = plain.score_device(opt, d_reads, d_refs).cpu().numpy()
        exp = cpu_ref.score(opt, reads, refs, osc, threads=8, affine=aff is not None)
        assert np.array_equal(a, exp) and np.array_equal(b, exp), opt
    fast.close()
    plain.close()


def test_small_calls_run_on_the_pinned_staging_directly(monkeypatch):
    """Calls of a few hundred KB skip the chunk pipeline: the kernels read the gathered sequences out of pinned host
    memory and (scores) write their results there.  Same results as the pipeline, which VALIGN_HIP_DEBUG direct_bytes=0 forces."""
    R, F, n = 64, 128, 1000                                       # BASELINE configs[0]
    reads, refs = synth.make_pairs(n, R, F, seed=5, indel_rate=0.03, n_run_frac=0.05, short_frac=0.1)
    exp = [cpu_ref.score(opt, reads, refs, threads=8) for opt in (0, 1)]
    exp_rows = [cpu_ref.align(opt, reads, refs, threads=8) for opt in (0, 1)]
    for direct in (True, False):
        if not direct:
            debug_switches(monkeypatch, direct_bytes=0)
        eng = hipkernel.Engine(R, F)
        for opt in (0, 1):
            assert np.array_equal(eng.score_host(opt, reads, refs, threads=2), exp[opt])
            assert eng.describe(opt, n)["direct_call"] == (1 if direct else 0)
            rows, idx = eng.align_host(opt, reads, refs, threads=2)
            assert np.array_equal(idx, exp_rows[opt][1]) and np.array_equal(rows, exp_rows[opt][0])
            assert eng.describe(opt, n)["direct_call"] == (2 if direct else 0)       # 2: fill + traceback in ONE launch
        # a large call on the same engine goes through the pipeline again
        big_r, big_f = np.tile(reads, (40, 1)), np.tile(refs, (40, 1))
        assert np.array_equal(eng.score_host(0, big_r, big_f, threads=4), np.tile(exp[0], 40))
        assert eng.describe(0, n)["direct_call"] == 0
        eng.close()


@pytest.mark.parametrize("gap,cells", [(-3, "f16"), (-10, "int16"), (-15, "int32")])
def test_nw_frame_keeps_inside_its_cell_range(gap, cells):
    """The NW score kernels keep cell (p, j) plus |gap| * (p + j) (tilted frame: gap steps cost nothing, the diagonal
    pays for both).  With a forced 64 x 32 geometry (2048 padded rows) that adds up to |gap| * 2350 at the far corner:
    half floats while the centred range stays inside +-2048, int16 while it stays inside the int16 range, int32 cells
    (strip path) beyond -- and the oracle's scores every time."""
    import torch
    R, F, n = 100, 300, 300
    reads, refs = synth.make_pairs(n, R, F, seed=123, indel_rate=0.02, n_run_frac=0.02, short_frac=0.05)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gap, gap), group_lanes=64, rows_per_lane=32)
    if gap == -3:        # 200 + 3 * 2350 = 7250: not even the centred half-float range -> int16 at this geometry
        cells = "int16"
    assert eng.describe(host.NW)["score_cells"] == cells
    got = eng.score_device(host.NW, torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()).cpu().numpy()
    assert np.array_equal(got, cpu_ref.score(host.NW, reads, refs, cpu_ref.Scoring.make(2, -1, gap, gap), threads=8))
    eng.close()
    if gap == -3:        # the engine's own geometry (7 x 16 = 112 rows): 200 + 3 * 414 centred = +-721 -> half floats
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gap, gap))
        assert eng.describe(host.NW)["score_cells"] == "f16"
        got = eng.score_device(host.NW, torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()).cpu().numpy()
        assert np.array_equal(got, cpu_ref.score(host.NW, reads, refs, cpu_ref.Scoring.make(2, -1, gap, gap), threads=8))
        eng.close()


def test_nw_half_float_kernel_serves_two_different_gap_scores(monkeypatch):
    """In the NW variant's tilted frame no gap constant is left in the recurrence (the row pays gap_ref, the column
    gap_read, the diagonal both through the profile), so the 3-instruction half-float kernel also serves
    gap_read != gap_ref; Smith-Waterman with two gap scores stays on int16.  Both forms equal the oracle."""
    import torch
    R, F, n = 150, 500, 2000
    reads, refs = _data(R, F, n, 97)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    exp = cpu_ref.score(host.NW, reads, refs, cpu_ref.Scoring.make(2, -1, -2, -4), threads=8)
    fast = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -2, -4))
    assert fast.describe(host.NW)["score_cells"] == "f16" and fast.describe(host.SW)["score_cells"] == "int16"
    assert np.array_equal(fast.score_device(host.NW, d_reads, d_refs).cpu().numpy(), exp)
    fast.close()
    debug_switches(monkeypatch, no_f16=1)
    plain = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -2, -4))
    assert plain.describe(host.NW)["score_cells"] == "int16"
    assert np.array_equal(plain.score_device(host.NW, d_reads, d_refs).cpu().numpy(), exp)
    plain.close()


@pytest.mark.parametrize("aff", [(0, 0, 0, 0), (-2, 0, -2, 0), None])
def test_nw_half_float_results_beyond_2048(aff):
    """Half-float NW cells live in a frame centred on zero, so a sweep whose best score is 170 x 13 = 2210 still runs
    on halves when the gap (extension) scores tilt the frame little or not at all -- but the SCORE itself is then beyond
    the exact range of a half: results are collected around half the best possible score and completed as integers
    (a fuzz soak found 2067 returned as 2068).  Odd scores above 2048 must come back exact."""
    import torch
    R, F, n = 170, 222, 300
    reads, refs = synth.make_pairs(n, R, F, seed=704, indel_rate=0.03, n_run_frac=0.05, short_frac=0.1)
    kw = {} if aff is None else dict(open_read=aff[0], ext_read=aff[1], open_ref=aff[2], ext_ref=aff[3])
    gap = 0 if aff is None else -7
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(13, 0, gap, gap, **kw))
    assert eng.describe(host.NW)["score_cells"] == "f16"
    got = eng.score_device(host.NW, torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()).cpu().numpy()
    exp = cpu_ref.score(host.NW, reads, refs, cpu_ref.Scoring.make(13, 0, gap, gap, **kw), threads=8, affine=aff is not None)
    assert exp.max() > 2048 and (exp[exp > 2048] % 2 == 1).any()
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:8]
    eng.close()
