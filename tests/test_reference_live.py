"""Oracle vs the reference's compiled kernels, live, on fresh random batches.

Runs wherever oracle/_ref exists (built from /root/reference by oracle/Makefile; the .so
files travel to the GPU box, the sources do not)."""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import host, synth

from conftest import ref_kernel


@pytest.mark.parametrize("R,F,seed", [(12, 20, 1), (33, 70, 2), (64, 128, 3), (150, 500, 4), (10, 8, 5)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_oracle_equals_reference(R, F, seed, gaps):
    default, sse = ref_kernel("Default"), ref_kernel("SSE")
    if not default or not sse:
        pytest.skip("oracle/_ref not built (no reference tree here)")
    n = 120
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02, n_run_frac=0.05, short_frac=0.1,
                                   lowercase_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    kw = dict(score_gap_read=gaps[0], score_gap_ref=gaps[1])
    with host.Plugin(default, R, F, **kw) as d, host.Plugin(sse, R, F, **kw) as s:
        for opt in (0, 1):
            mine = cpu_ref.score(opt, reads, refs, sc)
            assert np.array_equal(mine, s.score_alignments(opt, reads, refs))
            assert np.array_equal(mine & 0xFF, d.score_alignments(opt, reads, refs) & 0xFF)
            rows, idx = cpu_ref.align(opt, reads, refs, sc)
            drows, didx = d.compute_alignments(opt, reads, refs)
            assert np.array_equal(idx, didx) and np.array_equal(rows, drows)


def test_avx_kernel_agrees_with_sse():
    avx, sse = ref_kernel("AVX"), ref_kernel("SSE")
    if not avx or not sse:
        pytest.skip("oracle/_ref not built")
    reads, refs = synth.make_pairs(100, 40, 90, seed=8, n_run_frac=0.05, short_frac=0.1)
    with host.Plugin(avx, 40, 90) as a, host.Plugin(sse, 40, 90) as s:
        for opt in (0, 1):
            assert np.array_equal(a.score_alignments(opt, reads, refs), s.score_alignments(opt, reads, refs))
