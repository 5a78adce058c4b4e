"""Host-pointer path, round 3: 4-bit base classes across PCIe (score_alignments) and results copied straight into
page-locked caller buffers (valign_hip_align_host).  Both are transport changes: every result must equal the oracle
and the plain transport bit for bit."""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth
from conftest import debug_switches

pytestmark = pytest.mark.gpu


def _data(R, F, n, seed):
    return synth.make_pairs(n, R, F, seed=seed, indel_rate=0.02 if n <= 20000 else 0.0, n_run_frac=0.05, short_frac=0.08,
                            lowercase_frac=0.1, junk_frac=0.1)


@pytest.mark.parametrize("R,F,n,seed", [(150, 500, 6001, 1), (33, 71, 9000, 2), (101, 37, 7000, 3), (1, 1, 5000, 4),
                                        (64, 128, 10000, 5), (250, 301, 3000, 6)])
def test_packed_classes_give_identical_scores(monkeypatch, R, F, n, seed):
    """Every byte value, lower case, N runs, NUL padding, odd lengths: classes in, the scores of the bytes out.
    Small chunks so that several staging slots and a short last chunk are in play."""
    debug_switches(monkeypatch, direct_bytes=0)              # the chunk pipeline, whatever the size
    debug_switches(monkeypatch, chunk_bytes=1 << 18)
    reads, refs = _data(R, F, n, seed)
    for gaps in ((-3, -3), (-2, -4)):
        sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
        got = {}
        for packing in (1, 0):
            with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1], num_threads=6,
                             host_packing=packing) as hip:
                for opt in (host.SW, host.NW, host.SW):
                    got[packing, opt] = hip.score_alignments(opt, reads, refs)
                # (every call logs the engine's description first -- with the transport of the call BEFORE it)
                assert ('"packed_classes": 1' in hip.drain_log()) == (packing == 1)
        for opt in (host.SW, host.NW):
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8)
            assert np.array_equal(got[1, opt], exp), (opt, np.nonzero(got[1, opt] != exp)[0][:8])
            assert np.array_equal(got[0, opt], exp)


def test_packed_classes_affine_long_and_flat_entry(monkeypatch):
    """The same transport in front of the affine kernels, the strip (long-read) kernels and the flat entry point."""
    debug_switches(monkeypatch, direct_bytes=0)
    R, F, n = 150, 500, 5003
    reads, refs = _data(R, F, n, 11)
    aff = (-5, -1, -4, -2)
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    for opt in (0, 1):
        exp = cpu_ref.score(opt, reads, refs, osc, threads=8, affine=True)
        eng.set_host_packing(1)
        assert np.array_equal(eng.score_host(opt, reads, refs, threads=5), exp)
        assert eng.describe(opt, n)["packed_classes"] == 1
        eng.set_host_packing(0)
        assert np.array_equal(eng.score_host(opt, reads, refs, threads=5), exp)
        assert eng.describe(opt, n)["packed_classes"] == 0
    with pytest.raises(hipkernel.HipKernelError):
        eng.set_host_packing(2)
    eng.close()
    R, F, n = 3001, 2777, 40                     # odd lengths on the strip path
    reads, refs = synth.make_pairs(n, R, F, seed=12, sub_rate=0.1, n_run_frac=0.2, short_frac=0.2, lowercase_frac=0.2, junk_frac=0.2)
    eng = hipkernel.Engine(R, F)
    assert eng.describe(0, n)["long_mode"] == 1
    for opt in (0, 1):
        assert np.array_equal(eng.score_host(opt, reads, refs, threads=4), cpu_ref.score(opt, reads, refs, threads=8))
    assert eng.describe(0, n)["packed_classes"] == 1
    eng.close()


def test_flat_results_go_straight_into_registered_buffers(monkeypatch):
    """valign_hip_align_host into buffers the caller registered once (valign_hip_host_register): the device's copy
    engine writes them directly (`direct_out`), chunk after chunk; unregistered buffers take the staged path; both
    equal the oracle.  Default tie-breaks and the SSE policy, linear and affine."""
    debug_switches(monkeypatch, align_chunk_bytes=6 << 20)          # ~3000 pairs per chunk: several slots
    R, F, n = 150, 500, 20011
    reads, refs = synth.make_pairs(n, R, F, seed=21, n_run_frac=0.05, short_frac=0.08, lowercase_frac=0.05)
    AL = R + F
    rows = np.full((n, 2, AL), 0xEE, dtype=np.uint8)
    idx = np.full((n, 4), -1, dtype=np.int16)
    hipkernel.host_register(rows)
    hipkernel.host_register(idx)
    try:
        with pytest.raises(hipkernel.HipKernelError, match="overlaps"):
            hipkernel.host_register(rows[5:])
        for scoring, osc, affine in ((hipkernel.Scoring.make(), cpu_ref.Scoring.make(), False),
                                     (hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1),
                                      cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1), True)):
            eng = hipkernel.Engine(R, F, scoring)
            for opt in (host.SW, host.NW):
                exp_rows, exp_idx = cpu_ref.align(opt, reads, refs, osc, threads=8, affine=affine)
                rows[:] = 0xEE
                idx[:] = -1
                eng.align_host(opt, reads, refs, threads=6, out=(rows, idx))
                d = eng.describe(opt, n)
                assert d["direct_out"] == 1
                assert np.array_equal(idx, exp_idx) and np.array_equal(rows, exp_rows), (affine, opt)
                assert d["full_row_mb"] > 0 and d["d2h_row_mb"] == d["full_row_mb"]        # whole rows, straight from the copy engine
                # a part of the registered buffers works too (the call covers fewer pairs than were registered)
                rows[:] = 0xEE
                eng.align_host(opt, reads[:7001], refs[:7001], threads=6, out=(rows[:7001], idx[:7001]))
                assert eng.describe(opt, n)["direct_out"] == 1
                assert np.array_equal(rows[:7001], exp_rows[:7001]) and (rows[7001:] == 0xEE).all()
                plain = eng.align_host(opt, reads, refs, threads=6)                     # fresh, unregistered buffers
                d = eng.describe(opt, n)
                assert d["direct_out"] == 0
                # staged: only the columns that hold strings cross PCIe, packed on the device -- a third of the rows' bytes
                # for Smith-Waterman alignments of 150 bp reads in 650-byte rows; the scatter writes the zeros in front
                assert 0 < d["d2h_row_mb"] <= d["full_row_mb"]
                if opt == host.SW:
                    assert d["d2h_row_mb"] < 0.45 * d["full_row_mb"], d
                assert np.array_equal(plain[1], exp_idx) and np.array_equal(plain[0], exp_rows)
            eng.close()
    finally:
        hipkernel.host_unregister(rows)
        hipkernel.host_unregister(idx)
    with pytest.raises(hipkernel.HipKernelError, match="not the start"):
        hipkernel.host_unregister(rows)
    eng = hipkernel.Engine(R, F)                 # after unregistering: the staged path again
    got = eng.align_host(host.SW, reads[:5000], refs[:5000], threads=4, out=(rows[:5000], idx[:5000]))
    assert eng.describe(0, n)["direct_out"] == 0
    exp_rows, exp_idx = cpu_ref.align(host.SW, reads[:5000], refs[:5000], threads=8)
    assert np.array_equal(got[1], exp_idx) and np.array_equal(got[0], exp_rows)
    eng.close()
