"""Length-sorted batching: score calls that arrive with ragged, NUL-padded sequences are binned by
trimmed length ON THE DEVICE (ragged_kernels.hip.h: classification, packing by length class, one sweep per read class,
scores back in the caller's order) and swept at the bin's shape -- host-pointer calls chunk by chunk inside the pipeline,
device-resident batches in place.  The scores must be exactly those of the padded sweep -- checked against the oracle and
against the same library with the feature switched off -- and the engine must report that it swept fewer cells."""
import os

import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth

from conftest import ref_kernel, debug_switches

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_bins(monkeypatch):
    debug_switches(monkeypatch, ragged_min=64)      # read at engine creation: many bins at test sizes


CASES = [
    # (R, F, n, seed, threads)
    (150, 500, 20000, 41, 8),
    (150, 500, 5000, 42, 1),
    (64, 128, 3000, 43, 4),
    (250, 300, 4099, 44, 3),
    (33, 70, 2500, 45, 2),
]


@pytest.mark.parametrize("R,F,n,seed,threads", CASES)
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_ragged_scores_match_oracle(small_bins, R, F, n, seed, threads, gaps):
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=seed, n_run_frac=0.03, short_frac=0.02, junk_frac=0.02)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    exp = cpu_ref.score(host.SW, reads, refs, sc, threads=8)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gaps[0], gaps[1]))
    eng.set_ragged_batching(2)
    got = eng.score_host(host.SW, reads, refs, threads=threads)
    info = eng.describe(host.SW, n)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:8]
    assert info["ragged_batching"] == 2 and info["ragged_launches"] > 1
    assert info["ragged_cell_fraction"] < (0.75 if R >= 64 else 1.0)     # tiny shapes have two classes per side
    eng.set_ragged_batching(0)
    plain = eng.score_host(host.SW, reads, refs, threads=threads)
    assert np.array_equal(plain, exp)
    assert eng.describe(host.SW, n)["ragged_launches"] == 0
    eng.close()


def test_ragged_affine_and_nw(small_bins):
    R, F, n = 150, 500, 6000
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=46)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
    hsc = hipkernel.Scoring.make(2, -1, -3, -3, open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
    eng = hipkernel.Engine(R, F, hsc)
    eng.set_ragged_batching(1)
    got = eng.score_host(host.SW, reads, refs, threads=4)
    assert eng.describe()["ragged_launches"] > 1
    assert np.array_equal(got, cpu_ref.score(host.SW, reads, refs, sc, threads=8, affine=True))
    # the Needleman-Wunsch variant reads its result off the last row / column -- of the trimmed pair just as well: rows and
    # columns of padding score 0, every boundary value runs down its diagonal unchanged (round 3)
    got = eng.score_host(host.NW, reads, refs, threads=4)
    assert eng.describe()["ragged_launches"] > 1
    assert np.array_equal(got, cpu_ref.score(host.NW, reads, refs, sc, threads=8, affine=True))
    eng.close()


@pytest.mark.parametrize("R,F,n,seed,threads", CASES)
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_ragged_nw_scores_match_oracle(small_bins, R, F, n, seed, threads, gaps):
    """Length-sorted batches of the NW variant: identical to the padded sweep and to the oracle (N runs, junk bytes
    and fully padded sequences included), with fewer cells swept."""
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=seed + 100, n_run_frac=0.05, short_frac=0.03, junk_frac=0.03)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    exp = cpu_ref.score(host.NW, reads, refs, sc, threads=8)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gaps[0], gaps[1]))
    eng.set_ragged_batching(2)
    got = eng.score_host(host.NW, reads, refs, threads=threads)
    info = eng.describe(host.NW, n)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:8]
    assert info["ragged_launches"] > 1 and info["ragged_cell_fraction"] < (0.75 if R >= 64 else 1.0)
    eng.set_ragged_batching(0)
    assert np.array_equal(eng.score_host(host.NW, reads, refs, threads=threads), exp)
    eng.close()


def test_ragged_through_plugin_against_reference_sse(small_bins):
    """The plugin protocol end to end, checked against the reference's own SSE kernel (full scores)."""
    sse = ref_kernel("SSE")
    if not sse:
        pytest.skip("oracle/_ref not built")
    R, F, n = 150, 500, 4000
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=47)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, ragged_batching=1) as hip, host.Plugin(sse, R, F) as s, \
            host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4) as off:
        exp = s.score_alignments(host.SW, reads, refs)
        assert np.array_equal(hip.score_alignments(host.SW, reads, refs), exp)
        assert np.array_equal(off.score_alignments(host.SW, reads, refs), exp)
        assert '"ragged_batching": 0' in off.drain_log()


def test_uniform_batch_is_one_launch():
    """Full-length pairs: left alone by default; forced, one bin and one launch per chunk."""
    R, F, n = 150, 500, 3000
    reads, refs = synth.make_pairs(n, R, F, seed=48, n_run_frac=0.0, short_frac=0.0)
    exp = cpu_ref.score(host.SW, reads, refs, threads=8)
    eng = hipkernel.Engine(R, F)
    assert eng.describe()["ragged_batching"] == 0          # off unless asked for
    eng.set_ragged_batching(1)
    got = eng.score_host(host.SW, reads, refs, threads=2)
    assert eng.describe()["ragged_launches"] == 0          # mode 1: the sample says nothing to skip
    assert np.array_equal(got, exp)
    eng.set_ragged_batching(2)
    got = eng.score_host(host.SW, reads, refs, threads=2)
    assert eng.describe()["ragged_launches"] == 1
    assert np.array_equal(got, exp)
    with pytest.raises(hipkernel.HipKernelError):
        eng.set_ragged_batching(3)
    eng.close()


def test_all_empty_sequences():
    R, F, n = 64, 128, 500
    reads = np.zeros((n, R), dtype=np.uint8)
    refs = np.zeros((n, F), dtype=np.uint8)
    eng = hipkernel.Engine(R, F)
    assert not eng.score_host(host.SW, reads, refs).any()
    eng.close()


@pytest.mark.parametrize("alg", [host.SW, host.NW])
@pytest.mark.parametrize("R,F,n", [(150, 500, 30011), (64, 128, 9000), (37, 301, 4500)])
def test_ragged_device_resident_batch(small_bins, alg, R, F, n):
    """valign_hip_score_device on a batch that is already in HBM: classified, packed and swept by length class there."""
    import torch
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=51 + R, n_run_frac=0.03, short_frac=0.02, junk_frac=0.02)
    reads[7] = 0                                   # a pair of nothing but padding
    refs[7] = 0
    refs[11] = ord("N")                            # ... and a reference of nothing but N
    exp = cpu_ref.score(alg, reads, refs, threads=8)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    eng = hipkernel.Engine(R, F)
    plain = eng.score_device(alg, d_reads, d_refs).cpu().numpy()
    assert np.array_equal(plain, exp)
    for mode in (2, 1):
        eng.set_ragged_batching(mode)
        got = eng.score_device(alg, d_reads, d_refs).cpu().numpy()
        info = eng.describe(alg, n)
        assert np.array_equal(got, exp), (mode, np.nonzero(got != exp)[0][:8])
        assert info["ragged_launches"] > 1 and info["ragged_cell_fraction"] < (0.75 if R >= 64 else 1.0), info
    # a batch of full-length pairs: mode 1 looks at the device's histogram and sweeps the batch as it stands
    reads, refs = synth.make_pairs(n, R, F, seed=52, n_run_frac=0.0, short_frac=0.0)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    got = eng.score_device(alg, d_reads, d_refs).cpu().numpy()
    assert eng.describe(alg, n)["ragged_launches"] == 0
    assert np.array_equal(got, cpu_ref.score(alg, reads, refs, threads=8))
    eng.close()


def test_ragged_device_calls_on_two_streams_share_one_context_safely(small_bins):
    """One engine, two streams, back-to-back length-sorted device calls on different batches: the engine has ONE context
    (pinned histogram and tables, counters, packed buffers) for device-resident calls -- a call on another stream than the
    last one must first wait until that one is through with it, or it would rewrite the tables under its kernels."""
    import torch
    R, F, n = 150, 500, 40000
    batches = []
    for seed in (71, 72, 73, 74):
        reads, refs = synth.make_ragged_pairs(n, R, F, seed=seed, n_run_frac=0.03, short_frac=0.02)
        batches.append((torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda(), cpu_ref.score(host.SW, reads, refs, threads=8)))
    eng = hipkernel.Engine(R, F)
    eng.set_ragged_batching(2)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = []
    for k, (d_reads, d_refs, _) in enumerate(batches):             # alternating streams, nothing waited for in between
        outs.append(eng.score_device(host.SW, d_reads, d_refs, stream=streams[k & 1]))
    torch.cuda.synchronize()
    for k, (_, _, exp) in enumerate(batches):
        got = outs[k].cpu().numpy()
        assert np.array_equal(got, exp), (k, np.nonzero(got != exp)[0][:8])
    eng.close()


@pytest.mark.parametrize("packing", [1, 0])
def test_ragged_chunks_of_the_pipeline(small_bins, monkeypatch, packing):
    """Several chunks in flight: every chunk is classified while the host gathers the next one and swept one iteration
    later; affine scoring, both modes, 4-bit classes and ASCII across PCIe."""
    debug_switches(monkeypatch, chunk_bytes=1 << 20)          # ~1,600 pairs of 150 x 500 per chunk
    R, F, n = 150, 500, 23017
    reads, refs = synth.make_ragged_pairs(n, R, F, seed=53, n_run_frac=0.03, short_frac=0.02, junk_frac=0.02)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1))
    eng.set_host_packing(packing)
    eng.set_ragged_batching(2)
    for alg in (host.SW, host.NW):
        exp = cpu_ref.score(alg, reads, refs, sc, threads=8, affine=True)
        for threads in (1, 5):
            got = eng.score_host(alg, reads, refs, threads=threads)
            assert np.array_equal(got, exp), (alg, threads, np.nonzero(got != exp)[0][:8])
        info = eng.describe(alg, n)
        assert info["ragged_launches"] > 10 and info["ragged_cell_fraction"] < 0.75 and info["packed_classes"] == packing, info
    eng.close()
