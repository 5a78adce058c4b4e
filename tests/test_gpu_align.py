"""GPU parity of compute_alignments: gapped rows + coordinates, bit-exact against the oracle
(which is pinned to the reference Default kernel's tie-breaks) and against the committed
Default-kernel fixtures.  All calls go through the plugin ABI or the flat C entry point."""
import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth

from conftest import ref_kernel, debug_switches
from golden_util import golden_files, load

pytestmark = pytest.mark.gpu

SHAPES = [
    (64, 128, 500, 11),
    (150, 500, 333, 12),
    (12, 20, 300, 13),
    (33, 70, 301, 14),
    (16, 16, 65, 15),
    (1, 1, 5, 16),
    (100, 37, 129, 17),
    (250, 300, 33, 18),
]


def _data(R, F, n, seed):
    return synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.06, short_frac=0.1,
                            lowercase_frac=0.05, junk_frac=0.05)


def _assert_same(got, exp, what):
    rows, idx = got
    erows, eidx = exp
    bad = np.nonzero((idx != eidx).any(axis=1))[0]
    assert bad.size == 0, (what, "idx", bad[:5], idx[bad[:3]], eidx[bad[:3]])
    bad = np.nonzero((rows != erows).any(axis=(1, 2)))[0]
    assert bad.size == 0, (what, "rows", bad[:5])


@pytest.mark.parametrize("R,F,n,seed", SHAPES)
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_plugin_alignments_match_oracle(R, F, n, seed, gaps):
    reads, refs = _data(R, F, n, seed)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1],
                     num_threads=4) as hip:
        for opt in (host.SW, host.NW):
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            exp = cpu_ref.align(opt, reads, refs, sc, threads=8)
            _assert_same(got, exp, ("opt", opt))


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_plugin_alignments_match_default_kernel_fixtures(path):
    g = load(path)
    R, F = g["reads"].shape[1], g["refs"].shape[1]
    m, x, gr, gf = (int(v) for v in g["scoring"])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=m, score_mismatch=x, score_gap_read=gr,
                     score_gap_ref=gf) as hip:
        for opt, tag in ((0, "sw"), (1, "nw")):
            got = hip.compute_alignments(opt, g["reads"], g["refs"], normalise=False)
            _assert_same(got, (g["rows_" + tag], g["idx_" + tag]), (path, tag))
            assert np.array_equal(hip.score_alignments(opt, g["reads"], g["refs"]), g["score_" + tag])


def test_plugin_alignments_match_reference_default_live():
    default = ref_kernel("Default")
    if not default:
        pytest.skip("oracle/_ref not built")
    R, F, n = 150, 500, 200
    reads, refs = synth.make_pairs(n, R, F, seed=91, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip, host.Plugin(default, R, F) as d:
        for opt in (0, 1):
            _assert_same(hip.compute_alignments(opt, reads, refs), d.compute_alignments(opt, reads, refs), opt)


@pytest.mark.parametrize("geom", [(16, 10), (16, 12), (32, 8), (32, 10), (64, 12)])
def test_alignment_geometries(geom):
    R, F, n = 150, 500, 131
    reads, refs = _data(R, F, n, 23)
    with host.Plugin(build.HIP_PLUGIN, R, F, hip_group_lanes=geom[0], hip_rows_per_lane=geom[1]) as hip:
        for opt in (0, 1):
            _assert_same(hip.compute_alignments(opt, reads, refs, normalise=False),
                         cpu_ref.align(opt, reads, refs, threads=8), (geom, opt))


def test_align_device_entry_point_and_roundtrip_property():
    """Device entry point at a larger batch; plus a size-independent property: stripping the
    gaps from the two rows gives back substrings of the read / ref ending at the reported cell,
    and re-scoring the rows column by column reproduces the SW score."""
    import torch
    R, F, n = 150, 500, 20000
    reads, refs = synth.make_pairs(n, R, F, seed=61, indel_rate=0.01)
    eng = hipkernel.Engine(R, F)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    rows, idx = eng.align_device(0, d_reads, d_refs)
    scores = eng.score_device(0, d_reads, d_refs).cpu().numpy()
    rows, idx = rows.cpu().numpy(), idx.cpu().numpy()
    erows, eidx = cpu_ref.align(0, reads[:3000], refs[:3000], threads=8)
    _assert_same((rows[:3000], idx[:3000]), (erows, eidx), "device entry")
    cls = np.zeros(256, np.int64)
    for ch, c in zip(b"ATCG", (1, 2, 3, 4)):
        cls[ch] = c
        cls[ch | 0x20] = c
    AL = R + F
    assert (idx[:, 1] == AL - 1).all() and (idx[:, 3] == AL - 1).all() and (idx[:, 0] == idx[:, 2]).all()
    for i in range(0, n, 97):
        s = idx[i, 0]
        a, b = rows[i, 0, s:AL - 1], rows[i, 1, s:AL - 1]
        ca, cb = cls[a], cls[b]
        col = np.where((a == ord("-")), -3, np.where(b == ord("-"), -3,
                       np.where((ca > 0) & (cb > 0), np.where(ca == cb, 2, -1), 0)))
        assert col.sum() == scores[i]
        assert bytes(a[a != ord("-")]) in bytes(reads[i]) and bytes(b[b != ord("-")]) in bytes(refs[i])
    eng.close()


@pytest.mark.parametrize("R,F,n,seed", [(64, 128, 300, 31), (150, 500, 257, 32), (33, 70, 200, 33), (100, 37, 90, 34)])
@pytest.mark.parametrize("aff", [(-5, -1, -5, -1), (-3, -3, -3, -3), (-6, -2, -4, -1), (-2, -2, -4, -4)])
def test_affine_alignments(R, F, n, seed, aff):
    """BASELINE config 3 (NW affine + traceback) and its SW twin.  The affine model is an extension
    (parity unpinned by the reference): checked against the repo's Gotoh oracle, and for
    open == extend against the LINEAR oracle, i.e. the reference Default kernel's tie-breaks."""
    reads, refs = _data(R, F, n, seed)
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    keys = dict(score_gap_open_read=aff[0], score_gap_extend_read=aff[1],
                score_gap_open_ref=aff[2], score_gap_extend_ref=aff[3])
    with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
        for opt in (host.SW, host.NW):
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _assert_same(got, cpu_ref.align(opt, reads, refs, sc, threads=8, affine=True), (aff, opt))
            if aff[0] == aff[1] and aff[2] == aff[3]:
                lin = cpu_ref.Scoring.make(2, -1, aff[0], aff[2])
                _assert_same(got, cpu_ref.align(opt, reads, refs, lin, threads=8), ("degenerate", aff, opt))


def test_affine_alignment_rows_rescore_to_the_affine_score():
    """Property at a larger batch: re-scoring the emitted rows with the affine model (first gap
    base open, further ones extend) gives exactly the SW affine score of score_alignments."""
    import torch
    R, F, n = 150, 500, 30000
    reads, refs = synth.make_pairs(n, R, F, seed=62, indel_rate=0.02)
    aff = hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
    eng = hipkernel.Engine(R, F, aff)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    rows, idx = eng.align_device(0, d_reads, d_refs)
    scores = eng.score_device(0, d_reads, d_refs).cpu().numpy()
    rows, idx = rows.cpu().numpy(), idx.cpu().numpy()
    cls = np.zeros(256, np.int64)
    for ch, c in zip(b"ATCG", (1, 2, 3, 4)):
        cls[ch] = c
        cls[ch | 0x20] = c
    AL = R + F
    for i in range(0, n, 101):
        s = idx[i, 0]
        a, b = rows[i, 0, s:AL - 1], rows[i, 1, s:AL - 1]
        total, prev = 0, 0            # prev: 0 none/diag, 1 gap in read row, 2 gap in ref row
        for x, y in zip(a, b):
            if x == ord("-"):
                total += -1 if prev == 1 else -5
                prev = 1
            elif y == ord("-"):
                total += -1 if prev == 2 else -5
                prev = 2
            else:
                cx, cy = cls[x], cls[y]
                total += (2 if cx == cy else -1) if (cx and cy) else 0
                prev = 0
        assert total == scores[i], (i, total, scores[i])
    eng.close()


@pytest.mark.parametrize("R,F,n,seed", [(64, 128, 400, 41), (150, 500, 203, 42), (33, 70, 201, 43), (16, 16, 64, 44)])
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_sse_traceback_policy(R, F, n, seed, gaps):
    """traceback_policy = 1: the tie-breaks of the reference's SSE2/AVX2 kernels."""
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.15, short_frac=0.12,
                                   lowercase_frac=0.05, junk_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_gap_read=gaps[0], score_gap_ref=gaps[1],
                     traceback_policy=1) as hip:
        for opt in (0, 1):
            got = hip.compute_alignments(opt, reads, refs, normalise=False)
            _assert_same(got, cpu_ref.align(opt, reads, refs, sc, threads=8, policy="sse"), ("sse", opt))


@pytest.mark.parametrize("path", [p for p in golden_files() if "kat" not in p], ids=lambda p: p.split("/")[-1][:-4])
def test_sse_policy_matches_sse_kernel_fixtures(path):
    g = load(path)
    R, F = g["reads"].shape[1], g["refs"].shape[1]
    m, x, gr, gf = (int(v) for v in g["scoring"])
    with host.Plugin(build.HIP_PLUGIN, R, F, score_match=m, score_mismatch=x, score_gap_read=gr,
                     score_gap_ref=gf, traceback_policy=1) as hip:
        for opt, tag in ((0, "sw"), (1, "nw")):
            n8 = g["sse_idx_" + tag].shape[0]
            got = hip.compute_alignments(opt, g["reads"][:n8], g["refs"][:n8], normalise=False)
            _assert_same(got, (g["sse_rows_" + tag], g["sse_idx_" + tag]), (path, tag))


def test_sse_policy_live_against_sse_and_avx_kernels():
    sse, avx = ref_kernel("SSE"), ref_kernel("AVX")
    if not sse or not avx:
        pytest.skip("oracle/_ref not built")
    R, F, n = 150, 500, 160
    reads, refs = synth.make_pairs(n, R, F, seed=93, indel_rate=0.02, n_run_frac=0.1, short_frac=0.08)
    with host.Plugin(build.HIP_PLUGIN, R, F, traceback_policy=1) as hip, host.Plugin(sse, R, F) as s, \
            host.Plugin(avx, R, F) as a:
        for opt in (0, 1):
            got = hip.compute_alignments(opt, reads, refs)
            _assert_same(got, s.compute_alignments(opt, reads, refs), ("sse", opt))
            _assert_same(got, a.compute_alignments(opt, reads, refs), ("avx", opt))


def test_host_paths_with_several_chunks():
    """The host-pointer pipeline (gather -> pinned -> H2D -> kernels -> D2H -> scatter) with batches
    that span several staging chunks, against the device-resident entry points and the oracle."""
    import torch
    R, F, blk = 150, 500, 8192
    r0, f0 = synth.make_pairs(blk, R, F, seed=71, indel_rate=0.01, n_run_frac=0.02, short_frac=0.02)
    n = 300000
    reps = -(-n // blk)
    reads = np.tile(r0, (reps, 1))[:n]
    refs = np.tile(f0, (reps, 1))[:n]
    exp_scores = cpu_ref.score(0, r0, f0, threads=8)
    exp_rows, exp_idx = cpu_ref.align(1, r0, f0, threads=8)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=8) as hip:
        scores = hip.score_alignments(0, reads, refs)                       # 4 chunks of ~77k pairs
        rows, idx = hip.compute_alignments(1, reads, refs, normalise=False)  # 3 chunks of ~137k pairs
    for k in range(reps):
        lo, hi = k * blk, min(n, (k + 1) * blk)
        assert np.array_equal(scores[lo:hi], exp_scores[:hi - lo]), k
        assert np.array_equal(idx[lo:hi], exp_idx[:hi - lo]), k
        assert np.array_equal(rows[lo:hi], exp_rows[:hi - lo]), k


def test_full_size_properties_config3():
    """BASELINE config 3 at full size (1M pairs, 150 x 500, NW affine + traceback): the batch is 256
    copies of a 4096-pair block, so rows and coordinates must repeat with that period (checked on
    the device), and the first block must equal the oracle."""
    import torch
    R, F, blk, reps = 150, 500, 4096, 256
    reads, refs = synth.make_pairs(blk, R, F, seed=52, indel_rate=0.01)
    aff = (-5, -1, -5, -1)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    d_reads = torch.from_numpy(reads).cuda().repeat(reps, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(reps, 1).contiguous()
    rows, idx = eng.align_device(1, d_reads, d_refs)
    torch.cuda.synchronize()
    rows = rows.view(reps, blk, 2, R + F)
    idx = idx.view(reps, blk, 4)
    assert bool((rows == rows[0:1]).all()) and bool((idx == idx[0:1]).all())
    erows, eidx = cpu_ref.align(1, reads, refs, cpu_ref.Scoring.make(2, -1, -3, -3, *aff), threads=8, affine=True)
    assert np.array_equal(idx[0].cpu().numpy(), eidx) and np.array_equal(rows[0].cpu().numpy(), erows)
    eng.close()


@pytest.mark.parametrize("R,F,n,K", [(2000, 1200, 24, 32), (1500, 2500, 16, 24), (700, 3000, 33, 12), (2000, 1200, 24, 0), (1500, 2500, 16, 0)])
def test_largest_register_geometries(R, F, n, K):
    """Reads of up to 2048 rows fit the register sweep (64 lanes per pair, 12-32 rows per lane).  Since round 4 the engine
    prefers the long-read kernels beyond 1536 rows (scores) / row strips beyond 1024 rows (alignments) -- measured faster,
    profiles/r04_rate_sweep.txt --, so the tall register geometries are exercised by forcing them (K > 0) and the engine's
    own choice beside them (K = 0): scores and alignments of both modes against the oracle."""
    import torch
    reads, refs = synth.make_pairs(n, R, F, seed=R, indel_rate=0.01, n_run_frac=0.1, short_frac=0.2)
    eng = hipkernel.Engine(R, F, group_lanes=64 if K else 0, rows_per_lane=K)
    d = eng.describe()
    if K:
        assert d["group_lanes"] == 64 and d["rows_per_lane"] == K and not d["long_mode"]
    else:
        assert d["long_mode"] == (1 if R > 1536 else 0)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    for opt in (host.SW, host.NW):
        rows, idx = eng.align_device(opt, d_reads, d_refs)
        _assert_same((rows.cpu().numpy(), idx.cpu().numpy()), cpu_ref.align(opt, reads, refs, threads=8), ("opt", opt))
        assert np.array_equal(eng.score_device(opt, d_reads, d_refs).cpu().numpy(),
                              cpu_ref.score(opt, reads, refs, threads=8))
    eng.close()


@pytest.mark.parametrize("tuning", [0, 1, 2])
def test_host_malloc_tuning_key_changes_nothing_but_the_allocator(tuning):
    """host_malloc_tuning: 2 (default) grows glibc's arenas in 256 MB steps, 1 also stops trimming, 0 leaves the host's
    allocator alone -- the rows handed out are the same operator new[] blocks with the same contents."""
    R, F, n = 64, 128, 5000
    reads, refs = _data(R, F, n, 93)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, host_malloc_tuning=tuning) as hip:
        for _ in range(2):                                    # second call recycles the first call's blocks
            got = hip.compute_alignments(host.SW, reads, refs, normalise=False)
        _assert_same(got, cpu_ref.align(host.SW, reads, refs, threads=8), "tuned")


@pytest.mark.parametrize("kind", ["linear", "linear_asym", "sse", "affine", "affine_asym"])
def test_equality_test_kernels_stay_correct(monkeypatch, kind):
    """The tagged-cell kernels are the default; beyond their int16 headroom the engine falls back to the
    kernels that derive pointers by equality tests.  VALIGN_HIP_DEBUG no_tag forces that path: same alignments."""
    import torch
    debug_switches(monkeypatch, no_tag=1)
    R, F, n = 150, 500, 700
    reads, refs = _data(R, F, n, 97)
    kw, policy, gaps = {}, "default", (-3, -3)
    if kind == "linear_asym":
        gaps = (-2, -4)
    elif kind == "affine":
        kw = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
    elif kind == "affine_asym":
        kw = dict(open_read=-5, ext_read=-1, open_ref=-4, ext_ref=-2)
    elif kind == "sse":
        policy = "sse"
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, gaps[0], gaps[1], **kw))
    if policy == "sse":
        eng.set_traceback_policy(1)
    osc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1], **kw)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    for opt in (host.SW, host.NW):
        rows, idx = eng.align_device(opt, d_reads, d_refs)
        exp = cpu_ref.align(opt, reads, refs, osc, threads=8, affine=bool(kw), policy=policy)
        _assert_same((rows.cpu().numpy(), idx.cpu().numpy()), exp, (kind, opt))
    eng.close()


def test_scores_beyond_the_tag_headroom_take_the_fallback_kernels():
    """match 60 at 150 x 500: cells reach 9 000, four (eight) times that no longer fits int16, so the engine
    must pick the equality-test kernels by itself -- linear and affine, both modes."""
    import torch
    R, F, n = 150, 500, 400
    reads, refs = _data(R, F, n, 98)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    for kw in ({}, dict(open_read=-70, ext_read=-20, open_ref=-70, ext_ref=-20)):
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(60, -40, -50, -50, **kw))
        osc = cpu_ref.Scoring.make(60, -40, -50, -50, **kw)
        for opt in (host.SW, host.NW):
            rows, idx = eng.align_device(opt, d_reads, d_refs)
            exp = cpu_ref.align(opt, reads, refs, osc, threads=8, affine=bool(kw))
            _assert_same((rows.cpu().numpy(), idx.cpu().numpy()), exp, (bool(kw), opt))
            assert np.array_equal(eng.score_device(opt, d_reads, d_refs).cpu().numpy(),
                                  cpu_ref.score(opt, reads, refs, osc, threads=8, affine=bool(kw)))
        eng.close()


def test_in_plugin_device_shards():
    """hip_devices = N splits every call into N contiguous shards, one engine (device modulo the visible
    ones, so a one-GPU box runs them all on device 0) and host thread each: same scores and alignments."""
    R, F, n = 150, 500, 10007
    reads, refs = _data(R, F, n, 99)
    exp_scores = cpu_ref.score(host.SW, reads, refs, threads=8)
    exp = cpu_ref.align(host.NW, reads, refs, threads=8)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=6, hip_devices=3) as hip:
        # on a node with several GPUs the shards sit on DISTINCT devices (this is where that is exercised on real
        # hardware); a one-GPU box folds them onto device 0
        visible = hipkernel.lib().valign_hip_device_count()
        want = ", ".join(str(d % visible) for d in range(3))
        assert "shards on devices [%s] of %d visible" % (want, visible) in hip.drain_log()
        assert np.array_equal(hip.score_alignments(host.SW, reads, refs), exp_scores)
        _assert_same(hip.compute_alignments(host.NW, reads, refs, normalise=False), exp, "3 shards")
        assert np.array_equal(hip.score_alignments(host.SW, reads[:2], refs[:2]), exp_scores[:2])     # fewer pairs than shards
    with pytest.raises(host.PluginError):
        host.Plugin(build.HIP_PLUGIN, R, F, hip_devices=0)


def test_flat_host_entry_point_for_alignments():
    """valign_hip_align_host: host pointers in, contiguous rows / idx out (no operator new[] blocks)."""
    R, F, n = 150, 500, 9000
    reads, refs = _data(R, F, n, 101)
    eng = hipkernel.Engine(R, F)
    for opt in (host.SW, host.NW):
        _assert_same(eng.align_host(opt, reads, refs, threads=4), cpu_ref.align(opt, reads, refs, threads=8), ("flat", opt))
    eng.close()


@pytest.mark.parametrize("aff,order", [((-10, -6, -10, -6), (host.SW, host.NW)), ((0, 0, 0, 0), (host.NW, host.SW)),
                                       ((-10, -6, -10, -6), (host.NW, host.SW))])
def test_pointer_scratch_follows_the_widest_stream_of_an_engine(aff, order):
    """One engine, same pair count, two modes whose fill kernels stream different byte counts per pair (tagged
    4-step blocks vs untagged 8-step blocks with two code words): the scratch sized by the first call must grow
    for the second (round-1 advisor finding: capacity was tracked in pairs, the second call wrote past it)."""
    import torch
    R, F, n = 150, 500, 4096
    reads, refs = _data(R, F, n, 103)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, *aff))
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, *aff)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    canary = torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device="cuda")      # a neighbour allocation to trample
    for opt in order + order:
        rows, idx = eng.align_device(opt, d_reads, d_refs)
        exp = cpu_ref.align(opt, reads, refs, osc, threads=8, affine=True)
        _assert_same((rows.cpu().numpy(), idx.cpu().numpy()), exp, (aff, opt))
    assert bool((canary == 0x5A).all())
    eng.close()


@pytest.mark.parametrize("R,F,n,seed", [(64, 128, 1000, 1), (12, 20, 333, 2), (33, 70, 257, 3), (100, 37, 129, 4), (16, 16, 65, 5),
                                         (1, 1, 5, 6), (128, 300, 77, 7), (150, 200, 50, 8), (250, 120, 31, 9),
                                         (150, 500, 1000, 10), (150, 500, 3, 11), (256, 700, 41, 12), (200, 1000, 16, 13)])     # 64 x 4: a wave per pair-of-pairs
@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_fused_small_batch_kernel(monkeypatch, R, F, n, seed, gaps):
    """Small compute_alignments calls run fill + traceback in ONE launch with the pointer stream in LDS
    (align_fill_tag_kernel<..., FUSED>): same alignments as the oracle and as the three-kernel path
    (VALIGN_HIP_DEBUG no_fused), odd pair counts and half-empty waves included."""
    reads, refs = _data(R, F, n, seed)
    sc = hipkernel.Scoring.make(2, -1, gaps[0], gaps[1])
    osc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    eng = hipkernel.Engine(R, F, sc)
    debug_switches(monkeypatch, no_fused=1)
    plain = hipkernel.Engine(R, F, sc)
    debug_switches(monkeypatch, no_fused=None)
    for opt in (host.SW, host.NW):
        exp = cpu_ref.align(opt, reads, refs, osc, threads=8)
        got = eng.align_host(opt, reads, refs, threads=2)
        assert eng.describe(opt, n)["direct_call"] == 2
        _assert_same(got, exp, ("fused", opt))
        _assert_same(plain.align_host(opt, reads, refs, threads=2), exp, ("three kernels", opt))
        assert plain.describe(opt, n)["direct_call"] == 1
    eng.close()
    plain.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cap_mb,overlap", [(0, True), (1024, True), (1024, False)])
def test_tracebacks_beside_the_next_fill(monkeypatch, cap_mb, overlap):
    """Large device batches run the walk of one part on a helper stream beside the fill of the next: one 7/8 + 1/8 cut
    when the pointer scratch holds the batch, the two halves of the scratch in turn when it does not (forced here by
    VALIGN_HIP_DEBUG scratch_cap_mb).  Rows and coordinates must repeat with the period of the repeated block and equal the
    oracle on it, with and without the helper stream."""
    import torch
    if cap_mb:
        debug_switches(monkeypatch, scratch_cap_mb=cap_mb)
    if not overlap:
        debug_switches(monkeypatch, no_overlap=1)
    R, F, blk, reps = 150, 500, 2039, 70          # 142 730 pairs (not a multiple of a block) = 1.07e10 cells: above the overlap threshold
    reads, refs = synth.make_pairs(blk, R, F, seed=77, indel_rate=0.01, junk_frac=0.02)
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3))
    d_reads = torch.from_numpy(reads).cuda().repeat(reps, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(reps, 1).contiguous()
    for opt in (0, 1):
        for _ in range(2):                      # the second call reuses scratch the first call's walks were reading
            rows, idx = eng.align_device(opt, d_reads, d_refs)
        torch.cuda.synchronize()
        rows = rows.view(reps, blk, 2, R + F)
        idx = idx.view(reps, blk, 4)
        assert bool((rows == rows[0:1]).all()) and bool((idx == idx[0:1]).all())
        erows, eidx = cpu_ref.align(opt, reads, refs, cpu_ref.Scoring.make(2, -1, -3, -3), threads=8)
        assert np.array_equal(idx[0].cpu().numpy(), eidx) and np.array_equal(rows[0].cpu().numpy(), erows)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R,F,match,prof_key", [(150, 500, 2, True), (150, 500, 3, True), (150, 500, 4, False), (300, 600, 2, False)])
def test_sw_end_cell_key_in_the_profile_or_computed(monkeypatch, R, F, match, prof_key):
    """Smith-Waterman alignments pick the row-major first maximum from one key per lane (value, then earlier row).
    Where 64x the cell range fits int16 (min(R, F) * match <= 497) the key rides in the query profile; beyond, and with
    VALIGN_HIP_DEBUG no_prof_key, it is computed per register.  Same rows and coordinates as the oracle from both, on a batch
    with many equal maxima (perfect repeats make the first-maximum rule matter)."""
    import torch
    n = 777
    reads, refs = synth.make_pairs(n, R, F, seed=31, indel_rate=0.01, n_run_frac=0.03, short_frac=0.05)
    refs[::3, F // 2:F // 2 + R] = reads[::3, :min(R, F - F // 2)][:, :R]            # a second copy of the read: tied maxima
    osc = cpu_ref.Scoring.make(match, -1, -3, -3)
    erows, eidx = cpu_ref.align(0, reads, refs, osc, threads=8)
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    for computed in (False, True):
        if computed:
            debug_switches(monkeypatch, no_prof_key=1)
        eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(match, -1, -3, -3))
        rows, idx = eng.align_device(0, d_reads, d_refs)
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), eidx) and np.array_equal(rows.cpu().numpy(), erows), (computed, prof_key)
        eng.close()


@pytest.mark.gpu
def test_host_pointer_alignments_of_a_large_batch():
    """compute_alignments' host-pointer pipeline with chunks large enough for the helper-stream schedule (a chunk of
    256 MB of result rows is 1.15e10 cells at 150 x 500): rows and coordinates repeat with the period of the repeated
    block and equal the oracle on it, in both modes."""
    R, F, blk, reps = 150, 500, 1999, 100                     # 199 900 pairs: one full chunk and a short one
    reads, refs = synth.make_pairs(blk, R, F, seed=79, indel_rate=0.01, junk_frac=0.02)
    big_reads, big_refs = np.tile(reads, (reps, 1)), np.tile(refs, (reps, 1))
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3))
    for opt in (0, 1):
        rows, idx = eng.align_host(opt, big_reads, big_refs, threads=8)
        rows = rows.reshape(reps, blk, 2, R + F)
        idx = idx.reshape(reps, blk, 4)
        assert (rows == rows[0:1]).all() and (idx == idx[0:1]).all()
        erows, eidx = cpu_ref.align(opt, reads, refs, cpu_ref.Scoring.make(2, -1, -3, -3), threads=8)
        assert np.array_equal(idx[0], eidx) and np.array_equal(rows[0], erows)
    eng.close()


@pytest.mark.parametrize("R,F", [(75, 230), (100, 300), (200, 640), (700, 900)])
def test_calls_that_need_a_fallback_kernel_are_replanned_onto_a_full_geometry(monkeypatch, R, F):
    """Round 4 compiles the equality-test / SSE-policy / asymmetric-affine fill kernels for six "full" geometries only
    (kernel_instances.hip.h).  These shapes plan onto geometries that carry the fast set alone (8x10, 16x8, 32x8, 64x12):
    a call that needs one of the other kernels must be re-planned (Engine::align_plan_for) and return the oracle's rows."""
    n = 257
    reads, refs = _data(R, F, n, 300 + R)
    eng = hipkernel.Engine(R, F)
    d = eng.describe(0, n)
    eng.close()
    assert (d["group_lanes"], d["rows_per_lane"]) in ((8, 10), (16, 8), (32, 8), (64, 12)), d      # a fast-only geometry
    lin = cpu_ref.Scoring.make()
    for opt in (0, 1):
        # SSE2 / AVX2 tie-breaks
        with host.Plugin(build.HIP_PLUGIN, R, F, traceback_policy=1) as hip:
            _assert_same(hip.compute_alignments(opt, reads, refs, normalise=False),
                         cpu_ref.align(opt, reads, refs, lin, threads=8, policy="sse"), (R, F, opt, "sse"))
        # affine scores that differ by direction
        aff = cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -4, -2)
        keys = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-4, score_gap_extend_ref=-2)
        with host.Plugin(build.HIP_PLUGIN, R, F, **keys) as hip:
            _assert_same(hip.compute_alignments(opt, reads, refs, normalise=False),
                         cpu_ref.align(opt, reads, refs, aff, threads=8, affine=True), (R, F, opt, "affine_asym"))
    # equality-test kernels (what a scoring beyond the tagged cells' headroom takes), forced by the debug switch
    debug_switches(monkeypatch, no_tag=1)
    with host.Plugin(build.HIP_PLUGIN, R, F) as hip:
        for opt in (0, 1):
            _assert_same(hip.compute_alignments(opt, reads, refs, normalise=False),
                         cpu_ref.align(opt, reads, refs, lin, threads=8), (R, F, opt, "no_tag"))
    # ... and a scoring that really leaves the headroom: match 60 -> 4 * 75 * 60 > 32000 / 4
    if R <= 100:
        big = cpu_ref.Scoring.make(60, -40, -50, -50)
        debug_switches(monkeypatch, no_tag=None)
        with host.Plugin(build.HIP_PLUGIN, R, F, score_match=60, score_mismatch=-40, score_gap_read=-50, score_gap_ref=-50) as hip:
            _assert_same(hip.compute_alignments(0, reads, refs, normalise=False),
                         cpu_ref.align(0, reads, refs, big, threads=8), (R, F, "big scores"))
