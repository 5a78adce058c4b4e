"""BASELINE.json's configurations AS STATED, on the GPU -- the ones the round-2 verdict found unexercised:

  configs[0] ... 1k pairs, 64 x 128, NW linear-gap with int32 cells, against the Default kernel (oracle + live)
  configs[4] ... one GPU's share of it at FULL size: 32,768 pairs of 10 kbp x 10 kbp, band 512, int32 cells
  configs[3] ... sharded over several GPUs: runs whenever the box has more than one device (skips on one)

Size-independent properties where the oracle cannot cover the batch in seconds: the batch is a tiling of one
block, scores must repeat with the block's period (checked on the device) and the first block equals the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, hipkernel, host, synth

from conftest import ROOT, band_constants, ref_kernel

pytestmark = pytest.mark.gpu


def _devices():
    return hipkernel.lib().valign_hip_device_count()


@pytest.mark.parametrize("gaps", [(-3, -3), (-2, -4)])
def test_config1_nw_linear_int32_cells(gaps):
    """configs[0]: 1,000 pairs of 64 x 128, NW variant, linear gaps, int32 cells (score_width = 32) -- against the
    oracle, against the int16 cells the engine picks by itself, and live against the reference's compiled kernels
    (Default: low byte, DefaultKernel.cpp:199; SSE: the full short)."""
    R, F, n = 64, 128, 1000
    # (no bytes >= 0x80 in this batch: the Default kernel indexes its class table with a signed char there --
    # undefined, DefaultKernel.h:43-60 -- and is compared live below)
    reads, refs = synth.make_pairs(n, R, F, seed=101, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08, lowercase_frac=0.05)
    sc = cpu_ref.Scoring.make(2, -1, gaps[0], gaps[1])
    keys = dict(score_gap_read=gaps[0], score_gap_ref=gaps[1])
    got = {}
    with host.Plugin(build.HIP_PLUGIN, R, F, score_width=32, num_threads=4, **keys) as hip:
        for opt in (host.NW, host.SW):
            got[opt] = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got[opt], cpu_ref.score(opt, reads, refs, sc, threads=8)), opt
            assert np.array_equal(got[opt], cpu_ref.score(opt, reads, refs, sc, threads=8, wide=True)), opt
        assert '"score_cells": "int32"' in hip.drain_log()
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, **keys) as hip:
        for opt in (host.NW, host.SW):
            assert np.array_equal(hip.score_alignments(opt, reads, refs), got[opt])
        assert '"score_cells": "int32"' not in hip.drain_log()
    sse, default = ref_kernel("SSE"), ref_kernel("Default")
    if sse and default:
        with host.Plugin(sse, R, F, **keys) as s, host.Plugin(default, R, F, num_threads=4, **keys) as d:
            for opt in (host.NW, host.SW):
                assert np.array_equal(got[opt], s.score_alignments(opt, reads, refs)), opt
                assert np.array_equal(got[opt] & 0xFF, d.score_alignments(opt, reads, refs) & 0xFF), opt


def test_config1_int32_device_entry_point():
    """The same configuration through the flat device-resident entry point."""
    import torch
    R, F, n = 64, 128, 1000
    reads, refs = synth.make_pairs(n, R, F, seed=102, indel_rate=0.02, n_run_frac=0.05, short_frac=0.08)
    eng = hipkernel.Engine(R, F)
    eng.set_score_width(32)
    assert eng.describe(1, n)["score_cells"] == "int32"
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    for opt in (1, 0):
        assert np.array_equal(eng.score_device(opt, d_reads, d_refs).cpu().numpy(), cpu_ref.score(opt, reads, refs, threads=8))
    eng.close()


def test_config5_full_size_per_gpu():
    """configs[4], one GPU's share at full size: 32,768 pairs of 10 kbp x 10 kbp, banded Smith-Waterman (512
    diagonals), int32 cells -- the batch bench.py --workload long times.  1,024 copies of a 32-pair block: scores
    repeat with period 32 and the first block equals the oracle's statement of the documented band."""
    import torch
    R = F = 10000
    blk, reps = 32, 1024
    reads, refs = synth.make_pairs(blk, R, F, seed=59, sub_rate=0.1, indel_rate=0.002, n_run_frac=0.2, short_frac=0.1)
    block_rows, col_align = band_constants(R, F, 512)
    exp = cpu_ref.score_banded_sw(reads, refs, 512, threads=8, block_rows=block_rows, col_align=col_align)
    eng = hipkernel.Engine(R, F)
    eng.set_band_width(512)
    eng.set_score_width(32)
    assert eng.describe(0, blk * reps)["score_cells"] == "int32"
    d_reads = torch.from_numpy(reads).cuda().repeat(reps, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(reps, 1).contiguous()
    assert d_reads.shape == (32768, R)
    got = eng.score_device(0, d_reads, d_refs).view(reps, blk)
    assert bool((got == got[0:1]).all())
    assert np.array_equal(got[0].cpu().numpy(), exp)
    # the band really cuts: some pairs (those with indels drifting out of it) score less than unbanded
    assert (exp <= cpu_ref.score(0, reads, refs, threads=8)).all()
    eng.close()
    del d_reads, d_refs, got
    torch.cuda.empty_cache()


def test_hip_devices_beyond_the_visible_ones_is_announced_or_refused():
    """hip_devices = N with fewer than N devices visible: a WARNING line (shards folded, same results) -- or, with
    hip_devices_strict = 1, a refusal.  Runs on any box: asks for one device more than there are."""
    R, F, n = 64, 128, 301
    reads, refs = synth.make_pairs(n, R, F, seed=103)
    want = _devices() + 1
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, hip_devices=want) as hip:
        log = hip.drain_log()
        assert any(ln.startswith("WARNING") and "folded onto the visible devices" in ln for ln in log.splitlines()), log
        assert np.array_equal(hip.score_alignments(host.SW, reads, refs), cpu_ref.score(host.SW, reads, refs, threads=4))
    with pytest.raises(host.PluginError, match="hip_devices_strict"):
        host.Plugin(build.HIP_PLUGIN, R, F, hip_devices=want, hip_devices_strict=1)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, hip_devices=_devices(), hip_devices_strict=1) as hip:
        assert not any(ln.startswith("WARNING") and "folded" in ln for ln in hip.drain_log().splitlines())


def test_pointer_scratch_cap_key():
    """pointer_scratch_cap_mb bounds the device-side pointer scratch: the batch runs in chunks, same alignments."""
    R, F, n = 150, 500, 6000
    reads, refs = synth.make_pairs(n, R, F, seed=104, indel_rate=0.02)
    exp = cpu_ref.align(host.NW, reads, refs, threads=8)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4, pointer_scratch_cap_mb=16) as hip:      # 16 MiB: ~750 pairs per chunk
        rows, idx = hip.compute_alignments(host.NW, reads, refs, normalise=False)
    assert np.array_equal(idx, exp[1]) and np.array_equal(rows, exp[0])
    eng = hipkernel.Engine(R, F)
    eng.set_pointer_scratch_cap_mb(16)
    rows, idx = eng.align_host(host.SW, reads, refs, threads=4)
    exp = cpu_ref.align(host.SW, reads, refs, threads=8)
    assert np.array_equal(idx, exp[1]) and np.array_equal(rows, exp[0])
    with pytest.raises(hipkernel.HipKernelError):
        eng.set_pointer_scratch_cap_mb(-1)
    eng.close()


# ---- more than one device: these run by themselves wherever the box has them (configs[3]) ----

def _need_two_devices():
    if _devices() < 2:
        pytest.skip("one GPU visible: the multi-device legs need two")


def test_hip_devices_on_distinct_real_devices():
    """hip_devices = 2 on two REAL devices: the log names two different ordinals and both shards' results land in
    the caller's arrays, scores and alignments, identical to the oracle."""
    _need_two_devices()
    R, F, n = 150, 500, 20011
    reads, refs = synth.make_pairs(n, R, F, seed=105, indel_rate=0.01, n_run_frac=0.02, short_frac=0.02)
    aff = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=8, hip_devices=2, hip_devices_strict=1, **aff) as hip:
        assert "shards on devices [0, 1]" in hip.drain_log()
        assert np.array_equal(hip.score_alignments(host.SW, reads, refs), cpu_ref.score(host.SW, reads, refs, osc, threads=8, affine=True))
        rows, idx = hip.compute_alignments(host.NW, reads, refs, normalise=False)
        exp_rows, exp_idx = cpu_ref.align(host.NW, reads, refs, osc, threads=8, affine=True)
        assert np.array_equal(idx, exp_idx) and np.array_equal(rows, exp_rows)


def test_bench_on_two_real_gpus():
    """`python bench.py --gpus 2` as the driver runs it, on two real devices: two ranks seen after the RCCL
    all-gather, gathered checksum = sum of the ranks' checksums, timed scores verified against the oracle."""
    _need_two_devices()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--pairs", "131072", "--no-abi"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["multi_gpu"]["n_ranks_seen"] == 2 and line["multi_gpu"]["checksum_ok"]
    assert line["multi_gpu"]["gathered_pairs"] == 2 * 131072
    assert line["verified"]["mismatches"] == 0 and line["verified"]["pairs"] > 0


_RCCL_REHEARSAL = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VALIGN_ROOT"])
from versalignlib_amd import shard
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)                 # as bench.py: RCCL, one rank per GPU
gloo = dist.new_group(backend="gloo")
local = (torch.arange(100003, device=dev) % 30011 - 15000).to(torch.int16)
send = local.contiguous().view(torch.uint8)                     # scores travel as bytes: RCCL has no int16
recv = torch.empty(send.numel() * dist.get_world_size(), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(recv, send)
assert torch.equal(recv.view(torch.int16), local)
assert torch.equal(shard.all_gather_scores(local, n_total=local.numel()), local)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
s = local.to(torch.int64).sum().reshape(1)
dist.all_reduce(s)
assert float(t.item()) == 1.25 and int(s.item()) == int(local.to(torch.int64).sum().item())
dist.barrier()
torch.cuda.synchronize()
dist.barrier(group=gloo)
dist.destroy_process_group()
print("RCCL_REHEARSAL_OK")
"""


def test_torch_distributed_rccl_calls_of_the_bench_on_one_rank():
    """bench.py's N > 1 path has never met a second GPU (the driver's 8-GPU node is the first): what CAN run on one device is
    every torch.distributed call it makes -- process group on the `nccl` backend (= RCCL) bound to the device, a gloo group
    beside it, the byte-view all-gather of int16 scores, the MAX / SUM all-reduces, the barriers -- with a world of one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", VALIGN_ROOT=ROOT)
    proc = subprocess.run([sys.executable, "-c", _RCCL_REHEARSAL], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=600)
    assert proc.returncode == 0 and "RCCL_REHEARSAL_OK" in proc.stdout, (proc.stdout[-1000:], proc.stderr[-3000:])


def test_in_plugin_rccl_all_gather_of_shard_scores():
    """hip_devices_allgather = 1: every shard's scores stay on its device, an RCCL all-gather (librccl.so loaded by the
    plugin, one communicator per device in this one process) assembles the vector on every device and the host copy
    comes from the first.  With one device the communicator has one rank (the collective is then a copy, but the whole
    path -- dlopen, ncclCommInitAll, group call, device-resident shards, host copy -- runs); with two or more it is the
    north star's exchange step on real links.  Duplicate devices are refused: RCCL cannot have two ranks on one GPU."""
    R, F, n = 150, 500, 20011
    reads, refs = synth.make_pairs(n, R, F, seed=106, n_run_frac=0.02, short_frac=0.02)
    aff = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
    osc = cpu_ref.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
    devices = min(_devices(), 4)
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=8, hip_devices=devices, hip_devices_strict=1, hip_devices_allgather=1, **aff) as hip:
        assert "RCCL all-gather of the per-shard scores over %d device(s)" % devices in hip.drain_log()
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            assert np.array_equal(got, cpu_ref.score(opt, reads, refs, osc, threads=8, affine=True)), opt
        assert np.array_equal(hip.score_alignments(host.SW, reads[:5], refs[:5]), cpu_ref.score(host.SW, reads[:5], refs[:5], osc, threads=4, affine=True))
        assert "RCCL all-gather of %d shard(s)" % devices in hip.drain_log()
        rows, idx = hip.compute_alignments(host.NW, reads[:3000], refs[:3000], normalise=False)        # (alignments: no collective)
        exp_rows, exp_idx = cpu_ref.align(host.NW, reads[:3000], refs[:3000], osc, threads=8, affine=True)
        assert np.array_equal(idx, exp_idx) and np.array_equal(rows, exp_rows)
    with pytest.raises(host.PluginError, match="distinct device"):
        host.Plugin(build.HIP_PLUGIN, R, F, hip_devices=_devices() + 1, hip_devices_allgather=1)
