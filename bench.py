#!/usr/bin/env python3
"""bench.py -- GCUPS of the AlignmentKernel score path on MI355X (BASELINE.json metric).

One step = one pass of the hot path (Smith-Waterman affine-gap int16 score) over one
synthetic batch of 1,048,576 read/ref pairs of 150 x 500 bp per GPU (BASELINE.json
configs[1]), inputs already resident in HBM, launched through libHIPKernel.so's C ABI
(valign_hip_score_device).  With N > 1 (one rank per GPU, torch.distributed / RCCL) every
rank scores its own 1M-pair shard and the per-shard scores are all-gathered (configs[3]);
scaling is weak.  GCUPS = pairs * R * F / seconds / 1e9 over all ranks.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline ....... HBM roofline of the dominant kernel from ALGORITHMIC bytes
                   (R + F + 2 per pair) / average launch duration
                   measured with events on the launch stream.
  cpu_baseline ... the oracle (this repo's C port of the reference Default kernel, with the
                   same affine extension) timed on the host cores on a bounded sample.
  linear_gap ..... the same batch in the reference's own linear-gap model (the bit-exact
                   path), and the reference's compiled CPU kernels timed beside it when
                   oracle/_ref travelled with the repo: Default (OpenMP, all threads), AVX2
                   (one thread, as it runs on Linux) and AVX2 sharded over processes (all cores,
                   as it was meant to run).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from versalignlib_amd import build, hipkernel, shard

R, F = 150, 500
PAIRS_PER_GPU = 1 << 20
AFFINE = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
HBM_PEAK_GBPS = 8000.0
PMC_PROFILE = os.path.join(ROOT, "profiles", "r01_pmc_final.json")


def measured_traffic(kernel_key, pairs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/pmc_passes.sh: FETCH_SIZE and WRITE_SIZE in separate runs, in KB; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).  None when the profile is
    absent or was taken at another batch size."""
    try:
        with open(PMC_PROFILE) as f:
            prof = json.load(f)
        k = prof["kernels"][kernel_key]
        if int(prof["pairs"]) != int(pairs):
            return None
        return float(k["FETCH_SIZE"]) * 1024.0 * 2.0 + float(k["WRITE_SIZE"]) * 1024.0
    except (OSError, KeyError, ValueError, TypeError):
        return None


def measured_valu_issue(kernel_key, pairs):
    """The resource that does bind this kernel: share of all SIMD cycles spent issuing VALU
    instructions, from the same committed PMC passes.  Packed-int16 ops issue one wave64
    instruction per 4 cycles per SIMD (profiles/r01_valu_rate_microbench.txt); 256 CUs x 4 SIMDs;
    GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    try:
        with open(PMC_PROFILE) as f:
            prof = json.load(f)
        k = prof["kernels"][kernel_key]
        if int(prof["pairs"]) != int(pairs):
            return None
        issue = float(k["SQ_INSTS_VALU"]) * 4.0 / 1024.0
        cycles = float(k["GRBM_GUI_ACTIVE"]) / 8.0
        return {"bound": "valu-issue", "wave_instructions": float(k["SQ_INSTS_VALU"]),
                "issue_cycles_per_simd": round(issue), "kernel_cycles": round(cycles),
                "frac": round(issue / cycles, 4), "source": "profiles/r01_pmc_final.json"}
    except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def synth_on_device(n, device, seed):
    """SURVEY.md 8(d) batch, generated on the GPU: uniform ACGT refs; reads = ref window
    with 15% substitutions; 1% of pairs carry an N run, 1% are short and NUL padded."""
    g = torch.Generator(device=device).manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    refs = lut[torch.randint(0, 4, (n, F), device=device, generator=g)]
    off = torch.randint(0, F - R + 1, (n, 1), device=device, generator=g)
    col = torch.arange(R, device=device)[None, :]
    reads = torch.gather(refs, 1, off + col)
    sub = torch.rand((n, R), device=device, generator=g) < 0.15
    reads = torch.where(sub, lut[torch.randint(0, 4, (n, R), device=device, generator=g)], reads)
    pick = torch.rand((n, 1), device=device, generator=g)
    start = torch.randint(0, R - 5, (n, 1), device=device, generator=g)
    run = torch.randint(1, 6, (n, 1), device=device, generator=g)
    n_run = (pick < 0.01) & (col >= start) & (col < start + run)
    reads = torch.where(n_run, torch.full_like(reads, ord("N")), reads)
    keep_r = torch.randint(0, R + 1, (n, 1), device=device, generator=g)
    keep_f = torch.randint(0, F + 1, (n, 1), device=device, generator=g)
    short = (pick >= 0.01) & (pick < 0.02)
    reads = torch.where(short & (col >= keep_r), torch.zeros_like(reads), reads)
    fcol = torch.arange(F, device=device)[None, :]
    refs = torch.where(short & (fcol >= keep_f), torch.zeros_like(refs), refs)
    return reads.contiguous(), refs.contiguous()


def timed_steps(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def kernel_launch_ms(eng, opt, reads, refs, out, reps):
    """Average duration of one kernel launch, events on the stream the kernel runs on."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        eng.score_device(opt, reads, refs, out)
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / reps


def cpu_baseline(reads, refs, affine):
    """Oracle on the host cores over a bounded sample of the same batch (rank 0, N == 1)."""
    from oracle import cpu_ref
    cpu_ref.build()
    cores = min(cpu_ref.max_threads(), os.cpu_count() or 1)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, **(AFFINE if affine else {}))
    probe = 512 * cores
    h_reads, h_refs = reads[:probe].cpu().numpy(), refs[:probe].cpu().numpy()
    t0 = time.perf_counter()
    cpu_ref.score(0, h_reads, h_refs, sc, threads=cores, affine=affine)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    sample = int(min(reads.shape[0], max(probe, rate * 12.0)))          # ~12 s of CPU work
    h_reads, h_refs = reads[:sample].cpu().numpy(), refs[:sample].cpu().numpy()
    t0 = time.perf_counter()
    cpu_ref.score(0, h_reads, h_refs, sc, threads=cores, affine=affine)
    sec = time.perf_counter() - t0
    return {"value": round(sample * R * F / sec / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "first %d pairs of the rank-0 batch, SW %s int16, oracle/cpu_ref.c OpenMP over pairs, %.1f s"
                      % (sample, "affine-gap" if affine else "linear-gap", sec)}


def reference_cpu_kernels(reads, refs):
    """The reference's own compiled kernels (oracle/_ref), linear gap, through the plugin ABI."""
    from versalignlib_amd import host
    out = {}
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    for name, threads, sample in (("Default", cores, 4096 * max(1, cores // 4)), ("AVX", 1, 8192)):
        path = os.path.join(ref_dir, "lib%sKernel.so" % name)
        if not os.path.exists(path):
            continue
        try:
            h_reads, h_refs = reads[:sample].cpu().numpy(), refs[:sample].cpu().numpy()
            with host.Plugin(path, R, F, num_threads=threads) as p:
                t0 = time.perf_counter()
                p.score_alignments(0, h_reads, h_refs)
                sec = time.perf_counter() - t0
            out[name] = {"value": round(sample * R * F / sec / 1e9, 3), "unit": "GCUPS", "threads": threads,
                         "kind": "reference", "sample": "%d pairs, SW linear-gap" % sample}
        except Exception as e:      # a reference kernel that cannot run here is reported, not fatal
            out[name] = {"error": str(e)[:200]}
    try:
        out["AVX_sharded"] = reference_avx_sharded(reads, refs, min(cores, 32))
    except Exception as e:
        out["AVX_sharded"] = {"error": str(e)[:200]}
    return out


def reference_avx_sharded(reads, refs, procs, pairs_per_proc=32768):
    """What the reference's AVX2 kernel was meant to do -- use every core (its OpenMP loop is compiled
    out on Linux, AVXKernel.cpp:74-76): the sample is sharded over `procs` child processes, each
    running the compiled libAVXKernel.so on one thread through the plugin protocol
    (tools/ref_shard_worker.py: ctypes only, no torch, no GPU).  Children start together; the
    slowest one's time counts."""
    import subprocess
    import tempfile
    import numpy as np
    from versalignlib_amd import build as b
    plugin = os.path.join(ROOT, "oracle", "_ref", "libAVXKernel.so")
    if not os.path.exists(plugin):
        raise RuntimeError("oracle/_ref/libAVXKernel.so not built")
    sample = min(int(reads.shape[0]), procs * pairs_per_proc)
    per = sample // procs
    sample = per * procs
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        np.save(os.path.join(d, "reads.npy"), reads[:sample].cpu().numpy())
        np.save(os.path.join(d, "refs.npy"), refs[:sample].cpu().numpy())
        worker = os.path.join(ROOT, "tools", "ref_shard_worker.py")
        kids = [subprocess.Popen([sys.executable, worker, b.HOST_LIB, plugin, os.path.join(d, "reads.npy"),
                                  os.path.join(d, "refs.npy"), str(i * per), str((i + 1) * per), d, str(i)],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(procs)]
        deadline = time.time() + 120
        while sum(os.path.exists(os.path.join(d, "ready.%d" % i)) for i in range(procs)) < procs:
            if time.time() > deadline or any(k.poll() not in (None, 0) for k in kids):
                for k in kids:
                    k.kill()
                raise RuntimeError("reference shard workers did not come up")
            time.sleep(0.01)
        open(os.path.join(d, "go"), "w").close()
        secs = []
        for k in kids:
            out, err = k.communicate(timeout=300)
            if k.returncode != 0:
                raise RuntimeError("shard worker failed: " + err[-200:])
            secs.append(float(out.split()[0]))
    slowest = max(secs)
    return {"value": round(sample * R * F / slowest / 1e9, 3), "unit": "GCUPS", "processes": procs, "threads_each": 1,
            "kind": "reference", "sample": "%d pairs over %d processes, SW linear-gap, slowest shard %.2f s"
                                           % (sample, procs, slowest)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per GPU (default: config size)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one rank per GPU: launch with python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libHIPKernel.so has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    if not os.path.exists(build.HIP_PLUGIN):
        build.build_hip()
    n = args.pairs
    reads, refs = synth_on_device(n, device, seed=2000 + rank)
    affine_sc = hipkernel.Scoring.make(2, -1, -3, -3, **AFFINE)
    eng = hipkernel.Engine(R, F, affine_sc, device=local_rank)
    eng_lin = hipkernel.Engine(R, F, hipkernel.Scoring.make(), device=local_rank)
    local = torch.empty(n, dtype=torch.int16, device=device)
    gathered = [None]

    def step():
        eng.score_device(0, reads, refs, local)
        gathered[0] = shard.all_gather_scores(local, n_total=n * world) if world > 1 else local

    elapsed = timed_steps(step, args.steps, args.warmup, world)
    total_cells = float(n) * R * F * world * args.steps
    value = total_cells / elapsed / 1e9

    if rank == 0:
        assert gathered[0].numel() == n * world
        k_ms = kernel_launch_ms(eng, 0, reads, refs, local, max(3, min(args.steps, 10)))
        alg_bytes = float(n) * (R + F + 2)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        d = eng.describe(0, n)
        cells = d.get("score_cells", "int16")
        pmc_key = "score_kernel<%d, %d, 0, %d>" % (d["group_lanes"], d["rows_per_lane"], 4 if cells == "f16" else 3)
        cell_check = None
        if cells == "f16":
            # the half-float kernel must reproduce the int16 kernel bit for bit: check it on the whole
            # batch (outside the timed region) and time the int16 variant beside it
            os.environ["VALIGN_HIP_NO_F16"] = "1"
            eng_i16 = hipkernel.Engine(R, F, affine_sc, device=local_rank)
            del os.environ["VALIGN_HIP_NO_F16"]
            ref_scores = eng_i16.score_device(0, reads, refs)
            torch.cuda.synchronize()
            i16_ms = kernel_launch_ms(eng_i16, 0, reads, refs, ref_scores.clone(), 5)
            eng.score_device(0, reads, refs, local)
            torch.cuda.synchronize()
            cell_check = {"identical_to_int16_kernel": bool(torch.equal(local, ref_scores)),
                          "int16_kernel_ms": round(i16_ms, 4),
                          "int16_kernel_gcups": round(n * R * F / (i16_ms * 1e-3) / 1e9, 1),
                          "note": "cells are integers of magnitude <= 2048: exact in fp16"}
            eng_i16.close()
        line = {
            "metric": "GCUPS (giga DP cell updates/sec) SW affine-gap, 150x500 bp",
            "value": round(value, 1), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": cells, "data": "synthetic",
            "config": {"workload": "%d pairs/GPU, 150 bp x 500 bp, SW affine-gap int16 score (open -5, extend -1), "
                                   "inputs resident in HBM%s" % (n, ", RCCL all-gather of scores" if world > 1 else ""),
                       "pairs_per_gpu": n, "read_length": R, "ref_length": F, "kernel_geometry": "%dx%d" % (d["group_lanes"], d["rows_per_lane"])},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": measured_traffic(pmc_key, n),
                         "kernel": "score_kernel<%d,%d,SW,%s>" % (d["group_lanes"], d["rows_per_lane"],
                                                                  "affine-f16" if d.get("score_cells") == "f16" else "affine"),
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes": alg_bytes,
                         "kernel_gcups": round(n * R * F / (k_ms * 1e-3) / 1e9, 1),
                         "note": "integer VALU bound: %.4f B/cell algorithmic, HBM is idle by design" % ((R + F + 2) / (R * F)),
                         "valu": measured_valu_issue(pmc_key, n)},
        }
        if cell_check is not None:
            line["cell_check"] = cell_check
        lin_ms = kernel_launch_ms(eng_lin, 0, reads, refs, local, 5)
        line["linear_gap"] = {"kernel_ms": round(lin_ms, 4), "kernel_gcups": round(n * R * F / (lin_ms * 1e-3) / 1e9, 1),
                              "note": "reference's own gap model (bit-exact path), same batch"}
        # BASELINE config 3 (NW + traceback) on the same batch, device-resident: extra evidence,
        # never part of `value`
        try:
            AL = R + F
            rows = torch.empty((n, 2, AL), dtype=torch.uint8, device=device)
            idx = torch.empty((n, 4), dtype=torch.int16, device=device)
            line["alignments"] = {}
            for name, e in (("nw_affine_traceback", eng), ("nw_linear_traceback", eng_lin)):
                e.align_device(1, reads, refs, rows, idx)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e.align_device(1, reads, refs, rows, idx)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                line["alignments"][name] = {"ms": round(ms, 3), "gcups": round(n * R * F / (ms * 1e-3) / 1e9, 1),
                                            "algorithmic_GBps": round(n * (3 * AL + 8) / (ms * 1e-3) / 1e9, 1)}
            del rows, idx
        except hipkernel.HipKernelError as e:
            line["alignments"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(reads, refs, affine=True)
            line["linear_gap"]["cpu_port"] = cpu_baseline(reads, refs, affine=False)
            line["linear_gap"]["cpu_reference"] = reference_cpu_kernels(reads, refs)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
