#!/usr/bin/env python3
"""bench.py -- GCUPS of the AlignmentKernel score path on MI355X (BASELINE.json metric).

One step = one pass of the hot path (Smith-Waterman affine-gap int16 score) over one
synthetic batch of 1,048,576 read/ref pairs of 150 x 500 bp per GPU (BASELINE.json
configs[1]), inputs already resident in HBM, launched through libHIPKernel.so's C ABI
(valign_hip_score_device).  With N > 1 (one rank per GPU, torch.distributed / RCCL) every
rank scores its own 1M-pair shard and the per-shard scores are all-gathered (configs[3]);
scaling is weak.  GCUPS = pairs * R * F / seconds / 1e9 over all ranks.

`python bench.py --gpus N` from a plain shell starts its own ranks: the parent -- before it
touches the GPU in any way -- runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child process, relays rank 0's JSON line and exits with the child's code.  Launched by
torch.distributed.run directly (RANK / WORLD_SIZE in the environment) it is simply one rank.

`--workload long` is BASELINE.json configs[4] per GPU: 32,768 pairs of 10 kbp x 10 kbp, banded
Smith-Waterman (512 diagonals), int32 cells.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline ....... HBM roofline of the dominant kernel from ALGORITHMIC bytes
                   (R + F + 2 per pair) / average launch duration
                   measured with events on the launch stream.
  cpu_baseline ... the oracle (this repo's C port of the reference Default kernel, with the
                   same affine extension) timed on the host cores on a bounded sample.
  half_float ..... the same batch through the half-float-cell kernel (identical scores, checked on
                   the whole batch) -- the fast path the plugin takes by itself; `value` is the
                   int16-cell kernel BASELINE.json names.
  multi_gpu ...... N > 1: ranks seen after the RCCL all-gather, checksum of the gathered vector
                   against the sum of the ranks' own checksums, and the same K steps without
                   the all-gather.
  abi ............ PCIe-inclusive figures THROUGH the plugin ABI (spawn_alignment_kernel, scattered
                   host pointers): score_alignments and compute_alignments(SW) on the full batch,
                   and the reference host's own protocol (100 back-to-back calls, src/impl/main.cpp
                   :268-292) on BASELINE configs[0].  Never part of `value`.
  verified ....... the scores the LAST TIMED STEP wrote, compared with the oracle inside this run (the sample
                   cpu_baseline scored: hundreds of thousands of pairs at N = 1); a mismatch exits non-zero.
                   The alignment legs carry their own `verified` (first 2048 pairs, rows + coordinates).
  linear_gap ..... the same batch in the reference's own linear-gap model (the bit-exact
                   path), and the reference's compiled CPU kernels timed beside it when
                   oracle/_ref travelled with the repo.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R, F = 150, 500
PAIRS_PER_GPU = 1 << 20
LONG_R, LONG_F, LONG_PAIRS_PER_GPU, LONG_BAND = 10000, 10000, 32768, 512
AFFINE = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
HBM_PEAK_GBPS = 8000.0
PMC_PROFILES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_final.json")) +
                      glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_long_c5.json")))      # (--workload long's kernel)


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` from a plain shell (no GPU call may precede this)
# ---------------------------------------------------------------------------------------------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Start one rank per GPU as CHILD processes (never an exec: a process that has initialised the GPU
    must not be replaced, and this parent has not touched it), relay rank 0's JSON line, return the
    children's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["VALIGN_BENCH_LAUNCHED"] = "1"
    port = int(env.get("MASTER_PORT", 0)) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, cwd=ROOT)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)        # anything else a rank printed
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------

def source_hash():
    """sha256 over the kernel sources: PMC profiles are only quoted for the sources they were taken from."""
    h = hashlib.sha256()
    # the device code: every *kernels*.hip.h header and the instantiation lists -- not the engine / plugin (host code that
    # launches them) and not the host-only .cpp files
    for path in sorted(glob.glob(os.path.join(ROOT, "versalignlib_amd", "csrc", "*kernel*.hip*"))):
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(kernel_key, pairs):
    """Counters of `kernel_key` from the newest committed PMC profile that was taken from EXACTLY these
    kernel sources at this batch size (tools/pmc_passes.sh + tools/pmc_summary.py stamp `csrc_sha16`);
    None otherwise -- stale counters are not quoted."""
    want = source_hash()
    for path in reversed(PMC_PROFILES):
        try:
            with open(path) as f:
                prof = json.load(f)
            if prof.get("csrc_sha16") != want or int(prof["pairs"]) != int(pairs):
                continue
            kernels = prof["kernels"]
            if kernel_key not in kernels:           # (a trailing template argument chosen at run time: first match by prefix)
                kernel_key = next(k for k in kernels if k.startswith(kernel_key.rstrip(">")))
            return kernels[kernel_key], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, TypeError, StopIteration):
            continue
    return None, None


def measured_traffic(kernel_key, pairs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for wide coalesced reads on gfx950)."""
    k, _ = pmc_profile(kernel_key, pairs)
    if k is None:
        return None
    try:
        return float(k["FETCH_SIZE"]) * 1024.0 * 2.0 + float(k["WRITE_SIZE"]) * 1024.0
    except (KeyError, ValueError, TypeError):
        return None


def measured_valu_issue(kernel_key, pairs):
    """The resource that does bind this kernel: share of all SIMD cycles spent issuing VALU
    instructions.  The packed 16-bit ops issue one wave64 instruction per 4 cycles per SIMD
    (profiles/r02_valu_rate_microbench.txt); 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    k, path = pmc_profile(kernel_key, pairs)
    if k is None:
        return None
    try:
        issue = float(k["SQ_INSTS_VALU"]) * 4.0 / 1024.0
        cycles = float(k["GRBM_GUI_ACTIVE"]) / 8.0
        return {"bound": "valu-issue", "wave_instructions": float(k["SQ_INSTS_VALU"]),
                "issue_cycles_per_simd": round(issue), "kernel_cycles": round(cycles),
                "frac": round(issue / cycles, 4), "source": "committed profile " + path}
    except (KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def measured_lds_conflicts(kernel_key, pairs):
    """LDS bank-conflict cycles as a share of the cycles the LDS was busy (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
    the north star's second counter), from the same committed profile."""
    k, _ = pmc_profile(kernel_key, pairs)
    if k is None:
        return None
    try:
        return round(float(k["SQ_LDS_BANK_CONFLICT"]) / float(k["SQ_LDS_IDX_ACTIVE"]), 4)
    except (KeyError, ValueError, TypeError, ZeroDivisionError):
        return None


def synth_on_device(n, device, seed, R=R, F=F):
    """SURVEY.md 8(d) batch, generated on the GPU: uniform ACGT refs; reads = ref window
    with 15% substitutions; 1% of pairs carry an N run, 1% are short and NUL padded."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    reads = torch.empty((n, R), dtype=torch.uint8, device=device)
    refs = torch.empty((n, F), dtype=torch.uint8, device=device)
    step = max(1, min(n, (64 << 20) // max(R, F)))          # bounded temporaries for the long shapes
    col = torch.arange(R, device=device)[None, :]
    fcol = torch.arange(F, device=device)[None, :]
    for lo in range(0, n, step):
        m = min(step, n - lo)
        rf = lut[torch.randint(0, 4, (m, F), device=device, generator=g)]
        off = torch.randint(0, F - R + 1, (m, 1), device=device, generator=g)
        rd = torch.gather(rf, 1, off + col)
        sub = torch.rand((m, R), device=device, generator=g) < 0.15
        rd = torch.where(sub, lut[torch.randint(0, 4, (m, R), device=device, generator=g)], rd)
        pick = torch.rand((m, 1), device=device, generator=g)
        start = torch.randint(0, R - 5, (m, 1), device=device, generator=g)
        run = torch.randint(1, 6, (m, 1), device=device, generator=g)
        n_run = (pick < 0.01) & (col >= start) & (col < start + run)
        rd = torch.where(n_run, torch.full_like(rd, ord("N")), rd)
        keep_r = torch.randint(0, R + 1, (m, 1), device=device, generator=g)
        keep_f = torch.randint(0, F + 1, (m, 1), device=device, generator=g)
        short = (pick >= 0.01) & (pick < 0.02)
        reads[lo:lo + m] = torch.where(short & (col >= keep_r), torch.zeros_like(rd), rd)
        refs[lo:lo + m] = torch.where(short & (fcol >= keep_f), torch.zeros_like(rf), rf)
    return reads, refs


def timed_steps(fn, steps, warmup, world):
    import torch
    import torch.distributed as dist
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
    import torch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        eng.score_device(opt, reads, refs, out)
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / reps


def host_cpus():
    """What this process may really use: the CPUs of its affinity mask, capped by the cgroup's CFS quota
    (a container on a 256-thread host with `cpu.max = 1600000 100000` gets 16 CPUs' worth of time however many
    threads it starts -- more threads than that are throttled, not faster)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    quota, quota_text = None, "none"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2
            q, period = f.read().split()[:2]
        quota_text = "%s %s" % (q, period)
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                                            # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            quota_text = "%d %d" % (q, period)
            if q > 0:
                quota = q / float(period)
        except (OSError, ValueError):
            pass
    effective = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"effective": effective, "logical_cpus": logical, "affinity": affinity, "cpu_quota": quota_text}


def host_cores():
    return host_cpus()["effective"]


BAND_BLOCK = [160, 4]       # (block rows, column alignment) of the band the engine computes: set from its describe()


def cpu_simd_baseline(reads, refs, affine, cores, seconds):
    """The AVX2 inter-sequence sweep of oracle/cpu_simd.c (16 pairs per vector, OpenMP over batches: what the reference's
    AVX2 kernel does for the linear model on ONE thread, here for both gap models on all the CPUs this process has),
    on a bounded sample of the same batch -- checked against the scalar oracle on its first pairs."""
    import numpy as np
    from oracle import cpu_ref
    if cpu_ref.simd_lib() is None:
        return {"error": "no AVX2 on this host"}
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, **(AFFINE if affine else {}))
    probe = min(int(reads.shape[0]), 4096 * cores)
    h_reads, h_refs = reads[:probe].cpu().numpy(), refs[:probe].cpu().numpy()
    t0 = time.perf_counter()
    got = cpu_ref.score_simd(0, h_reads, h_refs, sc, threads=cores, affine=affine)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    check = min(probe, 2048)
    same = bool(np.array_equal(got[:check], cpu_ref.score(0, h_reads[:check], h_refs[:check], sc, threads=cores, affine=affine)))
    sample = int(min(reads.shape[0], max(probe, rate * seconds)))
    h_reads, h_refs = reads[:sample].cpu().numpy(), refs[:sample].cpu().numpy()
    t0 = time.perf_counter()
    cpu_ref.score_simd(0, h_reads, h_refs, sc, threads=cores, affine=affine)
    sec = time.perf_counter() - t0
    return {"value": round(float(sample) * R * F / sec / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "identical_to_scalar_oracle": same,
            "sample": "first %d pairs of the rank-0 batch, SW %s int16, oracle/cpu_simd.c (AVX2, 16 pairs per vector, OpenMP over "
                      "batches) on %d threads, %.1f s" % (sample, "affine-gap" if affine else "linear-gap", cores, sec)}


def cpu_baseline(reads, refs, affine, R=R, F=F, band=0, seconds=12.0, simd_seconds=6.0):
    """Oracle on the host cores over a bounded sample of the same batch (rank 0; N > 1: a short sample while the other
    ranks wait): threads = the CPUs this process really has (host_cpus).  Returns (record, oracle scores of the
    sample) -- the scores are what the timed kernel's output is verified against.  `simd`: the AVX2 sweep beside it."""
    from oracle import cpu_ref
    cpu_ref.build()
    cpus = host_cpus()
    cores = max(1, min(cpu_ref.max_threads(), cpus["effective"]))
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, **(AFFINE if affine else {}))

    def run(h_reads, h_refs):
        if band:
            return cpu_ref.score_banded_sw(h_reads, h_refs, band, sc, threads=cores, block_rows=BAND_BLOCK[0], col_align=BAND_BLOCK[1])   # the same cells as the GPU
        return cpu_ref.score(0, h_reads, h_refs, sc, threads=cores, affine=affine)

    probe = max(1, (512 * cores * 75000) // (R * F)) if not band else cores
    probe = min(probe, int(reads.shape[0]))
    h_reads, h_refs = reads[:probe].cpu().numpy(), refs[:probe].cpu().numpy()
    t0 = time.perf_counter()
    run(h_reads, h_refs)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    sample = int(min(reads.shape[0], max(probe, rate * seconds)))
    h_reads, h_refs = reads[:sample].cpu().numpy(), refs[:sample].cpu().numpy()
    t0 = time.perf_counter()
    scores = run(h_reads, h_refs)
    sec = time.perf_counter() - t0
    what = "SW %s int16" % ("affine-gap" if affine else "linear-gap") if not band else "SW linear-gap banded (%d diagonals)" % band
    cells = float(sample) * R * F
    simd = None
    if not band and simd_seconds > 0:
        try:
            simd = cpu_simd_baseline(reads, refs, affine, cores, simd_seconds)
        except Exception as e:      # the SIMD leg must never take the line with it
            simd = {"error": str(e)[:200]}
    return {"simd": simd, "value": round(cells / sec / 1e9, 3), "unit": "GCUPS", "cores": cores, "logical_cpus": cpus["logical_cpus"],
            "affinity_cpus": cpus["affinity"], "cpu_quota": cpus["cpu_quota"], "kind": "port",
            "sample": "first %d pairs of the rank-0 batch, %s, oracle/cpu_ref.c OpenMP over pairs on %d threads (= the CPUs "
                      "the cgroup quota and the affinity mask give this process), %.1f s%s"
                      % (sample, what, cores, sec, " (full-matrix cells counted, as for the GPU)" if band else "")}, scores


def verify_scores(device_scores, reads, refs, affine, R=R, F=F, band=0, oracle_scores=None, pairs=8192):
    """The scores of the timed kernel against the oracle, inside the driver's own run: `oracle_scores` (what
    cpu_baseline computed for the first len(oracle_scores) pairs) or a fresh oracle run over the first `pairs`."""
    import numpy as np
    from oracle import cpu_ref
    if oracle_scores is None:
        cpu_ref.build()
        m = int(min(pairs, reads.shape[0]))
        sc = cpu_ref.Scoring.make(2, -1, -3, -3, **(AFFINE if affine else {}))
        h_reads, h_refs = reads[:m].cpu().numpy(), refs[:m].cpu().numpy()
        th = max(1, min(cpu_ref.max_threads(), host_cores()))
        if band:
            oracle_scores = cpu_ref.score_banded_sw(h_reads, h_refs, band, sc, threads=th, block_rows=BAND_BLOCK[0], col_align=BAND_BLOCK[1])
        else:
            oracle_scores = cpu_ref.score(0, h_reads, h_refs, sc, threads=th, affine=affine)
    m = len(oracle_scores)
    got = device_scores[:m].cpu().numpy()
    bad = int(np.count_nonzero(got != np.asarray(oracle_scores, dtype=np.int16)))
    return {"pairs": m, "mismatches": bad, "against": "oracle/cpu_ref.c (restatement of DefaultKernel.cpp:83-202%s), "
            "first %d pairs of the timed batch, scores of the timed kernel" % (" + affine extension" if affine else "", m)}


def verify_alignments(rows, idx, opt, reads, refs, affine, pairs=2048):
    """Rows and coordinates of a device alignment leg against the oracle on the first `pairs` pairs."""
    import numpy as np
    from oracle import cpu_ref
    cpu_ref.build()
    m = int(min(pairs, reads.shape[0]))
    sc = cpu_ref.Scoring.make(2, -1, -3, -3, **(AFFINE if affine else {}))
    th = max(1, min(cpu_ref.max_threads(), host_cores()))
    exp_rows, exp_idx = cpu_ref.align(opt, reads[:m].cpu().numpy(), refs[:m].cpu().numpy(), sc, threads=th, affine=affine)
    got_rows, got_idx = rows[:m].cpu().numpy(), idx[:m].cpu().numpy()
    bad = int(np.count_nonzero((got_rows.reshape(m, -1) != exp_rows.reshape(m, -1)).any(axis=1) | (got_idx != exp_idx).any(axis=1)))
    return {"pairs": m, "mismatches": bad}


def reference_cpu_kernels(reads, refs):
    """The reference's own compiled kernels (oracle/_ref), linear gap, through the plugin ABI."""
    from versalignlib_amd import host
    out = {}
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    cores = host_cores()
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


def length_sorted_leg(reads, refs, scoring, device_index):
    """The timed batch with every read and reference cut to a uniformly drawn prefix (10-100 %) and NUL-padded, as
    pad() leaves a FASTA of mixed lengths (src/util/versalignUtil.cpp:17-33): swept padded and length-sorted."""
    import torch
    from versalignlib_amd import hipkernel
    n = int(reads.shape[0])
    g = torch.Generator(device=reads.device).manual_seed(77)
    def cut(t):
        L = t.shape[1]
        keep = (torch.rand((n, 1), device=t.device, generator=g) * 0.9 * L + 0.1 * L).to(torch.int64) + 1
        return torch.where(torch.arange(L, device=t.device)[None, :] < keep, t, torch.zeros((), dtype=t.dtype, device=t.device)).contiguous()
    r2, f2 = cut(reads), cut(refs)
    eng = hipkernel.Engine(int(reads.shape[1]), int(refs.shape[1]), scoring, device=device_index)
    out = {}
    scores = {}
    for mode in (0, 2):
        eng.set_ragged_batching(mode)
        scores[mode] = eng.score_device(0, r2, f2)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            eng.score_device(0, r2, f2, scores[mode])
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        d = eng.describe(0, n)
        out["padded_sweep" if mode == 0 else "length_sorted"] = {
            "ms": round(best * 1e3, 3), "gcups_on_padded_shape": round(n * reads.shape[1] * refs.shape[1] / best / 1e9, 1),
            "launches": d.get("ragged_launches"), "cell_fraction": d.get("ragged_cell_fraction")}
    out["identical_to_padded_sweep"] = bool(torch.equal(scores[0], scores[2]))
    out["note"] = ("valign_hip_score_device, wall time of the call (mode 2 waits for the device's length histogram); the engine's own cell "
                   "choice (half floats for this scoring); never part of `value`")
    eng.close()
    return out


def run_child_leg(flags, keep_devices=False):
    """A leg of the bench in a process of its own (`bench.py --child-...`): its JSON line, or what became of it."""
    drop = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"]
    if not keep_devices:
        drop += ["HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"]
    env = {k: v for k, v in os.environ.items() if k not in drop}
    cmd = [sys.executable, os.path.abspath(__file__)] + [str(f) for f in flags]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "child timed out after 300 s"}
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"error": "child exit code %d" % r.returncode, "stderr_tail": r.stderr[-400:]}


def in_plugin_shards_child(devices, threads):
    out = run_child_leg(["--child-in-plugin-shards", devices, "--abi-threads", threads])
    out.setdefault("devices", devices)
    return out


def child_in_plugin_shards(devices, threads):
    """The child: synthetic host sequences (no torch, no rank), the plugin with hip_devices = N: score_alignments with the
    host-side merge of the shards and with the in-plugin RCCL all-gather, checked against each other and against one device."""
    from versalignlib_amd import build, host, synth
    import numpy as np
    blk = 1 << 16
    r0, f0 = synth.make_pairs(blk, R, F, seed=2000)
    h_reads, h_refs = np.tile(r0, (PAIRS_PER_GPU // blk, 1)), np.tile(f0, (PAIRS_PER_GPU // blk, 1))
    n = int(h_reads.shape[0])
    keys = dict(score_gap_open_read=AFFINE["open_read"], score_gap_extend_read=AFFINE["ext_read"],
                score_gap_open_ref=AFFINE["open_ref"], score_gap_extend_ref=AFFINE["ext_ref"])
    out = {"devices": devices, "threads": threads, "pairs": n}
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, **keys) as k1:
        one = k1.score_alignments(0, h_reads, h_refs, scattered=True)[0].copy()
    for name, extra in (("score_alignments_sw", {}), ("score_alignments_sw_rccl_allgather", {"hip_devices_allgather": 1})):
        try:
            with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, hip_devices=devices, hip_devices_strict=1, **extra, **keys) as k:
                k.score_alignments(0, h_reads, h_refs, scattered=True)
                runs = [k.score_alignments(0, h_reads, h_refs, scattered=True) for _ in range(4)]
                best = min(r[1] for r in runs)
                out[name] = {"ms": round(best * 1e3, 2), "gcups": round(n * R * F / best / 1e9, 1),
                             "identical_to_one_device": bool(np.array_equal(runs[-1][0], one))}
        except Exception as e:
            out[name] = {"error": str(e)[:300]}
    print(json.dumps(out), flush=True)
    return 0


def abi_leg(reads, refs, threads):
    """PCIe-inclusive figures through the plugin ABI, exactly as a versalignLib host drives a backend:
    dlopen + set_parameters + set_logger + spawn_alignment_kernel, then the two virtuals on scattered
    heap blocks (pad(), src/util/versalignUtil.cpp:24-31).  Affine scoring as in `value`.  Warm: the first
    call of each kind sizes the pinned staging and is not counted."""
    import numpy as np
    from versalignlib_amd import build, host, synth
    n = int(reads.shape[0])
    h_reads, h_refs = (reads.cpu().numpy(), refs.cpu().numpy()) if hasattr(reads, "cpu") else (reads, refs)
    keys = dict(score_gap_open_read=AFFINE["open_read"], score_gap_extend_read=AFFINE["ext_read"],
                score_gap_open_ref=AFFINE["open_ref"], score_gap_extend_ref=AFFINE["ext_ref"])
    out = {"threads": threads, "pairs": n, "note": "through spawn_alignment_kernel with one heap block per sequence; "
                                                    "PCIe and host gather/scatter included; never part of `value`"}
    # compute_alignments(SW) hands out 2n operator new[] rows per call (include/AlignmentKernel.h:20-23).  What that costs
    # is glibc's arena growth: 128 KB steps (an mprotect each, 16 threads contending) by default -- the plugin's default
    # since round 4 (host_malloc_tuning = 0: it leaves the host's allocator alone) -- and 256 MB steps where the HOST opts
    # in: MALLOC_TOP_PAD_=268435456 in its environment, or the key host_malloc_tuning = 2, which is what this bench, as
    # the host, sets for the legs below (mallopt(M_TOP_PAD): process-wide and sticky, hence the untuned figures first).
    untuned = None
    if True:                    # (the first plugin object of the process: nothing has changed the allocator yet)
        with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, **keys) as k:
            floor_untuned, _ = host.alloc_probe(n, R + F, threads)
            k.time_calls(0, h_reads[:65536], h_refs[:65536], reps=1, align=True, free_between=False)
            _, per_call = k.time_calls(0, h_reads, h_refs, reps=2, align=True, free_between=False)
            untuned = {"ms_fresh_rows": round(min(per_call) * 1e3, 2), "ms_2n_fresh_new_rows_alone": round(floor_untuned * 1e3, 2),
                       "note": "the plugin's default (host_malloc_tuning = 0): glibc's own 128 KB arena steps"}
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, **keys) as k:
        k.score_alignments(0, h_reads[:65536], h_refs[:65536], scattered=True)
        k.score_alignments(0, h_reads, h_refs, scattered=True)
        secs = sorted(k.score_alignments(0, h_reads, h_refs, scattered=True)[1] for _ in range(8))
        phases = [ln for ln in k.drain_log().splitlines() if "score done" in ln]
        out["score_alignments_sw"] = {"ms": round(secs[0] * 1e3, 2), "ms_median_of_8": round(secs[4] * 1e3, 2),
                                      "gcups": round(n * R * F / secs[0] / 1e9, 1),
                                      "transport": "4-bit base classes (host_packing = 1, default)",
                                      "host_phases_last_call": json.loads(phases[-1].split("host phases ")[-1]) if phases else None}
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, host_packing=0, **keys) as k0:
        k0.score_alignments(0, h_reads, h_refs, scattered=True)
        secs = sorted(k0.score_alignments(0, h_reads, h_refs, scattered=True)[1] for _ in range(4))
        out["score_alignments_sw_ascii"] = {"ms": round(secs[0] * 1e3, 2), "gcups": round(n * R * F / secs[0] / 1e9, 1),
                                            "transport": "raw ASCII (host_packing = 0)"}
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, host_malloc_tuning=2, **keys) as k:
        # compute_alignments(SW): 2n operator new[] rows per call (include/AlignmentKernel.h:20-23).
        # "fresh": rows of earlier calls are still alive, as in the reference's timing loop, which leaks them
        # (main.cpp:280-285) -- every call gets memory the process has never touched.  "recycled": the host
        # destroyed the previous call's Alignment array first (what a host that consumes its results does).
        # the floor the contract sets on this host, no plugin involved: 2n rows allocated and written by `threads` threads
        floor_fresh, _ = host.alloc_probe(n, R + F, threads)
        total, per_call = k.time_calls(0, h_reads, h_refs, reps=3, align=True, free_between=False)
        fresh = min(per_call[1:])
        phases = [ln for ln in k.drain_log().splitlines() if "align done" in ln]
        out["compute_alignments_sw"] = {"ms_fresh_rows": round(fresh * 1e3, 2), "gcups_fresh_rows": round(n * R * F / fresh / 1e9, 1),
                                        "ms_2n_fresh_new_rows_alone": round(floor_fresh * 1e3, 2),
                                        "host_malloc_tuning": "2, set by this bench as the host (= MALLOC_TOP_PAD_=268435456); the plugin's default is 0",
                                        "host_malloc_tuning_0": untuned,
                                        "host_phases_last_call": json.loads(phases[-1].split("host phases ")[-1]) if phases else None}
        total, per_call = k.time_calls(0, h_reads, h_refs, reps=3, align=True, free_between=True)
        recycled = min(per_call[1:])
        out["compute_alignments_sw"].update({"ms_recycled_rows": round(recycled * 1e3, 2),
                                             "gcups_recycled_rows": round(n * R * F / recycled / 1e9, 1)})
    # the same call through the flat C entry point (include/valign_hip.h: valign_hip_align_host): caller-provided
    # contiguous result buffers instead of 2n heap blocks -- what an FFI binding that does not need Alignment objects uses
    from versalignlib_amd import hipkernel
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **AFFINE))
    eng.align_host(0, h_reads[:65536], h_refs[:65536], threads=threads)
    bufs = eng.align_host(0, h_reads, h_refs, threads=threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.align_host(0, h_reads, h_refs, threads=threads, out=bufs)
        best = min(best, time.perf_counter() - t0)
    out["compute_alignments_sw_flat_buffers"] = {"ms": round(best * 1e3, 2), "gcups": round(n * R * F / best / 1e9, 1),
                                                 "note": "valign_hip_align_host into the caller's own (reused) result buffers: "
                                                         "pinned staging + host copy"}
    # ... and with those buffers registered once (valign_hip_host_register): the device's copy engine writes them directly
    t0 = time.perf_counter()
    hipkernel.host_register(bufs[0])
    hipkernel.host_register(bufs[1])
    reg_s = time.perf_counter() - t0
    try:
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            eng.align_host(0, h_reads, h_refs, threads=threads, out=bufs)
            best = min(best, time.perf_counter() - t0)
        d = eng.describe(0, n)
        out["compute_alignments_sw_flat_registered"] = {
            "ms": round(best * 1e3, 2), "gcups": round(n * R * F / best / 1e9, 1), "direct_out": d.get("direct_out"),
            "register_once_ms": round(reg_s * 1e3, 1),
            "host_phases_last_call": {k: d.get(k) for k in ("host_gather_ms", "host_wait_ms", "host_drain_ms")},
            "note": "valign_hip_align_host into result buffers registered once with valign_hip_host_register: D2H straight "
                    "into the caller's memory, no host-side copy"}
    finally:
        hipkernel.host_unregister(bufs[0])
        hipkernel.host_unregister(bufs[1])
    eng.close()
    del bufs
    # the reference host's own protocol on BASELINE configs[0]: 1,000 pairs of 64 x 128, linear gaps,
    # 100 back-to-back compute_alignments(SW) calls, microseconds per call (main.cpp:66-69, 268-292)
    r1, f1 = synth.make_pairs(1000, 64, 128, seed=1)
    with host.Plugin(build.HIP_PLUGIN, 64, 128, num_threads=threads) as k:
        for opt in (0, 1):                                   # first use of a kernel loads its code object: not timed
            k.time_calls(opt, r1, f1, reps=5, align=True)
            k.time_calls(opt, r1, f1, reps=5, align=False)
        total, _ = k.time_calls(0, r1, f1, reps=100, align=True)
        total_s, _ = k.time_calls(0, r1, f1, reps=100, align=False)
        total_nw, _ = k.time_calls(1, r1, f1, reps=100, align=False)
        out["reference_protocol_config0"] = {"pairs": 1000, "shape": "64x128", "calls": 100,
                                             "compute_alignments_sw_us_per_call": round(total / 100 * 1e6, 1),
                                             "score_alignments_sw_us_per_call": round(total_s / 100 * 1e6, 1),
                                             "score_alignments_nw_us_per_call": round(total_nw / 100 * 1e6, 1)}
    return out


# ---------------------------------------------------------------------------------------------
# launcher self-test (CPU, gloo): no alignment is computed -- it exists so that the path
# `python bench.py --gpus N` -> child ranks -> all-gather -> one JSON line can be tested without a GPU
# ---------------------------------------------------------------------------------------------

def roofline_record(n, RR, FF, k_ms, kernel_name, pmc_key, extra=None):
    """The `roofline` object of the line: algorithmic bytes of one launch (R + F + 2 per pair) over the
    dominant kernel's launch time (HIP events on its own stream, kernel_launch_ms) against the 8 TB/s HBM peak; measured
    traffic / VALU issue / LDS conflicts quoted from the committed PMC profile of exactly these kernel sources.
    k_ms None (the launcher's CPU self-test, which runs no kernel): the same keys, nothing achieved."""
    alg_bytes = float(n) * (RR + FF + 2)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms else None
    rec = {"bound": "hbm", "achieved": round(achieved, 2) if achieved is not None else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBPS, 6) if achieved is not None else None,
           "traffic": measured_traffic(pmc_key, n) if k_ms else None,
           "kernel": kernel_name, "kernel_ms": round(k_ms, 4) if k_ms else None, "algorithmic_bytes": alg_bytes,
           "kernel_gcups": round(n * RR * FF / (k_ms * 1e-3) / 1e9, 1) if k_ms else None,
           "note": "integer VALU bound: %.5f B/cell algorithmic, HBM is idle by design; traffic / valu are quoted "
                   "from a committed PMC profile only when it was taken from these exact kernel sources (csrc %s)"
                   % ((RR + FF + 2) / (RR * FF), source_hash()),
           "valu": measured_valu_issue(pmc_key, n) if k_ms else None,
           "lds_bank_conflict_frac": measured_lds_conflicts(pmc_key, n) if k_ms else None}
    if extra:
        rec.update(extra)
    return rec


def selftest_rank(args, world, rank):
    import torch
    import torch.distributed as dist
    from versalignlib_amd import shard
    dist.init_process_group("gloo")
    n = args.pairs
    local = ((torch.arange(n, dtype=torch.int64) * 7 + rank * 1000003) % 30011).to(torch.int16)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = shard.all_gather_scores(local, n_total=n * world)
    elapsed = time.perf_counter() - t0
    mine = torch.tensor([int(local.to(torch.int64).sum())], dtype=torch.int64)
    dist.all_reduce(mine)
    if rank == 0:
        # the CPU leg of a real N > 1 line, on a small synthetic batch: rank 0 runs the oracle while the others wait below
        from versalignlib_amd import synth
        h_reads, h_refs = synth.make_pairs(512, R, F, seed=2000)
        cpu, _ = cpu_baseline(torch.from_numpy(h_reads), torch.from_numpy(h_refs), affine=True, seconds=0.2, simd_seconds=0.1)
        print(json.dumps({"metric": "launcher self-test (no alignment computed)", "value": 0.0, "unit": "none",
                          "n_gpus": world, "steps": args.steps, "warmup": 0, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "data": "launcher-selftest", "backend": "gloo",
                          "roofline": roofline_record(n, R, F, None, "(none: the self-test runs no kernel)", ""),
                          "cpu_baseline": cpu,
                          "multi_gpu": {"n_ranks_seen": dist.get_world_size(), "gathered_pairs": int(full.numel()),
                                        "gather_checksum": int(full.to(torch.int64).sum()),
                                        "sum_of_rank_checksums": int(mine.item()),
                                        "checksum_ok": int(full.to(torch.int64).sum()) == int(mine.item())}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------

def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("short", "long"), default="short",
                    help="short: BASELINE configs[1]/[3] (150x500 SW affine, default); long: configs[4] per GPU "
                         "(10 kbp x 10 kbp banded SW, int32 cells)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the config's size)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-abi", action="store_true", help="skip the plugin-ABI (PCIe-inclusive) leg")
    ap.add_argument("--abi-threads", type=int, default=0, help="num_threads of the ABI leg (default: host cores, <= 16)")
    ap.add_argument("--child-in-plugin-shards", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU/gloo self-test of launcher + all-gather + result relay; computes no alignment")
    args = ap.parse_args(argv)

    if args.child_in_plugin_shards:
        return child_in_plugin_shards(args.child_in_plugin_shards, args.abi_threads or 16)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)          # nothing above touched the GPU
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.selftest_launcher:
        if not args.pairs:
            args.pairs = 1001
        selftest_rank(args, world, rank)
        return 0

    import torch
    import torch.distributed as dist
    from versalignlib_amd import build, hipkernel, shard

    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: libHIPKernel.so has no CPU path", file=sys.stderr)
        return 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    gloo = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
        gloo = dist.new_group(backend="gloo")          # host-side waits that keep the GPUs free

    if not os.path.exists(build.HIP_PLUGIN):
        build.build_hip()
    long_mode = args.workload == "long"
    RR, FF = (LONG_R, LONG_F) if long_mode else (R, F)
    n = args.pairs or (LONG_PAIRS_PER_GPU if long_mode else PAIRS_PER_GPU)
    reads, refs = synth_on_device(n, device, seed=2000 + rank, R=RR, F=FF)
    affine_sc = hipkernel.Scoring.make(2, -1, -3, -3, **AFFINE)
    lin_sc = hipkernel.Scoring.make()
    if long_mode:
        eng = hipkernel.Engine(RR, FF, lin_sc, device=local_rank)
        eng.set_band_width(LONG_BAND)
        eng.set_score_width(32)
        dd = eng.describe(0, n)
        BAND_BLOCK[:] = [dd["band_block_rows"], dd["band_col_align"]]
    else:
        # `value` is the int16-cell kernel BASELINE.json names; the half-float-cell kernel the engine would
        # pick by itself for this scoring is reported beside it (half_float)
        eng = hipkernel.Engine(RR, FF, affine_sc, device=local_rank)
        eng.set_half_float_cells(0)
    local = torch.empty(n, dtype=torch.int16, device=device)
    gathered = [None]

    def step():
        eng.score_device(0, reads, refs, local)
        gathered[0] = shard.all_gather_scores(local, n_total=n * world) if world > 1 else local

    def step_no_gather():
        eng.score_device(0, reads, refs, local)

    elapsed = timed_steps(step, args.steps, args.warmup, world)
    total_cells = float(n) * RR * FF * world * args.steps
    value = total_cells / elapsed / 1e9
    timed_scores = local.clone() if rank == 0 else None       # what the last timed step wrote (later legs reuse `local`)

    multi = None
    if world > 1:
        # evidence that the collective did its job: every rank's shard is in the gathered vector
        mine = local.to(torch.int64).sum().reshape(1)
        dist.all_reduce(mine)
        got = gathered[0].to(torch.int64).sum()
        elapsed_ng = timed_steps(step_no_gather, args.steps, 1, world)
        multi = {"n_ranks_seen": dist.get_world_size(), "gathered_pairs": int(gathered[0].numel()),
                 "gather_checksum": int(got.item()), "sum_of_rank_checksums": int(mine.item()),
                 "checksum_ok": int(got.item()) == int(mine.item()),
                 "without_all_gather": {"ms_per_step": round(elapsed_ng / args.steps * 1e3, 3),
                                        "value": round(total_cells / elapsed_ng / 1e9, 1)},
                 "all_gather_bytes_per_rank": 2 * n, "devices_visible_per_rank": torch.cuda.device_count()}

    rc = 0
    if rank == 0:
        assert gathered[0].numel() == n * world
        k_ms = kernel_launch_ms(eng, 0, reads, refs, local, max(3, min(args.steps, 10)))
        d = eng.describe(0, n)
        cells = d.get("score_cells", "int16")
        if long_mode:
            chain = d.get("band_block_rows") == 16
            kernel_name = "score_band_kernel<16,shared-gap> (cyclic block chain, int32 cells)" if chain else "score_long_kernel<16,10,SW,shared-gap,int32>"
            pmc_key = "score_band_kernel<16, true>" if chain else "score_long_kernel<16, 10, 0, true, true, false>"
            workload = ("%d pairs/GPU, 10 kbp x 10 kbp, SW linear-gap banded (%d diagonals), int32 cells, inputs resident in HBM%s"
                        % (n, LONG_BAND, ", RCCL all-gather of scores" if world > 1 else ""))
            metric = "GCUPS (giga DP cell updates/sec, full-matrix cells) SW banded, 10 kbp x 10 kbp"
        else:
            kernel_name = "score_kernel<%d,%d,SW,affine-sym>" % (d["group_lanes"], d["rows_per_lane"])
            pmc_key = "score_kernel<%d, %d, 0, 3>" % (d["group_lanes"], d["rows_per_lane"])
            workload = ("%d pairs/GPU, 150 bp x 500 bp, SW affine-gap int16 score (open -5, extend -1), inputs resident in HBM%s"
                        % (n, ", RCCL all-gather of scores" if world > 1 else ""))
            metric = "GCUPS (giga DP cell updates/sec) SW affine-gap, 150x500 bp"
        line = {
            "metric": metric,
            "value": round(value, 1), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": cells, "data": "synthetic",
            "config": {"workload": workload, "pairs_per_gpu": n, "read_length": RR, "ref_length": FF,
                       "kernel_geometry": "%dx%d" % (d["group_lanes"], d["rows_per_lane"])},
            # banded workload: `value` counts the cells of the FULL matrix (the metric of BASELINE config 5 as north_star words
            # it); the cells the band actually holds -- what the kernel sweeps -- are far fewer: both rates, side by side
            **({"swept_gcups": round(world * n * d.get("band_cells_per_pair", 0) / elapsed * args.steps / 1e9, 1),
                "band_cells_per_pair": d.get("band_cells_per_pair"), "full_matrix_cells_per_pair": RR * FF} if long_mode else {}),
            "roofline": roofline_record(n, RR, FF, k_ms, kernel_name, pmc_key,
                                        {"band_waves_per_cu": d.get("band_waves_per_cu"), "band_lds_per_wave": d.get("band_lds_per_wave")} if long_mode else None),
        }
        if multi is not None:
            line["multi_gpu"] = multi
            if not multi["checksum_ok"] or multi["n_ranks_seen"] != world:
                rc = 3
        if not long_mode:
            # the half-float-cell kernel (what the plugin picks by itself for this scoring): identical scores
            eng_f16 = hipkernel.Engine(RR, FF, affine_sc, device=local_rank)
            f16_scores = eng_f16.score_device(0, reads, refs)
            torch.cuda.synchronize()
            f16_ms = kernel_launch_ms(eng_f16, 0, reads, refs, f16_scores, 5)
            eng.score_device(0, reads, refs, local)
            torch.cuda.synchronize()
            line["half_float"] = {"cells": eng_f16.describe(0, n).get("score_cells"),
                                  "identical_to_int16_kernel": bool(torch.equal(local, f16_scores)),
                                  "kernel_ms": round(f16_ms, 4), "kernel_gcups": round(n * RR * FF / (f16_ms * 1e-3) / 1e9, 1),
                                  "note": "v_pk_maximum3_f16 recurrence on integers of magnitude <= 2048 (exact in fp16); "
                                          "the engine's own choice for this scoring, bit-identical results"}
            eng_f16.close()
            del f16_scores
            eng_lin = hipkernel.Engine(RR, FF, lin_sc, device=local_rank)
            lin_ms = kernel_launch_ms(eng_lin, 0, reads, refs, local, 5)
            line["linear_gap"] = {"kernel_ms": round(lin_ms, 4), "kernel_gcups": round(n * RR * FF / (lin_ms * 1e-3) / 1e9, 1),
                                  "cells": eng_lin.describe(0, n).get("score_cells"),
                                  "note": "reference's own gap model (bit-exact path), same batch"}
            # BASELINE config 3 (NW + traceback) on the same batch, device-resident: extra evidence,
            # never part of `value`
            try:
                AL = RR + FF
                rows = torch.empty((n, 2, AL), dtype=torch.uint8, device=device)
                idx = torch.empty((n, 4), dtype=torch.int16, device=device)
                line["alignments"] = {}
                # (sw_linear: the call the reference's own timing loop makes, src/impl/main.cpp:278-287, device-resident here)
                for name, e, opt, aff in (("nw_affine_traceback", eng, 1, True), ("nw_linear_traceback", eng_lin, 1, False),
                                          ("sw_linear_traceback", eng_lin, 0, False)):
                    e.align_device(opt, reads, refs, rows, idx)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    e.align_device(opt, reads, refs, rows, idx)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    line["alignments"][name] = {"ms": round(ms, 3), "gcups": round(n * RR * FF / (ms * 1e-3) / 1e9, 1),
                                                "algorithmic_GBps": round(n * (3 * AL + 8) / (ms * 1e-3) / 1e9, 1),
                                                "verified": verify_alignments(rows, idx, opt, reads, refs, aff)}
                    if line["alignments"][name]["verified"]["mismatches"]:
                        rc = 4
                del rows, idx
            except hipkernel.HipKernelError as e:
                line["alignments"] = {"error": str(e)[:200]}
            eng_lin.close()
            # a mixed-length batch of the same size, classified, packed and swept by length class on
            # the device (ragged_batching = 2) against the padded sweep of the same bytes -- identical scores required
            try:
                line["length_sorted"] = length_sorted_leg(reads, refs, affine_sc, local_rank)
                if not line["length_sorted"]["identical_to_padded_sweep"]:
                    rc = 4
            except hipkernel.HipKernelError as e:
                line["length_sorted"] = {"error": str(e)[:200]}
            if not args.no_abi:
                try:
                    threads = args.abi_threads or min(16, host_cores())
                    line["abi"] = abi_leg(reads, refs, threads)
                    # the plugin picks the half-float-cell kernel for this scoring: that is the kernel the call contains
                    for key in ("score_alignments_sw", "score_alignments_sw_ascii"):
                        line["abi"][key]["ratio_to_kernel_ms"] = round(line["abi"][key]["ms"] / line["half_float"]["kernel_ms"], 3)
                except Exception as e:
                    line["abi"] = {"error": str(e)[:300]}
                visible = torch.cuda.device_count()
                if world > 1 and visible > 1:
                    # one host process driving several devices through the plugin key hip_devices (and the RCCL all-gather
                    # inside the plugin), on the real devices of this node while the other ranks wait on the host.  In a
                    # CHILD process: this path has never met more than one real device, and whatever it does there must
                    # not take the rank that prints the line with it.
                    line["abi_in_plugin_shards"] = in_plugin_shards_child(min(world, visible), args.abi_threads or min(16, host_cores()))
        oracle_scores = None
        if not args.no_cpu:
            # N == 1: the full CPU legs (10-30 s).  N > 1: the same oracle on a short sample (~1.5 s + ~1 s of the AVX2
            # sweep) on rank 0's host CPUs while the other ranks wait at the final barrier -- the CPU path timed "in the
            # same run" at every N, without holding N - 1 GPUs for half a minute
            short = world > 1
            if long_mode:
                line["cpu_baseline"], oracle_scores = cpu_baseline(reads, refs, affine=False, R=RR, F=FF, band=LONG_BAND,
                                                                   seconds=1.5 if short else 12.0)
            else:
                line["cpu_baseline"], oracle_scores = cpu_baseline(reads, refs, affine=True, seconds=1.5 if short else 12.0,
                                                                   simd_seconds=1.0 if short else 6.0)
                if not short:
                    line["linear_gap"]["cpu_port"], _ = cpu_baseline(reads, refs, affine=False, simd_seconds=3.0)
                    line["linear_gap"]["cpu_reference"] = reference_cpu_kernels(reads, refs)
        # the timed kernel's own output against the oracle, in this very run: the sample cpu_baseline scored (or a
        # small fresh one when the CPU legs are off / N > 1); a mismatch fails the run
        line["verified"] = verify_scores(timed_scores, reads, refs, affine=not long_mode, R=RR, F=FF,
                                         band=LONG_BAND if long_mode else 0, oracle_scores=oracle_scores,
                                         pairs=8192 if not long_mode else 16)
        if line["verified"]["mismatches"]:
            rc = 4
        print(json.dumps(line), flush=True)
        if rc == 4:
            print("bench.py: results differ from the oracle (see `verified`)", file=sys.stderr)
    if world > 1:
        dist.barrier(group=gloo)
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
