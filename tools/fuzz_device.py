#!/usr/bin/env python3
"""Randomised sweep of the flat device / host API (developer tool; run on the GPU box): hipkernel.Engine objects created and
destroyed by the hundred, score_device / align_device on torch tensors and random streams, score_host / align_host on numpy
arrays (with and without a registered destination), length-sorted batching, bands, pointer-scratch caps -- every result
against oracle/cpu_ref.  What tools/fuzz_parity.py does for the plugin ABI, for the entry points the tests of the device path
use; a process that runs it for minutes also exercises the library's set-up and tear-down far more often than the suite.

    python -X faulthandler tools/fuzz_device.py --seconds 300 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                                       # noqa: E402

from oracle import cpu_ref                                          # noqa: E402
from versalignlib_amd import hipkernel, synth                      # noqa: E402


def draw(rng):
    kind = rng.choice(["short", "short", "mid", "long", "longref", "tiny"])
    if kind == "tiny":
        R, F = int(rng.integers(1, 24)), int(rng.integers(1, 40))
    elif kind == "short":
        R, F = int(rng.integers(8, 400)), int(rng.integers(8, 900))
    elif kind == "mid":
        R, F = int(rng.integers(400, 2100)), int(rng.integers(200, 3000))
    elif kind == "longref":
        R, F = int(rng.integers(20, 400)), int(rng.integers(3000, 12000))
    else:
        R, F = int(rng.integers(2049, 5000)), int(rng.integers(50, 4000))
    n = int(max(1, min(rng.integers(1, 600), 5_000_000 // max(R * F, 1))))
    affine = bool(rng.random() < 0.45)
    match, mismatch = int(rng.integers(1, 6)), -int(rng.integers(0, 7))
    gr = -int(rng.integers(1, 9))
    gf = gr if rng.random() < 0.5 else -int(rng.integers(1, 9))
    aff = {}
    if affine:
        er, ef = -int(rng.integers(1, 4)), -int(rng.integers(1, 4))
        orr, of = er - int(rng.integers(0, 8)), ef - int(rng.integers(0, 8))
        if rng.random() < 0.6:
            of, ef = orr, er
        aff = dict(open_read=orr, ext_read=er, open_ref=of, ext_ref=ef)
    return dict(R=R, F=F, n=n, affine=affine, match=match, mismatch=mismatch, gr=gr, gf=gf, aff=aff,
                seed=int(rng.integers(1, 1 << 30)), ragged=int(rng.integers(0, 3)) if rng.random() < 0.3 else 0,
                band=int(rng.integers(8, 300)) * 2 if (rng.random() < 0.12 and R >= 64) else 0,
                stream=bool(rng.random() < 0.5), policy=1 if (not affine and rng.random() < 0.2) else 0,
                cap_mb=int(rng.integers(1, 64)) if rng.random() < 0.15 else 0, host=bool(rng.random() < 0.3),
                threads=int(rng.integers(1, 9)))


def run(c):
    R, F, n = c["R"], c["F"], c["n"]
    reads, refs = synth.make_pairs(n, R, F, seed=c["seed"], sub_rate=0.1, indel_rate=0.02 if n * R < 300_000 else 0.0, n_run_frac=0.1,
                                   short_frac=0.3, lowercase_frac=0.05, junk_frac=0.05)
    osc = cpu_ref.Scoring.make(c["match"], c["mismatch"], c["gr"], c["gf"], *(c["aff"].values() if c["affine"] else ()))
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(c["match"], c["mismatch"], c["gr"], c["gf"], **c["aff"]))
    try:
        if c["ragged"]:
            eng.set_ragged_batching(c["ragged"])
        if c["policy"]:
            eng.set_traceback_policy(1)
        if c["cap_mb"]:
            eng.set_pointer_scratch_cap_mb(c["cap_mb"])
        stream = torch.cuda.Stream() if c["stream"] else None
        d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
        torch.cuda.synchronize()
        if c["band"]:
            eng.set_band_width(c["band"])
            d = eng.describe(0, n)
            got = eng.score_device(0, d_reads, d_refs, stream=stream)
            if stream is not None:
                stream.synchronize()
            exp = cpu_ref.score_banded_sw(reads, refs, c["band"], osc, threads=8, block_rows=d["band_block_rows"], col_align=d["band_col_align"], affine=c["affine"])
            if not np.array_equal(got.cpu().numpy(), exp):
                return "banded score_device differs"
            return None
        for opt in (0, 1):
            exp = cpu_ref.score(opt, reads, refs, osc, threads=8, affine=c["affine"], wide=True)
            got = eng.score_device(opt, d_reads, d_refs, stream=stream)
            if stream is not None:
                stream.synchronize()
            if not np.array_equal(got.cpu().numpy(), exp):
                return "score_device opt %d differs" % opt
            if c["host"]:
                if not np.array_equal(eng.score_host(opt, reads, refs, threads=c["threads"]), exp):
                    return "score_host opt %d differs" % opt
        if R * F * n <= 30_000_000:
            akw = dict(affine=True) if c["affine"] else dict(policy="sse" if c["policy"] else "default")
            for opt in (0, 1):
                e16 = cpu_ref.align(opt, reads, refs, osc, threads=8, **akw)
                e32 = cpu_ref.align(opt, reads, refs, osc, threads=8, wide=True, **akw)
                rows, idx = eng.align_device(opt, d_reads, d_refs, stream=stream)
                if stream is not None:
                    stream.synchronize()
                rows, idx = rows.cpu().numpy(), idx.cpu().numpy()
                ok = (np.array_equal(rows, e16[0]) and np.array_equal(idx, e16[1])) or (np.array_equal(rows, e32[0]) and np.array_equal(idx, e32[1]))
                if not ok:
                    return "align_device opt %d differs" % opt
                if c["host"]:
                    hrows, hidx = eng.align_host(opt, reads, refs, threads=c["threads"])
                    ok = (np.array_equal(hrows, e16[0]) and np.array_equal(hidx, e16[1])) or (np.array_equal(hrows, e32[0]) and np.array_equal(hidx, e32[1]))
                    if not ok:
                        return "align_host opt %d differs" % opt
        return None
    finally:
        if c["seed"] % 7:                  # (most engines are closed here; the rest are left to the finalizer / the atexit hook)
            eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    i = done = 0
    while time.time() - t0 < a.seconds:
        c = draw(rng)
        if i >= a.first:
            if a.verbose:
                print("case %d: %r" % (i, c), flush=True)
            try:
                err = run(c)
            except hipkernel.HipKernelError as e:
                err = "library error: %s" % e
            done += 1
            if err:
                print("MISMATCH case %d (seed %d): %s\n  %r" % (i, a.seed, err, c), flush=True)
                return 1
            if done % 25 == 0:
                print("%d cases, %.0f s" % (done, time.time() - t0), flush=True)
        i += 1
    print("ok: %d cases in %.0f s (seed %d)" % (done, time.time() - t0, a.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
