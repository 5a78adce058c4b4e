#!/usr/bin/env python3
"""Randomised parity sweep (developer tool; run on the GPU box): libHIPKernel.so through the plugin ABI against oracle/cpu_ref
on random shapes, scorings, modes and plugin keys, for a bounded time.

    python tools/fuzz_parity.py --seconds 300 --seed 1

Every case is drawn from one seeded generator and printed with what is needed to repeat it (--only N runs case N alone).
The oracle's cell width follows the arithmetic the plugin documents: int16 where the mode's cells fit (the reference's own
arithmetic), the int32 restatement where they do not (scores saturate the ABI's short, alignments are the int32 ones).
Exit code 1 on the first mismatch, the case's parameters on stdout."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cpu_ref                                          # noqa: E402
from versalignlib_amd import build, host, synth                    # noqa: E402


def draw_case(rng, big=False, kinds=None):
    kind = "big" if big else rng.choice(kinds or ["short", "short", "short", "mid", "long", "tiny"])
    if kind == "big":                                     # many pairs: the chunked host pipeline, tail chunks, every slot
        R, F = int(rng.integers(30, 200)), int(rng.integers(60, 600))
    elif kind == "tiny":
        R, F = int(rng.integers(1, 20)), int(rng.integers(1, 30))
    elif kind == "short":
        R, F = int(rng.integers(8, 400)), int(rng.integers(8, 900))
    elif kind == "mid":
        R, F = int(rng.integers(400, 2100)), int(rng.integers(200, 3000))
    elif kind == "longref":                               # short reads against a reference that starves LDS
        R, F = int(rng.integers(20, 600)), int(rng.integers(5000, 20000))
    else:                                                 # row strips (alignments), long-read score kernels
        R, F = int(rng.integers(2049, 6000)), int(rng.integers(1, 5000))
    cells = R * F
    n = int(max(1, min(rng.integers(1, 400), 6_000_000 // max(cells, 1))))
    if kind == "big":
        n = int(rng.integers(40_000, 300_000))
    match = int(rng.integers(1, 6))
    mismatch = -int(rng.integers(0, 7))
    affine = bool(rng.random() < 0.45)
    policy = "sse" if (not affine and rng.random() < 0.25) else "default"
    gr, gf = -int(rng.integers(1, 9)), -int(rng.integers(1, 9))
    if rng.random() < 0.5:
        gf = gr
    keys = dict(score_match=match, score_mismatch=mismatch, score_gap_read=gr, score_gap_ref=gf)
    aff = ()
    if affine:
        er, ef = -int(rng.integers(1, 4)), -int(rng.integers(1, 4))
        orr, of = er - int(rng.integers(0, 8)), ef - int(rng.integers(0, 8))
        if rng.random() < 0.6:
            of, ef = orr, er
        aff = (orr, er, of, ef)
        keys.update(score_gap_open_read=orr, score_gap_extend_read=er, score_gap_open_ref=of, score_gap_extend_ref=ef)
    scale = 1
    if rng.random() < 0.15:                               # scores large enough that cells leave int16: the int32 paths
        scale = int(rng.integers(40, 300))
        keys = {k: (v * scale if k.startswith("score_") and k != "score_width" else v) for k, v in keys.items()}
        match, mismatch, gr, gf = match * scale, mismatch * scale, gr * scale, gf * scale
        aff = tuple(v * scale for v in aff)
    band = 0
    if rng.random() < 0.15 and R >= 64:                   # banded Smith-Waterman scores (linear or affine gaps)
        band = int(rng.integers(8, 400)) * 2
    if policy == "sse":
        keys["traceback_policy"] = 1
    if rng.random() < 0.2:
        keys["ragged_batching"] = int(rng.integers(1, 3))
    if rng.random() < 0.15:
        keys["host_packing"] = 0
    if rng.random() < 0.3:
        keys["num_threads"] = int(rng.integers(1, 9))
    if rng.random() < 0.15:
        keys["score_width"] = 32                          # int32 score cells whatever the range
    if rng.random() < 0.15:
        keys["half_float_cells"] = int(rng.integers(0, 2))
    if rng.random() < 0.1:
        keys["hip_devices"] = 1                           # the in-plugin shard path with one shard
    data = dict(seed=int(rng.integers(1, 1 << 30)), sub_rate=float(rng.choice([0.02, 0.1, 0.3])), indel_rate=float(rng.choice([0.0, 0.02])) if n * R < 400_000 else 0.0,
                n_run_frac=float(rng.choice([0.0, 0.1])), short_frac=float(rng.choice([0.0, 0.2, 0.6])), lowercase_frac=0.05, junk_frac=0.05)
    return dict(R=R, F=F, n=n, keys=keys, aff=aff, affine=affine, policy=policy, data=data, gaps=(gr, gf), match=match, mismatch=mismatch, band=band)


def band_blocks(c):
    """The block shape libHIPKernel.so reports for this shape / band / scoring (include/valign_hip.h documents both)."""
    from versalignlib_amd import hipkernel
    kw = dict(zip(("open_read", "ext_read", "open_ref", "ext_ref"), c["aff"])) if c["affine"] else {}
    eng = hipkernel.Engine(c["R"], c["F"], hipkernel.Scoring.make(c["match"], c["mismatch"], c["gaps"][0], c["gaps"][1], **kw))
    eng.set_band_width(c["band"])
    d = eng.describe(0, 1)
    eng.close()
    return d["band_block_rows"], d["band_col_align"]


def run_case(c, verbose=False):
    R, F, n = c["R"], c["F"], c["n"]
    reads, refs = synth.make_pairs(n, R, F, **c["data"])
    sc = cpu_ref.Scoring.make(c["match"], c["mismatch"], c["gaps"][0], c["gaps"][1], *c["aff"])
    okw = dict(affine=c["affine"])
    if c["band"]:
        rows_, align_ = band_blocks(c)
        with host.Plugin(build.HIP_PLUGIN, R, F, band_width=c["band"], **c["keys"]) as hip:
            got = hip.score_alignments(host.SW, reads, refs)
        exp = cpu_ref.score_banded_sw(reads, refs, c["band"], sc, threads=8, block_rows=rows_, col_align=align_, affine=c["affine"])
        if not np.array_equal(got, exp):
            bad = np.nonzero(got != exp)[0]
            return "banded score (blocks %d / %d): %d of %d differ, first %s got %s exp %s" % (rows_, align_, bad.size, n, bad[:4], got[bad[:4]], exp[bad[:4]])
        return None
    with host.Plugin(build.HIP_PLUGIN, R, F, **c["keys"]) as hip:
        for opt in (host.SW, host.NW):
            got = hip.score_alignments(opt, reads, refs)
            exp = cpu_ref.score(opt, reads, refs, sc, threads=8, wide=True, **okw)
            if not np.array_equal(got, exp):
                bad = np.nonzero(got != exp)[0]
                return "score opt %d: %d of %d differ, first %s got %s exp %s" % (opt, bad.size, n, bad[:4], got[bad[:4]], exp[bad[:4]])
        if R * F * n <= 40_000_000 or (n >= 40_000 and R * F * n <= 12_000_000_000):     # (the oracle keeps a pointer matrix per thread)
            for opt in (host.SW, host.NW):
                try:
                    rows, idx = hip.compute_alignments(opt, reads, refs, normalise=False)
                except host.PluginError as e:
                    return "align opt %d refused: %s" % (opt, e)
                akw = dict(affine=c["affine"]) if c["affine"] else dict(policy=c["policy"])
                e16 = cpu_ref.align(opt, reads, refs, sc, threads=8, **akw)
                e32 = cpu_ref.align(opt, reads, refs, sc, threads=8, wide=True, **akw)
                same16 = np.array_equal(rows, e16[0]) and np.array_equal(idx, e16[1])
                same32 = np.array_equal(rows, e32[0]) and np.array_equal(idx, e32[1])
                if not (same16 or same32):
                    bad = np.nonzero((idx != e32[1]).any(axis=1) | (rows != e32[0]).any(axis=(1, 2)))[0]
                    return "align opt %d: %d of %d differ from the int32 oracle (int16 oracle equal: %s), first %s" % (opt, bad.size, n, same16, bad[:4])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--first", type=int, default=0, help="skip the cases before this index (they are still drawn)")
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs")
    ap.add_argument("--big-every", type=int, default=0, help="every K-th case has 40k-300k pairs (0: none)")
    ap.add_argument("--kinds", default="", help="comma-separated shape kinds to draw from (tiny, short, mid, long, longref); default: a mix")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    i = 0
    done = 0
    while time.time() - t0 < a.seconds or (a.only >= 0 and i <= a.only):
        c = draw_case(rng, big=a.big_every > 0 and i % a.big_every == a.big_every - 1, kinds=a.kinds.split(",") if a.kinds else None)
        if (a.only < 0 or i == a.only) and i >= a.first:
            if a.verbose:
                print("case %d: %r" % (i, c), flush=True)
            try:
                err = run_case(c)
            except host.PluginError as e:
                err = "plugin error: %s" % e
            done += 1
            if err:
                print("MISMATCH case %d (seed %d): %s\n  %r" % (i, a.seed, err, c), flush=True)
                return 1
            if done % 25 == 0:
                print("%d cases, %.0f s" % (done, time.time() - t0), flush=True)
            if i == a.only:
                break
        i += 1
    print("ok: %d cases in %.0f s (seed %d)" % (done, time.time() - t0, a.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
