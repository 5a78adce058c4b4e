#!/usr/bin/env python3
"""Step-by-step model of score_band_kernel's schedule (versalignlib_amd/csrc/band_kernels.hip.h) in plain Python.

The banded long-read kernel runs a lane group as a CYCLIC systolic chain: lane l owns row block b = m * G + l of K
rows in "strip" m and sweeps only that block's own band window; block b starts d steps after block b - 1, the bottom
row of a block reaches the next lane through an LDS delay ring (read D_b = d - (lo_b - lo_{b-1}) steps after it was
written), and lane 0 of strip m + 1 follows lane G - 1 of strip m through the very same ring -- no strip boundary
rows in HBM, every lane busy in its own window.  This file states that schedule with rings, events and masks exactly
as the kernel has them and checks it against the oracle's block-band definition (block_rows = K, col_align = 1) --
the off-by-ones are settled here, on the CPU, before a GPU sees them.

    python tools/band_schedule_model.py          # random shapes, asserts equality with oracle/cpu_ref
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cls(ch):
    u = ch & 0xDF
    return {65: 0, 84: 1, 67: 2, 71: 3}.get(u, 4) if ch < 0x80 else 4        # A T C G, else "none"


def plan(R, F, w, G, K):
    """Host-side plan: strips, padding, per-block windows, step distance d, ring depths."""
    rows = G * K
    M = max(1, (R + rows - 1) // rows)
    pad = M * rows - R
    nb = M * G
    lo_raw, lo, hi = [], [], []
    for b in range(nb):
        r_lo, r_hi = b * K - pad, (b + 1) * K - pad - 1
        if r_hi < 0:                                   # a block of padding rows only: nothing to compute
            lo_raw.append(None)
            lo.append(1)
            hi.append(0)
            continue
        r_lo = max(r_lo, 0)
        r_hi = min(r_hi, R - 1)
        a = r_lo * F // R - w
        lo_raw.append(a)
        lo.append(max(a, 0))
        hi.append(min(r_hi * F // R + w, F - 1))
    # blocks of padding take the start of the first real block (they compute nothing; their lanes write zeros)
    first = next(x for x in lo_raw if x is not None)
    pads = sum(1 for x in lo_raw if x is None)                          # (the links out of padding blocks carry zeros at any delay)
    lo_raw = [first if x is None else x for x in lo_raw]
    start = [x - 1 for x in lo_raw]                    # one warm-up column: the diagonal neighbour of the first cell
    dlo = [start[b] - start[b - 1] for b in range(1, nb)]
    width = max(hi[b] - start[b] + 1 for b in range(nb))            # steps a block needs (warm-up included)
    dmax = max(dlo) if dlo else 0
    # a block reads its predecessor up to dmax steps "late"; the predecessor may already have begun its next block by
    # then, but only with that block's warm-up step (which writes 0, the value the band gives the cell): - 1
    d = max(-(-max(width, width + dmax - 1) // G), dmax + 1)          # (and a lane finishes its own block first)
    dmin = min(dlo[pads:]) if len(dlo) > pads else dmax
    unit = dmin == dmax and d == dmax + 1                               # every delay 1: the kernel's DPP variant
    if not unit:
        d = max(d, dmax + 2)                                            # the ring variant reads one step ahead: delays >= 2
    return dict(unit=unit, M=M, pad=pad, nb=nb, start=start, lo=lo, hi=hi, d=d, P=G * d)


def model_score(read, ref, w, match, mismatch, gap, G, K):
    R, F = len(read), len(ref)
    pl = plan(R, F, w, G, K)
    d, P, nb = pl["d"], pl["P"], pl["nb"]
    depth = 1
    while depth < d + 1:
        depth *= 2
    ring = np.zeros((G, depth), dtype=np.int64)        # lane l's bottom-row outputs, slot = step & (depth - 1)
    H = np.zeros((G, K), dtype=np.int64)               # previous column of the lane's K rows
    up0 = np.zeros(G, dtype=np.int64)
    col = np.full(G, -10 ** 9, dtype=np.int64)
    lo = np.ones(G, dtype=np.int64)
    hi = np.zeros(G, dtype=np.int64)
    delay = np.ones(G, dtype=np.int64)
    rows_cls = np.full((G, K), 4, dtype=np.int64)      # class of the read base of each of the lane's rows (the "profile")
    best = 0
    total_steps = (nb + G) * d
    for t in range(total_steps):
        if t % d == 0:                                 # ---- event: block b = t / d starts on lane b % G ----
            b = t // d
            l = b % G
            H[l, :] = 0
            up0[l] = 0
            if b < nb:
                col[l] = pl["start"][b]
                lo[l], hi[l] = pl["lo"][b], pl["hi"][b]
                delay[l] = d - (pl["start"][b] - pl["start"][b - 1]) if b > 0 else 1
                for q in range(K):
                    r = b * K - pl["pad"] + q
                    rows_cls[l, q] = cls(read[r]) if 0 <= r < R else 4
            else:
                lo[l], hi[l] = 1, 0                    # no block left: idle, writes zeros
        out = np.zeros(G, dtype=np.int64)
        new_up = np.zeros(G, dtype=np.int64)
        for l in range(G):                             # all lanes in lockstep: reads first (ring state of step t - 1)
            pred = (l - 1) % G
            new_up[l] = ring[pred, (t - delay[l]) & (depth - 1)]
        for l in range(G):
            diag0 = up0[l]
            up0[l] = new_up[l]
            j = col[l]
            if lo[l] <= j <= hi[l]:
                c = cls(ref[j])
                h_up, d_in = up0[l], diag0
                for q in range(K):
                    a = rows_cls[l, q]
                    s = 0 if (a == 4 or c == 4) else (match if a == c else mismatch)
                    left = H[l, q]
                    h = max(d_in + s, left - gap, h_up - gap, 0)
                    d_in = left                        # this row's left neighbour is the next row's diagonal one
                    H[l, q] = h
                    h_up = h
                    best = max(best, h)
                out[l] = h_up
            else:
                if j > hi[l]:
                    H[l, :] = 0                        # (the kernel leaves them; they are reset at the next event)
            col[l] += 1
        ring[:, t & (depth - 1)] = out
    return min(best, 32767), pl


def main():
    from oracle import cpu_ref
    from versalignlib_amd import synth
    cpu_ref.build()
    rng = np.random.default_rng(int(os.environ.get("BAND_MODEL_SEED", "5")))
    cases = 0
    for G, K in ((4, 2), (8, 2), (4, 4), (32, 16)):
        for _ in range(60 if G < 32 else 3):
            R = int(rng.integers(1, 90)) if G < 32 else int(rng.integers(400, 1300))
            F = int(rng.integers(1, 120)) if G < 32 else int(rng.integers(300, 1500))
            w = int(rng.integers(1, 12)) if G < 32 else int(rng.integers(1, 40))
            reads, refs = synth.make_pairs(2, R, F, seed=int(rng.integers(1, 1 << 30)), sub_rate=0.1, indel_rate=0.05 if R > 8 else 0.0,
                                           n_run_frac=0.3, short_frac=0.3)
            exp = cpu_ref.score_banded_sw(reads, refs, 2 * w, threads=1, block_rows=K, col_align=1)
            for p in range(2):
                got, pl = model_score(bytes(reads[p]), bytes(refs[p]), w, 2, -1, 3, G, K)
                assert got == exp[p], (G, K, R, F, w, p, got, int(exp[p]), pl["d"])
            cases += 2
    print("band schedule model == oracle block band on %d random pairs" % cases)


if __name__ == "__main__":
    main()
