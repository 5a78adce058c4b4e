#!/usr/bin/env python3
"""Host-pointer path of libHIPKernel.so, 1,048,576 pairs of 150 x 500 (affine scoring as bench.py): wall time of
valign_hip_score_host (4-bit classes vs ASCII, chunk size, ramp) and valign_hip_align_host (staged vs registered
result buffers, chunk size, walks chained beside the next fill or in stream order).  Run on the GPU box."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from versalignlib_amd import hipkernel, synth      # noqa: E402

AFF = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)


def engine(env, R, F):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **AFF))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def best(fn, reps):
    fn()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        out.append(time.perf_counter() - t0)
    return min(out) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--what", default="score,align")
    args = ap.parse_args()
    R, F, n = 150, 500, args.n
    blk = 65536
    r0, f0 = synth.make_pairs(blk, R, F, seed=3)
    reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
    if "score" in args.what:
        print("# valign_hip_score_host, %d pairs, %d threads: ms (best of %d)" % (n, args.threads, args.reps))
        for pack in (1, 0):
            for mb in (16, 32, 48, 96):
                for ramp in (1, 0):
                    env = {"VALIGN_HIP_CHUNK_BYTES": mb << 20}
                    if not ramp:
                        env["VALIGN_HIP_NO_RAMP"] = "1"
                    eng = engine(env, R, F)
                    eng.set_host_packing(pack)
                    ms = best(lambda: eng.score_host(0, reads, refs, threads=args.threads), args.reps)
                    d = eng.describe(0, n)
                    print("pack %d chunk %3d MB ramp %d: %7.2f ms   gather %.2f wait %.2f drain %.2f" %
                          (pack, mb, ramp, ms, d["host_gather_ms"], d["host_wait_ms"], d["host_drain_ms"]), flush=True)
                    eng.close()
    if "align" in args.what:
        print("# valign_hip_align_host (SW), %d pairs, %d threads: ms (best of %d)" % (n, args.threads, args.reps))
        rows = np.zeros((n, 2, R + F), dtype=np.uint8)
        idx = np.zeros((n, 4), dtype=np.int16)
        for registered in (1, 0):
            if registered:
                hipkernel.host_register(rows)
                hipkernel.host_register(idx)
            for mb in (64, 128, 256):
                for chained, primed, issuer in ((1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 1)):
                    env = {"VALIGN_HIP_ALIGN_CHUNK_BYTES": mb << 20}
                    if not chained:
                        env["VALIGN_HIP_NO_OVERLAP"] = "1"
                    if not primed:
                        env["VALIGN_HIP_NO_ENGINE_PRIMING"] = "1"
                    if not issuer:
                        env["VALIGN_HIP_D2H_ON_STREAM"] = "1"
                    eng = engine(env, R, F)
                    ms = best(lambda: eng.align_host(0, reads, refs, threads=args.threads, out=(rows, idx)), args.reps)
                    d = eng.describe(0, n)
                    print("registered %d chunk %3d MB chained walks %d engines primed %d copy issuer %d: %7.2f ms   direct_out %d gather %.2f wait %.2f drain %.2f" %
                          (registered, mb, chained, primed, issuer, ms, d["direct_out"], d["host_gather_ms"], d["host_wait_ms"], d["host_drain_ms"]), flush=True)
                    eng.close()
            if registered:
                hipkernel.host_unregister(rows)
                hipkernel.host_unregister(idx)


if __name__ == "__main__":
    main()
