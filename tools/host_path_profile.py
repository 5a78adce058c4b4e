"""Where the host-pointer path of score_alignments spends its time (developer tool): packing on the
host threads, blocked on the device, copy-out -- for uniform and mixed-length batches."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from versalignlib_amd import hipkernel, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 20)
    ap.add_argument("--threads", default="8,32")
    a = ap.parse_args()
    R, F, blk = 150, 500, 4096
    for kind in ("uniform", "ragged"):
        make = synth.make_pairs if kind == "uniform" else synth.make_ragged_pairs
        r0, f0 = make(blk, R, F, seed=3)
        reads = np.ascontiguousarray(np.tile(r0, (a.pairs // blk, 1)))
        refs = np.ascontiguousarray(np.tile(f0, (a.pairs // blk, 1)))
        for on in (1, 0):
            for threads in (int(t) for t in a.threads.split(",")):
                eng = hipkernel.Engine(R, F)
                eng.set_ragged_batching(on)
                eng.score_host(0, reads, refs, threads=threads)
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    eng.score_host(0, reads, refs, threads=threads)
                    sec = time.perf_counter() - t0
                    if best is None or sec < best[0]:
                        best = (sec, eng.describe(0, a.pairs))
                d = best[1]
                print(json.dumps({"input": kind, "ragged_batching": on, "threads": threads, "ms": round(best[0] * 1e3, 2),
                                  "gather_ms": d["host_gather_ms"], "classify_ms": d["host_classify_ms"], "wait_ms": d["host_wait_ms"],
                                  "drain_ms": d["host_drain_ms"], "launches": d["ragged_launches"],
                                  "cell_fraction": d["ragged_cell_fraction"]}))
                eng.close()


if __name__ == "__main__":
    main()
