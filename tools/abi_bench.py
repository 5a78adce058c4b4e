"""End-to-end rate THROUGH THE HOST ABI (PCIe-inclusive): score_alignments / compute_alignments
of libHIPKernel.so called with scattered host pointers, exactly as the reference host calls a
backend (src/impl/main.cpp:268-287).  Never bench.py's `value`; recorded in DESIGN.md."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from versalignlib_amd import build, host, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 18)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--align-pairs", type=int, default=1 << 16)
    ap.add_argument("--malloc-tuning", type=int, default=0, help="plugin key host_malloc_tuning")
    ap.add_argument("--ragged", action="store_true",
                    help="mixed-length, NUL-padded sequences; times score_alignments with length-sorted "
                         "batching on and off (GCUPS counted on the padded shape, as the reference does)")
    a = ap.parse_args()
    R, F = 150, 500
    blk = 4096
    r0, f0 = synth.make_pairs(blk, R, F, seed=3)
    reads = np.tile(r0, (a.pairs // blk, 1))
    refs = np.tile(f0, (a.pairs // blk, 1))
    if a.ragged:
        r0, f0 = synth.make_ragged_pairs(blk, R, F, seed=3)
        reads = np.tile(r0, (a.pairs // blk, 1))
        refs = np.tile(f0, (a.pairs // blk, 1))
        for on in (1, 0):
            with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=a.threads, ragged_batching=on) as k:
                k.score_alignments(0, reads, refs, scattered=True)
                best = min(k.score_alignments(0, reads, refs, scattered=True)[1] for _ in range(3))
                print(json.dumps({"call": "score_alignments(SW) via ABI, ragged input", "ragged_batching": on,
                                  "pairs": a.pairs, "threads": a.threads, "seconds": round(best, 4),
                                  "padded_gcups_pcie_inclusive": round(synth.gcups(a.pairs, R, F, best), 1)}))
        return
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=a.threads, host_malloc_tuning=a.malloc_tuning) as k:
        k.score_alignments(0, reads[:blk], refs[:blk])
        for rep in range(3):
            _, sec = k.score_alignments(0, reads, refs, scattered=True)
            print(json.dumps({"call": "score_alignments(SW) via ABI, scattered host pointers", "pairs": a.pairs,
                              "threads": a.threads, "seconds": round(sec, 4),
                              "gcups_pcie_inclusive": round(synth.gcups(a.pairs, R, F, sec), 1)}))
        n = a.align_pairs
        k.compute_alignments(0, reads[:n], refs[:n], normalise=False)      # sizes the pinned staging once
        for rep in range(2):
            t0 = time.perf_counter()
            k.compute_alignments(0, reads[:n], refs[:n], normalise=False)
            sec = time.perf_counter() - t0
        inside = k.last_call_seconds()
        phases = [ln for ln in k.drain_log().splitlines() if "align done" in ln]
        print(json.dumps({"call": "compute_alignments(SW) via ABI, 2n new[] rows", "pairs": n,
                          "seconds_in_plugin": round(inside, 4),
                          "gcups_pcie_inclusive": round(synth.gcups(n, R, F, inside), 1),
                          "seconds_with_harness_copy_out": round(sec, 4),
                          "host_phases": phases[-1].split("host phases ")[-1] if phases else None}))


if __name__ == "__main__":
    main()
