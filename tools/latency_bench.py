"""Small-batch latency through the plugin ABI (developer tool): BASELINE config 1 (1 000 pairs of
64 x 128, NW linear) and a few other small calls, median of many repetitions."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versalignlib_amd import build, host, synth


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # num_threads of the plugin (host gather / scatter)
    for R, F, n, opt, name in ((64, 128, 1000, 1, "config 1: 1k x 64x128 NW score"),
                               (64, 128, 1000, 0, "1k x 64x128 SW score"),
                               (150, 500, 1000, 0, "1k x 150x500 SW score"),
                               (150, 500, 16, 0, "16 x 150x500 SW score")):
        reads, refs = synth.make_pairs(n, R, F, seed=1)
        with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads) as k:
            for _ in range(5):
                k.score_alignments(opt, reads, refs, scattered=True)
            ts = sorted(k.score_alignments(opt, reads, refs, scattered=True)[1] for _ in range(200))
            k.compute_alignments(opt, reads, refs, normalise=False)
            ta = []
            for _ in range(50):
                k.compute_alignments(opt, reads, refs, normalise=False)
                ta.append(k.last_call_seconds())
            print(json.dumps({"call": name, "num_threads": threads, "score_us_median": round(statistics.median(ts) * 1e6, 1),
                              "score_us_min": round(ts[0] * 1e6, 1),
                              "align_us_median": round(statistics.median(ta) * 1e6, 1),
                              "gcups_score": round(synth.gcups(n, R, F, statistics.median(ts)), 2)}))


if __name__ == "__main__":
    main()
