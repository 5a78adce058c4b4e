"""Where a small call spends its time (developer tool): BASELINE config 0 (1 000 pairs of 64 x 128) through the
flat host entry points, engine-reported host phases + wall time, median of many calls."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versalignlib_amd import hipkernel, synth


def main():
    for R, F, n in ((64, 128, 1000), (150, 500, 1000), (64, 128, 16)):
        reads, refs = synth.make_pairs(n, R, F, seed=1)
        eng = hipkernel.Engine(R, F)
        for name, fn in (("score_host SW", lambda: eng.score_host(0, reads, refs, threads=1)),
                         ("score_host NW", lambda: eng.score_host(1, reads, refs, threads=1)),
                         ("align_host SW", lambda: eng.align_host(0, reads, refs, threads=1))):
            for _ in range(10):
                fn()
            ts, ph = [], []
            for _ in range(200):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
                d = eng.describe(0, n)
                ph.append((d["host_gather_ms"], d["host_wait_ms"], d["host_drain_ms"]))
            med = statistics.median(ts)
            print(json.dumps({"shape": "%dx%d x %d" % (R, F, n), "call": name, "wall_us_median_incl_python": round(med * 1e6, 1),
                              "gather_us": round(statistics.median(p[0] for p in ph) * 1e3, 1),
                              "wait_us": round(statistics.median(p[1] for p in ph) * 1e3, 1),
                              "drain_us": round(statistics.median(p[2] for p in ph) * 1e3, 1),
                              "geometry": "%dx%d" % (d["group_lanes"], d["rows_per_lane"])}))
        eng.close()


if __name__ == "__main__":
    main()
