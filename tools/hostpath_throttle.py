#!/usr/bin/env python3
"""Is the spread of the host-pointer path's wall time the cgroup's CPU quota?  1,048,576 pairs of 150 x 500 through
valign_hip_score_host (4-bit classes) at several host thread counts, 12 calls each; beside every call the number of CFS
periods in which the cgroup was throttled (cpu.stat: nr_throttled, throttled_usec) while it ran.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import hipkernel, synth      # noqa: E402
from tools.hostpath_sweep import engine             # noqa: E402


def cpu_stat():
    out = {}
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            for line in f:
                k, v = line.split()
                out[k] = int(v)
    except OSError:
        pass
    return out


def main():
    R, F, n, blk = 150, 500, 1 << 20, 65536
    r0, f0 = synth.make_pairs(blk, R, F, seed=3)
    reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
    try:
        print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError:
        print("cpu.max: unreadable")
    what = sys.argv[1] if len(sys.argv) > 1 else "score"
    rows = idx = None
    if what == "align":
        rows = np.zeros((n, 2, R + F), dtype=np.uint8)
        idx = np.zeros((n, 4), dtype=np.int16)
        hipkernel.host_register(rows)
        hipkernel.host_register(idx)
    for th in (8, 10, 12, 13, 14, 15, 16):
        eng = engine({}, R, F)
        call = (lambda: eng.score_host(0, reads, refs, threads=th)) if what == "score" else \
               (lambda: eng.align_host(0, reads, refs, threads=th, out=(rows, idx)))
        call()
        times, thr, usec = [], [], []
        for _ in range(12):
            a = cpu_stat()
            t0 = time.perf_counter()
            call()
            times.append((time.perf_counter() - t0) * 1e3)
            b = cpu_stat()
            thr.append(b.get("nr_throttled", 0) - a.get("nr_throttled", 0))
            usec.append((b.get("throttled_usec", 0) - a.get("throttled_usec", 0)) / 1e3)
            time.sleep(0.12)                               # let a fresh quota period begin
        s = sorted(times)
        print("%s %2d threads: min %6.2f  median %6.2f  max %6.2f ms | throttled periods per call %s | throttled ms per call %s" %
              (what, th, s[0], s[len(s) // 2], s[-1], thr, ["%.0f" % u for u in usec]), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
