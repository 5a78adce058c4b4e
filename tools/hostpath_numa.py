#!/usr/bin/env python3
"""Is the run-to-run spread of the host-pointer path NUMA placement?  For every NUMA node of the box: confine this process
(and so the plugin's worker threads, which inherit the mask) to the node's CPUs, allocate the sequences there (first touch),
time valign_hip_score_host over 1,048,576 pairs of 150 x 500; then the same with the whole mask.  Run on the GPU box."""
import glob
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import hipkernel, synth      # noqa: E402
from tools.hostpath_sweep import engine             # noqa: E402


def cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def main():
    R, F, n, blk = 150, 500, 1 << 20, 65536
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    whole = os.sched_getaffinity(0)
    nodes = {}
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist")):
        k = int(re.search(r"node(\d+)", path).group(1))
        cpus = cpulist(open(path).read()) & whole
        if cpus:
            nodes[k] = cpus
    print("nodes:", {k: len(v) for k, v in nodes.items()}, "mask", len(whole), flush=True)
    try:
        for dev in glob.glob("/sys/class/drm/card*/device/numa_node"):
            print(dev, open(dev).read().strip())
    except OSError:
        pass
    r0, f0 = synth.make_pairs(blk, R, F, seed=3)
    cases = [("node %d" % k, v, v) for k, v in nodes.items()]
    for k, v in nodes.items():
        cases.append(("data on node %d, threads anywhere" % k, v, whole))
    for name, data_mask, run_mask in cases:
        os.sched_setaffinity(0, data_mask)
        reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))          # first touch under data_mask
        os.sched_setaffinity(0, run_mask)
        for rep in range(3):                                                          # three engines: three sets of worker threads
            eng = engine({}, R, F)
            eng.score_host(0, reads, refs, threads=threads)
            t = []
            for _ in range(8):
                t0 = time.perf_counter()
                eng.score_host(0, reads, refs, threads=threads)
                t.append((time.perf_counter() - t0) * 1e3)
            d = eng.describe(0, n)
            print("%-36s engine %d: min %6.2f median %6.2f ms  gather %.2f wait %.2f" %
                  (name, rep, min(t), sorted(t)[4], d["host_gather_ms"], d["host_wait_ms"]), flush=True)
            eng.close()
        del reads, refs
    os.sched_setaffinity(0, whole)


if __name__ == "__main__":
    main()
