"""One shard of a reference CPU kernel run (child process of bench.py's `cpu_reference` leg).

The reference's AVX2 kernel has its OpenMP loop compiled out on Linux (AVXKernel.cpp:74-76), so one
process scores with one thread.  What its authors intended -- all cores -- is reproduced by sharding
the pairs over processes, each loading the kernel through the plugin protocol (libvalignhost.so,
plain ctypes: no torch, no GPU).  Usage:
    ref_shard_worker.py <libvalignhost.so> <plugin.so> <reads.npy> <refs.npy> <begin> <end> <dir> <id>
Writes <dir>/ready.<id>, waits for <dir>/go, scores its shard once, prints the seconds it took."""
import ctypes
import os
import sys
import time

import numpy as np


def main():
    host_lib, plugin, reads_path, refs_path, begin, end, sync_dir, ident = sys.argv[1:9]
    begin, end = int(begin), int(end)
    reads = np.ascontiguousarray(np.load(reads_path, mmap_mode="r")[begin:end])
    refs = np.ascontiguousarray(np.load(refs_path, mmap_mode="r")[begin:end])
    n, R = reads.shape
    F = refs.shape[1]
    L = ctypes.CDLL(host_lib)
    L.vh_open.restype = ctypes.c_void_p
    L.vh_open.argtypes = [ctypes.c_char_p]
    L.vh_set_param.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.vh_spawn.argtypes = [ctypes.c_void_p]
    L.vh_score.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.vh_close.argtypes = [ctypes.c_void_p]
    p = L.vh_open(os.fsencode(plugin))
    if not p:
        raise SystemExit("cannot open " + plugin)
    for key, val in ((b"read_length", R), (b"ref_length", F), (b"num_threads", 1)):
        L.vh_set_param(p, key, val)
    if L.vh_spawn(p) != 0:
        raise SystemExit("spawn failed")
    scores = np.zeros(n, dtype=np.int16)
    warm = min(n, 256)
    L.vh_score(p, 0, warm, reads.ctypes.data, refs.ctypes.data, scores.ctypes.data)
    open(os.path.join(sync_dir, "ready." + ident), "w").close()
    while not os.path.exists(os.path.join(sync_dir, "go")):
        time.sleep(0.002)
    t0 = time.perf_counter()
    if L.vh_score(p, 0, n, reads.ctypes.data, refs.ctypes.data, scores.ctypes.data) != 0:
        raise SystemExit("score failed")
    sec = time.perf_counter() - t0
    L.vh_close(p)
    print("%.6f %d" % (sec, int(scores.astype(np.int64).sum())))


if __name__ == "__main__":
    main()
