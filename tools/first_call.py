"""Time from a cold process to the first result through the plugin ABI: dlopen + set_parameters, spawn_alignment_kernel,
first score_alignments / compute_alignments call (code-object load, staging allocation), second call
(developer tool; run on the GPU box)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from versalignlib_amd import build, host, synth          # noqa: E402

t1 = time.perf_counter()
R, F, n = 150, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reads, refs = synth.make_pairs(n, R, F, seed=3)
t2 = time.perf_counter()
k = host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4)
k.__enter__()
t3 = time.perf_counter()
k.score_alignments(0, reads, refs)
t4 = time.perf_counter()
k.score_alignments(0, reads, refs)
t5 = time.perf_counter()
k.compute_alignments(0, reads, refs)
t6 = time.perf_counter()
k.compute_alignments(0, reads, refs)
t7 = time.perf_counter()
k.__exit__(None, None, None)
# a second kernel object in the same (now warm) process: what a host that spawns per batch pays (src/impl/main.cpp:261-265)
t8 = time.perf_counter()
k2 = host.Plugin(build.HIP_PLUGIN, R, F, num_threads=4)
t9 = time.perf_counter()
k2.score_alignments(0, reads, refs)
t10 = time.perf_counter()
k2.close()
t11 = time.perf_counter()
print(json.dumps({"pairs": n, "import_ms": round((t1 - t0) * 1e3, 1), "dlopen_spawn_ms": round((t3 - t2) * 1e3, 1),
                  "first_score_ms": round((t4 - t3) * 1e3, 1), "second_score_ms": round((t5 - t4) * 1e3, 2),
                  "first_align_ms": round((t6 - t5) * 1e3, 1), "second_align_ms": round((t7 - t6) * 1e3, 2),
                  "respawn_ms": round((t9 - t8) * 1e3, 2), "respawn_first_score_ms": round((t10 - t9) * 1e3, 2),
                  "delete_ms": round((t11 - t10) * 1e3, 2)}))
