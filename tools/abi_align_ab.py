#!/usr/bin/env python3
"""compute_alignments(SW, affine) through the plugin ABI, 1,048,576 pairs of 150 x 500, fresh result rows (the rows of earlier
calls stay alive, as in the reference's timing loop): wall time of the call and the host's phases, three calls per process.
A/B by environment (VALIGN_HIP_COPY_WHOLE_ROWS, host_malloc_tuning through argv[2]).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import build, host, synth      # noqa: E402

R, F, n, blk = 150, 500, 1 << 20, 65536
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tuning = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r0, f0 = synth.make_pairs(blk, R, F, seed=3)
reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
keys = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, host_malloc_tuning=tuning, **keys) as k:
    k.time_calls(0, reads[:65536], refs[:65536], reps=1, align=True, free_between=False)
    total, per_call = k.time_calls(0, reads, refs, reps=4, align=True, free_between=False)
    phases = [ln for ln in k.drain_log().splitlines() if "align done" in ln]
    print(json.dumps({"whole_rows": os.environ.get("VALIGN_HIP_COPY_WHOLE_ROWS") is not None, "host_malloc_tuning": tuning,
                      "ms_per_call": [round(t * 1e3, 2) for t in per_call],
                      "host_phases_last_call": json.loads(phases[-1].split("host phases ")[-1]) if phases else None}), flush=True)
