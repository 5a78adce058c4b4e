#!/usr/bin/env python3
"""score_alignments(SW, affine) through the plugin ABI on a mixed-length batch (1,048,576 pairs of 150 x 500, prefixes of
10-100 %): ragged_batching 0 / 2 against the staging chunk size (VALIGN_HIP_DEBUG chunk_bytes).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import build, host, synth      # noqa: E402

R, F, n, blk = 150, 500, 1 << 20, 1 << 16
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r0, f0 = synth.make_ragged_pairs(blk, R, F, seed=3)
reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
keys = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
for rnd in range(2):
    for mb in (48, 96, 192, 384):
        for mode in (0, 2):
            os.environ["VALIGN_HIP_DEBUG"] = "chunk_bytes=%d" % (mb << 20)
            with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, ragged_batching=mode, **keys) as k:
                k.score_alignments(0, reads, refs, scattered=True)
                best = min(k.score_alignments(0, reads, refs, scattered=True)[1] for _ in range(6))
                phases = [ln for ln in k.drain_log().splitlines() if "score done" in ln]
                print(json.dumps({"chunk_MB": mb, "ragged_batching": mode, "ms": round(best * 1e3, 2),
                                  "host_phases": json.loads(phases[-1].split("host phases ")[-1]) if phases else None}), flush=True)
