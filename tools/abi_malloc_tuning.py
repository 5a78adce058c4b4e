"""compute_alignments(SW) through the plugin ABI with rows of earlier calls alive (the reference's leaking timing loop),
1M pairs of 150 x 500, for one value of the plugin key host_malloc_tuning (argv[1]: 0, 1, 2) -- and the bare 2n new[]
loop before and after the spawn (developer tool; run on the GPU box, one process per value: mallopt is sticky)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from versalignlib_amd import build, host
n = 1 << 20
reads, refs = bench.synth_on_device(n, torch.device("cuda:0"), seed=2000)
h_reads, h_refs = reads.cpu().numpy(), refs.cpu().numpy()
tune = int(sys.argv[1])
keys = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
keys["host_malloc_tuning"] = tune
floor0, _ = host.alloc_probe(n, 650, 16)
with host.Plugin(build.HIP_PLUGIN, 150, 500, num_threads=16, **keys) as k:
    floor1, _ = host.alloc_probe(n, 650, 16)
    total, per_call = k.time_calls(0, h_reads, h_refs, reps=4, align=True, free_between=False)
    print(json.dumps({"tuning": tune, "env": os.environ.get("MALLOC_TOP_PAD_"), "bare_new_before_spawn_ms": round(floor0*1e3,1), "bare_new_after_spawn_ms": round(floor1*1e3,1), "fresh_rows_ms_per_call": [round(x*1e3,1) for x in per_call]}))
