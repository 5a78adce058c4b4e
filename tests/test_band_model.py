"""The schedule of the banded block chain (versalignlib_amd/csrc/band_kernels.hip.h), stated in plain Python
(tools/band_schedule_model.py: plan, events, delay rings, masks exactly as the kernel has them), against the oracle's
block band -- on the CPU.  The GPU tests (tests/test_gpu_band.py) check the kernel; this one pins the logic the kernel was
written from: period and delay formulas, the warm-up column, padding blocks, the cyclic hand-over from the last lane of a
strip to the first lane of the next."""
import os
import sys

import numpy as np

from oracle import cpu_ref
from versalignlib_amd import synth

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import band_schedule_model as model       # noqa: E402


def test_plan_of_baseline_config5():
    """10 kbp x 10 kbp at 512 diagonals on 32 lanes x 16 rows: blocks start 17 steps apart, every hand-over is one step
    (the kernel's DPP variant), 640 blocks + 240 rows of top padding."""
    p = model.plan(10000, 10000, 256, 32, 16)
    assert (p["d"], p["P"], p["nb"], p["pad"], p["unit"]) == (17, 544, 640, 240, True)
    p = model.plan(10000, 5000, 256, 32, 16)          # slope 1/2: windows advance 8 columns per block -> the delay ring
    assert not p["unit"] and p["d"] >= 8 + 2


def test_schedule_model_matches_the_block_band():
    rng = np.random.default_rng(17)
    checked = 0
    for G, K in ((4, 2), (8, 2), (4, 4), (8, 4)):
        for _ in range(25):
            R, F, w = int(rng.integers(1, 90)), int(rng.integers(1, 120)), int(rng.integers(1, 12))
            reads, refs = synth.make_pairs(2, R, F, seed=int(rng.integers(1, 1 << 30)), sub_rate=0.1, indel_rate=0.05 if R > 8 else 0.0,
                                           n_run_frac=0.3, short_frac=0.3)
            exp = cpu_ref.score_banded_sw(reads, refs, 2 * w, threads=1, block_rows=K, col_align=1)
            for p in range(2):
                got, pl = model.model_score(bytes(reads[p]), bytes(refs[p]), w, 2, -1, 3, G, K)
                assert got == exp[p], (G, K, R, F, w, p, got, int(exp[p]), pl["d"])
                checked += 1
    assert checked == 200
