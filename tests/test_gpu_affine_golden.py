"""libHIPKernel.so against the enumeration fixtures (tests/golden/affine): affine scores must be the
exhaustively enumerated optimum, alignments the rows whose re-scoring the generator confirmed.  Through the
plugin ABI (spawn_alignment_kernel)."""
import glob
import os

import numpy as np
import pytest

from versalignlib_amd import build, hipkernel, host

from conftest import ROOT

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "affine", "affine_enum_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_affine_equals_enumerated_optimum(path):
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    R, F = g["reads"].shape[1], g["refs"].shape[1]
    for s, sc in enumerate(g["scorings"]):
        m, x, o_r, e_r, o_f, e_f = (int(v) for v in sc)
        with host.Plugin(build.HIP_PLUGIN, R, F, score_match=m, score_mismatch=x, score_gap_read=o_r, score_gap_ref=o_f,
                         score_gap_open_read=o_r, score_gap_extend_read=e_r, score_gap_open_ref=o_f,
                         score_gap_extend_ref=e_f) as hip:
            assert np.array_equal(hip.score_alignments(host.SW, g["reads"], g["refs"]), g["sw_score_%d" % s]), (path, sc)
            assert np.array_equal(hip.score_alignments(host.NW, g["reads"], g["refs"]), g["nw_score_%d" % s]), (path, sc)
            for opt, tag in ((host.SW, "sw"), (host.NW, "nw")):
                rows, idx = hip.compute_alignments(opt, g["reads"], g["refs"], normalise=False)
                assert np.array_equal(idx, g["idx_%s_%d" % (tag, s)]), (path, sc, tag)
                assert np.array_equal(rows, g["rows_%s_%d" % (tag, s)]), (path, sc, tag)


def test_extension_dearer_than_opening_is_refused():
    with pytest.raises(host.PluginError, match="extend >= open"):
        host.Plugin(build.HIP_PLUGIN, 5, 8, score_gap_open_read=-1, score_gap_extend_read=-3,
                    score_gap_open_ref=-2, score_gap_extend_ref=-2)
    with pytest.raises(hipkernel.HipKernelError, match="extend >= open"):
        hipkernel.Engine(5, 8, hipkernel.Scoring.make(1, -1, -1, -2, -1, -3, -2, -2))
