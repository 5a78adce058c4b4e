"""valign-bench, the command-line counterpart of the reference host program: same output
files and line formats as src/impl/main.cpp:133-189, for any plugin given by path."""
import os
import subprocess

import numpy as np
import pytest

from oracle import cpu_ref
from versalignlib_amd import build, host, synth

from conftest import ref_kernel


def _write_fasta(path, seqs, width=60):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">seq%d\n" % i)
            for k in range(0, len(s), width):
                f.write(s[k:k + width].decode() + "\n")


def _batch(tmp_path, n=40, R=48, F=90, seed=5):
    reads, refs = synth.make_pairs(n, R, F, seed=seed, indel_rate=0.03, n_run_frac=0.1, short_frac=0.0)
    # ragged lengths: trim some sequences so that pad() has work to do
    # (indels leave NUL padding at the end of a read: FASTA text cannot carry it, strip it)
    rd = [bytes(r[:R - (i % 7)]).rstrip(b"\0") for i, r in enumerate(reads)]
    rf = [bytes(r[:F - (i % 5)]).rstrip(b"\0") for i, r in enumerate(refs)]
    rd[0] = rd[0].ljust(R, b"A")                             # keep the maxima
    rf[0] = rf[0].ljust(F, b"C")
    _write_fasta(tmp_path / "reads.fa", rd)
    _write_fasta(tmp_path / "refs.fa", rf)
    return host.pad(rd), host.pad(rf)


def _expected_text(opt, reads, refs):
    scores = cpu_ref.score(opt, reads, refs)
    rows, idx = cpu_ref.align(opt, reads, refs)
    aln = []
    for i in range(reads.shape[0]):
        s = idx[i, 0]
        aln.append(bytes(rows[i, 0, s:]).split(b"\0")[0].decode() + "\n" +
                   bytes(rows[i, 1, s:]).split(b"\0")[0].decode() + "\n\n")
    names = [bytes(r).split(b"\0")[0].decode() for r in reads]
    return scores, names, "".join(aln)


def _run_cli(kernel, tmp_path, extra=()):
    out = tmp_path / "out"
    out.mkdir(exist_ok=True)
    cmd = [build.BENCH_CLI, "--kernel", kernel, "--reads", str(tmp_path / "reads.fa"), "--refs",
           str(tmp_path / "refs.fa"), "--out-dir", str(out), "--threads", "2", "--ladder", "1,2", "--loops", "2"]
    res = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return out, res.stdout


def _check_outputs(out, reads, refs, full_scores):
    for opt, tag in ((0, "smith_waterman"), (1, "needleman_wunsch")):
        scores, names, aln = _expected_text(opt, reads, refs)
        assert (out / ("alignments_%s.txt" % tag)).read_text() == aln
        lines = (out / ("scores_%s.txt" % tag)).read_text().splitlines()
        assert len(lines) == len(names)
        for line, name, sc in zip(lines, names, scores):
            got_name, got_score = line.split("\t")
            assert got_name == name
            assert int(got_score) == (int(sc) if full_scores else int(sc) & 0xFF)


def test_cli_with_reference_default_kernel(tmp_path):
    default = ref_kernel("Default")
    if not default:
        pytest.skip("oracle/_ref not built")
    reads, refs = _batch(tmp_path)
    out, stdout = _run_cli(default, tmp_path, extra=("--cigar",))
    _check_outputs(out, reads, refs, full_scores=False)
    for opt, tag in ((0, "smith_waterman"), (1, "needleman_wunsch")):      # --cigar: one more file per mode
        rows, idx = cpu_ref.align(opt, reads, refs)
        lines = (out / ("cigars_%s.txt" % tag)).read_text().splitlines()
        assert [ln.split("\t")[1] if "\t" in ln else "" for ln in lines] == \
            [c for c in host.cigars(rows, idx, extended=True)]
    table = stdout.splitlines()
    assert table[0] == "Threads\t1\t2" and table[1].startswith(default) and len(table[1].split("\t")) == 3


def test_cli_rejects_unequal_sets(tmp_path):
    _batch(tmp_path)
    with open(tmp_path / "refs.fa", "a") as f:
        f.write(">extra\nACGT\n")
    res = subprocess.run([build.BENCH_CLI, "--kernel", "/nonexistent.so", "--reads", str(tmp_path / "reads.fa"),
                          "--refs", str(tmp_path / "refs.fa")], stderr=subprocess.PIPE, text=True)
    assert res.returncode != 0 and "Unequal sizes of reads and ref set (40 vs 41)" in res.stderr


@pytest.mark.gpu
def test_cli_with_hip_kernel(tmp_path):
    reads, refs = _batch(tmp_path, n=300, R=150, F=500, seed=9)
    out, stdout = _run_cli(build.HIP_PLUGIN, tmp_path)
    _check_outputs(out, reads, refs, full_scores=True)
    assert stdout.splitlines()[0] == "Threads\t1\t2"


def _ref_host():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_host")
    return path if os.path.exists(path) else None


def _run_ref_host(kernel, tmp_path):
    out = tmp_path / "refout"
    out.mkdir(exist_ok=True)
    # (the reference's own binaries are not ours to sanitize: under tools/sanitize.sh's preloaded ASan runtime the
    # reference host's strlen() over a Default-kernel SW row -- which carries no NUL, DefaultKernel.cpp:441-451 vs
    # :484-485 -- is reported as the heap overflow it is)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    res = subprocess.run([_ref_host(), kernel, str(tmp_path / "reads.fa"), str(tmp_path / "refs.fa"), str(out), "2"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    return out


def test_reference_host_code_drives_the_reference_kernel(tmp_path):
    """oracle/_ref/ref_host is built from the reference's OWN host code (its headers, DLL_init, pad,
    FastaProvider); sanity: with the reference Default kernel it writes what the oracle predicts."""
    default = ref_kernel("Default")
    if not default or not _ref_host():
        pytest.skip("oracle/_ref not built")
    reads, refs = _batch(tmp_path)
    _check_outputs(_run_ref_host(default, tmp_path), reads, refs, full_scores=False)


@pytest.mark.gpu
def test_reference_host_code_drives_the_hip_plugin(tmp_path):
    """The drop-in claim, end to end: a host compiled against the reference's headers (not this
    repo's restatement of them) dlopen()s libHIPKernel.so through the reference's DLL_init and gets
    the oracle's scores and alignments."""
    if not _ref_host():
        pytest.skip("oracle/_ref not built")
    reads, refs = _batch(tmp_path, n=200, R=150, F=500, seed=19)
    _check_outputs(_run_ref_host(build.HIP_PLUGIN, tmp_path), reads, refs, full_scores=True)
