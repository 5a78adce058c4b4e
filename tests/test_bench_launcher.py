"""`python bench.py --gpus N` from a plain shell (what the round-end driver may run): the parent starts its
own ranks as child processes, relays rank 0's single JSON line and the children's exit code.  Exercised here
with two gloo ranks on the CPU in the launcher's self-test mode, which computes no alignment (there is no CPU
path to compute one with) -- launcher, rendezvous, all-gather, checksum and relay are the real ones."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*argv, timeout=300):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_plain_command_starts_two_ranks_and_relays_one_line():
    p = _run("--gpus", "2", "--selftest-launcher", "--steps", "2", "--pairs", "1001")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout, everything else on stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["data"] == "launcher-selftest"
    m = line["multi_gpu"]
    assert m["n_ranks_seen"] == 2 and m["gathered_pairs"] == 2002
    assert m["checksum_ok"] and m["gather_checksum"] == m["sum_of_rank_checksums"]
    # an N > 1 line carries the CPU baseline (rank 0 times the oracle on a short sample while the other ranks wait)
    # and the roofline object, with the same keys as the N = 1 line
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["unit"] == "GCUPS" and cpu["cores"] >= 1 and cpu["kind"] == "port" and "sample" in cpu
    assert cpu["simd"] is None or "error" in cpu["simd"] or (cpu["simd"]["value"] > 0 and cpu["simd"]["identical_to_scalar_oracle"])
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["peak"] == 8000.0


def test_real_line_builds_its_cpu_leg_for_every_world_size():
    """bench.py's N > 1 path must not skip `cpu_baseline` (round 3 did): the condition in main() is `not args.no_cpu`,
    with the short sample for world > 1."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "if world == 1 and not args.no_cpu" not in src
    assert "short = world > 1" in src and "seconds=1.5 if short else 12.0" in src


def test_failing_ranks_fail_the_command():
    """Without a GPU the real workload refuses to run (no CPU fallback): the launcher must hand that failure on."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU is present")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-abi")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "no CPU path" in p.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_in_plugin_shards_child_cannot_take_the_rank_down():
    """The in-plugin multi-device leg of an N > 1 run is a child process of rank 0: whatever becomes of it -- here, on a
    machine without a GPU, the plugin refuses to spawn a kernel -- the parent gets a dict to put in its line."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU is present")
    sys.path.insert(0, ROOT)
    import bench
    out = bench.in_plugin_shards_child(2, 2)
    assert isinstance(out, dict) and out.get("devices") == 2
    legs = [out.get("score_alignments_sw"), out.get("score_alignments_sw_rccl_allgather")]
    assert "error" in out or all(isinstance(x, dict) and "error" in x for x in legs), out
