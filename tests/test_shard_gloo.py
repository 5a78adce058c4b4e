"""The N > 1 path on CPU: two gloo ranks shard a batch, score their ranges (the oracle
stands in for the device here -- test infrastructure only) and all-gather the scores."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu_ref
    from versalignlib_amd import shard, synth
    reads, refs = synth.make_pairs(n, 24, 40, seed=77, n_run_frac=0.05, short_frac=0.1)

    def score_fn(rd, rf):
        return torch.from_numpy(cpu_ref.score(0, rd.numpy(), rf.numpy()))

    full = shard.score_sharded(score_fn, torch.from_numpy(reads), torch.from_numpy(refs))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(n, world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import cpu_ref
    from versalignlib_amd import synth
    reads, refs = synth.make_pairs(n, 24, 40, seed=77, n_run_frac=0.05, short_frac=0.1)
    exp = cpu_ref.score(0, reads, refs)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.dtype == np.int16 and np.array_equal(got, exp)


def test_two_ranks_even(tmp_path):
    _run(64, 2, tmp_path)


def test_two_ranks_ragged_tail(tmp_path):
    _run(51, 2, tmp_path)      # shards of 26 and 25: padded for the collective, trimmed after
