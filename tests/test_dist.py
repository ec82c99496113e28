"""Multi-process (gloo, world_size 2, CPU) tests of the N>1 path: sharding of images and
(image, class) problems with no data-path collective, and the flat-bucket gradient all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from simpledet_amd import dist as sdd
    r, _, w = sdd.init("gloo")
    assert (r, w) == (rank, world)
    # 1. images shard contiguously, every image exactly once
    lo, hi = sdd.shard_range(5, rank, world)
    mine = list(range(lo, hi))
    allr = sdd.gather_ragged(mine)
    # 2. gradient buckets: rank-dependent grads, averaged
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(300,), (17, 5), (1000, 3), (2,)]]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    red = sdd.GradBucketReducer(params, bucket_mb=0.004)  # forces several buckets
    red.reduce()
    ok = all(torch.allclose(p.grad, torch.full_like(p, (i + 1) * (world + 1) / 2.0))
             for i, p in enumerate(params))
    # 3. (image, class) soft-NMS problems round robin; results gathered ragged
    probs = sdd.shard_round_robin(7, rank, world)
    res = sdd.gather_ragged([(p, p * p) for p in probs])
    if rank == 0:
        q.put((allr, len(red.buckets), ok, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, nb, ok, res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(sum(allr, [])) == list(range(5))
    assert nb >= 3 and ok
    flat = sorted(sum(res, []))
    assert flat == [(p, p * p) for p in range(7)]


def test_shard_helpers_cover_everything_once():
    from simpledet_amd import dist as sdd
    for n in (0, 1, 7, 16):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                lo, hi = sdd.shard_range(n, r, w)
                got += list(range(lo, hi))
            assert got == list(range(n))
            rr = sorted(sum([sdd.shard_round_robin(n, r, w) for r in range(w)], []))
            assert rr == list(range(n))
