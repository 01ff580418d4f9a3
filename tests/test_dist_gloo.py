"""CPU tests of the N > 1 path: world_size-2 (and 3) gloo process groups exercise the pair sharding and the
record gather that bench.py and the batch driver use on GPUs over RCCL.  No GPU compute here: the records
are synthetic."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as graft


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, npairs, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    lo, hi = pkg.shard.shard_range(npairs, world, rank)
    recs = (pkg.IcpResult * (hi - lo))()
    for i in range(hi - lo):
        g = lo + i
        for k in range(16):
            recs[i].T[k] = float(g * 100 + k)
        recs[i].fitness = g * 0.5
        recs[i].iterations = g + 1
        recs[i].converged = g % 2
    local = pkg.shard.records_to_array(recs, lo)
    allr = pkg.shard.gather_records(local, npairs, world, rank)
    out = pkg.shard.array_to_records(allr, pkg.IcpResult)
    ok = len(out) == npairs
    for g, r in enumerate(out):
        ok = ok and r.pair_id == g and r.iterations == g + 1 and r.fitness == g * 0.5 and r.T[5] == float(g * 100 + 5)
    # timing protocol of bench.py: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == float(world)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,npairs", [(2, 8), (2, 5), (3, 7)])
def test_shard_and_gather_gloo(world, npairs):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), npairs, ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)


def test_shard_range_partitions():
    pkg = graft.load_package()
    for n in (1, 7, 8, 8192):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = pkg.shard.shard_range(n, w, r)
                assert 0 <= lo <= hi <= n and hi - lo <= pkg.shard.max_shard(n, w)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert C.sizeof(pkg.IcpResult) == pkg.shard.RECORD_BYTES
