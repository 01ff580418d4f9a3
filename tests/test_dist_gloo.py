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
    # full-registration records (kss_register_result: s, R, t, ...) travel through the same gather
    regs = (pkg.binding.RegisterResult * (hi - lo))()
    for i in range(hi - lo):
        regs[i].scale = 1.0 + (lo + i); regs[i].t[2] = float(lo + i); regs[i].icp_iterations = lo + i
    allg = pkg.shard.gather_records(pkg.shard.records_to_array(regs, lo), npairs, world, rank, record_bytes=C.sizeof(pkg.binding.RegisterResult))
    back = pkg.shard.array_to_records(allg, pkg.binding.RegisterResult)
    ok = ok and len(back) == npairs and all(b.scale == 1.0 + g and b.t[2] == float(g) and b.icp_iterations == g for g, b in enumerate(back))
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


def _sums_numpy(moved, tgt, idx, d2, max_d2=1.0):
    """The KSS_NSUMS correspondence sums of one rank's rows (layout of include/kssicp.h), in f64."""
    s = np.zeros(20)
    keep = ~(d2.astype(np.float64) > max_d2)
    p, q = moved[keep].astype(np.float64), tgt[idx[keep]].astype(np.float64)
    s[0] = keep.sum(); s[1:4] = p.sum(0); s[4:7] = q.sum(0); s[7:16] = (p[:, :, None] * q[:, None, :]).sum(0).reshape(-1)
    s[16] = d2[keep].astype(np.float64).sum(); s[17] = d2.astype(np.float64).sum(); s[18] = np.sqrt(d2.astype(np.float64)).sum()
    return s


def _split_worker(rank, world, port, iters, ret):
    """Source rows split over ranks, target replicated, ONE all-reduce of the sums per iteration through the
    kss_allreduce_fn callback (gloo here, RCCL on GPUs).  The per-rank NN pass is done by the oracle (test stand-in for
    the HIP pass); what is tested is the exchange step: same T on every rank, equal to the unsharded registration."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package(); O = graft.load_oracle(); S = pkg.synth
    src, tgt = S.make_pair(7, 3000, R=S.rot_axis_angle([0.1, 0.3, 1.0], np.deg2rad(8.0)), t=(0.01, 0.0, -0.02), shape="bumpy")
    lo, hi = pkg.shard.shard_range(len(src), world, rank)
    rows = src[lo:hi]
    cb = pkg.shard.make_allreduce(pkg.binding)
    fin = np.eye(4, dtype=np.float32)
    cur = rows.copy()
    for _ in range(iters):
        idx, d2 = O.nn_brute(cur, tgt)
        sums = _sums_numpy(cur, tgt, idx, d2)
        buf = (C.c_double * 20)(*sums)
        assert cb(None, buf, 20) == 0
        Tk = O.rigid_from_sums(np.array(buf[:]))
        fin = O.mat4_mul(Tk, fin)
        cur = O.transform_points_f32(Tk, cur)                 # PCL transforms the cloud incrementally
    allT = [torch.zeros(16, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(allT, torch.from_numpy(np.ascontiguousarray(fin, dtype=np.float32).reshape(-1)))
    same = all(torch.equal(allT[0], t) for t in allT)
    ref = O.icp(src, tgt, O.icp_params(max_iterations=iters, fixed_iterations=1))
    err = float(np.abs(ref["T"] - fin).max())
    ret[rank] = (bool(same), err)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_source_exchange_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_split_worker, args=(world, _free_port(), 6, ret), nprocs=world, join=True)
    for r in range(world):
        same, err = ret[r]
        assert same and err < 2e-6, dict(ret)      # f64 sums grouped per rank instead of serially: float round-off of T


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


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` with no rank environment must RUN two ranks (the launcher spawns them before anything
    touches torch / HIP) and print ONE JSON line with n_gpus = 2.  --dry-run-dist swaps the registration for a stand-in
    so that the protocol (gloo group, barriers, MAX over ranks, padded record gather) runs on CPU."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run-dist", "--steps", "3", "--warmup", "1", "--c5-pairs", "96"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    # the C5 leg's protocol (every rank's shard, ONE gather, records in global pair order) with stand-in records
    c5 = d["secondary"]["c5"]
    assert c5["n_gpus"] == 2 and c5["pairs_total"] == 192 and c5["gathered_pair_ids_in_order"] is True
    assert d["records_gathered_ok"] is True and d["scaling"] == "weak" and d["metric"] == "icp_iterations_per_sec"
    # the time is the slowest rank's: rank 1 sleeps 4 ms per step
    assert d["ms_per_step"] >= 4.0
    assert abs(d["value"] - 3 * 50 * 2 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_rejects_a_rank_count_that_disagrees_with_the_environment():
    """Under an existing rank environment (torch.distributed.run) --gpus must match WORLD_SIZE: a mismatch is an error,
    never a silently smaller job."""
    r = _run_bench(["--gpus", "4", "--dry-run-dist", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
    r = _run_bench(["--gpus", "1", "--dry-run-dist", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout
