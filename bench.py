#!/usr/bin/env python3
"""bench.py -- headline benchmark of the KSS-ICP registration core on MI355X.

Metric (BASELINE.json): ICP iterations/s (+ point-pairs/s) on a 100k x 100k cloud pair, 1/2/4/8 GPUs.
Workload = config C2: one synthetic uniform-sphere pair per GPU (R_z(10 deg), 1e-3 jitter), 50 fixed
ICP iterations (PCL convergence tests off) followed by the getFitnessScore() pass, exactly as
KSSICP::shapeRegistration_ICP(int) drives PCL (KSS_ICP.hpp:133-183).  A "step" is one such registration.
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, independent pairs
(weak scaling), the (R,t) records all-gathered over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector == f32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=100000, help="points per cloud (C2: 100000)")
    ap.add_argument("--iters", type=int, default=50, help="fixed ICP iterations per registration")
    ap.add_argument("--fma", type=int, default=0, help="1: fused distance form (not bit-parity)")
    ap.add_argument("--spt", type=int, default=0, help="NN sources per thread (0 = auto)")
    ap.add_argument("--splits", type=int, default=0, help="NN target splits (0 = auto)")
    ap.add_argument("--mode", default="auto", choices=["auto", "brute", "grid"], help="NN search structure (same results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(n, iters, src, tgt):
    """The oracle's PCL-style ICP (kd-tree, 1 thread = PCL 1.8.1's serial correspondence loop) on the
    SAME workload, timed on this box's host cores.  Checker code used as the baseline, never shipped."""
    O = graft.load_oracle()
    p = O.icp_params(max_iterations=iters, fixed_iterations=1, use_kdtree=1, nthreads=1, compute_fitness=1)
    t0 = time.perf_counter()
    r = O.icp(src, tgt, p)
    dt = time.perf_counter() - t0
    out = {"value": iters / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": "the full workload once: %dx%d pair, %d fixed iterations + fitness pass, kd-tree build included "
                     "(%.3f s build, %.2f s in NN queries, %.2f s total)" % (n, n, iters, r["build_seconds"], r["nn_seconds"], dt),
           "T": [float(x) for x in r["T"].reshape(-1)]}
    ncpu = os.cpu_count() or 1
    if ncpu > 1:
        p2 = O.icp_params(max_iterations=iters, fixed_iterations=1, use_kdtree=1, nthreads=min(ncpu, 64), compute_fitness=1)
        t0 = time.perf_counter()
        O.icp(src, tgt, p2)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": iters / dt2, "cores": min(ncpu, 64), "note": "OpenMP over queries; not what PCL 1.8.1 does"}
    return out


def read_traffic():
    """HBM bytes per nn_sweep launch from the committed PMC summary (collected in separate rocprofv3
    --pmc passes and corrected per MI355X_MICROARCH.md HBM section), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as f:
            return json.load(f).get("nn_sweep", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    a = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the KSS-ICP core has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    pkg = graft.load_package()
    S = pkg.synth
    # one independent pair per rank (pair_id = rank): the path shards over pairs with no data exchange
    src, tgt = S.make_pair(rank, a.n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
    d_src = torch.from_numpy(src).to(dev)
    d_tgt = torch.from_numpy(tgt).to(dev)
    torch.cuda.synchronize()

    ctx = pkg.Context(local_rank)       # raises without libkssicp.so / GPU
    params = ctx.icp_params(max_iterations=a.iters, fixed_iterations=1, compute_fitness=1, nn_fma=a.fma,
                            nn_sources_per_thread=a.spt, nn_target_splits=a.splits,
                            nn_mode={"auto": pkg.NN_AUTO, "brute": pkg.NN_BRUTE, "grid": pkg.NN_GRID}[a.mode])
    RecArr = pkg.IcpResult * 1

    def step():
        res = ctx.icp_dev(d_src.data_ptr(), a.n, d_tgt.data_ptr(), a.n, params)
        if world > 1:       # final gather of the (R,t) records over RCCL/xGMI: one 96-B record per pair
            local = pkg.shard.records_to_array(RecArr(res), rank)
            pkg.shard.gather_records(local, world, world, rank, device=dev)
        return res

    for _ in range(a.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nn_ms, nn_launches = ctx.profile_get(pkg.K_NN_SWEEP)
    red_ms, red_launches = ctx.profile_get(pkg.K_CORR_REDUCE)
    grid_ms, grid_launches = ctx.profile_get(pkg.K_GRID_NN)
    build_ms, build_launches = ctx.profile_get(pkg.K_GRID_BUILD)
    ctx.profile_enable(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        iters_total = a.steps * a.iters * world
        value = iters_total / dt
        nn_avg_s = (nn_ms / max(1, nn_launches)) * 1e-3
        flops_per_launch = 8.0 * a.n * a.n                      # 8 flop per (source, target) pair
        achieved_tf = flops_per_launch / nn_avg_s / 1e12 if nn_avg_s > 0 else 0.0
        # algorithmic bytes of one sweep: targets re-streamed once per source block + sources + keys
        spt = params.nn_sources_per_thread if params.nn_sources_per_thread else 4
        src_blocks = -(-a.n // (256 * spt))
        alg_bytes = 16.0 * a.n * src_blocks + 16.0 * a.n * 2 + 8.0 * a.n
        out = {
            "metric": "icp_iterations_per_sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: one %dx%d uniform-sphere pair per GPU, %d fixed ICP iterations + fitness pass "
                                   "(brute-force exact NN + cov reduce), R_z(10deg), jitter 1e-3" % (a.n, a.n, a.iters),
                       "n_src": a.n, "n_tgt": a.n, "icp_iters_per_step": a.iters, "pairs_per_gpu": 1,
                       "nn_arithmetic": "fma" if a.fma else "reference (no fma)"},
            "point_pairs_evaluated_per_sec": float(a.n) * a.n * (a.iters + 1) * a.steps * world / dt,
            "correspondences_per_sec": float(a.n) * (a.iters + 1) * a.steps * world / dt,
            "roofline": {"kernel": "nn_sweep_kernel", "bound": "valu-fp32",
                         "bound_note": "min-reduction on the FP32 vector ALU; peak = 157.3 TFLOP/s, numerically the "
                                       "f32-input MFMA peak; MFMA is not used (north star)",
                         "achieved": achieved_tf, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP32_VALU_PEAK_TFLOPS, "traffic": read_traffic(),
                         "avg_launch_ms": nn_avg_s * 1e3, "launches": nn_launches,
                         "flops_per_launch": flops_per_launch,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "achieved": alg_bytes / nn_avg_s / 1e9 if nn_avg_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": (alg_bytes / nn_avg_s / 1e9 / HBM_PEAK_GBS) if nn_avg_s > 0 else 0.0}},
            "corr_reduce": {"avg_launch_ms": red_ms / max(1, red_launches), "launches": red_launches},
            "grid_nn": {"avg_launch_ms": grid_ms / max(1, grid_launches), "launches": grid_launches,
                        "build_avg_ms": build_ms / max(1, build_launches), "builds": build_launches},
            "nn_mode": a.mode,
            "result": {"iterations": int(last.iterations), "fitness": float(last.fitness)},
        }
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(a.n, a.iters, src, tgt)
            Tg = np.array(last.T, dtype=np.float64)
            out["parity_vs_cpu_baseline_max_abs_T"] = float(np.abs(Tg - np.array(cb.pop("T"))).max())
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1core"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
