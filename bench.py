#!/usr/bin/env python3
"""bench.py -- headline benchmark of the KSS-ICP registration core on MI355X.

Metric (BASELINE.json): ICP iterations/s (+ point-pairs/s) on a 100k x 100k cloud pair, 1/2/4/8 GPUs.
Workload = config C2: one synthetic uniform-sphere pair per GPU (R_z(10 deg), 1e-3 jitter), 50 fixed
ICP iterations (PCL convergence tests off) followed by the getFitnessScore() pass, exactly as
KSSICP::shapeRegistration_ICP(int) drives PCL (KSS_ICP.hpp:133-183).  A "step" is one such registration,
including the per-target setup (packing, cell-list build).  Inputs are resident in HBM before the timed
region.  N > 1: one process per GPU, independent pairs (weak scaling), the 96-byte (R,t) records
all-gathered over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: N child processes (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set), spawned BEFORE this process imports torch or touches a GPU; rank 0's JSON
line is the output.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks exist
already and each process is one of them; a --gpus that disagrees with WORLD_SIZE is an error.

Two NN engines produce bit-identical correspondences (tests/test_gpu_grid.py):
  * default (--mode auto/grid): exact uniform cell list + brute-force fallback  -> `value`, `roofline`
  * --mode brute: the north star's LDS-tiled source x target sweep              -> `brute_force` object
    (always measured too, on a few steps, because it is the kernel graded against the FP32 VALU roofline)
`secondary` (N = 1): GPU-only legs of the other BASELINE configs -- C3 (1024 x 10k x 10k batch), C4 (1M x 1M: pre-shape,
one NN pass, 10 iterations), the pre-shape kernels streaming 64M points -- each with its kernel's launch time -- and
`register`: the reference's one live entry point (KSSICP_Registration on <= 2000-point samples) with the oracle's one-core
time beside it.  N > 1: `secondary.c5` -- every rank registers its 1024-pair shard in one batch call, ONE all-gather of the
records (BASELINE config 5; KSS_BENCH_FORCE_DIST=1 runs the same code on one rank over a real RCCL group).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector == f32-input MFMA peak
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a streaming copy reaches)
L2_PEAK_GBS = 34500.0             # MI355X_MICROARCH.md: L2, all eight XCDs


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed registrations (one step = 50 ICP iterations + fitness pass, ~1 ms)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=100000, help="points per cloud (C2: 100000)")
    ap.add_argument("--iters", type=int, default=50, help="fixed ICP iterations per registration")
    ap.add_argument("--fma", type=int, default=0, help="1: fused distance form (not bit-parity)")
    ap.add_argument("--spt", type=int, default=0, help="brute force: sources per thread (0 = auto)")
    ap.add_argument("--splits", type=int, default=0, help="brute force: target splits (0 = auto)")
    ap.add_argument("--mode", default="auto", choices=["auto", "brute", "grid"], help="NN engine of the timed path")
    ap.add_argument("--brute-steps", type=int, default=3, help="steps of the secondary brute-force measurement (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 / streaming / register / C5 legs")
    ap.add_argument("--legs", default="c3,c4,stream,register,c5", help="secondary legs to run (comma separated; c5 needs N > 1 or KSS_BENCH_FORCE_DIST=1)")
    ap.add_argument("--c5-pairs", type=int, default=1024, help="pairs per rank of the C5 leg (N > 1 ranks: 1024 = BASELINE config 5 at N = 8)")
    ap.add_argument("--split-source", action="store_true",
                    help="N > 1: ONE registration, its source rows split over the ranks (target replicated), one RCCL "
                         "all-reduce of the 20 sums per iteration (SURVEY 8e alternative; strong scaling).  Default: one "
                         "independent pair per rank (weak scaling, no data-path collective)")
    ap.add_argument("--prof-stride", type=int, default=53,
                    help="HIP-event timing of every n-th kernel launch inside the timed region (1 = every launch).  A bracketed "
                         "launch cannot be pre-enqueued behind the running one and costs the iteration ~20 us instead of ~14 "
                         "(rocprof trace, DESIGN.md section 4), so the default samples about one launch per registration: 53 is "
                         "coprime with the 51 launches of a step, the samples walk through all iteration positions")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="CPU rehearsal of the N-rank protocol (launcher, gloo process group, barriers, max-over-ranks timing, "
                         "record gather, one JSON line): the step is a stand-in, nothing is measured")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without an existing rank environment
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(a):
    """Start a.gpus child ranks of this script and relay rank 0's line.  Runs before anything imports torch / HIP."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def cpu_baseline(n, iters, src, tgt):
    """The oracle's PCL-style ICP (kd-tree, 1 thread = PCL 1.8.1's serial correspondence loop) on the
    SAME workload, timed on this box's host cores.  Checker code used as the baseline, never shipped.
    Built here, on the box, with the flags BASELINE.md section 3 states (-O3 -march=native); the portable build the tests
    use (-O2 -march=x86-64-v2) is the fall-back if that compile fails."""
    import __graft_entry__ as graft
    flags = "portable test build (oracle/Makefile default)"
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["KSS_ORACLE_SO"] = os.path.join(ROOT, "oracle", "libkss_oracle_native.so")
        flags = open(os.path.join(ROOT, "oracle", "libkss_oracle_native.flags")).read().strip()
    except Exception:
        pass
    O = graft.load_oracle()
    p = O.icp_params(max_iterations=iters, fixed_iterations=1, use_kdtree=1, nthreads=1, compute_fitness=1)
    reps, total, r = 0, 0.0, None
    while reps < 12 and (total < 10.0 or reps < 2):   # ~10 s of single-core work: the registration repeated
        t0 = time.perf_counter()
        r = O.icp(src, tgt, p)
        total += time.perf_counter() - t0
        reps += 1
    dt = total / reps
    out = {"value": iters / dt, "unit": "iterations/s", "cores": 1, "kind": "port", "build": flags,
           "sample": "the full workload (%dx%d pair, %d fixed iterations + fitness pass, kd-tree build included) run %d times, "
                     "%.1f s of CPU work in all; last run: %.3f s build, %.2f s in NN queries, %.2f s total"
                     % (n, n, iters, reps, total, r["build_seconds"], r["nn_seconds"], dt),
           "T": [float(x) for x in r["T"].reshape(-1)]}
    ncpu = os.cpu_count() or 1
    if ncpu > 1:
        nth = min(ncpu, 64)
        p2 = O.icp_params(max_iterations=iters, fixed_iterations=1, use_kdtree=1, nthreads=nth, compute_fitness=1)
        t0 = time.perf_counter()
        O.icp(src, tgt, p2)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": iters / dt2, "cores": nth, "note": "OpenMP over queries; not what PCL 1.8.1 does"}
    return out


def read_traffic(key):
    """HBM bytes per launch of a kernel from the committed PMC summary (separate rocprofv3 --pmc passes,
    corrected per MI355X_MICROARCH.md HBM section; tools/parse_prof.py), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def run_steps(step, n_steps, dist, world, sync, finish=None):
    """K steps between barrier + synchronize brackets; `finish` (the batch's one result gather) runs INSIDE the bracket."""
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(n_steps):
        last = step()
    if finish is not None:
        finish()
    if world > 1:
        dist.barrier()
    sync()
    return time.perf_counter() - t0, last


def dry_run(a, world, rank):
    """The N-rank protocol on CPU (gloo): same barriers, MAX over ranks, record gather and single JSON line as the GPU
    path, around a stand-in step.  tests/test_dist_gloo.py drives it with --gpus 2."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    if world > 1:
        dist.init_process_group(backend="gloo")
    done = []

    def step():
        r = pkg.IcpResult()
        r.iterations = a.iters
        r.fitness = float(rank)
        for k in range(16):
            r.T[k] = float(rank * 16 + k)
        time.sleep(0.002 * (rank + 1))      # uneven ranks: the reported time must be the slowest rank's
        done.append(r)
        return r

    gathered = []

    def finish():
        if world > 1 and done:
            recs = (pkg.IcpResult * len(done))(*done)
            local = pkg.shard.records_to_array(recs, rank * len(done))
            gathered.append(pkg.shard.gather_records(local, world * len(done), world, rank))
        done.clear()

    for _ in range(a.warmup):
        step()
    finish()
    dt, last = run_steps(step, a.steps, dist, world, lambda: None, finish)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the C5 leg's protocol with stand-in records: every rank's shard of a.c5_pairs records, ONE gather, ids in global order
    c5 = None
    if world > 1 and "c5" in a.legs.split(","):
        shard = (pkg.IcpResult * a.c5_pairs)()
        for i in range(a.c5_pairs):
            shard[i].fitness = float(rank * a.c5_pairs + i)
            shard[i].iterations = 20
        dist.barrier()
        t0 = time.perf_counter()
        local = pkg.shard.records_to_array(shard, rank * a.c5_pairs)
        allrec = pkg.shard.gather_records(local, world * a.c5_pairs, world, rank)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        recs = pkg.shard.array_to_records(allrec, pkg.IcpResult)
        c5 = {"workload": "DRY RUN of the C5 leg's protocol (stand-in records)", "n_gpus": world, "pairs_total": world * a.c5_pairs,
              "gather_us_max_over_ranks": float(t.item()) * 1e6,
              "gathered_pair_ids_in_order": bool(len(recs) == world * a.c5_pairs and all(r.pair_id == i and r.fitness == float(i) for i, r in enumerate(recs)))}
    if rank == 0:
        ok = True
        if world > 1:
            allr = pkg.shard.array_to_records(gathered[-1], pkg.IcpResult)
            ok = len(allr) == world * a.steps and all(r.pair_id == i for i, r in enumerate(allr)) and allr[-1].fitness == float(world - 1)
        print(json.dumps({"metric": "icp_iterations_per_sec", "value": a.steps * a.iters * world / dt, "unit": "iterations/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "DRY RUN of the rank protocol on CPU (gloo): no registration is computed"},
                          "dry_run": True, "records_gathered_ok": bool(ok), "slowest_rank_sleep_ms": 2.0 * world,
                          "secondary": {"c5": c5} if c5 else {}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def register_leg(pkg, ctx, np, cpu=True):
    """The reference's ONE live entry point (Main_KSS_ICP.cpp:79-82 -> KSSICP_Registration, KSS_ICP.hpp:69-131) on C1-size
    inputs: AIVS down-sampling of both clouds to pNumber = min(n) / 2 (2000-point clouds -> 1000-point samples, accurate = 8)
    and kss_register (pre-shape, 729-candidate rotation search, judge ICP, candidate ICP batch, final transform), and
    kss_register alone on 2000 x 2000 samples (the reference's cap, KSS_ICP.hpp:64-66).  The oracle's one-core time of the same
    call stands beside each; rot_search_kernel is event-timed and priced against the FP32 vector peak (8 flop per pair)."""
    import __graft_entry__ as graft
    S = pkg.synth
    O = graft.load_oracle() if cpu else None      # (the checker, timed as this leg's CPU baseline AFTER the GPU measurement; never in it)
    out = {}
    for name, n, sub in (("c1_2000_to_1000", 2000, True), ("samples_2000x2000", 2000, False)):
        src, tgt = S.make_pair(4242 + n + int(sub), n, R=S.rot_axis_angle([0.3, 0.2, 1.0], np.deg2rad(30.0)), t=(0.05, -0.02, 0.03), shape="bumpy")
        src = src.astype(np.float64); tgt = tgt.astype(np.float64)
        m = min(len(src), len(tgt)) // 2

        def gpu():
            if sub:
                (t_, _), (s_, _) = ctx.downsample_aivs_pair(tgt, m, src, m)      # both clouds at once (KSS_ICP.hpp:71-81 does the target, then the source; independent)
            else:
                s_, t_ = src, tgt
            return s_, t_, ctx.register(s_, t_, src, 8.0, 1000)
        gpu()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            ss, tt, r = gpu()
        dt = (time.perf_counter() - t0) / reps
        ctx.profile_enable(True); ctx.profile_reset()
        ctx.register(ss, tt, src, 8.0, 1000)
        rms, rn = ctx.profile_get(pkg.K_ROT_SEARCH)
        ctx.profile_enable(False)
        cpu_dt, k = None, None
        if cpu:
            t0 = time.perf_counter()
            if sub:
                os_, ot_ = src[O.aivs(src, m)], tgt[O.aivs(tgt, m)]
            else:
                os_, ot_ = src, tgt
            k = O.kssicp_register(os_, ot_, src, 8.0, 1000)
            cpu_dt = time.perf_counter() - t0
        g = int(r["grid"])
        flops = 8.0 * g ** 3 * len(ss) * len(tt)
        rs = rms / rn * 1e-3 if rn else None
        out[name] = {"workload": "KSSICP_Registration on a %d x %d bumpy pair, 30 deg: %s kss_register (pre-shape, %d-candidate rotation search, judge ICP, "
                                 "%d-candidate ICP batch, transform of the full cloud)" % (n, n, "AIVS to %d-point samples + " % m if sub else "", g ** 3, int(r["n_angle_list"])),
                     "ms_per_registration": dt * 1e3, "registrations_per_sec": 1.0 / dt,
                     "cpu_oracle_1core_ms": cpu_dt * 1e3 if cpu else None, "speedup_vs_cpu_1core": cpu_dt / dt if cpu else None,
                     "max_abs_R_diff_vs_oracle": float(np.abs(r["R"] - k["R"]).max()) if cpu else None,
                     "same_angle_index": bool(r["angle_index"] == k["angle_index"]) if cpu else None,
                     "n_src_samples": len(ss), "n_tgt_samples": len(tt), "candidates": int(r["n_angle_list"]), "icp_iterations": int(r["icp_iterations"]),
                     "rot_search_kernel": {"avg_launch_ms": rms / rn if rn else None, "launches": rn, "flops_per_launch": flops,
                                           "achieved_TFLOPs": flops / rs / 1e12 if rs else None, "peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
                                           "frac_of_fp32_valu_peak": flops / rs / 1e12 / FP32_VALU_PEAK_TFLOPS if rs else None}}
    return out


def c5_leg(pkg, ctx, torch, np, dist, world, rank, dev, pairs_per_rank=1024, n=10000, iters=20):
    """C5: every rank registers ITS shard of `pairs_per_rank` ModelNet40-scale pairs (pair ids rank * shard + i, no input
    exchange) in one kss_icp_batch_dev call, then ONE all-gather of the 96-byte records over RCCL.  Run by every rank; the time
    is the MAX over ranks of (batch + gather) between barriers."""
    S = pkg.synth
    src = np.empty((pairs_per_rank * n, 3), np.float32); tgt = np.empty((pairs_per_rank * n, 3), np.float32)
    for i in range(pairs_per_rank):
        s, t = S.config_c3_pair(rank * pairs_per_rank + i, n)
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    off = np.arange(pairs_per_rank + 1, dtype=np.int64) * n
    d_src = torch.from_numpy(src).to(dev); d_tgt = torch.from_numpy(tgt).to(dev)
    p = ctx.icp_params(max_iterations=iters, fixed_iterations=1)
    ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)        # warm-up (workspace, worker threads)
    reps = 3
    times, gathers, allrec, res = [], [], None, None
    for _ in range(reps):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
        t1 = time.perf_counter()
        local = pkg.shard.records_to_array(res, rank * pairs_per_rank)
        allrec = pkg.shard.gather_records(local, world * pairs_per_rank, world, rank, device=dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        dist.barrier()
        t = torch.tensor([t2 - t0, t2 - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times.append(float(t[0].item())); gathers.append(float(t[1].item()))
    out = None
    if rank == 0:
        dt = min(times)
        recs = pkg.shard.array_to_records(allrec, pkg.IcpResult)
        mine_ok = all(np.array_equal(np.array(recs[i].T), np.array(res[i].T)) and recs[i].fitness == res[i].fitness and recs[i].pair_id == i
                      for i in range(pairs_per_rank))
        ids_ok = len(recs) == world * pairs_per_rank and all(r.pair_id == i for i, r in enumerate(recs))
        out = {"workload": "C5: %d pairs of %dx%d per rank x %d ranks, %d fixed ICP iterations + fitness, one kss_icp_batch_dev call per rank, "
                           "one all-gather of the %d 96-byte records over RCCL" % (pairs_per_rank, n, n, world, iters, world * pairs_per_rank),
               "n_gpus": world, "pairs_total": world * pairs_per_rank, "ms_per_batch_max_over_ranks": dt * 1e3,
               "registrations_per_sec": world * pairs_per_rank / dt, "pair_iterations_per_sec": world * pairs_per_rank * iters / dt,
               "gather_us_max_over_ranks": min(gathers) * 1e6, "gathered_records_equal_local": bool(mine_ok), "gathered_pair_ids_in_order": bool(ids_ok),
               "scaling": "weak", "result_iterations": int(res[0].iterations)}
    del d_src, d_tgt
    return out


def secondary_legs(pkg, ctx, torch, np, legs=("c3", "c4", "stream")):
    """GPU-only legs of the other BASELINE configs (each a fraction of a second of GPU time)."""
    S = pkg.synth
    out = {}
    sync = torch.cuda.synchronize
    if "c3" not in legs and "c4" not in legs and "stream" not in legs:
        return out

    def kernel_time(k):
        ms, n = ctx.profile_get(k)
        return (ms / n if n else None), n

    if "c3" in legs:
        # ---- C3: 1024 x (10k x 10k), 20 fixed iterations + fitness, one kss_icp_batch_dev call -----------------------
        npairs, n, iters = 1024, 10000, 20
        src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
        for i in range(npairs):
            s, t = S.config_c3_pair(i, n)
            src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
        off = np.arange(npairs + 1, dtype=np.int64) * n
        d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
        p = ctx.icp_params(max_iterations=iters, fixed_iterations=1)
        ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
        ctx.profile_enable(True); ctx.profile_reset()
        reps = 3
        sync(); t0 = time.perf_counter()
        for _ in range(reps):
            res = ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
        sync(); dt = (time.perf_counter() - t0) / reps
        kms, kn = kernel_time(pkg.K_GRID_NN)
        rms, rn = kernel_time(pkg.K_RESIDENT)
        _, runits = ctx.profile_get(pkg.K_RESIDENT_PASS)
        bms, bn = kernel_time(pkg.K_GRID_BUILD)
        ctx.profile_enable(False)
        comp = 12.0 * (2 * npairs * n) + 8.0 * npairs * n           # SURVEY 8d compulsory bytes of one pass over the batch
        out["c3"] = {"workload": "C3: %d independent %dx%d pairs in one kss_icp_batch_dev call, %d fixed ICP iterations + fitness pass, setup included" % (npairs, n, n, iters),
                     "ms_per_batch": dt * 1e3, "pair_iterations_per_sec": npairs * iters / dt, "registrations_per_sec": npairs / dt,
                     "correspondences_per_sec": npairs * n * (iters + 1) / dt, "cell_list_build_ms": bms,
                     "result_iterations": int(res[0].iterations)}
        if rn:   # the pair-resident engine: every pass of every pair in ONE launch, or -- more pairs than the device runs at
            # once -- in two (passes 0-2 of every pair, then the rest, longest pairs first: kss_engine.hip, resident_loop)
            passes = iters + 1
            lpb = max(1, int(round(rn / float(reps))))   # launches per batch
            rms = rms * lpb                              # kernel time per batch (the launches of a batch follow each other)
            pass_eq = rms / passes                       # ... per pass of the whole batch
            tr = read_traffic("resident_icp")
            if tr:
                tr = tr * lpb                            # (the counters are averaged per launch)
            out["c3"].update({"kernel": "resident_icp_kernel", "avg_launch_ms": rms, "launches": rn, "launches_per_batch": lpb,
                              "avg_launch_ms_is": "the kernel time of one batch (sum over its launches)",
                              "pair_passes_per_launch": runits / rn * lpb if rn else None,
                              "ms_per_pass_of_the_batch": pass_eq,
                              "bound": "VALU / LDS latency inside one CU per pair (targets and cell table in LDS, sources in registers: a pass moves no "
                                       "source or target through HBM; the canonical f64 sum tree is ~60 % of a pass's instructions)",
                              "queries_per_sec_in_kernel": npairs * n * passes / (rms * 1e-3),
                              "compulsory_bytes_per_pass": comp, "frac_of_hbm_peak": comp / (pass_eq * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic": tr, "traffic_per_pass": tr / passes if tr else None,
                              "frac_by_pmc_traffic": (tr / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None})
        else:
            out["c3"].update({"kernel": "gridb_pass_kernel", "avg_launch_ms": kms, "launches": kn,
                              "bound": "residency / HBM (64 B of traffic per source and pass: float4 source in and out, winner, skip state)",
                              "queries_per_sec_in_kernel": npairs * n / (kms * 1e-3) if kms else None,
                              "compulsory_bytes_per_launch": comp, "frac_of_hbm_peak": comp / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None,
                              "traffic": read_traffic("gridb_pass"),
                              "frac_by_pmc_traffic": (read_traffic("gridb_pass") / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (kms and read_traffic("gridb_pass")) else None})
        del d_src, d_tgt, src, tgt

    if "c4" in legs:
        # ---- C4: 1M x 1M, 2x scale + 60 deg: pre-shape of both clouds, one NN pass, 10 ICP iterations ------------------
        n = 1000000
        src, tgt = S.config_c4(n)
        d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
        for _ in range(3):
            (cS, rS), (cT, rT) = ctx.preshape_stats_pair_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, pkg.binding.F32)
        ctx.profile_enable(True); ctx.profile_reset()
        reps = 20
        sync(); t0 = time.perf_counter()
        for _ in range(reps):
            (cS, rS), (cT, rT) = ctx.preshape_stats_pair_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, pkg.binding.F32)
        sync(); pre_dt = (time.perf_counter() - t0) / reps
        pms, pn = kernel_time(pkg.K_PRESHAPE)
        # S' on the device (f64 as the reference holds clouds), narrowed for the NN engine
        d_s64 = d_src.to(torch.float64); d_sp = torch.empty_like(d_s64)
        pose = ctx.make_pose([cT[k] - cS[k] for k in range(3)], cT, rT / rS, [0.0, 0.0, 0.0])
        ctx.pose_apply_dev(d_s64.data_ptr(), n, pose, d_sp.data_ptr()); ctx.synchronize()
        d_pre = d_sp.to(torch.float32).contiguous()
        d_idx = torch.empty(n, dtype=torch.int32, device="cuda"); d_d2 = torch.empty(n, dtype=torch.float32, device="cuda")
        ctx.nn_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, d_idx.data_ptr(), d_d2.data_ptr())
        sync(); t0 = time.perf_counter()
        ctx.nn_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, d_idx.data_ptr(), d_d2.data_ptr())
        sync(); nn_dt = time.perf_counter() - t0
        p = ctx.icp_params(max_iterations=10, fixed_iterations=1)
        ctx.icp_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, p)
        ctx.profile_enable(5)        # (a bracketed launch is a plain one: sample a few of the 11, keep the others gated)
        ctx.profile_reset()
        sync(); t0 = time.perf_counter()
        r = ctx.icp_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, p)
        sync(); icp_dt = time.perf_counter() - t0
        kms, kn = kernel_time(pkg.K_GRID_NN)
        ctx.profile_enable(False)
        comp = 12.0 * (2 * n) + 8.0 * n
        out["c4"] = {"workload": "C4: one %dx%d pair, source = 2 x R(60 deg) x target + t: pre-shape statistics of both clouds (one call), one exact NN pass (setup included), 10 fixed ICP iterations + fitness (setup included)" % (n, n),
                     "preshape_stats_ms": pre_dt * 1e3, "preshape_event_ms": pms, "preshape_algorithmic_bytes": 24.0 * 2 * n,
                     "preshape_frac_of_hbm_peak": 24.0 * 2 * n / pre_dt / 1e9 / HBM_PEAK_GBS, "scale_estimate": rT / rS,
                     "nn_pass_ms": nn_dt * 1e3, "icp_10_iters_ms": icp_dt * 1e3, "icp_iterations_per_sec": 10 / icp_dt,
                     "kernel": "grid_pass_kernel", "avg_launch_ms": kms, "launches": kn, "bound": "vector-memory pipe / L2 latency",
                     "queries_per_sec_in_kernel": n / (kms * 1e-3) if kms else None,
                     "compulsory_bytes_per_launch": comp, "frac_of_hbm_peak": comp / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None,
                     "traffic": read_traffic("grid_pass_c4"),
                     "frac_by_pmc_traffic": (read_traffic("grid_pass_c4") / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (kms and read_traffic("grid_pass_c4")) else None,
                     "fitness": float(r.fitness)}
        del d_src, d_tgt, d_s64, d_sp, d_pre, d_idx, d_d2

    if "stream" in legs:
        # ---- streaming: pre-shape statistics (two read passes) of 64M f32 points = 768 MB, past every cache ------------------
        n = 64 * 1024 * 1024
        x = torch.rand((n, 3), dtype=torch.float32, device="cuda")
        ctx.preshape_stats_dev(x.data_ptr(), pkg.binding.F32, n)
        ctx.profile_enable(True); ctx.profile_reset()
        reps = 5
        sync(); t0 = time.perf_counter()
        for _ in range(reps):
            ctx.preshape_stats_dev(x.data_ptr(), pkg.binding.F32, n)
        sync(); dt = (time.perf_counter() - t0) / reps
        pms, pn = kernel_time(pkg.K_PRESHAPE)
        ctx.profile_enable(False)
        out["stream_64m"] = {"workload": "pre-shape statistics of one %d-point f32 cloud (768 MB, two read passes: sum + centroid, radius)" % n,
                             "kernel": "preshape_sum_kernel + preshape_radius_kernel", "bound": "hbm", "ms": dt * 1e3, "event_ms_both_launches": pms,
                             "algorithmic_bytes": 24.0 * n, "achieved_GBps": 24.0 * n / dt / 1e9, "frac_of_hbm_peak": 24.0 * n / dt / 1e9 / HBM_PEAK_GBS}
        del x
    return out


def main():
    a = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        return launch_ranks(a)            # nothing below has run yet: no torch, no HIP in this process
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus\n" % (a.gpus, world))
        return 2
    if a.dry_run_dist:
        return dry_run(a, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the KSS-ICP core has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("KSS_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    pkg = graft.load_package()
    S = pkg.synth
    split = a.split_source and (world > 1 or force_dist)
    if split:
        # ONE pair for the whole job: every rank holds the target and its contiguous block of source rows
        src, tgt = S.make_pair(0, a.n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
        lo, hi = pkg.shard.shard_range(a.n, world, rank)
        src = np.ascontiguousarray(src[lo:hi])
    else:
        # one independent pair per rank (pair_id = rank): the path shards over pairs with no data exchange
        src, tgt = S.make_pair(rank, a.n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
    n_src = len(src)
    d_src = torch.from_numpy(src).to(dev)
    d_tgt = torch.from_numpy(tgt).to(dev)
    torch.cuda.synchronize()

    ctx = pkg.Context(local_rank)       # raises without libkssicp.so / GPU
    modes = {"auto": pkg.NN_AUTO, "brute": pkg.NN_BRUTE, "grid": pkg.NN_GRID}
    rccl_lib = rccl_comm = None
    if split:
        rccl_lib, rccl_comm = pkg.shard.rccl_comm(rank, world, device=dev)

    keepalive = []

    def make_step(mode):
        kw = dict(max_iterations=a.iters, fixed_iterations=1, compute_fitness=1, nn_fma=a.fma,
                  nn_sources_per_thread=a.spt, nn_target_splits=a.splits, nn_mode=modes[mode])
        if split:   # the exchange step: ncclAllReduce of the 20 sums after every pass, inside the library
            params, link = pkg.shard.rccl_allreduce_params(ctx, pkg.binding, pkg.load_library(), rccl_comm, **kw)
            keepalive.append(link)
        else:
            params = ctx.icp_params(**kw)
        done = []

        def step():
            res = ctx.icp_dev(d_src.data_ptr(), n_src, d_tgt.data_ptr(), a.n, params)
            done.append(res)
            return res

        def finish():
            # The K registrations of the timed region are this rank's shard of a K x N-pair batch: ONE all-gather of the
            # 96-byte (R, t) records over RCCL/xGMI at the end of the batch (one collective per batch, never
            # per pair), inside the timed bracket.  (split: every rank already holds the same record.)
            if split or not (world > 1 or force_dist) or not done:
                done.clear()
                return
            recs = (pkg.IcpResult * len(done))(*done)
            local = pkg.shard.records_to_array(recs, rank * len(done))
            pkg.shard.gather_records(local, world * len(done), world, rank, device=dev)
            done.clear()
        return step, finish

    # ---- primary measurement ---------------------------------------------------------------------------
    step, finish = make_step(a.mode)
    for _ in range(a.warmup):
        step()
    finish()
    ctx.profile_enable(a.prof_stride)   # HIP events around every n-th launch of the timed region
    ctx.profile_reset()
    dt, last = run_steps(step, a.steps, dist, world, torch.cuda.synchronize, finish)
    prof = {k: ctx.profile_get(getattr(pkg, k)) for k in ("K_NN_SWEEP", "K_CORR_REDUCE", "K_GRID_NN", "K_GRID_BUILD", "K_GRID_CHAIN", "K_GRID_CHAIN_PASS")}
    gstats = ctx.grid_stats()
    ctx.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    used_grid = prof["K_GRID_NN"][1] > 0
    ev_over_s = ctx.profile_event_overhead() * 1e-3 if rank == 0 else 0.0   # event pair around an empty launch
    evals_per_launch = None
    if used_grid and rank == 0 and not split:   # (split: a step is collective, every rank would have to join)
        # untimed diagnostic step: the kernel counts its distance evaluations (mean over the step's launches)
        import ctypes as C
        lib = pkg.load_library()
        lib.kss_debug_grid_evals.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        buf = (C.c_double * 2)()
        os.environ["KSS_GRID_STAMPS"] = "1"
        try:
            lib.kss_debug_grid_evals(ctx.h, buf)
            step()
            lib.kss_debug_grid_evals(ctx.h, buf)
            finish() if world == 1 and not force_dist else None   # (drops the record; never a collective here)
        finally:
            del os.environ["KSS_GRID_STAMPS"]
        if buf[1] > 0:
            evals_per_launch = buf[0] / buf[1]

    # ---- secondary: the brute-force sweep (north-star kernel), single-GPU runs only ------------------------
    brute = None
    if world == 1 and a.brute_steps > 0 and used_grid:
        bstep, bfinish = make_step("brute")
        bstep()
        bfinish()
        ctx.profile_enable(True)
        ctx.profile_reset()
        bdt, blast = run_steps(bstep, a.brute_steps, dist, world, torch.cuda.synchronize, bfinish)
        bms, bn = ctx.profile_get(pkg.K_NN_SWEEP)
        ctx.profile_enable(False)
        brute = (bdt, bms, bn, blast)

    # ---- C5 (world > 1, or KSS_BENCH_FORCE_DIST=1 with one rank): every rank's 1024-pair shard + ONE gather -- all ranks take part
    c5 = None
    if (world > 1 or force_dist) and not split and not a.no_secondary and "c5" in a.legs.split(","):
        c5 = c5_leg(pkg, ctx, torch, np, dist, world, rank, dev, pairs_per_rank=a.c5_pairs)

    if rank == 0:
        value = a.steps * a.iters * (1 if split else world) / dt   # split: the ranks share ONE registration
        passes = a.iters + 1

        def valu_roofline(raw_s, launches):
            flops = 8.0 * a.n * a.n          # 8 flop per (source, target) pair
            avg_s = raw_s
            ach = flops / avg_s / 1e12 if raw_s > 0 else 0.0
            return {"kernel": "nn_sweep_kernel", "bound": "valu-fp32",
                    "bound_note": "min-reduction on the FP32 vector ALU (MFMA not used); peak 157.3 TFLOP/s counts an fma as 2 "
                                  "flop, the parity arithmetic (FLANN L2_Simple, no fma) issues one instruction per flop, so "
                                  "0.5 is its ceiling; see DESIGN.md",
                    "achieved": ach, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_VALU_PEAK_TFLOPS,
                    "traffic": read_traffic("nn_sweep"), "avg_launch_ms": avg_s * 1e3, "launches": launches,
                    "event_pair_around_empty_kernel_ms": ev_over_s * 1e3, "flops_per_launch": flops}

        out = {
            "metric": "icp_iterations_per_sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: one %dx%d uniform-sphere pair per GPU, %d fixed ICP iterations + fitness pass per step "
                                   "(exact NN + cov reduce + host 3x3 SVD), R_z(10deg), jitter 1e-3" % (a.n, a.n, a.iters),
                       "n_src": a.n, "n_tgt": a.n, "icp_iters_per_step": a.iters, "pairs_per_gpu": 1,
                       "parallelism": ("source rows of one pair split over %d ranks, ncclAllReduce of 20 f64 per iteration" % world) if split
                                      else "independent pairs sharded over ranks (%d registrations per rank in the timed region), no data-path "
                                           "collective, one all-gather of the 96-byte records per batch" % a.steps,
                       "nn_engine": "cell list + brute-force fallback" if used_grid else "brute-force sweep",
                       "nn_arithmetic": "fma" if a.fma else "reference (no fma)"},
            "correspondences_per_sec": float(a.n) * passes * a.steps * (1 if split else world) / dt,
            "result": {"iterations": int(last.iterations), "fitness": float(last.fitness)},
        }
        if used_grid:
            gms, gn = prof["K_GRID_NN"]
            raw_s = gms / max(1, gn) * 1e-3
            # chained launches (one kernel runs the remaining iterations of a registration, waiting at an in-kernel gate for
            # each transform): bracketed as a whole, every one of them; per pass = that time / the passes they ran
            cms, cn = prof["K_GRID_CHAIN"]
            _, cpasses = prof["K_GRID_CHAIN_PASS"]
            chained = cn > 0 and cpasses > 0
            plain_s = raw_s
            if chained:
                raw_s = cms / cpasses * 1e-3
            # launch duration = the HIP-event bracket as measured.  For a ~12 us kernel the bracket itself is not free:
            # `event_pair_around_empty_kernel_ms` is what the same bracket reads around an EMPTY launch (dispatch + event
            # packets), so rocprofv3's kernel trace (profiles/) reads 1-2 us less than avg_launch_ms.  Not subtracted.
            avg_s = raw_s
            ev = evals_per_launch if evals_per_launch is not None else gstats["evaluations_per_pass"]
            # compulsory bytes of one pass = 12 B per source and per target + 8 B of result per source
            comp = 12.0 * (n_src + a.n) + 8.0 * n_src
            # what the lanes REQUEST from the L1/L2 hierarchy (not HBM): per source 16 B read + 16 B transformed write +
            # 2 x 4 B previous-winner position + the previous winner (16 B) + 9 cell-bound loads of 16 B + 16 B per point
            # walked (the kernel counts its distance evaluations in an untimed diagnostic step)
            req = n_src * (16.0 + 16.0 + 8.0 + 9 * 16.0) + ev * 16.0
            traffic = read_traffic("grid_pass")
            if chained:   # PMC bytes of a chained launch / the passes it ran (the profiled command runs the same chains)
                ct = read_traffic("grid_pass_chain")
                traffic = ct / (cpasses / cn) if ct else None
            out["roofline"] = {
                "kernel": "grid_pass_kernel<.., CHAIN = true>" if chained else "grid_pass_kernel", "bound": "latency",
                "bound_note": "skip test / cell-list search of the few sources that need one / correspondence sums / pair reduction; "
                              "196 workgroups of 512 sources.  Chained form: one launch runs the remaining iterations of a "
                              "registration and every pass waits at an in-kernel gate for the transform the host solves from the "
                              "previous pass's sums, so a pass = host round trip (PCIe both ways + 3x3 SVD) + a chain of dependent "
                              "L2 round trips (cell bounds, points, row hand-over, row sums, publish).  `avg_launch_ms` is the whole "
                              "launch, `avg_pass_ms` = launch time / passes (gate wait included) is what `achieved` / `frac` price "
                              "SURVEY 8(d)'s compulsory bytes against, as the contract asks; the kernel is nowhere near HBM-bound "
                              "and is not meant to be (its working set lives in L2 and in registers).  `plain_launches`: the first "
                              "pass of a registration and the fitness pass, sampled",
                "achieved": comp / avg_s / 1e9 if avg_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (comp / avg_s / 1e9 / HBM_PEAK_GBS) if avg_s > 0 else 0.0,
                "traffic": traffic,
                "frac_by_pmc_traffic": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_s > 0) else None,
                "compulsory_bytes_per_launch": comp,
                "queries_per_sec": n_src / avg_s if avg_s > 0 else 0.0,
                "l1_requested_bytes_per_launch": req, "l2_request_GBps": req / avg_s / 1e9 if avg_s > 0 else 0.0,
                "l2_peak_GBps": L2_PEAK_GBS, "frac_of_l2_peak": (req / avg_s / 1e9 / L2_PEAK_GBS) if avg_s > 0 else 0.0,
                "avg_launch_ms": (cms / cn) if chained else avg_s * 1e3, "launches": cn if chained else gn,
                "passes_per_launch": (cpasses / cn) if chained else 1.0, "avg_pass_ms": avg_s * 1e3,
                "plain_launches": {"avg_launch_ms": plain_s * 1e3, "launches": gn, "launches_timed_every": a.prof_stride},
                "event_pair_around_empty_kernel_ms": ev_over_s * 1e3,
                "distance_evaluations_per_launch": ev, "distance_evaluations_unpruned_27_cells": gstats["evaluations_per_pass"],
                "grid": {k: gstats[k] for k in ("h", "gx", "gy", "gz", "occupied_cells")}}
            out["setup"] = {"cell_list_build_avg_ms": prof["K_GRID_BUILD"][0] / max(1, prof["K_GRID_BUILD"][1]),
                            "builds": prof["K_GRID_BUILD"][1], "fallback_sweep_launches": prof["K_NN_SWEEP"][1]}
        else:
            nms, nn_ = prof["K_NN_SWEEP"]
            out["roofline"] = valu_roofline(nms / max(1, nn_) * 1e-3, nn_)
        if brute is not None:
            bdt, bms, bn, blast = brute
            out["brute_force"] = {"value": a.brute_steps * a.iters / bdt, "unit": "iterations/s", "steps": a.brute_steps,
                                  "ms_per_step": bdt / a.brute_steps * 1e3,
                                  "point_pairs_evaluated_per_sec": float(a.n) * a.n * passes * a.brute_steps / bdt,
                                  "roofline": valu_roofline(bms / max(1, bn) * 1e-3, bn),
                                  "max_abs_T_diff_vs_default": float(np.abs(np.array(blast.T) - np.array(last.T)).max())}
        if world == 1 and not a.no_secondary:
            del d_src, d_tgt
            legs = set(a.legs.split(","))
            out["secondary"] = secondary_legs(pkg, ctx, torch, np, legs)
            if "register" in legs:
                out["secondary"]["register"] = register_leg(pkg, ctx, np, cpu=not a.no_cpu_baseline)
        if c5 is not None:
            out.setdefault("secondary", {})["c5"] = c5
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(a.n, a.iters, src, tgt)
            Tg = np.array(last.T, dtype=np.float64)
            out["parity_vs_cpu_baseline_max_abs_T"] = float(np.abs(Tg - np.array(cb.pop("T"))).max())
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1core"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if rccl_comm is not None:
        rccl_lib.ncclCommDestroy(rccl_comm)
    ctx.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
