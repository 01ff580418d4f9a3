#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (not the headline bench line): C3 (batch of 10k x 10k pairs on one GPU),
C4 (1M x 1M pair: pre-shape + one NN sweep + 10 ICP iterations) and the streaming kernels at large N.
Prints one JSON object per config.  usage: python tools/bench_configs.py [c3 c4 stream] [--pairs 1024]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def c3(pkg, ctx, torch, npairs, iters=20, grid_only=False):
    S = pkg.synth
    n = 10000
    t0 = time.perf_counter()
    src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
    for i in range(npairs):
        s, t = S.config_c3_pair(i, n)
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    gen = time.perf_counter() - t0
    off = np.arange(npairs + 1, dtype=np.int64) * n
    d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
    out = {"config": "C3: %d independent 10k x 10k pairs on one GPU, %d fixed ICP iterations + fitness" % (npairs, iters), "gen_s": gen}
    res = {}
    for mode, m in (("grid", pkg.NN_GRID), ("brute", pkg.NN_BRUTE)):
        if grid_only and mode == "brute":
            continue
        p = ctx.icp_params(max_iterations=iters, fixed_iterations=1, nn_mode=m)
        ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)      # warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[mode] = r
        out[mode] = {"seconds": dt, "pair_iterations_per_sec": npairs * iters / dt, "registrations_per_sec": npairs / dt,
                     "correspondences_per_sec": npairs * n * (iters + 1) / dt}
    if grid_only:
        return out
    out["max_abs_T_diff_grid_vs_brute"] = float(max(np.abs(res["grid"][i].matrix() - res["brute"][i].matrix()).max() for i in range(npairs)))
    # CPU baseline on a 16-pair sample (1 core, kd-tree)
    O = graft.load_oracle()
    t0 = time.perf_counter()
    m = min(16, npairs)
    for i in range(m):
        O.icp(src[i * n:(i + 1) * n], tgt[i * n:(i + 1) * n], O.icp_params(max_iterations=iters, fixed_iterations=1))
    cdt = (time.perf_counter() - t0) / m
    out["cpu_1core_pair_iterations_per_sec"] = iters / cdt
    return out


def c4(pkg, ctx, torch, n=1000000, iters=10):
    S = pkg.synth
    src, tgt = S.config_c4(n)
    d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
    out = {"config": "C4: one %dx%d pair, scale 2 + 60 deg: pre-shape stats, one NN sweep, %d ICP iterations" % (n, n, iters)}
    for _ in range(3):
        (cS, rS), (cT, rT) = ctx.preshape_stats_pair_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        (cS, rS), (cT, rT) = ctx.preshape_stats_pair_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, 0)
    torch.cuda.synchronize(); out["preshape_stats_ms"] = (time.perf_counter() - t0) * 1e2
    out["scale_estimate"] = rT / rS
    # S' on the device in f64 (the reference's cloud type), narrowed to f32 for the NN engine
    d_s64 = d_src.to(torch.float64); d_sp = torch.empty_like(d_s64)
    pose = ctx.make_pose([cT[k] - cS[k] for k in range(3)], cT, rT / rS, [0.0, 0.0, 0.0])
    ctx.pose_apply_dev(d_s64.data_ptr(), n, pose, d_sp.data_ptr()); ctx.synchronize()
    d_pre = d_sp.to(torch.float32).contiguous()
    d_idx = torch.empty(n, dtype=torch.int32, device="cuda"); d_d2 = torch.empty(n, dtype=torch.float32, device="cuda")
    for mode, m in (("grid", pkg.NN_GRID), ("brute", pkg.NN_BRUTE)):
        ctx.set_nn_mode(m)
        ctx.nn_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, d_idx.data_ptr(), d_d2.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.nn_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, d_idx.data_ptr(), d_d2.data_ptr())
        torch.cuda.synchronize(); out["nn_sweep_ms_" + mode] = (time.perf_counter() - t0) * 1e3
        if mode == "grid":
            gi = d_idx.cpu().numpy().copy(); gd = d_d2.cpu().numpy().copy()
        else:
            out["nn_identical_grid_vs_brute"] = bool(np.array_equal(gi, d_idx.cpu().numpy()) and np.array_equal(gd, d_d2.cpu().numpy()))
    ctx.set_nn_mode(pkg.NN_AUTO)
    p = ctx.icp_params(max_iterations=iters, fixed_iterations=1)
    ctx.icp_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.icp_dev(d_pre.data_ptr(), n, d_tgt.data_ptr(), n, p)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["icp_%d_iters_ms" % iters] = dt * 1e3
    out["icp_iterations_per_sec"] = iters / dt
    out["fitness"] = r.fitness
    return out


def stream(pkg, ctx, torch, n=64 * 1024 * 1024):
    """(a) pre-shape statistics and pose application at a size far beyond the caches (HBM roofline)."""
    out = {"config": "streaming kernels at %d points" % n}
    x = torch.rand((n, 3), dtype=torch.float32, device="cuda")
    ctx.preshape_stats_dev(x.data_ptr(), 0, n)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(5):
        ctx.preshape_stats_dev(x.data_ptr(), 0, n)
    ms, k = ctx.profile_get(pkg.K_PRESHAPE)
    ctx.profile_enable(False)
    out["preshape_f32"] = {"ms": ms / k, "algorithmic_bytes": 24.0 * n, "GBps": 24.0 * n / (ms / k * 1e-3) / 1e9,
                           "frac_of_8TBps": 24.0 * n / (ms / k * 1e-3) / 8e12}
    del x
    m = n // 4
    y = torch.rand((m, 3), dtype=torch.float64, device="cuda"); z = torch.empty_like(y)
    pose = ctx.make_pose([0.1, 0.2, 0.3], [0.5, 0.5, 0.5], 1.25, [0.3, 0.2, 0.1])
    import ctypes as C
    L = pkg.load_library()
    def run():
        rc = L.kss_pose_apply_dev(ctx.h, C.c_void_p(y.data_ptr()), C.c_int64(m), C.byref(pose), C.c_void_p(z.data_ptr()))
        assert rc == 0
    run(); torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(5):
        run()
    ms, k = ctx.profile_get(pkg.K_POSE_APPLY)
    ctx.profile_enable(False)
    out["pose_apply_f64"] = {"ms": ms / k, "algorithmic_bytes": 48.0 * m, "GBps": 48.0 * m / (ms / k * 1e-3) / 1e9,
                             "frac_of_8TBps": 48.0 * m / (ms / k * 1e-3) / 8e12}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c3", "c4", "stream"])
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--grid-only", action="store_true", help="C3: skip the brute-force leg and the CPU sample")
    a = ap.parse_args()
    import torch
    pkg = graft.load_package()
    ctx = pkg.Context(0)
    for w in a.which:
        if w == "c3":
            print(json.dumps(c3(pkg, ctx, torch, a.pairs, grid_only=a.grid_only)), flush=True)
        elif w == "c4":
            print(json.dumps(c4(pkg, ctx, torch)), flush=True)
        elif w == "stream":
            print(json.dumps(stream(pkg, ctx, torch)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
