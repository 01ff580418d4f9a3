#!/usr/bin/env python3
"""Pair-resident batch engine (kss_resident.hip): records against the one-pair path (bit for bit) and the time of a batch.

    python tools/resident_check.py [npairs] [points] [iters]

KSS_RESIDENT=0 in the environment runs the launch-per-pass engine instead (A/B)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    import torch
    pkg = graft.load_package()
    ctx = pkg.Context(0)
    S = pkg.synth
    src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
    for i in range(npairs):
        s, t = S.config_c3_pair(i, n)
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    off = np.arange(npairs + 1, dtype=np.int64) * n
    d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
    for kw in (dict(max_iterations=iters, fixed_iterations=1), dict()):
        p = ctx.icp_params(**kw)
        res = ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
        bad = 0
        step = max(1, npairs // 64)
        for i in range(0, npairs, step):
            one = ctx.icp_dev(d_src.data_ptr() + 12 * int(off[i]), n, d_tgt.data_ptr() + 12 * int(off[i]), n, p)
            same = (np.array_equal(np.array(res[i].T), np.array(one.T)) and res[i].iterations == one.iterations and res[i].state == one.state
                    and res[i].fitness == one.fitness and res[i].last_mse == one.last_mse)
            if not same:
                bad += 1
                if bad <= 5:
                    print("MISMATCH pair %d: |dT| %.3e iters %d/%d state %d/%d fitness %.17g / %.17g" % (
                        i, float(np.abs(np.array(res[i].T) - np.array(one.T)).max()), res[i].iterations, one.iterations, res[i].state, one.state,
                        res[i].fitness, one.fitness), flush=True)
        print("params %s: %d pairs checked, %d mismatches; iterations of pair 0: %d" % (kw, len(range(0, npairs, step)), bad, res[0].iterations), flush=True)
    p = ctx.icp_params(max_iterations=iters, fixed_iterations=1)
    ctx.profile_enable(True); ctx.profile_reset()
    reps = 5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        ctx.icp_batch_dev(d_src.data_ptr(), off, d_tgt.data_ptr(), off, p)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    out = {k: ctx.profile_get(getattr(pkg, k)) for k in ("K_RESIDENT", "K_RESIDENT_PASS", "K_GRID_NN", "K_GRID_BUILD")}
    print("batch of %d x %d x %d, %d iterations + fitness: %.3f ms per batch; kernels (ms, launches): %s" % (npairs, n, n, iters, dt * 1e3, out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
