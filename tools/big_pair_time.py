#!/usr/bin/env python3
"""ICP on one large well-posed pair (the full-resolution ICP that ends a registration): ms per iteration.
usage: python tools/big_pair_time.py [n=1000000] [iters=30]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
src, tgt = S.make_pair(0, n, R=S.rot_axis_angle([0, 0, 1], np.deg2rad(10.0)))
ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
ctx = pkg.Context(0)
for it in (iters, 2 * iters):
    p = ctx.icp_params(max_iterations=it, fixed_iterations=1, compute_fitness=0, nn_mode=pkg.NN_GRID)
    ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.icp_dev(ds.data_ptr(), n, dt.data_ptr(), n, p)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("n %d, %d iterations: %.3f ms" % (n, it, t * 1e3), flush=True)
    if it == iters: t1 = t
    else: print("per iteration (difference): %.1f us = %.2e sources/s; setup + first pass %.3f ms" % ((t - t1) / iters * 1e6, n / ((t - t1) / iters), (2 * t1 - t) * 1e3))
