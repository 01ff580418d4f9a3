#!/usr/bin/env python3
"""Small batches (where should AUTO pick the batched cell lists?): npairs x n points, ICP to convergence, both engines."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); S = pkg.synth
ctx = pkg.Context(0)
for npairs, n in ((4, 600), (16, 600), (4, 1500), (16, 1500), (64, 1500), (8, 4000), (32, 4000)):
    src = np.empty((npairs * n, 3), np.float32); tgt = np.empty((npairs * n, 3), np.float32)
    for i in range(npairs):
        s, t = S.make_pair(200 + i, n, R=S.rot_axis_angle([0.2, 1.0, 0.3], np.deg2rad(8.0 + i % 5)), t=(0.02, -0.01, 0.01), shape="bumpy")
        src[i * n:(i + 1) * n] = s; tgt[i * n:(i + 1) * n] = t
    off = np.arange(npairs + 1, dtype=np.int64) * n
    ds = torch.from_numpy(src).cuda(); dt = torch.from_numpy(tgt).cuda()
    out = {}
    for name, m in (("grid", pkg.NN_GRID), ("brute", pkg.NN_BRUTE)):
        p = ctx.icp_params(nn_mode=m)
        ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = ctx.icp_batch_dev(ds.data_ptr(), off, dt.data_ptr(), off, p)
        torch.cuda.synchronize(); out[name] = ((time.perf_counter() - t0) / 5, r)
    its = max(x.iterations for x in out["grid"][1])
    print("%3d pairs x %5d: cell lists %.2f ms, brute force %.2f ms (max %d iterations)" % (npairs, n, out["grid"][0] * 1e3, out["brute"][0] * 1e3, its), flush=True)
